// errors.h -- one thread-local error string behind is3d_last_error() (include/is3d_amd.h).
#pragma once
namespace is3d {
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}
