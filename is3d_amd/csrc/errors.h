// errors.h -- one thread-local error string behind is3d_last_error() (include/is3d_amd.h).
#pragma once
namespace is3d {
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// Create the HIP contexts of the listed devices (all visible ones for n == 0) ahead of their first use: the run driver calls it on
// a thread of its own while it parses the surface file (a cold context costs a quarter of a second).  Errors are left to the real calls.
void warm_devices(const int *devices, int n);
// process-wide resource counters behind is3d_resource_counters() (include/is3d_amd.h): what: 0 plans created, 1 device allocations
void count_resource(int what);
}
// Developer switches (A/B timing of kernel parts, cycle accounting, alternative writers and chunk rules) exist only in a -DIS3D_DEV build
// (`make DEV=1` -> is3d_amd/lib_dev/, loaded by the Python binding when IS3D_USE_DEV_LIB=1): the shipped library reads none of these
// variables, so a stray variable in a production job cannot change its chunking, its summation order or the validity of its spectrum.
#include <cstdlib>
namespace is3d {
#ifdef IS3D_DEV
inline const char *dev_env(const char *name) { return std::getenv(name); }
constexpr bool kDevBuild = true;
#else
inline const char *dev_env(const char *) { return nullptr; }
constexpr bool kDevBuild = false;
#endif
}
struct is3d_plan;
namespace is3d {
// opts.accumulate of a plan (cf_multi.hip refuses it in front of a collective)
int plan_accumulate(const is3d_plan *plan);
// HIP device ordinal the plan lives on
int plan_device(const is3d_plan *plan);
}
