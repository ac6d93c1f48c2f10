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
struct is3d_plan;
namespace is3d {
// opts.accumulate of a plan (cf_multi.hip refuses it in front of a collective)
int plan_accumulate(const is3d_plan *plan);
// HIP device ordinal the plan lives on
int plan_device(const is3d_plan *plan);
}
