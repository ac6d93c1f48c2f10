// cf_multi.hip -- the cell-axis split over several GPUs, inside the library (include/is3d_amd.h, "Multi-GPU").
//
// The reference has no distributed code; what makes the split legal is that its spectrum is a plain sum over cells
// (/root/reference/src/cpp/emissionfunction_smooth_kernels.cpp:363-375: dN_pTdpTdphidy[iS3D] += dN_pTdpTdphidy_tmp, one
// chunk of cells after the other).  Here a shard of cells plays the role of a chunk: every GPU runs prep -> main -> finalize on
// its contiguous block of cells and the per-bin spectra are added once at the end -- either in shard order by a device kernel
// (IS3D_REDUCE_ORDERED: bitwise reproducible, any device list) or by one RCCL all-reduce (IS3D_REDUCE_RCCL, and the
// one-process-per-GPU form is3d_comm_* / is3d_plan_execute_allreduce).  No other communication exists on this path.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the rccl.h types only at compile time): a single-GPU host needs no RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/is3d_amd.h"
#include "errors.h"

#define fail is3d::set_error

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(IS3D_ENODEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- RCCL, bound lazily ----
struct Rccl {
    void *handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;   // optional: polled by the waits behind a collective
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // IS3D_RCCL_LIBRARY: an explicit path (sites that keep RCCL elsewhere; the test suite's process-level test double, tests/cpp/fake_rccl.cpp)
        if (const char *path = getenv("IS3D_RCCL_LIBRARY")) {
            if (*path) {
                r.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
                if (!r.handle) {
                    const char *e = dlerror();
                    r.error = std::string("IS3D_RCCL_LIBRARY = ") + path + " cannot be loaded: " + (e ? e : "?");
                    return;
                }
            }
        }
        // a copy the host process has already mapped (e.g. the one PyTorch bundles) is reused, so that the process holds ONE RCCL
        if (!r.handle)
            for (const char *name : {"librccl.so", "librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
                if (r.handle) break;
            }
        if (!r.handle)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.handle) break;
            }
        if (!r.handle) {
            const char *e = dlerror();
            r.error = std::string("librccl.so.1 cannot be loaded: ") + (e ? e : "?");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p) { ok = false; r.error = std::string("librccl lacks ") + n; }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (ok) r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(r.handle, "ncclCommGetAsyncError");
        if (!ok) { dlclose(r.handle); r.handle = nullptr; }
    });
    return r;
}

int rccl_ready()
{
    Rccl &r = rccl();
    if (!r.handle) return fail(IS3D_ENODEVICE, "RCCL is not available: %s", r.error.c_str());
    return IS3D_OK;
}

#define NCCL_TRY(expr)                                                                                                   \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess) return fail(IS3D_ENODEVICE, "%s failed: %s", #expr, rccl().GetErrorString(r_));          \
    } while (0)

static_assert(sizeof(ncclUniqueId) == IS3D_COMM_ID_BYTES, "IS3D_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

// dst[i] += src[i]: the shard-order sum of IS3D_REDUCE_ORDERED (one 16-byte load per operand and lane, coalesced)
__global__ void __launch_bounds__(256) cf_add_spectrum(double *__restrict__ dst, const double *__restrict__ src, int64_t n)
{
    const int64_t i = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 < n) {
        double2 a = *(const double2 *)(dst + i);
        const double2 b = *(const double2 *)(src + i);
        a.x += b.x;
        a.y += b.y;
        *(double2 *)(dst + i) = a;
    } else if (i < n) {
        dst[i] += src[i];
    }
}

hipError_t launch_add_spectrum(double *dst, const double *src, int64_t n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int64_t pairs = (n + 1) / 2;
    hipLaunchKernelGGL(cf_add_spectrum, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, dst, src, n);
    return hipGetLastError();
}

}  // namespace

struct is3d_comm {
    ncclComm_t comm = nullptr;
    int32_t n_ranks = 1, rank = 0, device = 0;
    // error word that travels with every is3d_plan_execute_allreduce: d_flag[0] = 1.0 on a rank whose execute failed, summed over the
    // ranks next to the spectrum, read back into h_flag (pinned) on the same stream.  d_flag[1], d_flag[2] hold the constants 0.0, 1.0.
    double *d_flag = nullptr, *h_flag = nullptr;
    int64_t peer_errors = 0;          // sum of the error words of all collectives since the last is3d_comm_check
    bool flag_in_flight = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the last collective (is3d_comm_timings)
    bool timed = false, aborted = false;
    int pending = 0;                  // collectives enqueued since the last host-side wait that saw the stream (or ev1) complete
    double timeout_s = 300.0;         // deadline of every host-side wait behind a collective (is3d_comm_set_timeout; IS3D_COMM_TIMEOUT_S at creation)
};

namespace is3d {
void warm_devices(const int *devices, int n)
{
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return;
    const int cnt = (devices && n > 0) ? n : visible;
    for (int i = 0; i < cnt; i++) {
        const int d = (devices && n > 0) ? devices[i] : i;
        if (d < 0 || d >= visible) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        (void)hipFree(nullptr);                 // creates the context
    }
    (void)hipGetLastError();
}
}  // namespace is3d

extern "C" int is3d_shard_bounds(int64_t n_cells, int32_t rank, int32_t n_ranks, int64_t *lo, int64_t *hi)
{
    if (!lo || !hi || n_cells < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(IS3D_EINVAL, "is3d_shard_bounds: bad argument");
    const int64_t base = n_cells / n_ranks, rem = n_cells % n_ranks;
    *lo = rank * base + std::min<int64_t>(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return IS3D_OK;
}

extern "C" int is3d_comm_unique_id(uint8_t id[IS3D_COMM_ID_BYTES])
{
    if (!id) return fail(IS3D_EINVAL, "null id");
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId u;
    NCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return IS3D_OK;
}

namespace {
// the error word of a collective (is3d_comm): mode 0: f[0] = v before the sum; mode 1: f[3] += f[0] after it (the running total
// is3d_comm_check reads); one thread
__global__ void cf_comm_flag(double *f, int mode, double v)
{
    if (mode == 0) f[0] = v;
    else f[3] += f[0];
}

void comm_abort(is3d_comm *c)
{
    if (c->comm && rccl().handle && rccl().CommAbort) {
        (void)hipSetDevice(c->device);
        (void)rccl().CommAbort(c->comm);   // frees the communicator and makes THIS rank's collective kernels exit; the peers are not told:
                                           // each finds out through its own comm_wait (asynchronous error or deadline)
    }
    c->comm = nullptr;
    c->aborted = true;
}

// spectrum + error word in one group on the stream; on an RCCL failure the communicator is aborted
int comm_allreduce_flagged(is3d_comm *c, double *dN_dev, int64_t n, bool local_error, hipStream_t st)
{
    hipLaunchKernelGGL(cf_comm_flag, dim3(1), dim3(1), 0, st, c->d_flag, 0, local_error ? 1.0 : 0.0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev0, st));
    ncclResult_t r = rccl().GroupStart();
    if (r == ncclSuccess && n > 0) r = rccl().AllReduce(dN_dev, dN_dev, (size_t)n, ncclDouble, ncclSum, c->comm, st);
    if (r == ncclSuccess) r = rccl().AllReduce(c->d_flag, c->d_flag, 1, ncclDouble, ncclSum, c->comm, st);
    const ncclResult_t r2 = rccl().GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) {
        const std::string what = rccl().GetErrorString(r);
        comm_abort(c);
        return fail(IS3D_ENODEVICE, "ncclAllReduce failed on rank %d of %d: %s; the communicator was aborted", c->rank, c->n_ranks, what.c_str());
    }
    HIP_TRY(hipEventRecord(c->ev1, st));
    c->timed = true;
    c->pending++;
    hipLaunchKernelGGL(cf_comm_flag, dim3(1), dim3(1), 0, st, c->d_flag, 1, 0.0);
    HIP_TRY(hipGetLastError());
    c->flag_in_flight = true;
    return IS3D_OK;
}

// Host-side wait for work that sits behind a collective -- the stream (ev == nullptr) or one event -- WITH A DEADLINE.  RCCL enqueues an
// all-reduce and returns at once; if a peer never joins (it crashed, or it aborted its own communicator: a local ncclCommAbort does not
// unblock the other ranks' kernels), the collective's kernel spins on the device and hipStreamSynchronize would block this process for
// ever.  So the wait polls hipStreamQuery / hipEventQuery together with ncclCommGetAsyncError and, on an asynchronous RCCL error or after
// timeout_s, aborts THIS rank's communicator (which makes its own collective kernel exit) and returns IS3D_ENODEVICE -- the caller is
// expected to exit non-zero so that the launcher tears the job down.
int comm_wait(is3d_comm *c, hipStream_t st, hipEvent_t ev, const char *what)
{
    // What the deadline times (round 5).  The wait sits behind everything queued on the stream -- this rank's own kernels first, then the
    // collective.  With exactly ONE collective in flight since the last completed wait, ev0 (recorded just before it) is unambiguous: the clock
    // starts when ev0 completes, so only time spent in or behind the collective counts and a long local execute (a large surface in several
    // passes, culling off) cannot be taken for a dead peer.  With several collectives queued ev0 is the LAST one's and would never complete
    // behind an earlier one that hangs: then, and as a backstop while ev0 has not completed, the clock runs from entry against
    // kQueuedFactor x timeout_s -- the caller's timeout must exceed 1/kQueuedFactor of the longest compute it queues ahead of a wait
    // (include/is3d_amd.h, is3d_comm_set_timeout).
    constexpr double kQueuedFactor = 20.0;
    const auto t_entry = std::chrono::steady_clock::now();
    auto t0 = t_entry;
    bool started = !(c->pending == 1 && c->timed && c->ev0);
    int spins = 0;
    for (;;) {
        const hipError_t q = ev ? hipEventQuery(ev) : hipStreamQuery(st);
        if (q == hipSuccess) { c->pending = 0; return IS3D_OK; }
        if (q != hipErrorNotReady) {
            (void)hipGetLastError();
            comm_abort(c);
            return fail(IS3D_ENODEVICE, "%s: %s while waiting behind a collective on rank %d of %d; the communicator was aborted", what, hipGetErrorString(q),
                        c->rank, c->n_ranks);
        }
        (void)hipGetLastError();   // hipErrorNotReady is sticky in hipGetLastError otherwise
        if (!started) {
            if (hipEventQuery(c->ev0) != hipErrorNotReady) { started = true; t0 = std::chrono::steady_clock::now(); }   // the collective has been reached
            (void)hipGetLastError();
        }
        if (c->comm && rccl().CommGetAsyncError) {
            ncclResult_t ar = ncclSuccess;
            if (rccl().CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                const std::string w = rccl().GetErrorString(ar);
                comm_abort(c);
                return fail(IS3D_ENODEVICE, "%s: RCCL reported an asynchronous error on rank %d of %d (%s); the communicator was aborted", what, c->rank,
                            c->n_ranks, w.c_str());
            }
        }
        const auto now = std::chrono::steady_clock::now();
        const double waited = std::chrono::duration<double>(now - t0).count(), since_entry = std::chrono::duration<double>(now - t_entry).count();
        if (started ? waited > c->timeout_s : since_entry > kQueuedFactor * c->timeout_s) {
            comm_abort(c);
            return fail(IS3D_ENODEVICE, "%s: rank %d of %d waited %.3g s %s (a peer that never joined?); the communicator was aborted -- "
                        "this rank should exit so that the launcher ends the job", what, c->rank, c->n_ranks, started ? waited : since_entry,
                        started ? "in or behind a collective" : "for the work queued ahead of a collective (more than 20 x the communicator's timeout)");
        }
        if (++spins < 2000) std::this_thread::yield();                              // the usual case: a few hundred microseconds
        else std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 50 : 500));
    }
}

// running total of the error words since the last read; waits for the stream (with the communicator's deadline)
int comm_read_errors(is3d_comm *c, hipStream_t st, double *total)
{
    *total = 0.0;
    if (!c->flag_in_flight) return IS3D_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->h_flag, c->d_flag + 3, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(c->d_flag + 3, 0, sizeof(double), st));
    if (int rc = comm_wait(c, st, nullptr, "is3d_comm_check")) return rc;
    *total = c->h_flag[0];
    c->flag_in_flight = false;
    return IS3D_OK;
}
}  // namespace

extern "C" int is3d_comm_create(is3d_comm **out, const uint8_t id[IS3D_COMM_ID_BYTES], int32_t n_ranks, int32_t rank, int32_t device)
{
    if (!out || !id) return fail(IS3D_EINVAL, "null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(IS3D_EINVAL, "is3d_comm_create: rank %d of %d", rank, n_ranks);
    if (int rc = rccl_ready()) return rc;
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    is3d_comm *c = new is3d_comm;
    c->n_ranks = n_ranks; c->rank = rank; c->device = dev;
    ncclResult_t r = rccl().CommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(IS3D_ENODEVICE, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, n_ranks, dev, rccl().GetErrorString(r));
    }
    is3d::count_resource(1);
    const hipError_t e1 = hipMalloc((void **)&c->d_flag, 4 * sizeof(double));
    const hipError_t e2 = hipHostMalloc((void **)&c->h_flag, sizeof(double), hipHostMallocDefault);
    const hipError_t e3 = e1 == hipSuccess ? hipMemset(c->d_flag, 0, 4 * sizeof(double)) : e1;
    const hipError_t e4 = hipEventCreate(&c->ev0), e5 = hipEventCreate(&c->ev1);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
        is3d_comm_destroy(c);
        return fail(IS3D_ENODEVICE, "is3d_comm_create: cannot allocate the communicator's error word / events");
    }
    if (const char *t = getenv("IS3D_COMM_TIMEOUT_S")) {
        const double v = atof(t);
        if (v > 0.0) c->timeout_s = v;
    }
    *out = c;
    return IS3D_OK;
}

extern "C" int is3d_comm_set_timeout(is3d_comm *c, double seconds)
{
    if (!c || !(seconds > 0.0)) return fail(IS3D_EINVAL, "is3d_comm_set_timeout: null communicator or seconds <= 0");
    c->timeout_s = seconds;
    return IS3D_OK;
}

// wait for everything enqueued on hip_stream (kernels, collectives) with the communicator's deadline: what a host calls instead of
// hipStreamSynchronize behind an is3d_plan_execute_allreduce whose peers may have died
extern "C" int is3d_comm_synchronize(is3d_comm *c, void *hip_stream)
{
    if (!c) return fail(IS3D_EINVAL, "null communicator");
    if (c->aborted) return fail(IS3D_ENODEVICE, "the communicator was aborted");
    HIP_TRY(hipSetDevice(c->device));
    return comm_wait(c, (hipStream_t)hip_stream, nullptr, "is3d_comm_synchronize");
}

extern "C" int is3d_comm_rank(const is3d_comm *c, int32_t *rank, int32_t *n_ranks)
{
    if (!c) return fail(IS3D_EINVAL, "null communicator");
    if (rank) *rank = c->rank;
    if (n_ranks) *n_ranks = c->n_ranks;
    return IS3D_OK;
}

extern "C" int is3d_comm_allreduce(is3d_comm *c, double *dN_dev, int64_t n, void *hip_stream)
{
    if (!c || !dN_dev || n < 0) return fail(IS3D_EINVAL, "is3d_comm_allreduce: bad argument");
    if (c->aborted) return fail(IS3D_ENODEVICE, "is3d_comm_allreduce: the communicator was aborted");
    if (n == 0) return IS3D_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, (hipStream_t)hip_stream));
    NCCL_TRY(rccl().AllReduce(dN_dev, dN_dev, (size_t)n, ncclDouble, ncclSum, c->comm, (hipStream_t)hip_stream));
    HIP_TRY(hipEventRecord(c->ev1, (hipStream_t)hip_stream));
    c->timed = true;
    c->pending++;
    return IS3D_OK;
}

extern "C" int is3d_comm_abort(is3d_comm *c)
{
    if (!c) return fail(IS3D_EINVAL, "null communicator");
    if (!c->aborted) comm_abort(c);
    return IS3D_OK;
}

extern "C" int is3d_comm_check(is3d_comm *c, void *hip_stream, int32_t *n_failed)
{
    if (!c) return fail(IS3D_EINVAL, "null communicator");
    if (n_failed) *n_failed = 0;
    if (c->aborted) return fail(IS3D_ENODEVICE, "the communicator was aborted");
    double total = 0.0;
    if (int rc = comm_read_errors(c, (hipStream_t)hip_stream, &total)) return rc;
    c->peer_errors += (int64_t)(total + 0.5);
    const int64_t k = c->peer_errors;
    c->peer_errors = 0;
    if (n_failed) *n_failed = (int32_t)std::min<int64_t>(k, 0x7fffffff);
    if (k > 0) return fail(IS3D_EPEER, "%lld rank-executes since the last check reported an error before their all-reduce: the summed spectra are incomplete", (long long)k);
    return IS3D_OK;
}

extern "C" int is3d_comm_timings(is3d_comm *c, double *ms_allreduce)
{
    if (!c || !ms_allreduce) return fail(IS3D_EINVAL, "null argument");
    *ms_allreduce = 0.0;
    if (!c->timed || c->aborted) return IS3D_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = comm_wait(c, nullptr, c->ev1, "is3d_comm_timings")) return rc;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_allreduce = ms;
    return IS3D_OK;
}

extern "C" void is3d_comm_destroy(is3d_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && rccl().handle) (void)rccl().CommDestroy(c->comm);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

extern "C" int is3d_plan_execute_allreduce(is3d_plan *plan, const is3d_cells *shard, double *dN_out, is3d_comm *comm, void *hip_stream,
                                           is3d_status *status)
{
    if (!comm) return is3d_plan_execute(plan, shard, dN_out, hip_stream, status);
    if (comm->aborted) return fail(IS3D_ENODEVICE, "is3d_plan_execute_allreduce: the communicator was aborted");
    hipStream_t st = (hipStream_t)hip_stream;
    // A rank should never leave its peers waiting in ncclAllReduce.  Whatever happens locally, it joins the collective if it can -- with its
    // error word set, so that EVERY rank learns the sum is incomplete.  When it cannot (no buffer to reduce, a HIP error: the stream may be
    // dead) it aborts its own communicator and returns an error; its host is expected to exit non-zero so that the launcher ends the job.
    // The PEERS are protected by their own deadline, not by this rank's abort: every host-side wait behind a collective (comm_wait) polls
    // the stream together with ncclCommGetAsyncError and gives up after the communicator's timeout.
    if (!plan || !dN_out) {
        comm_abort(comm);
        return fail(IS3D_EINVAL, "is3d_plan_execute_allreduce: null plan or spectrum; the communicator was aborted so that the other ranks do not wait");
    }
    if (is3d::plan_device(plan) != comm->device) {   // the spectrum would live on another device than the communicator's rank: it cannot be reduced
        const int pd = is3d::plan_device(plan), cd = comm->device;
        comm_abort(comm);
        return fail(IS3D_EINVAL, "is3d_plan_execute_allreduce: the plan is on device %d, the communicator on device %d; the communicator was aborted so that "
                    "the other ranks do not wait", pd, cd);
    }
    int rc;
    if (is3d::plan_accumulate(plan))   // the old contents of dN_out would be summed n_ranks times
        rc = fail(IS3D_EINVAL, "is3d_plan_execute_allreduce needs a plan with opts.accumulate = 0");
    else
        // the status read-back of is3d_plan_execute synchronises the stream: the collective is enqueued after it, so a rank whose
        // shard has a domain error still takes part in the all-reduce and reports the error afterwards
        rc = is3d_plan_execute(plan, shard, dN_out, hip_stream, status);
    const std::string kept = rc ? is3d_last_error() : "";
    const int64_t nout = is3d_plan_output_size(plan);
    if (rc == IS3D_ENODEVICE) {
        comm_abort(comm);
        return fail(rc, "%s; the communicator was aborted so that the other ranks do not wait", kept.c_str());
    }
    if (hipSetDevice(comm->device) != hipSuccess ||
        (rc != IS3D_OK && rc != IS3D_EDOMAIN && hipMemsetAsync(dN_out, 0, sizeof(double) * (size_t)nout, st) != hipSuccess)) {
        comm_abort(comm);   // argument error and the neutral contribution cannot be enqueued either
        return fail(rc ? rc : IS3D_ENODEVICE, "%s; the communicator was aborted so that the other ranks do not wait", kept.c_str());
    }
    if (int rc2 = comm_allreduce_flagged(comm, dN_out, nout, rc != IS3D_OK, st)) {
        if (!comm->aborted) comm_abort(comm);
        return rc2;
    }
    if (rc) return fail(rc, "%s", kept.c_str());
    if (status) {   // the caller asked for a synchronous answer: include the other ranks'
        double total = 0.0;
        if (int rc3 = comm_read_errors(comm, st, &total)) return rc3;
        if (total > 0.5) {
            status->code = IS3D_EPEER;
            return fail(IS3D_EPEER, "%d of the %d ranks reported an error before the all-reduce: the summed spectrum is incomplete", (int)(total + 0.5), comm->n_ranks);
        }
    }
    return IS3D_OK;
}

// ------------------------------------------------------------------------------------------------
// one process, several devices
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kCellArrays = 23;

// everything a shard owns for the life of a multi-device plan
struct Shard {
    int device = 0;
    int64_t cap = 0;                   // cells this shard's plan, device block and staging block are sized for
    int64_t lo = 0, hi = 0;            // its cells in the current execute
    is3d_plan *plan = nullptr;
    double *d_cells = nullptr, *d_out = nullptr, *d_tmp = nullptr;   // d_tmp: the partner's spectrum in a round of the tree sum
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e_sum = nullptr;
    is3d_status st{};
    int rc = IS3D_OK;
    std::string err;
};

void shard_release(Shard &s)
{
    (void)hipSetDevice(s.device);
    if (s.plan) is3d_plan_destroy(s.plan);
    if (s.d_cells) (void)hipFree(s.d_cells);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.d_tmp) (void)hipFree(s.d_tmp);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    for (hipEvent_t e : {s.e0, s.e1, s.e_sum})
        if (e) (void)hipEventDestroy(e);
    s = Shard{};
}

// communicators of IS3D_REDUCE_RCCL are kept for the life of the process, keyed by the device list (creating one costs ~0.1-1 s)
struct CommSet { std::vector<int> devices; std::vector<ncclComm_t> comms; };
std::mutex g_commset_mutex;
std::vector<CommSet *> g_commsets;

int commset_for(const std::vector<int> &devs, CommSet **out)
{
    if (int rc = rccl_ready()) return rc;
    std::vector<int> sorted = devs;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
        return fail(IS3D_EINVAL, "IS3D_REDUCE_RCCL needs distinct devices (a communicator holds one rank per GPU); use IS3D_REDUCE_ORDERED");
    std::lock_guard<std::mutex> lock(g_commset_mutex);
    for (CommSet *c : g_commsets)
        if (c->devices == devs) { *out = c; return IS3D_OK; }
    CommSet *n = new CommSet;
    n->devices = devs;
    n->comms.resize(devs.size());
    ncclResult_t r = rccl().CommInitAll(n->comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) {
        delete n;
        return fail(IS3D_ENODEVICE, "ncclCommInitAll over %d devices failed: %s", (int)devs.size(), rccl().GetErrorString(r));
    }
    g_commsets.push_back(n);
    *out = n;
    return IS3D_OK;
}

}  // namespace

struct is3d_multi_plan {
    std::vector<Shard> sh;
    int reduce = IS3D_REDUCE_ORDERED;
    int64_t max_cells = 0, nout = 0;
    bool diff = false, accumulate = false;
    int n_arrays = 18;                 // cell arrays a shard uploads (23 with baryon diffusion)
    CommSet *comms = nullptr;
    std::vector<double> h_acc;         // accumulate: the device sum lands here first

    ~is3d_multi_plan()
    {
        for (auto &s : sh) shard_release(s);
    }
};

namespace {

int shard_create(Shard &s, const is3d_species *sp, const is3d_grid *grid, const is3d_df_tables *df, const is3d_feqmod_tables *fq,
                 const is3d_options *opts, bool need_tmp)
{
    HIP_TRY(hipSetDevice(s.device));
    is3d_options o = *opts;
    o.device = s.device;
    o.accumulate = 0;
    int rc = fq ? is3d_plan_create_feqmod(&s.plan, sp, grid, df, fq, &o, s.cap) : is3d_plan_create(&s.plan, sp, grid, df, &o, s.cap);
    if (rc) return rc;
    (void)is3d_plan_set_timing(s.plan, 1);
    const int64_t nout = is3d_plan_output_size(s.plan);
    HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void **)&s.d_cells, sizeof(double) * kCellArrays * (size_t)s.cap));
    HIP_TRY(hipMalloc((void **)&s.d_out, sizeof(double) * (size_t)nout));
    if (need_tmp) HIP_TRY(hipMalloc((void **)&s.d_tmp, sizeof(double) * (size_t)nout));
    for (int k = 0; k < (need_tmp ? 3 : 2); k++) is3d::count_resource(1);
    HIP_TRY(hipEventCreate(&s.e0));
    HIP_TRY(hipEventCreate(&s.e1));
    HIP_TRY(hipEventCreateWithFlags(&s.e_sum, hipEventDisableTiming));
    return IS3D_OK;
}

// upload the shard's slices, run the plan; the spectrum stays on the device (s.d_out), the stream is synchronised (is3d_plan_execute
// reads the status back).  The uploads go straight from the caller's pageable arrays: the runtime pins them in place and reaches
// 40-53 GB/s (144 MB in 2.7-3.6 ms); a pinned staging block filled by memcpy was measured slower (18 MB: 3.6 ms for the memcpy alone)
int shard_run(Shard &s, const is3d_cells *cells, bool diff)
{
    HIP_TRY(hipSetDevice(s.device));
    const int64_t n = s.hi - s.lo;
    const double *src[kCellArrays] = {cells->tau, cells->eta, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un,
                                      cells->T, cells->P, cells->E, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn, cells->bulkPi,
                                      diff ? cells->muB : nullptr, diff ? cells->nB : nullptr, diff ? cells->Vx : nullptr,
                                      diff ? cells->Vy : nullptr, diff ? cells->Vn : nullptr};
    const double *dptr[kCellArrays];
    HIP_TRY(hipEventRecord(s.e0, s.stream));
    for (int a = 0; a < kCellArrays; a++) {
        dptr[a] = nullptr;
        if (src[a] && n > 0) {
            HIP_TRY(hipMemcpyAsync(s.d_cells + (size_t)a * n, src[a] + s.lo, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s.stream));
            dptr[a] = s.d_cells + (size_t)a * n;
        }
    }
    HIP_TRY(hipEventRecord(s.e1, s.stream));
    is3d_cells dc{};
    dc.n_cells = n;
    dc.tau = dptr[0]; dc.eta = dptr[1]; dc.dat = dptr[2]; dc.dax = dptr[3]; dc.day = dptr[4]; dc.dan = dptr[5];
    dc.ux = dptr[6]; dc.uy = dptr[7]; dc.un = dptr[8]; dc.T = dptr[9]; dc.P = dptr[10]; dc.E = dptr[11];
    dc.pixx = dptr[12]; dc.pixy = dptr[13]; dc.pixn = dptr[14]; dc.piyy = dptr[15]; dc.piyn = dptr[16]; dc.bulkPi = dptr[17];
    dc.muB = dptr[18]; dc.nB = dptr[19]; dc.Vx = dptr[20]; dc.Vy = dptr[21]; dc.Vn = dptr[22];
    const int rc = is3d_plan_execute(s.plan, &dc, s.d_out, s.stream, &s.st);
    if (rc) return rc;
    is3d_status t{};
    (void)is3d_plan_timings(s.plan, &t);
    s.st.ms_prep = t.ms_prep; s.st.ms_main = t.ms_main; s.st.ms_finalize = t.ms_finalize;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s.e0, s.e1));
    s.st.ms_h2d = ms;
    return IS3D_OK;
}

int rccl_allreduce_shards(std::vector<Shard> &sh, CommSet *cs, int64_t nout)
{
    NCCL_TRY(rccl().GroupStart());
    for (size_t i = 0; i < sh.size(); i++) {
        (void)hipSetDevice(sh[i].device);
        ncclResult_t r = rccl().AllReduce(sh[i].d_out, sh[i].d_out, (size_t)nout, ncclDouble, ncclSum, cs->comms[i], sh[i].stream);
        if (r != ncclSuccess) {
            (void)rccl().GroupEnd();
            return fail(IS3D_ENODEVICE, "ncclAllReduce failed: %s", rccl().GetErrorString(r));
        }
    }
    NCCL_TRY(rccl().GroupEnd());
    for (auto &s : sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipStreamSynchronize(s.stream));
    }
    return IS3D_OK;
}

// Pairwise tree in a fixed order: round r adds shard i + 2^r into shard i for every i that is a multiple of 2^(r+1); the pairs of a
// round run concurrently on the streams of their receiving shards, a receiver waits for its partner's previous round through an event.
// The sum ends in shard 0; the order of the additions depends on the shard count only.
int tree_sum_shards(std::vector<Shard> &sh, int64_t nout)
{
    const size_t n = sh.size();
    for (size_t stride = 1; stride < n; stride *= 2) {
        for (size_t i = 0; i + stride < n; i += 2 * stride) {
            Shard &dst = sh[i], &src = sh[i + stride];
            HIP_TRY(hipSetDevice(dst.device));
            if (stride > 1) HIP_TRY(hipStreamWaitEvent(dst.stream, src.e_sum, 0));   // round 0: every shard's stream is already synchronised
            const double *from = src.d_out;
            if (src.device != dst.device) {
                HIP_TRY(hipMemcpyPeerAsync(dst.d_tmp, dst.device, src.d_out, src.device, sizeof(double) * (size_t)nout, dst.stream));
                from = dst.d_tmp;
            }
            HIP_TRY(launch_add_spectrum(dst.d_out, from, nout, dst.stream));
            HIP_TRY(hipEventRecord(dst.e_sum, dst.stream));
        }
    }
    HIP_TRY(hipSetDevice(sh[0].device));
    HIP_TRY(hipStreamSynchronize(sh[0].stream));
    return IS3D_OK;
}

int check_devices(const int32_t *devices, int32_t &n_devices, std::vector<int> &dev)
{
    const int visible = is3d_device_count();
    if (visible < 1) return fail(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");
    if (n_devices <= 0) { n_devices = visible; devices = nullptr; }
    if (n_devices > 1024) return fail(IS3D_EINVAL, "n_devices = %d", n_devices);
    dev.resize(n_devices);
    for (int i = 0; i < n_devices; i++) {
        dev[i] = devices ? devices[i] : i;
        if (dev[i] < 0 || dev[i] >= visible) return fail(IS3D_EINVAL, "device %d is not one of the %d visible HIP devices", dev[i], visible);
    }
    return IS3D_OK;
}

}  // namespace

extern "C" int is3d_multi_plan_create(is3d_multi_plan **out, const is3d_species *species, const is3d_grid *grid, const is3d_df_tables *df,
                                      const is3d_feqmod_tables *fq, const is3d_options *opts, const int32_t *devices, int32_t n_devices,
                                      int32_t reduce, int64_t max_cells)
{
    if (!out || !opts) return fail(IS3D_EINVAL, "null argument");
    *out = nullptr;
    if (reduce != IS3D_REDUCE_ORDERED && reduce != IS3D_REDUCE_RCCL) return fail(IS3D_EINVAL, "reduce must be IS3D_REDUCE_ORDERED or IS3D_REDUCE_RCCL");
    if (max_cells < 0) return fail(IS3D_EINVAL, "max_cells < 0");
    std::vector<int> dev;
    if (int rc = check_devices(devices, n_devices, dev)) return rc;
    std::unique_ptr<is3d_multi_plan> M(new is3d_multi_plan);
    M->reduce = reduce;
    M->max_cells = max_cells;
    M->accumulate = opts->accumulate != 0;
    M->diff = opts->include_baryon && opts->include_baryondiff_deltaf;
    M->sh.resize(n_devices);
    if (reduce == IS3D_REDUCE_RCCL && n_devices > 1)
        if (int rc = commset_for(dev, &M->comms)) return rc;
    for (int i = 0; i < n_devices; i++) {
        M->sh[i].device = dev[i];
        M->sh[i].cap = std::max<int64_t>((max_cells + n_devices - 1) / n_devices, 1);
        M->sh[i].st.bad_cell = -1;
    }
    // a shard needs a receive buffer if it takes a partner from another device in some round of the tree
    std::vector<char> need_tmp(n_devices, 0);
    if (reduce == IS3D_REDUCE_ORDERED)
        for (int stride = 1; stride < n_devices; stride *= 2)
            for (int i = 0; i + stride < n_devices; i += 2 * stride)
                if (dev[i] != dev[i + stride]) need_tmp[i] = 1;
    {
        std::vector<std::thread> th;
        for (int i = 0; i < n_devices; i++)
            th.emplace_back([&, i] {
                Shard &s = M->sh[i];
                s.rc = shard_create(s, species, grid, df, fq, opts, need_tmp[i] != 0);
                if (s.rc) s.err = is3d_last_error();
            });
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < n_devices; i++)
        if (M->sh[i].rc) return fail(M->sh[i].rc, "shard %d (device %d): %s", i, M->sh[i].device, M->sh[i].err.c_str());
    M->nout = is3d_plan_output_size(M->sh[0].plan);
    *out = M.release();
    return IS3D_OK;
}

extern "C" int32_t is3d_multi_plan_shards(const is3d_multi_plan *M) { return M ? (int32_t)M->sh.size() : 0; }
extern "C" int64_t is3d_multi_plan_output_size(const is3d_multi_plan *M) { return M ? M->nout : 0; }
extern "C" void is3d_multi_plan_destroy(is3d_multi_plan *M) { delete M; }

extern "C" int is3d_multi_plan_execute(is3d_multi_plan *M, const is3d_cells *cells, double *dN_out, is3d_status *status,
                                       is3d_status *shard_status)
{
    if (!M || !cells || !dN_out) return fail(IS3D_EINVAL, "null argument");
    const int n_devices = (int)M->sh.size();
    if (status) { memset(status, 0, sizeof *status); status->bad_cell = -1; }
    if (shard_status) memset(shard_status, 0, sizeof(is3d_status) * (size_t)n_devices);
    if (cells->n_cells < 0 || cells->n_cells > M->max_cells)
        return fail(IS3D_EINVAL, "n_cells = %lld outside the multi-device plan's max_cells = %lld", (long long)cells->n_cells, (long long)M->max_cells);
    std::vector<Shard> &sh = M->sh;
    for (int i = 0; i < n_devices; i++) {
        (void)is3d_shard_bounds(cells->n_cells, i, n_devices, &sh[i].lo, &sh[i].hi);
        sh[i].st = is3d_status{};
        sh[i].st.bad_cell = -1;
        sh[i].rc = IS3D_OK;
        sh[i].err.clear();
    }
    if (n_devices == 1) {
        sh[0].rc = shard_run(sh[0], cells, M->diff);
        if (sh[0].rc) sh[0].err = is3d_last_error();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < n_devices; i++)
            th.emplace_back([&, i] {
                sh[i].rc = shard_run(sh[i], cells, M->diff);
                if (sh[i].rc) sh[i].err = is3d_last_error();   // the error text is thread-local
            });
        for (auto &t : th) t.join();
    }
    // aggregate (also on failure, so that the caller sees which cell was bad)
    int rc_first = IS3D_OK;
    std::string err_first;
    is3d_status agg{};
    agg.bad_cell = -1;
    for (int i = 0; i < n_devices; i++) {
        const is3d_status &t = sh[i].st;
        if (shard_status) { shard_status[i] = t; shard_status[i].code = sh[i].rc; }
        if (sh[i].rc && !rc_first) { rc_first = sh[i].rc; err_first = "shard " + std::to_string(i) + " (device " + std::to_string(sh[i].device) + "): " + sh[i].err; }
        agg.n_classes = std::max(agg.n_classes, t.n_classes);
        agg.n_cells_skipped += t.n_cells_skipped;
        agg.n_passes = std::max(agg.n_passes, t.n_passes);
        agg.kernel_variant = t.kernel_variant ? t.kernel_variant : agg.kernel_variant;
        agg.ms_prep = std::max(agg.ms_prep, t.ms_prep);
        agg.ms_main = std::max(agg.ms_main, t.ms_main);
        agg.ms_finalize = std::max(agg.ms_finalize, t.ms_finalize);
        agg.ms_h2d = std::max(agg.ms_h2d, t.ms_h2d);
        agg.n_wave_rows += t.n_wave_rows;
        agg.n_wave_rows_culled += t.n_wave_rows_culled;
        agg.n_cells_breakdown += t.n_cells_breakdown;
        agg.n_cells_narrow += t.n_cells_narrow;
        if (t.bad_cell >= 0 && (agg.bad_cell < 0 || sh[i].lo + t.bad_cell < agg.bad_cell)) agg.bad_cell = sh[i].lo + t.bad_cell;
    }
    agg.code = rc_first;
    if (rc_first) {
        if (status) *status = agg;
        return fail(rc_first, "%s", err_first.c_str());
    }
    const int64_t nout = M->nout;
    HIP_TRY(hipSetDevice(sh[0].device));
    HIP_TRY(hipEventRecord(sh[0].e0, sh[0].stream));
    int rc = IS3D_OK;
    if (n_devices > 1) rc = (M->reduce == IS3D_REDUCE_RCCL) ? rccl_allreduce_shards(sh, M->comms, nout) : tree_sum_shards(sh, nout);
    if (rc) { agg.code = rc; if (status) *status = agg; return rc; }
    HIP_TRY(hipSetDevice(sh[0].device));
    if (M->accumulate) {   // reference semantics: dN += result (smooth_kernels.cpp:375)
        M->h_acc.resize((size_t)nout);
        HIP_TRY(hipMemcpyAsync(M->h_acc.data(), sh[0].d_out, sizeof(double) * (size_t)nout, hipMemcpyDeviceToHost, sh[0].stream));
        HIP_TRY(hipStreamSynchronize(sh[0].stream));
        for (int64_t i = 0; i < nout; i++) dN_out[i] += M->h_acc[(size_t)i];
    } else {
        HIP_TRY(hipMemcpyAsync(dN_out, sh[0].d_out, sizeof(double) * (size_t)nout, hipMemcpyDeviceToHost, sh[0].stream));
    }
    HIP_TRY(hipEventRecord(sh[0].e1, sh[0].stream));
    HIP_TRY(hipEventSynchronize(sh[0].e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, sh[0].e0, sh[0].e1));
    agg.ms_d2h = ms;
    if (status) *status = agg;
    return IS3D_OK;
}

extern "C" int is3d_smooth_spectra_multi(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                                         const is3d_df_tables *df, const is3d_feqmod_tables *fq, const is3d_options *opts,
                                         const int32_t *devices, int32_t n_devices, int32_t reduce, double *dN_out,
                                         is3d_status *status, is3d_status *shard_status)
{
    if (!cells || !opts || !dN_out) return fail(IS3D_EINVAL, "null argument");
    if (reduce != IS3D_REDUCE_ORDERED && reduce != IS3D_REDUCE_RCCL) return fail(IS3D_EINVAL, "reduce must be IS3D_REDUCE_ORDERED or IS3D_REDUCE_RCCL");
    if (cells->n_cells < 0) return fail(IS3D_EINVAL, "n_cells < 0");
    std::vector<int> dev;
    if (int rc = check_devices(devices, n_devices, dev)) return rc;
    if (status) { memset(status, 0, sizeof *status); status->bad_cell = -1; }
    if (shard_status) memset(shard_status, 0, sizeof(is3d_status) * (size_t)n_devices);

    if (n_devices == 1 && reduce == IS3D_REDUCE_ORDERED) {
        is3d_options o = *opts;
        o.device = dev[0];
        is3d_status st{};
        const int rc = fq ? is3d_smooth_spectra_feqmod(cells, species, grid, df, fq, &o, dN_out, &st)
                          : is3d_smooth_spectra(cells, species, grid, df, &o, dN_out, &st);
        if (status) *status = st;
        if (shard_status) shard_status[0] = st;
        return rc;
    }
    // the one-shot form: a multi-device plan created, executed once and destroyed (hosts that call more than once keep the plan)
    std::vector<int32_t> dev32(dev.begin(), dev.end());
    is3d_multi_plan *M = nullptr;
    if (int rc = is3d_multi_plan_create(&M, species, grid, df, fq, opts, dev32.data(), n_devices, reduce, cells->n_cells)) return rc;
    const int rc = is3d_multi_plan_execute(M, cells, dN_out, status, shard_status);
    const std::string kept = rc ? is3d_last_error() : "";
    is3d_multi_plan_destroy(M);
    if (rc) return fail(rc, "%s", kept.c_str());
    return IS3D_OK;
}

// ------------------------------------------------------------------------------------------------
// particle sampler over several devices
// ------------------------------------------------------------------------------------------------
extern "C" int is3d_sample_particles_multi(const is3d_cells *cells, const is3d_species *species, const is3d_df_tables *df,
                                           const is3d_sampler_inputs *in, const is3d_options *opts, const int32_t *devices,
                                           int32_t n_devices, is3d_particle *particles, int64_t capacity, int64_t *n_particles,
                                           is3d_sampler_stats *stats)
{
    if (!cells || !in || !opts || !n_particles) return fail(IS3D_EINVAL, "null argument");
    *n_particles = 0;
    if (stats) memset(stats, 0, sizeof *stats);
    if (cells->n_cells < 0) return fail(IS3D_EINVAL, "n_cells < 0");
    const int visible = is3d_device_count();
    if (visible < 1) return fail(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");
    if (n_devices <= 0) { n_devices = visible; devices = nullptr; }
    if (n_devices > 1024) return fail(IS3D_EINVAL, "n_devices = %d", n_devices);
    std::vector<int> dev(n_devices);
    for (int i = 0; i < n_devices; i++) {
        dev[i] = devices ? devices[i] : i;
        if (dev[i] < 0 || dev[i] >= visible) return fail(IS3D_EINVAL, "device %d is not one of the %d visible HIP devices", dev[i], visible);
    }
    if (particles == nullptr) capacity = 0;
    if (n_devices == 1) {
        is3d_options o = *opts;
        o.device = dev[0];
        return is3d_sample_particles(cells, species, df, in, &o, particles, capacity, n_particles, stats);
    }
    struct SShard {
        int device = 0;
        int64_t lo = 0, hi = 0, count = 0;
        std::vector<is3d_particle> list;
        is3d_sampler_stats st{};
        int rc = IS3D_OK;
        std::string err;
    };
    std::vector<SShard> sh(n_devices);
    const bool fill = capacity > 0;
    {
        std::vector<std::thread> th;
        for (int i = 0; i < n_devices; i++) {
            sh[i].device = dev[i];
            (void)is3d_shard_bounds(cells->n_cells, i, n_devices, &sh[i].lo, &sh[i].hi);
            th.emplace_back([&, i] {
                SShard &s = sh[i];
                is3d_cells c = *cells;
                c.n_cells = s.hi - s.lo;
                const double **fields[] = {&c.tau, &c.eta, &c.dat, &c.dax, &c.day, &c.dan, &c.ux, &c.uy, &c.un, &c.T, &c.P, &c.E, &c.pixx, &c.pixy,
                                           &c.pixn, &c.piyy, &c.piyn, &c.bulkPi, &c.muB, &c.nB, &c.Vx, &c.Vy, &c.Vn};
                for (auto f : fields)
                    if (*f) *f += s.lo;
                is3d_sampler_inputs si = *in;
                si.first_cell = in->first_cell + s.lo;
                if (si.x) si.x += s.lo;
                if (si.y) si.y += s.lo;
                is3d_options o = *opts;
                o.device = s.device;
                s.rc = is3d_sample_particles(&c, species, df, &si, &o, nullptr, 0, &s.count, &s.st);
                if (!s.rc && fill && s.count > 0) {
                    s.list.resize((size_t)s.count);
                    s.rc = is3d_sample_particles(&c, species, df, &si, &o, s.list.data(), s.count, &s.count, &s.st);
                }
                if (s.rc) s.err = is3d_last_error();
            });
        }
        for (auto &t : th) t.join();
    }
    int64_t total = 0;
    is3d_sampler_stats agg{};
    for (int i = 0; i < n_devices; i++) {
        if (sh[i].rc) return fail(sh[i].rc, "shard %d (device %d): %s", i, sh[i].device, sh[i].err.c_str());
        total += sh[i].count;
        const is3d_sampler_stats &t = sh[i].st;
        agg.n_cells_skipped += t.n_cells_skipped; agg.n_hadrons_drawn += t.n_hadrons_drawn;
        agg.n_momentum_samples += t.n_momentum_samples; agg.n_acceptances += t.n_acceptances;
        agg.n_classes = std::max(agg.n_classes, t.n_classes); agg.n_cells_breakdown += t.n_cells_breakdown;
        agg.ms_h2d = std::max(agg.ms_h2d, t.ms_h2d); agg.ms_prep = std::max(agg.ms_prep, t.ms_prep);
        agg.ms_count = std::max(agg.ms_count, t.ms_count); agg.ms_fill = std::max(agg.ms_fill, t.ms_fill);
    }
    *n_particles = total;
    if (stats) *stats = agg;
    if (!fill) return IS3D_OK;
    // merge: every shard list is ordered by (event, cell, draw) and the shards are ascending cell ranges, so the single-device order is,
    // event by event, shard 0's hadrons of that event, then shard 1's, ...
    std::vector<size_t> pos(n_devices, 0);
    int64_t out = 0;
    for (int32_t ev = 0; ev < in->n_events; ev++)
        for (int i = 0; i < n_devices; i++) {
            const std::vector<is3d_particle> &l = sh[i].list;
            size_t p = pos[i];
            while (p < l.size() && l[p].event == ev) {
                if (out < capacity) particles[out] = l[p];
                out++;
                p++;
            }
            pos[i] = p;
        }
    if (total > capacity) return fail(IS3D_ENOMEM, "particle buffer too small: %lld particles, capacity %lld", (long long)total, (long long)capacity);
    return IS3D_OK;
}
