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
#include <cstring>
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
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the host process has already mapped (e.g. the one PyTorch bundles) is reused, so that the process holds ONE RCCL
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
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
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
    *out = c;
    return IS3D_OK;
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
    if (n == 0) return IS3D_OK;
    HIP_TRY(hipSetDevice(c->device));
    NCCL_TRY(rccl().AllReduce(dN_dev, dN_dev, (size_t)n, ncclDouble, ncclSum, c->comm, (hipStream_t)hip_stream));
    return IS3D_OK;
}

extern "C" void is3d_comm_destroy(is3d_comm *c)
{
    if (!c) return;
    if (c->comm && rccl().handle) {
        (void)hipSetDevice(c->device);
        (void)rccl().CommDestroy(c->comm);
    }
    delete c;
}

extern "C" int is3d_plan_execute_allreduce(is3d_plan *plan, const is3d_cells *shard, double *dN_out, is3d_comm *comm, void *hip_stream,
                                           is3d_status *status)
{
    // the status read-back of is3d_plan_execute synchronises the stream: the collective is enqueued after it, so a rank whose
    // shard has a domain error still takes part in the all-reduce (no rank is left waiting) and reports the error afterwards
    const int rc = is3d_plan_execute(plan, shard, dN_out, hip_stream, status);
    if (rc != IS3D_OK && rc != IS3D_EDOMAIN) return rc;
    std::string kept = rc ? is3d_last_error() : "";
    if (comm) {
        const int rc2 = is3d_comm_allreduce(comm, dN_out, is3d_plan_output_size(plan), hip_stream);
        if (rc2) return rc2;
    }
    if (rc) return fail(rc, "%s", kept.c_str());
    return IS3D_OK;
}

// ------------------------------------------------------------------------------------------------
// one process, several devices
// ------------------------------------------------------------------------------------------------
namespace {

struct Shard {
    int device = 0;
    int64_t lo = 0, hi = 0;
    is3d_plan *plan = nullptr;
    double *d_cells = nullptr, *d_out = nullptr;
    hipStream_t stream = nullptr;
    is3d_status st{};
    int rc = IS3D_OK;
    std::string err;
    float ms_h2d = 0.f;
};

void shard_release(Shard &s)
{
    (void)hipSetDevice(s.device);
    if (s.plan) is3d_plan_destroy(s.plan);
    if (s.d_cells) (void)hipFree(s.d_cells);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s.plan = nullptr; s.d_cells = nullptr; s.d_out = nullptr; s.stream = nullptr;
}

// upload the shard's slices, run the plan; the spectrum stays on the device (s.d_out), the stream is synchronised
int shard_run(Shard &s, const is3d_cells *cells, const is3d_species *sp, const is3d_grid *grid, const is3d_df_tables *df,
              const is3d_feqmod_tables *fq, const is3d_options *opts)
{
    HIP_TRY(hipSetDevice(s.device));
    is3d_options o = *opts;
    o.device = s.device;
    o.accumulate = 0;
    const int64_t n = s.hi - s.lo;
    int rc = fq ? is3d_plan_create_feqmod(&s.plan, sp, grid, df, fq, &o, std::max<int64_t>(n, 1))
                : is3d_plan_create(&s.plan, sp, grid, df, &o, std::max<int64_t>(n, 1));
    if (rc) return rc;
    (void)is3d_plan_set_timing(s.plan, 1);
    const int64_t nout = is3d_plan_output_size(s.plan);
    HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void **)&s.d_cells, sizeof(double) * 23 * (size_t)std::max<int64_t>(n, 1)));
    HIP_TRY(hipMalloc((void **)&s.d_out, sizeof(double) * (size_t)nout));
    const bool diff = opts->include_baryon && opts->include_baryondiff_deltaf;
    const double *src[23] = {cells->tau, cells->eta, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un,
                             cells->T, cells->P, cells->E, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn, cells->bulkPi,
                             diff ? cells->muB : nullptr, diff ? cells->nB : nullptr, diff ? cells->Vx : nullptr,
                             diff ? cells->Vy : nullptr, diff ? cells->Vn : nullptr};
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
    HIP_TRY(hipEventRecord(e0, s.stream));
    const double *dptr[23];
    for (int a = 0; a < 23; a++) {
        dptr[a] = nullptr;
        if (src[a] && n > 0) {
            HIP_TRY(hipMemcpyAsync(s.d_cells + (size_t)a * n, src[a] + s.lo, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s.stream));
            dptr[a] = s.d_cells + (size_t)a * n;
        }
    }
    HIP_TRY(hipEventRecord(e1, s.stream));
    is3d_cells dc{};
    dc.n_cells = n;
    dc.tau = dptr[0]; dc.eta = dptr[1]; dc.dat = dptr[2]; dc.dax = dptr[3]; dc.day = dptr[4]; dc.dan = dptr[5];
    dc.ux = dptr[6]; dc.uy = dptr[7]; dc.un = dptr[8]; dc.T = dptr[9]; dc.P = dptr[10]; dc.E = dptr[11];
    dc.pixx = dptr[12]; dc.pixy = dptr[13]; dc.pixn = dptr[14]; dc.piyy = dptr[15]; dc.piyn = dptr[16]; dc.bulkPi = dptr[17];
    dc.muB = dptr[18]; dc.nB = dptr[19]; dc.Vx = dptr[20]; dc.Vy = dptr[21]; dc.Vn = dptr[22];
    rc = is3d_plan_execute(s.plan, &dc, s.d_out, s.stream, &s.st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s.stream));
    is3d_status t{};
    (void)is3d_plan_timings(s.plan, &t);
    s.st.ms_prep = t.ms_prep; s.st.ms_main = t.ms_main; s.st.ms_finalize = t.ms_finalize;
    HIP_TRY(hipEventElapsedTime(&s.ms_h2d, e0, e1));
    s.st.ms_h2d = s.ms_h2d;
    return IS3D_OK;
}

// communicators of IS3D_REDUCE_RCCL are kept for the life of the process, keyed by the device list (creating one costs ~0.1-1 s)
struct CommSet { std::vector<int> devices; std::vector<ncclComm_t> comms; };
std::mutex g_commset_mutex;
std::vector<CommSet> g_commsets;

int rccl_allreduce_shards(std::vector<Shard> &sh, int64_t nout)
{
    if (int rc = rccl_ready()) return rc;
    std::vector<int> devs;
    for (auto &s : sh) devs.push_back(s.device);
    std::vector<int> sorted = devs;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
        return fail(IS3D_EINVAL, "IS3D_REDUCE_RCCL needs distinct devices (a communicator holds one rank per GPU); use IS3D_REDUCE_ORDERED");
    std::lock_guard<std::mutex> lock(g_commset_mutex);
    CommSet *cs = nullptr;
    for (auto &c : g_commsets)
        if (c.devices == devs) cs = &c;
    if (!cs) {
        CommSet n;
        n.devices = devs;
        n.comms.resize(devs.size());
        NCCL_TRY(rccl().CommInitAll(n.comms.data(), (int)devs.size(), devs.data()));
        g_commsets.push_back(n);
        cs = &g_commsets.back();
    }
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

// ((s0 + s1) + s2) + ... on shard 0's device
int ordered_sum_shards(std::vector<Shard> &sh, int64_t nout)
{
    Shard &s0 = sh[0];
    HIP_TRY(hipSetDevice(s0.device));
    double *tmp = nullptr;
    bool need_tmp = false;
    for (size_t i = 1; i < sh.size(); i++) need_tmp |= sh[i].device != s0.device;
    if (need_tmp) HIP_TRY(hipMalloc((void **)&tmp, sizeof(double) * (size_t)nout));
    struct Guard { double *p; ~Guard() { if (p) (void)hipFree(p); } } guard{tmp};
    for (size_t i = 1; i < sh.size(); i++) {
        const double *src = sh[i].d_out;
        if (sh[i].device != s0.device) {
            HIP_TRY(hipMemcpyPeerAsync(tmp, s0.device, sh[i].d_out, sh[i].device, sizeof(double) * (size_t)nout, s0.stream));
            src = tmp;
        }
        HIP_TRY(launch_add_spectrum(s0.d_out, src, nout, s0.stream));
    }
    HIP_TRY(hipStreamSynchronize(s0.stream));
    return IS3D_OK;
}

}  // namespace

extern "C" int is3d_smooth_spectra_multi(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                                         const is3d_df_tables *df, const is3d_feqmod_tables *fq, const is3d_options *opts,
                                         const int32_t *devices, int32_t n_devices, int32_t reduce, double *dN_out,
                                         is3d_status *status, is3d_status *shard_status)
{
    if (!cells || !opts || !dN_out) return fail(IS3D_EINVAL, "null argument");
    if (reduce != IS3D_REDUCE_ORDERED && reduce != IS3D_REDUCE_RCCL) return fail(IS3D_EINVAL, "reduce must be IS3D_REDUCE_ORDERED or IS3D_REDUCE_RCCL");
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

    std::vector<Shard> sh(n_devices);
    for (int i = 0; i < n_devices; i++) {
        sh[i].device = dev[i];
        (void)is3d_shard_bounds(cells->n_cells, i, n_devices, &sh[i].lo, &sh[i].hi);
        sh[i].st.bad_cell = -1;
    }
    struct Release { std::vector<Shard> &v; ~Release() { for (auto &s : v) shard_release(s); } } release{sh};
    {
        std::vector<std::thread> th;
        for (int i = 0; i < n_devices; i++)
            th.emplace_back([&, i] {
                sh[i].rc = shard_run(sh[i], cells, species, grid, df, fq, opts);
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
    const int64_t nout = is3d_plan_output_size(sh[0].plan);
    hipEvent_t e0, e1;
    HIP_TRY(hipSetDevice(sh[0].device));
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
    HIP_TRY(hipEventRecord(e0, sh[0].stream));
    int rc = (reduce == IS3D_REDUCE_RCCL) ? rccl_allreduce_shards(sh, nout) : ordered_sum_shards(sh, nout);
    if (rc) { agg.code = rc; if (status) *status = agg; return rc; }
    HIP_TRY(hipSetDevice(sh[0].device));
    if (opts->accumulate) {   // reference semantics: dN += result (smooth_kernels.cpp:375)
        std::vector<double> h((size_t)nout);
        HIP_TRY(hipMemcpyAsync(h.data(), sh[0].d_out, sizeof(double) * (size_t)nout, hipMemcpyDeviceToHost, sh[0].stream));
        HIP_TRY(hipStreamSynchronize(sh[0].stream));
        for (int64_t i = 0; i < nout; i++) dN_out[i] += h[(size_t)i];
    } else {
        HIP_TRY(hipMemcpyAsync(dN_out, sh[0].d_out, sizeof(double) * (size_t)nout, hipMemcpyDeviceToHost, sh[0].stream));
    }
    HIP_TRY(hipEventRecord(e1, sh[0].stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    agg.ms_d2h = ms;
    if (status) *status = agg;
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
