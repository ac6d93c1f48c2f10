// cf_launch.h -- launch entry points of cf_kernels.hip (HIP translation units only).
#pragma once
#include "cf_device.h"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstdint>
namespace is3d {
constexpr int kWavesPerBlock = 4;
// Cap on the derived streams of one pass over the cell axis, shared by every plan (cf_plan.cpp, cf_vah.hip): the caller's opts.workspace_bytes, else
// max(16 GiB, 45 % of the device's TOTAL memory) -- the total, not what happens to be free: the pass count, the chunk count and with them the summation
// order of a surface that needs several passes must not depend on what else occupies the GPU at the moment -- and never more than the device can hold
// beside the partial slab (<= kPartialCapBytes by default) and the fixed buffers: streams + slab <= 90 % of the total.
constexpr int64_t kPartialCapBytes = (int64_t)12 << 30;
inline int64_t default_stream_cap_bytes(int64_t requested)
{
    if (requested > 0) return requested;
    int64_t ws = (int64_t)16 << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        ws = std::max<int64_t>(ws, (int64_t)((double)total_b * 0.45));
        ws = std::min<int64_t>(ws, std::max<int64_t>((int64_t)1 << 30, (int64_t)((double)total_b * 0.9) - kPartialCapBytes));
    }
    return ws;
}
size_t prep_lds_bytes(int nT, int nspl, int J, int K, int baryon, int rec, int dim3);
hipError_t launch_prep(const PrepParams &p, hipStream_t stream);
void main_tile_shape(int variant, int dim3, int *JT, int *KT);
hipError_t launch_main(int variant, int ce, int dim3, int outflow, int reg, const MainArgs &a, hipStream_t st);  // a.g.baryon selects the B kernels
hipError_t launch_cull_floor(const double *partial, int nA, int J, int K, int Kacc, int Lpad, int JT, int R, int jtiles, int ktiles,
                             const int32_t *lane_pe, double unscale, double *floor_out, hipStream_t stream);
hipError_t launch_finalize(double *partial /* chunk 0 receives the sum over chunks */, const int *cls, const double *degeneracy, double *out,
                           int64_t nout, int npart, int npT, int J, int Kacc, int Lpad, int nch, double prefactor,
                           int accumulate, const unsigned long long *pds_bound, hipStream_t stream, int split = 1, int Lbins = 0);
// *out (zeroed by the caller) = bits of a bound on |p.dsigma| over all lanes, bins and the given cells
hipError_t launch_pds_bound(const CellPtrs &cells, int64_t n_cells, int is_dim3, double kmin, double kmax, double gw2d, double mTmax,
                            double pTmax, unsigned long long *out, hipStream_t st);
const char *main_kernel_name(int variant);
int tile3e_units_per_batch(int JT, int R, int npT, int wpb, int baryon, int e2g = 0);   // variants 5, 6 (10: e2g = 1, records only): units per LDS batch
int tile3e_stream_slack_doubles(int JT, int R);                  // variant 5: over-read slack behind TS and TE
hipError_t launch_observables(const double *dN, const double *phi_w, const double *pT_w, const double *coskphi,
                              const double *sinkphi, double *dndy, double *spec2pi, double *vn, int npart, int npT, int J,
                              int ny, hipStream_t st);
// which: 0 exp_full | 1 exp_p9 | 2 exp_p9_sat | 3 exp_full_sat | 4 sqrt_g1 | 5 sqrt_nr | 6 rcp_nr1 | 7 rcp_nr  (cf_math.h), y[i] = f(x[i])
hipError_t launch_math_probe(int which, int64_t n, const double *x, double *y, hipStream_t st);
hipError_t launch_clock_probe(unsigned long long ref_ticks, unsigned long long *out /* 2 x 8 */, hipStream_t st);
hipError_t launch_fold_status(const unsigned long long *status /* [8] */, unsigned long long *sticky /* [2] */, hipStream_t st);
}  // namespace is3d
