// cf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the smooth Cooper-Frye path.
//
// Replaces the two hot loops of EmissionFunctionArray::calculate_dN_pTdpTdphidy
// (/root/reference/src/cpp/emissionfunction_smooth_kernels.cpp:106-349 compute, :354-383 reduce).
//
// Design:
//   cf_prep      lanes <-> cells.  Coalesced SoA reads of the 18 cell arrays, delta-f splines staged
//                in LDS, writes the per-cell *derived coefficient streams* S1/S2/S3 (cf_device.h):
//                everything in the integrand that does not depend on the species/pT of a lane is
//                folded into ~4 numbers per (cell, phi, y) so the hot loop is ~20 fp64 VALU ops/eval.
//   cf_main_*    lanes <-> (species class, pT) bins, loop over cells.  Cell coefficients are
//                wave-uniform: cf_main_direct takes them through scalar loads, the tile kernels through an LDS-staged stream
//                of (phi tile x y tile) unit records; no cross-lane reduction exists in the hot loop and the accumulators
//                stay in VGPRs for a whole cell chunk.  fp64 VALU bound (fp64 MFMA measured: it blocks the VALU for its whole
//                duration, tools/ubench_mfma64.hip).  Rows and units that provably cannot change a bit of any accumulator are
//                skipped (exact zeros; terms below half an ulp of every accumulator they would be added to): 58 % of the rows
//                of BASELINE config 3.
//                  cf_main_tile3e (3+1D, the default there): the phi-side exponentials come from the E2 table stream cf_prep
//                  writes once per (cell, phi tile) instead of being recomputed by every lane; direct-to-LDS staging
//                  (global_load_lds_dwordx4); the rows of a unit are tested for liveness before their exponentials.
//                  cf_main_tile (2+1D, include_baryon): with few momentum bins every bin gets S lane slots that take the units
//                  u = s (mod S) of the stream (unit-strided lanes), so that the waves are full.
//   cf_finalize  fixed-order sum over cell chunks (bitwise reproducible), x prefactor x degeneracy,
//                scatter from class layout to the reference's species-fastest layout.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "cf_launch.h"
#include "cf_math.h"
#include "errors.h"

namespace is3d {


// Tail of one integrand evaluation, shared by both main kernels.
//   z = exp(-p.u/T), x = p.u/T, br = mT^2 alpha + mT pT beta + pT^2 gamma, pds = p.dsigma (weighted)
// reference lines (smooth_kernels.cpp): feq :289 = z/(1 + sign z); feqbar :290 = 1/(1 + sign z);
// df :303-321; regulate :328; f :330; outflow :285.
template <bool CE, bool OUTFLOW, bool REG>
__device__ __forceinline__ double eval_tail(double z, double x, double br, double kappa, double sign, double pds)
{
    double d = __builtin_fma(sign, z, 1.0);
    double r, df;
    if (CE) {
        // one reciprocal for both 1/(1 + sign z) and 1/x
        double R = rcp_nr(d * x);
        r = R * x;
        double invx = R * d;
        df = r * __builtin_fma(br, invx, kappa * x);
    } else {
        r = rcp_nr(d);
        df = r * br;
    }
    if (REG) df = __builtin_fmax(-1.0, __builtin_fmin(df, 1.0));
    double feq = z * r;
    double f = __builtin_fma(feq, df, feq);
    if (OUTFLOW) pds = __builtin_fmax(pds, 0.0);
    return pds * f;
}

// ------------------------------------------------------------------------------------------------
// cf_pds_bound: a bound on |p.dsigma| = |mT (ch dat + sh dan/tau) w + pT w (cos dax + sin day)| over all lanes, bins and
// cells of an execute, for the power-of-two scale of the tiled stream (cf_device.h):
//   mTmax (|dat| + |dan|/tau) cosh(max_k |y_k - eta|) + pTmax (|dax| + |day|)     (x max eta weight in 2+1D)
// Bits of a non-negative double order like unsigned integers, so atomicMax on the bit pattern is a float max.
// Non-finite cells are ignored (they are skipped or reported by cf_prep).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cf_pds_bound(CellPtrs cells, int64_t n_cells, int dim3, double kmin, double kmax, double gw2d, double mTmax, double pTmax,
             unsigned long long *__restrict__ out)
{
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += (int64_t)gridDim.x * blockDim.x) {
        double g = gw2d;
        if (dim3) {
            const double eta = cells.eta[i];
            g = cosh(fmax(fabs(kmin - eta), fabs(kmax - eta)));
        }
        const double b = (mTmax * (fabs(cells.dat[i]) + fabs(cells.dan[i] / cells.tau[i])) + pTmax * (fabs(cells.dax[i]) + fabs(cells.day[i]))) * g;
        if (b < 1.0e300) m = fmax(m, b);   // false for NaN
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// 2^-e (and optionally 2^e) with |p.dsigma| < 2^e; 1 when there is no bound (empty or all-zero surface, flat streams)
__device__ __forceinline__ double pds_scale(const unsigned long long *bound_bits, double *inverse)
{
    int e = 0;
    if (bound_bits) {
        const double b = __longlong_as_double((long long)*bound_bits);
        if (b > 0.0) (void)frexp(b, &e);
    }
    if (inverse) *inverse = ldexp(1.0, e);
    return ldexp(1.0, -e);
}

hipError_t launch_pds_bound(const CellPtrs &cells, int64_t n_cells, int is_dim3, double kmin, double kmax, double gw2d, double mTmax,
                            double pTmax, unsigned long long *out, hipStream_t st)
{
    if (n_cells <= 0) return hipSuccess;
    const int grid = (int)std::min<int64_t>((n_cells + 255) / 256, 2048);
    hipLaunchKernelGGL(cf_pds_bound, dim3(grid), dim3(256), 0, st, cells, n_cells, is_dim3, kmin, kmax, gw2d, mTmax, pTmax, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// cf_prep
// ------------------------------------------------------------------------------------------------
constexpr int kPrepCB = 4;          // cells per workgroup batch (2+1D: K = 241 rows per cell fill the LDS)
constexpr int kPrepCB3 = 16;       // 3+1D: larger batches give longer contiguous runs per stream of the tiled output
constexpr int kPrepThreads = 256;
// wave 0's share of a batch's units, in % of another wave's, while it runs the next batch's phase 1: measured 40 / 60 / 77 / 100:
// 4.27 / 4.17 / 4.23 / 4.27 ms (3+1D, 1e6 cells, 16 cells per batch); 2+1D (4 cells per batch, 24 x 404-double records per cell: the
// phase 1 is a tenth of a wave's batch): 60 / 85 / 93 / 100: 4.17 / 3.98 / 3.95 / 3.96 ms per 1e5 cells
constexpr int kPrepW0Share3 = 60, kPrepW0Share2 = 93;
constexpr int kPrepWriterDefault = 3;   // record writer: 3 = duo / rows writers (raw and beta elements in separate trips with fixed lane roles; where eligible, else 1) | 1 = two elements per lane and trip | 0 = one (A/B: IS3D_PREP_PAIR)

struct CellScal {
    double dat, dax, day, dan_tau, ut, ux, uy, tau_un, invT, eta;
    double pitt, pitx, pity, tpitn, pixx, pixy, tpixn, piyy, tpiyn, t2pinn;
    double shear, Pb0, Pb2, kappa;
    double wvalid;  // 1.0 valid, 0.0 neutralised (skipped or out-of-table)
    double psc;     // tiled stream: p.dsigma is stored times psc = 2^-e (cf_device.h)
    // include_baryon: alpha_B = mu_B/T; V^mu (V^tau reconstructed, :193); b1P = bulk1_coeff*Pi;
    // cLin / cQ: coefficients of the parts of the diffusion term that are linear / quadratic in the momentum
    double alphaB, Vt, Vx, Vy, tVn, b1P, cLin, cQ;
};

template <int CB, bool NT>
__global__ void __launch_bounds__(kPrepThreads) cf_prep(PrepParams p)
{
    extern __shared__ double lds[];
    const int nT = p.spl.n, nspl = p.spl.nspl, J = p.J, K = p.K;
    double *sx = lds;                         // [nT]
    double *sy = sx + nT;                     // [nspl][nT]
    double *sc = sy + nspl * nT;              // [nspl][nT]
    CellScal *cs = (CellScal *)(sc + nspl * nT);  // [2][CB]
    // per (cell, k): A, Cp, alpha, W, ch, sh, C ; per (cell, j): B, Dp, gamma, D, E, F
    // (the two last arrays of each group exist with include_baryon only: a 241-row eta table would pay 15 KB for them, and two
    // workgroups per CU need <= 80 KB each)
    const int NKA = p.baryon ? 9 : 7, NJA = p.baryon ? 8 : 6;
    double *lk = (double *)(cs + 2 * CB);    // [NKA][CB][K]   (cs: two buffers, the batch in hand and the next one's)
    double *lj = lk + NKA * CB * K;      // [NJA][CB][J]
    const int CK = CB * K, CJ = CB * J;
    double *l_A = lk, *l_Cp = lk + CK, *l_al = lk + 2 * CK, *l_W = lk + 3 * CK, *l_ch = lk + 4 * CK, *l_sh = lk + 5 * CK, *l_C = lk + 6 * CK;
    double *l_B = lj, *l_Dp = lj + CJ, *l_ga = lj + 2 * CJ, *l_D = lj + 3 * CJ, *l_E = lj + 4 * CJ, *l_F = lj + 5 * CJ;
    double *l_V1 = lk + 7 * CK, *l_Lk = lk + 8 * CK, *l_V2 = lj + 6 * CJ, *l_L2 = lj + 7 * CJ;   // include_baryon only
    double *l_bD = lj + NJA * CJ;              // [CB][jtiles] max_j Dp_j of a phi tile   (unit-level cull bounds, 3+1D)
    double *l_bC = l_bD + CB * ((J + 1) / 2);  // [CB][rblocks] min_k Cp_k of a row block
    // element descriptors of a unit record (tiled stream), one int2 per record element, filled once per workgroup (below)
    // 16-byte aligned (two descriptors per read): an even double offset from the block's base -- NOT a round trip through uintptr_t, which
    // loses the LDS address space: the descriptor reads became flat_load + s_waitcnt vmcnt(0), i.e. every trip of the writer waited for
    // all of the wave's outstanding stores (gfx9 counts stores in vmcnt)
    int2 *desc = (int2 *)(lds + (((int)((p.dim3 ? l_bC + CB * K : l_bD) - lds) + 1) & ~1));

    const int tid = threadIdx.x;
    for (int i = tid; i < nT; i += kPrepThreads) {
        sx[i] = p.spl.x[i];
        for (int s = 0; s < nspl; s++) {
            sy[s * nT + i] = p.spl.y[s][i];
            sc[s * nT + i] = p.spl.c[s][i];
        }
    }
    __syncthreads();

    // Record-element descriptors (tiled stream).  Every element of a unit record is a copy of one LDS double (possibly times the
    // p.dsigma scale), an exact zero, or a beta_jk; WHICH is the same for every unit, so the decode is done once:
    //   x: bits 0-15 offset of the source array in the LDS block | 16-18 how the index continues (0: c J + j, 1: c K + k, 2: c jtiles + jt,
    //      3: c rblocks + rb, 4: c sizeof(CellScal)/8) | 19 scaled | 20 beta | 21 zero when the row is padding (k >= K) | 22 always zero
    //   y: jj (header entry / beta column) | r << 16 (row)
    // (As nested conditionals inside the writer loop the decode compiled to ~30 exec-masked branches with an LDS round trip in each arm:
    // 1 400 cycles per 64 elements, 91 % of the 2+1D prep kernel.)
    if (p.tiled) {
        const int JT = p.JT, R = p.R;
        const bool unit_bounds = p.dim3 && JT >= 2;
        const int HDR = 4 * JT + (p.baryon ? 2 : 0), RS = p.baryon ? 6 : 4, RWD = RS + JT, REC = HDR + R * RWD;
        const int o_A = (int)(l_A - lds), o_Cp = (int)(l_Cp - lds), o_al = (int)(l_al - lds), o_W = (int)(l_W - lds), o_Lk = (int)(l_Lk - lds);
        const int o_B = (int)(l_B - lds), o_Dp = (int)(l_Dp - lds), o_ga = (int)(l_ga - lds), o_L2 = (int)(l_L2 - lds);
        const int o_bD = (int)(l_bD - lds), o_bC = (int)(l_bC - lds);
        const int o_alphaB = (int)(&cs[0].alphaB - lds);
        constexpr int ZERO = 1 << 22, PADZ = 1 << 21, BETA = 1 << 20, SCALED = 1 << 19;
        for (int e = tid; e < REC; e += kPrepThreads) {
            int x = ZERO, y = 0;
            if (e < 4 * JT) {
                const int jj = e >> 2, f = e & 3;
                y = jj;
                x = f == 0 ? (o_B | SCALED) : f == 1 ? o_Dp : f == 2 ? o_ga : (p.baryon ? o_L2 : ZERO);
                if (unit_bounds && !p.baryon && e == 3) x = o_bD | (2 << 16);         // max_j D'_j of the tile: bmax = pT Dmax
                else if (unit_bounds && !p.baryon && e == 7) x = o_bC | (3 << 16);    // min_k C'_k of the unit's rows
            } else if (e < HDR) {
                x = (e == 4 * JT) ? (o_alphaB | (4 << 16)) : ((unit_bounds && e == 4 * JT + 1) ? (o_bD | (2 << 16)) : ZERO);
            } else {
                const int q = e - HDR, r = q / RWD, f = q - r * RWD;
                y = r << 16;
                if (f < 4) x = (f == 0 ? (o_A | SCALED | PADZ) : f == 1 ? o_Cp : f == 2 ? (o_al | PADZ) : (o_W | PADZ)) | (1 << 16);   // a padding row keeps Cp of row K-1
                else if (f < RS) x = (f == 4) ? (o_Lk | (1 << 16) | PADZ) : ((unit_bounds && r == 0) ? (o_bC | (3 << 16) | PADZ) : ZERO);
                else { x = BETA | PADZ; y |= f - RS; }
            }
            desc[e] = int2{x, y};
        }
        __syncthreads();
    }

    const int nbatch = (p.n_cells + CB - 1) / CB;
    // ---- phase 1: per-cell scalars (smooth_kernels.cpp:118-242), one lane per cell of the batch ----
    // A serial chain (22 loads, a square root, seven divisions, three spline searches, a cosh: ~7 us) that only 16 lanes of the workgroup
    // work on: the batch loop below runs it for the NEXT batch on wave 0 while the other waves write the current batch's records
    // (two CellScal buffers; wave 0 takes a smaller share of the records), so only a workgroup's first batch waits for it.
    auto phase1 = [&](const int cbase, CellScal *dst) {
        {
            const int64_t gi = p.cell0 + cbase + tid;
            CellScal s;
            double tau = p.cells.tau[gi], tau2 = tau * tau;
            double dat = p.cells.dat[gi], dax = p.cells.dax[gi], day = p.cells.day[gi], dan = p.cells.dan[gi];
            double ux = p.cells.ux[gi], uy = p.cells.uy[gi], un = p.cells.un[gi];
            double ut = sqrt(1.0 + ux * ux + uy * uy + tau2 * un * un);          // :133
            double udsigma = ut * dat + ux * dax + uy * day + un * dan;           // :135
            bool valid = udsigma > 0.0;                                           // :137
            if (!valid) atomicAdd(&p.status[1], 1ULL);
            double T = p.cells.T[gi];
            double muB = 0.0, nB = 0.0, Vx = 0.0, Vy = 0.0, Vn = 0.0;
            if (valid && p.baryon && p.baryondiff) {                              // :186-197
                muB = p.cells.muB[gi]; nB = p.cells.nB[gi];
                Vx = p.cells.Vx[gi]; Vy = p.cells.Vy[gi]; Vn = p.cells.Vn[gi];
            }
            double bl[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            if (valid && p.baryon) {
                if (!bilinear5(p.bil, T, muB, bl)) {                               // outside the (T, mu_B) table
                    atomicMin(&p.status[0], (unsigned long long)gi);
                    valid = false;
                }
            } else if (valid && !(T >= sx[0] && T <= sx[nT - 1])) {               // GSL domain error
                atomicMin(&p.status[0], (unsigned long long)gi);
                valid = false;
            }
            if (valid) {
                // exponent range of exp_core: p.u/T <= mTmax (u^tau + |tau u^eta|) cosh(max |y - eta|) / T must stay below 1e9
                // (a Lorentz factor of several hundred at T = 0.1 GeV; the reference's exp() would return inf there)
                const double eta0 = p.dim3 ? p.cells.eta[gi] : 0.0;
                const double xb = p.mTmax * (ut + fabs(tau * un)) * cosh(fmax(fabs(p.kmin - eta0), fabs(p.kmax - eta0))) / T;
                if (!(xb < 1.0e9)) {
                    atomicMin(&p.status[7], (unsigned long long)gi);
                    valid = false;
                }
            }
            if (valid) {
                double P = p.cells.P[gi], E = p.cells.E[gi];
                double ux2 = ux * ux, uy2 = uy * uy, ut2 = ut * ut;
                double utperp2 = 1.0 + ux * ux + uy * uy;                         // utperp^2, :142
                double pixx = 0, pixy = 0, pixn = 0, piyy = 0, piyn = 0, pinn = 0, pitn = 0, pity = 0, pitx = 0, pitt = 0;
                if (p.include_shear) {                                            // :159-171
                    pixx = p.cells.pixx[gi]; pixy = p.cells.pixy[gi]; pixn = p.cells.pixn[gi];
                    piyy = p.cells.piyy[gi]; piyn = p.cells.piyn[gi];
                    pinn = (pixx * (ux2 - ut2) + piyy * (uy2 - ut2) + 2.0 * (pixy * ux * uy + tau2 * un * (pixn * ux + piyn * uy))) / (tau2 * utperp2);
                    pitn = (pixn * ux + piyn * uy + tau2 * pinn * un) / ut;
                    pity = (pixy * ux + piyy * uy + tau2 * piyn * un) / ut;
                    pitx = (pixx * ux + pixy * uy + tau2 * pixn * un) / ut;
                    pitt = (pitx * ux + pity * uy + tau2 * pitn * un) / ut;
                }
                double bulkPi = p.include_bulk ? p.cells.bulkPi[gi] : 0.0;       // :173-175
                double T2 = T * T, T4 = T2 * T2;
                double shear, b0, b1 = 0.0, b2, kappa = 0.0, cLin = 0.0, cQ = 0.0;
                const double T3 = T2 * T;
                if (!p.ce) {                                                      // :222-229
                    double c0, c2;
                    if (p.baryon) {                                               // deltafReader.cpp:436-452
                        c0 = bl[0] / T4; b1 = bl[1] / T3; c2 = bl[2] / T4;
                        cLin = bl[3] / T4;                                        // c3
                        cQ = bl[4] / (T4 * T);                                    // c4
                    } else {                                                      // deltafReader.cpp:337-344
                        c0 = spline_eval_lds(nT, sx, sy, sc, T) / T4;
                        c2 = spline_eval_lds(nT, sx, sy + nT, sc + nT, T) / T4;
                    }
                    shear = 0.5 / (T2 * (E + P));
                    b0 = c0 - c2;
                    b2 = 4.0 * c2 - c0;
                } else {                                                          // :230-237
                    double F, betabulk, betapi;
                    if (p.baryon) {                                               // deltafReader.cpp:454-468
                        F = bl[0] * T; betabulk = bl[2] * T4; betapi = bl[4] * T4;
                        const double G = bl[1], betaV = bl[3] * T3;
                        b1 = G / betabulk;
                        cLin = -1.0 / (T * betaV);                                // -(b/pu) V.p / betaV, pu = T x
                        cQ = (nB / (E + P)) / betaV;                              // baryon_enthalpy_ratio V.p / betaV
                    } else {                                                      // deltafReader.cpp:352-358
                        F = spline_eval_lds(nT, sx, sy, sc, T) * T;
                        betabulk = spline_eval_lds(nT, sx, sy + nT, sc + nT, T) * T4;
                        betapi = spline_eval_lds(nT, sx, sy + 2 * nT, sc + 2 * nT, T) * T4;
                    }
                    shear = 0.5 / (betapi * T);
                    b0 = F / (T2 * betabulk);
                    b2 = 1.0 / (3.0 * T * betabulk);
                    kappa = (b0 + b2) * bulkPi * T;
                }
                s.dat = dat; s.dax = dax; s.day = day; s.dan_tau = dan / tau;
                s.ut = ut; s.ux = ux; s.uy = uy; s.tau_un = tau * un; s.invT = 1.0 / T;
                s.eta = p.dim3 ? p.cells.eta[gi] : 0.0;
                s.pitt = pitt; s.pitx = pitx; s.pity = pity; s.tpitn = tau * pitn;
                s.pixx = pixx; s.pixy = pixy; s.tpixn = tau * pixn; s.piyy = piyy; s.tpiyn = tau * piyn;
                s.t2pinn = tau2 * pinn;
                s.shear = shear; s.Pb0 = bulkPi * b0; s.Pb2 = bulkPi * b2; s.kappa = kappa;
                s.alphaB = muB / T;                                               // :195
                s.Vx = Vx; s.Vy = Vy; s.tVn = tau * Vn;
                s.Vt = (Vx * ux + Vy * uy + tau2 * Vn * un) / ut;                 // :193
                s.b1P = b1 * bulkPi; s.cLin = cLin; s.cQ = cQ;
                s.wvalid = 1.0;
            } else {
                // neutral cell: p.dsigma == 0 for every momentum, finite distribution -> contributes 0
                s.dat = s.dax = s.day = s.dan_tau = 0.0;
                s.ut = 1.0; s.ux = s.uy = s.tau_un = 0.0; s.invT = 1.0; s.eta = 0.0;
                s.pitt = s.pitx = s.pity = s.tpitn = s.pixx = s.pixy = s.tpixn = s.piyy = s.tpiyn = s.t2pinn = 0.0;
                s.shear = s.Pb0 = s.Pb2 = s.kappa = 0.0;
                s.alphaB = s.Vt = s.Vx = s.Vy = s.tVn = s.b1P = s.cLin = s.cQ = 0.0;
                s.wvalid = 0.0;
            }
            dst[tid] = s;
        }
    };
    CellScal *const cs_base = cs;
    if ((int)blockIdx.x < nbatch && tid < min(CB, p.n_cells - (int)blockIdx.x * CB)) phase1(blockIdx.x * CB, cs);
    __syncthreads();
    for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const int cbase = batch * CB;
        const int ncb = min(CB, p.n_cells - cbase);
        const int cs_cur = (int)((const double *)cs - (const double *)cs_base);      // the buffer in use, as an offset in doubles
        const int batch_next = batch + gridDim.x;
        const bool prefetch = batch_next < nbatch;

        // ---- phase 2: (cell, k) quantities ----
        for (int idx = tid; idx < ((p.dev_skip & 4) ? 0 : ncb * K); idx += kPrepThreads) {
            const int c = idx / K, k = idx - c * K;
            const CellScal &s = cs[c];
            double dlt, w;
            (void)dlt;
            double ch, sh;
            if (p.dim3) { dlt = p.kgrid[k] - s.eta; w = s.wvalid; ch = cosh(dlt); sh = sinh(dlt); }      // y - eta_cell, :279
            else {
                // y = 0, eta = table node (:75-80): cosh / sinh(0 - eta_k) do not depend on the cell -- tabulated once per plan
                // (host libm, like the reference) instead of 2 x 241 device libm calls per cell
                w = p.kweight[k] * s.wvalid; ch = p.kch[k]; sh = p.ksh[k];
            }
            double C = ch * s.ut - sh * s.tau_un;
            double Q0 = s.pitt * ch * ch + s.t2pinn * sh * sh - 2.0 * s.tpitn * ch * sh;
            l_A[c * K + k] = w * (ch * s.dat + sh * s.dan_tau);
            l_Cp[c * K + k] = C * s.invT;
            // Chapman-Enskog: the kappa*x term is folded in, (N' + kappa x^2)/x with x^2 = mT^2 Cp^2 - 2 mT pT Cp Dp + pT^2 Dp^2
            const double Cpk = C * s.invT;
            double alpha = p.ce ? ((s.shear * Q0 - s.Pb2) * s.invT + s.kappa * Cpk * Cpk) : (s.shear * Q0 + s.Pb2 * C * C + s.Pb0);
            if (p.baryon) {
                // V.p = mT V1_k - pT V2_j.  14-moment: (c3 b + c4 pu) V.p;  Chapman-Enskog, times x: (rho x - b/T) V.p / betaV.
                // The part without b is a quadratic form (folded into alpha, beta, gamma); the part with b is linear:
                // b (mT L_k + pT L2_j), together with the bulk term b1 b pu Pi (14-moment) / b1 b Pi x (Chapman-Enskog).
                const double V1 = ch * s.Vt - sh * s.tVn;
                const double Cq = p.ce ? Cpk : C;
                alpha += s.cQ * Cq * V1;
                l_V1[c * K + k] = V1;
                l_Lk[c * K + k] = s.b1P * Cq + s.cLin * V1;
            }
            l_al[c * K + k] = alpha;
            l_W[c * K + k] = w;
            l_ch[c * K + k] = ch; l_sh[c * K + k] = sh; l_C[c * K + k] = C;
        }
        // ---- phase 2b: (cell, j) quantities ----
        for (int idx = tid; idx < ((p.dev_skip & 4) ? 0 : ncb * J); idx += kPrepThreads) {
            const int c = idx / J, j = idx - c * J;
            const CellScal &s = cs[c];
            double cp = p.cosphi[j], sp = p.sinphi[j];
            double D = cp * s.ux + sp * s.uy;
            double Q2 = s.pixx * cp * cp + s.piyy * sp * sp + 2.0 * s.pixy * cp * sp;
            l_B[c * J + j] = cp * s.dax + sp * s.day;
            l_Dp[c * J + j] = D * s.invT;
            const double Dpj = D * s.invT;
            double gamma = p.ce ? ((s.shear * Q2 + s.Pb2) * s.invT + s.kappa * Dpj * Dpj) : (s.shear * Q2 + s.Pb2 * D * D - s.Pb0);
            if (p.baryon) {
                const double V2 = cp * s.Vx + sp * s.Vy;
                const double Dq = p.ce ? Dpj : D;
                gamma += s.cQ * Dq * V2;
                l_V2[c * J + j] = V2;
                l_L2[c * J + j] = -(s.b1P * Dq + s.cLin * V2);
            }
            l_ga[c * J + j] = gamma;
            l_D[c * J + j] = D;
            l_E[c * J + j] = -2.0 * (s.pitx * cp + s.pity * sp);
            l_F[c * J + j] = 2.0 * (s.tpixn * cp + s.tpiyn * sp);
        }
        __syncthreads();
        // 3+1D records carry the bounds of the main kernel's unit-level cull in free slots: without baryon slots the x of header
        // entries jj = 0, 1; "B" records the free double behind alpha_B and the free scalar of row 0 (cf_device.h)
        if (p.tiled && p.dim3 && p.JT >= 2) {
            const int JT = p.JT, R = p.R;
            for (int idx = tid; idx < ncb * p.jtiles; idx += kPrepThreads) {
                const int c = idx / p.jtiles, jt = idx - c * p.jtiles;
                double v = -1.0e300;
                for (int q2 = 0; q2 < JT; q2++) v = fmax(v, l_Dp[c * J + min(jt * JT + q2, J - 1)]);
                l_bD[idx] = v;
            }
            for (int idx = tid; idx < ncb * p.rblocks; idx += kPrepThreads) {
                const int c = idx / p.rblocks, rb = idx - c * p.rblocks;
                double v = 1.0e300;
                for (int q2 = 0; q2 < R; q2++) v = fmin(v, l_Cp[c * K + min(rb * R + q2, K - 1)]);   // padding rows repeat row K-1
                l_bC[idx] = v;
            }
            __syncthreads();
        }
        // phase 1 of the workgroup's next batch, on wave 0, under the other waves' record writing
        if (prefetch && tid < min(CB, p.n_cells - batch_next * CB)) phase1(batch_next * CB, cs == cs_base ? cs_base + CB : cs_base);
        // units (and E2 tables) per wave: wave 0 takes w0 % of another wave's share while it has a phase 1 to run
        const int w0 = prefetch ? p.w0_share : 100;
        auto wave_lo = [&](int n_items, int w) { return w == 0 ? 0 : (int)(((int64_t)n_items * (w0 + 100 * (w - 1))) / (w0 + 100 * (kPrepThreads / 64 - 1))); };
        // beta_jk = shear X_jk - 2 Pi b2 C_k D_j (14-moment) | shear X_jk / T (Chapman-Enskog)
        // ij = c J + j, ik = c K + k (the duo writer forms them with 24-bit multiplies)
        auto beta_at = [&](const CellScal &s, int ij, int ik) {
            double X = l_E[ij] * l_ch[ik] + l_F[ij] * l_sh[ik];
            double beta = p.ce ? (s.shear * X * s.invT - 2.0 * s.kappa * l_Cp[ik] * l_Dp[ij])
                               : (s.shear * X - 2.0 * s.Pb2 * l_C[ik] * l_D[ij]);
            if (p.baryon) {
                const double Cq = p.ce ? l_Cp[ik] : l_C[ik], Dq = p.ce ? l_Dp[ij] : l_D[ij];
                beta -= s.cQ * (Cq * l_V2[ij] + Dq * l_V1[ik]);
            }
            return beta;
        };
        auto beta_of = [&](int c, int j, int k) { return beta_at(cs[c], c * J + j, c * K + k); };
        if (!p.tiled) {
            // ---- phase 3 (flat): S1[cell][k][4], S2[cell][j][4], S3[cell][j][k] ----
            for (int idx = tid; idx < ncb * K; idx += kPrepThreads) {
                double *o = p.S1 + ((int64_t)cbase * K + idx) * kS1Rec;
                o[0] = l_A[idx]; o[1] = l_Cp[idx]; o[2] = l_al[idx]; o[3] = l_W[idx];
            }
            for (int idx = tid; idx < ncb * J; idx += kPrepThreads) {
                double *o = p.S2 + ((int64_t)cbase * J + idx) * kS2Rec;
                o[0] = l_B[idx]; o[1] = l_Dp[idx]; o[2] = l_ga[idx]; o[3] = 0.0;   // kappa is folded into alpha/beta/gamma
            }
            const int JK = J * K;
            for (int idx = tid; idx < ncb * JK; idx += kPrepThreads) {
                const int c = idx / JK, r = idx - c * JK, j = r / K, k = r - j * K;
                p.S3[(int64_t)(cbase + c) * JK + r] = beta_of(c, j, k);
            }
        } else {
            // ---- phase 3 (tiled): one unit record per wave at a time, consecutive lanes -> consecutive doubles.  The (cell,
            // tile, row block) loops are wave-uniform and the only division left, element -> (row, field), is a multiplication
            // (q < 2^15, RWD < 64: q * ceil(2^20 / RWD) >> 20 is exact); decoding a flat index with five runtime divisions per
            // element made this kernel VALU-bound (200 instructions per double written).
            const int JT = p.JT, R = p.R;
            const bool unit_bounds = p.dim3 && JT >= 2;
            const double psc = pds_scale(p.pds_bound, nullptr);
            const int HDR = 4 * JT + (p.baryon ? 2 : 0), RS = p.baryon ? 6 : 4, RWD = RS + JT, REC = HDR + R * RWD;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // wave-uniform unit ranges: scalar loop arithmetic
            // units in (stream, cell) order, a contiguous range per wave: a wave writes the records of consecutive cells of ONE stream
            // back to back (ncb x REC x 8 bytes contiguous in 3+1D) instead of hopping to another stream after every 928-byte record
            const int n_units_b = ncb * p.jtiles * p.rblocks;
            const int n_lo = wave_lo(n_units_b, wave), n_hi = wave_lo(n_units_b, wave + 1);
            // a record of at most 128 doubles (3+1D: 116) is ONE trip of the pair writer, lane <-> element pair fixed for every unit: its
            // descriptors are read and decoded once, outside the unit loop
            // duo writer (pair_writer 3): TWO unit records per trip, one per half wave, and a record's raw elements (header + row scalars: one
            // LDS read each) and its beta elements (six LDS reads each) in separate trips of 16-byte stores -- no lane computes a beta_of it
            // does not store, no beta lane decodes a descriptor, and both trips run with ~60 of 64 lanes busy (the one-trip pair writer: 58 of
            // 64 lanes, every lane paying for a beta_of).  Needs pairs that stay inside a field group: JT even, no baryon slots.
            // record stores: PLAIN by default.  The duo and rows writers store a record in 32-, 64- and 256-byte pieces from different trips; as
            // non-temporal stores those pieces leave the L2 as partial-line writes (2+1D: 4.2 ms per 1e5 cells, 7.8 GB), as plain stores they
            // merge in the L2 first (1.9 ms); 3+1D 4.2 -> 3.8 ms.  The whole-run writers (pair_writer 0, 1) do not care (3.43 / 3.75 ms either way).
            auto st_ = [&](double v, double *q) { if constexpr (NT) __builtin_nontemporal_store(v, q); else *q = v; };
            const int NRAWP = (HDR + R * RS) / 2, NBETAP = (R * JT) / 2;
            const bool duo = p.pair_writer == 3 && !p.baryon && !(JT & 1) && NRAWP <= 32 && NBETAP <= 32;
            // otherwise: a record of at most 128 doubles is ONE trip of the pair writer, lane <-> element pair fixed for every unit: its
            // descriptors are read and decoded once, outside the unit loop
            const bool one_trip = (p.pair_writer == 1 || p.pair_writer == 3) && !duo && REC / 2 <= 64;   // (pair_writer 3 with neither the duo nor the rows writer eligible: baryon slots, odd JT)
            int4 dd0 = int4{1 << 22, 0, 1 << 22, 0};
            if (one_trip && lane < REC / 2) dd0 = ((const int4 *)desc)[lane];
            // a lane's raw elements never change (duo and rows writers): their descriptors are decoded ONCE into a branch-free address rule
            //   lds[off + c mc + min(jt a + rb b + q, lim)] * mul,   zero when (always | padding row and rb R + r >= K)
            struct RawRule { int off, mc, a, b, q, lim, r, zero, padz; double mul; };
            auto rule_of = [&](const int2 d) {
                RawRule u;
                const int jj = d.y & 0xffff, r = d.y >> 16, msel = (d.x >> 16) & 7;
                u.off = (d.x & 0xffff) + (msel == 4 ? cs_cur : 0); u.r = r;
                u.mc = msel == 0 ? J : msel == 1 ? K : msel == 2 ? p.jtiles : msel == 3 ? p.rblocks : (int)(sizeof(CellScal) / sizeof(double));
                u.a = msel == 0 ? JT : msel == 2 ? 1 : 0;
                u.b = msel == 1 ? R : msel == 3 ? 1 : 0;
                u.q = msel == 0 ? jj : msel == 1 ? r : 0;
                u.lim = msel == 0 ? J - 1 : msel == 1 ? K - 1 : 0x7fffffff;
                u.zero = (d.x >> 22) & 1; u.padz = (d.x >> 21) & 1;
                u.mul = ((d.x >> 19) & 1) ? psc : 1.0;
                return u;
            };
            auto raw_value = [&](const RawRule &u, int c, int jt, int rb) {
                const bool zero = u.zero | (u.padz & (__mul24(rb, R) + u.r >= K));
                const int at = u.off + __mul24(c, u.mc) + min(__mul24(jt, u.a) + __mul24(rb, u.b) + u.q, u.lim);
                const double v = lds[zero ? 0 : at] * u.mul;
                return zero ? 0.0 : v;
            };
            constexpr int CSD = (int)(sizeof(CellScal) / sizeof(double));
            const int cs_off = (int)((const double *)cs - lds);
            // rows writer (pair_writer 3, records too long for the duo writer: the 2+1D tiles, 31 or 61 eta rows): fixed lane roles in three
            // kinds of trip, all hoisted -- the headers of 64 / (HDR / 2) consecutive units in one trip (lane <-> (unit, header pair)); then per
            // unit one trip for the row scalars (lane <-> (row, pair)) and one or two for the beta elements (lane <-> (row, column pair)).
            // No descriptor read, no beta_of on a raw lane, 13 trips instead of 16 per four 404-double records, every trip >= 94 % full.
            const int NH = HDR / 2, NRW = (R * RS) / 2;
            const bool rows_w = p.pair_writer == 3 && !duo && !p.baryon && !(JT & 1) && NH <= 64 && NRW <= 64 && NBETAP <= 128;
            if (rows_w && !(p.dev_skip & 1)) {
                const int UH = 64 / NH;
                const int hsub = lane / NH, hp = lane - hsub * NH;
                const bool hdr_lane = hsub < UH;
                const RawRule h0 = rule_of(desc[2 * hp]), h1 = rule_of(desc[2 * hp + 1]);
                const int lw = min(lane, NRW - 1), rw_r = lw / (RS / 2), eW = HDR + rw_r * RWD + 2 * (lw - rw_r * (RS / 2));
                const RawRule w0 = rule_of(desc[eW]), w1 = rule_of(desc[eW + 1]);
                const int q0 = min(lane, NBETAP - 1), q1 = min(lane + 64, NBETAP - 1);
                const int rB0 = (2 * q0) / JT, jB0 = 2 * q0 - rB0 * JT, eB0 = HDR + rB0 * RWD + RS + jB0;
                const int rB1 = (2 * q1) / JT, jB1 = 2 * q1 - rB1 * JT, eB1 = HDR + rB1 * RWD + RS + jB1;
                const bool row_lane = lane < NRW, b0_lane = lane < NBETAP, b1_lane = lane + 64 < NBETAP;
                auto decode = [&](int nn, int &jt, int &rb, int &c) {
                    if (p.dim3) {
                        const int sidx = nn / ncb;
                        c = nn - sidx * ncb; jt = sidx / p.rblocks; rb = sidx - jt * p.rblocks;
                    } else {
                        const int per_jt = ncb * p.rblocks;
                        jt = nn / per_jt;
                        const int r2 = nn - jt * per_jt;
                        c = r2 / p.rblocks; rb = r2 - c * p.rblocks;
                    }
                };
                const int64_t wrap_stride = p.dim3 ? (int64_t)(p.n_cells - ncb + 1) * REC : (int64_t)(1 + (int64_t)(p.n_cells - ncb) * p.rblocks) * REC;
                auto step = [&](int &jt, int &rb, int &c, double *&o) {
                    bool wrap;
                    if (p.dim3) {
                        const int c1 = c + 1; wrap = c1 == ncb;
                        c = wrap ? 0 : c1;
                        const int rb1 = rb + (wrap ? 1 : 0); const bool w2 = rb1 == p.rblocks;
                        rb = w2 ? 0 : rb1; jt += w2 ? 1 : 0;
                    } else {
                        const int rb1 = rb + 1; const bool w1 = rb1 == p.rblocks;
                        rb = w1 ? 0 : rb1;
                        const int c1 = c + (w1 ? 1 : 0); wrap = c1 == ncb;
                        c = wrap ? 0 : c1; jt += wrap ? 1 : 0;
                    }
                    o += wrap ? wrap_stride : (int64_t)REC;
                };
                auto first_record = [&](int jt, int rb, int c) {
                    const int64_t unit0 = p.dim3 ? (int64_t)(jt * p.rblocks + rb) * p.n_cells + (cbase + c) : ((int64_t)jt * p.n_cells + (cbase + c)) * p.rblocks + rb;
                    return p.TS + unit0 * REC;
                };
                // the unit in hand (wave-uniform) and the header lanes' own unit (n + hsub): decoded once, stepped
                int jt, rb, c, hjt, hrb, hc;
                decode(min(n_lo, max(n_hi - 1, 0)), jt, rb, c);
                decode(min(n_lo + hsub, max(n_hi - 1, 0)), hjt, hrb, hc);
                double *o = first_record(jt, rb, c), *ho = first_record(hjt, hrb, hc);
                for (int n = n_lo; n < n_hi; n += UH) {
                    if (hdr_lane && n + hsub < n_hi) {
                        const double vx = raw_value(h0, hc, hjt, hrb), vy = raw_value(h1, hc, hjt, hrb);
                        st_(vx, &ho[2 * hp]);
                        st_(vy, &ho[2 * hp + 1]);
                    }
                    for (int i = 0; i < UH; i++) step(hjt, hrb, hc, ho);
                    const int cnt = min(UH, n_hi - n);
                    for (int sub = 0; sub < cnt; sub++, step(jt, rb, c, o)) {
                        const double vx = raw_value(w0, c, jt, rb), vy = raw_value(w1, c, jt, rb);
                        const int rbR = __mul24(rb, R), jtJT = __mul24(jt, JT), cK = __mul24(c, K), cJ = __mul24(c, J);
                        const CellScal &sc_ = *(const CellScal *)(lds + (cs_off + __mul24(c, CSD)));
                        const int k0 = rbR + rB0, ik0 = cK + min(k0, K - 1);
                        const double b00 = beta_at(sc_, cJ + min(jtJT + jB0, J - 1), ik0), b01 = beta_at(sc_, cJ + min(jtJT + jB0 + 1, J - 1), ik0);
                        if (row_lane) {
                            st_(vx, &o[eW]);
                            st_(vy, &o[eW + 1]);
                        }
                        if (b0_lane) {
                            st_(k0 >= K ? 0.0 : b00, &o[eB0]);
                            st_(k0 >= K ? 0.0 : b01, &o[eB0 + 1]);
                        }
                        if (NBETAP > 64) {
                            const int k1 = rbR + rB1, ik1 = cK + min(k1, K - 1);
                            const double b10 = beta_at(sc_, cJ + min(jtJT + jB1, J - 1), ik1), b11 = beta_at(sc_, cJ + min(jtJT + jB1 + 1, J - 1), ik1);
                            if (b1_lane) {
                                st_(k1 >= K ? 0.0 : b10, &o[eB1]);
                                st_(k1 >= K ? 0.0 : b11, &o[eB1 + 1]);
                            }
                        }
                    }
                }
            }
            if (duo && !(p.dev_skip & 1)) {
                const int l = lane & 31, half = lane >> 5;
                const int lr = min(l, NRAWP - 1), lb = min(l, NBETAP - 1);
                const int eR = lr < HDR / 2 ? 2 * lr : HDR + ((lr - HDR / 2) >> 1) * RWD + ((lr - HDR / 2) & 1) * 2;
                const RawRule u0 = rule_of(desc[eR]), u1 = rule_of(desc[eR + 1]);
                const int rB = (2 * lb) / JT, jB = 2 * lb - rB * JT, eB = HDR + rB * RWD + RS + jB;
                // the half wave's unit (jt, rb, c): decoded once, then stepped (two units per trip) -- per-lane integer divisions in the
                // loop would cost more than the record's arithmetic
                int jt, rb, c;
                {
                    const int nn = min(n_lo + half, max(n_hi - 1, n_lo));
                    if (p.dim3) {
                        const int sidx = nn / ncb;
                        c = nn - sidx * ncb; jt = sidx / p.rblocks; rb = sidx - jt * p.rblocks;
                    } else {
                        const int per_jt = ncb * p.rblocks;
                        jt = nn / per_jt;
                        const int r2 = nn - jt * per_jt;
                        c = r2 / p.rblocks; rb = r2 - c * p.rblocks;
                    }
                }
                const bool raw_lane = l < NRAWP, beta_lane = l < NBETAP;
                // the record pointer is stepped with the unit (a 64-bit multiply per trip otherwise), and every index product of the loop is a
                // 24-bit multiply (full rate; v_mul_lo_u32 is quarter rate and the loop is bound by VALU issue): all factors are < 2^15
                const int64_t unit0 = p.dim3 ? (int64_t)(jt * p.rblocks + rb) * p.n_cells + (cbase + c) : ((int64_t)jt * p.n_cells + (cbase + c)) * p.rblocks + rb;
                double *o = p.TS + unit0 * REC;
                const int64_t wrap_stride = p.dim3 ? (int64_t)(p.n_cells - ncb + 1) * REC : (int64_t)(1 + (int64_t)(p.n_cells - ncb) * p.rblocks) * REC;
                auto step = [&]() {   // selects, no branches: (stream, cell) order in 3+1D, (phi tile, cell, row block) order in 2+1D
                    bool wrap;
                    if (p.dim3) {
                        const int c1 = c + 1; wrap = c1 == ncb;
                        c = wrap ? 0 : c1;
                        const int rb1 = rb + (wrap ? 1 : 0); const bool w2 = rb1 == p.rblocks;
                        rb = w2 ? 0 : rb1; jt += w2 ? 1 : 0;
                    } else {
                        const int rb1 = rb + 1; const bool w1 = rb1 == p.rblocks;
                        rb = w1 ? 0 : rb1;
                        const int c1 = c + (w1 ? 1 : 0); wrap = c1 == ncb;
                        c = wrap ? 0 : c1; jt += wrap ? 1 : 0;
                    }
                    o += wrap ? wrap_stride : (int64_t)REC;
                };
                for (int n = n_lo; n < n_hi; n += 2, step(), step()) {
                    if (n + half >= n_hi) continue;          // an odd range's last trip: the upper half wave has no unit
                    const int rbR = __mul24(rb, R), jtJT = __mul24(jt, JT);
                    // both trips' LDS reads are issued before either store
                    const double vx = raw_value(u0, c, jt, rb), vy = raw_value(u1, c, jt, rb);
                    const int k = rbR + rB, ik = __mul24(c, K) + min(k, K - 1), cJ = __mul24(c, J);
                    const int j0 = min(jtJT + jB, J - 1), j1 = min(jtJT + jB + 1, J - 1);
                    const CellScal &sc_ = *(const CellScal *)(lds + (cs_off + __mul24(c, CSD)));
                    const double b0 = beta_at(sc_, cJ + j0, ik), b1 = beta_at(sc_, cJ + j1, ik);
                    if (raw_lane) {
                        st_(vx, &o[eR]);
                        st_(vy, &o[eR + 1]);
                    }
                    if (beta_lane) {
                        st_(k >= K ? 0.0 : b0, &o[eB]);
                        st_(k >= K ? 0.0 : b1, &o[eB + 1]);
                    }
                }
            }
            for (int n = n_lo; n < (((p.dev_skip & 1) || duo || rows_w) ? n_lo : n_hi); n++) {
                int jt, rb, c;
                if (p.dim3) {         // stream = (jt, rb): the cells of the batch are consecutive records
                    const int sidx = n / ncb;
                    c = n - sidx * ncb; jt = sidx / p.rblocks; rb = sidx - jt * p.rblocks;
                } else {              // stream = jt: a cell's eta row blocks are consecutive records, then the next cell's
                    const int per_jt = ncb * p.rblocks;
                    jt = n / per_jt;
                    const int r2 = n - jt * per_jt;
                    c = r2 / p.rblocks; rb = r2 - c * p.rblocks;
                }
                const int64_t cell = cbase + c;
                {
                    {
                        int64_t unit;
                        if (p.dim3) unit = (int64_t)(jt * p.rblocks + rb) * p.n_cells + cell;       // s = jt*rblocks + rb
                        else unit = ((int64_t)jt * p.n_cells + cell) * p.rblocks + rb;              // s = jt
                        double *o = p.TS + unit * REC;
                        // one LDS round trip per 64 elements: the descriptor, then the source double next to the six reads of a
                        // (clamped) beta_of; no divergent branch
                        const int cJ = c * J, cK = c * K, cT = c * p.jtiles + jt, cR = c * p.rblocks + rb, cS = cs_cur + c * (int)(sizeof(CellScal) / sizeof(double));
                        // two consecutive elements per lane: one 16-byte descriptor read, one 16-byte store (records are 16-byte
                        // multiples, cf_device.h), the two elements' LDS reads in flight together -- half the trips per record
                        auto element = [&](const int2 d) {
                            const int jj = d.y & 0xffff, r = d.y >> 16;
                            const int jcl = min(jt * JT + jj, J - 1);
                            const int k = rb * R + r, kcl = min(k, K - 1);
                            const int msel = (d.x >> 16) & 7;
                            const int add = msel == 0 ? cJ + jcl : msel == 1 ? cK + kcl : msel == 2 ? cT : msel == 3 ? cR : cS;
                            const bool zero = ((d.x >> 22) & 1) | (((d.x >> 21) & 1) & (k >= K));
                            const double raw = lds[zero ? 0 : (d.x & 0xffff) + add];
                            const double bet = beta_of(c, jcl, kcl);
                            double v = ((d.x >> 19) & 1) ? raw * psc : raw;
                            v = ((d.x >> 20) & 1) ? bet : v;
                            return zero ? 0.0 : v;
                        };
                        if (one_trip) {
                            if (lane < REC / 2) {
                                const double vx = element(int2{dd0.x, dd0.y}), vy = element(int2{dd0.z, dd0.w});
                                st_(vx, &o[2 * lane]);
                                st_(vy, &o[2 * lane + 1]);
                            }
                        } else if (p.pair_writer) {
                            const int4 *desc2 = (const int4 *)desc;
                            for (int e2 = lane; e2 < REC / 2; e2 += 64) {
                                const int4 dd = desc2[e2];
                                double2 v2;
                                v2.x = element(int2{dd.x, dd.y});
                                v2.y = element(int2{dd.z, dd.w});
                                st_(v2.x, &o[2 * e2]);
                                st_(v2.y, &o[2 * e2 + 1]);
                            }
                        } else {
                            for (int e = lane; e < REC; e += 64) st_(element(desc[e]), &o[e]);
                        }
                    }
                }
            }
            if (p.TE && unit_bounds && !(p.dev_skip & 2)) {
                // E2 table stream (cf_device.h): exp(pT Dp_j - pT Dmax) for every pT of the grid, with the roundings the main
                // kernel's own header used (explicit mul / sub, no contraction): pT Dmax == max_j (pT Dp_j) for pT >= 0
                // One (cell, phi tile) table per wave at a time, lanes <-> (jj, ipT) with ipT fastest (kE2Stride = 32: shifts, no division;
                // the lane's pT is loop-invariant: 64 lanes = two jj rows of the 32 pT columns).
                const int NPJ = kE2Stride * JT;
                static_assert(kE2Stride == 32, "lane -> (jj, ipT) decode below");
                const int wv = tid >> 6, ln = tid & 63;
                const double pTl = p.pTgrid[min(ln & 31, p.npT - 1)];           // [jj][ipT], columns past the grid repeat the last
                const int n_tab_b = ncb * p.jtiles;
                const int m_lo = wave_lo(n_tab_b, wv), m_hi = wave_lo(n_tab_b, wv + 1);
                for (int m = m_lo; m < m_hi; m++) {   // (phi tile, cell) order: consecutive tables of one tile stream per wave
                    const int jt = m / ncb, c = m - jt * ncb;
                    {
                        const double bmax = __dmul_rn(pTl, l_bD[c * p.jtiles + jt]);
                        double *t = p.TE + ((int64_t)jt * p.n_cells + (cbase + c)) * NPJ;
                        for (int e = ln; e < NPJ; e += 64) {
                            const int jj = e >> 5;
                            const double pTD = __dmul_rn(pTl, l_Dp[c * J + min(jt * JT + jj, J - 1)]);
                            __builtin_nontemporal_store(exp_full(__dsub_rn(pTD, bmax)), &t[e]);   // whole 512-byte runs: non-temporal measures 0.07 ms better than plain
                        }
                    }
                }
            }
        }
        __syncthreads();
        cs = cs == cs_base ? cs_base + CB : cs_base;
    }
}

// cells per batch: 16 while (9 K + 8 J) doubles per cell stay small (3+1D grids), else 4 (2+1D, 241 eta rows: 1 / 2 / 4 cells per
// batch measure 6.5 / 5.7 / 5.5 ms per 1e5 cells -- occupancy is not what limits it)
static int prep_batch_cells(int K)
{
    if (K > 32) return kPrepCB;
    // dev switch (A/B only): IS3D_PREP_CB3 = 4 | 16 cells per workgroup batch in 3+1D (8 measured between the two)
    static const int env = [] { const char *e = dev_env("IS3D_PREP_CB3"); const int v = e ? atoi(e) : 0; return (v == 4 || v == 16) ? v : 0; }();
    return env ? env : kPrepCB3;
}

size_t prep_lds_bytes(int nT, int nspl, int J, int K, int baryon, int rec, int dim3)
{
    const int cb = prep_batch_cells(K);
    const int nka = baryon ? 9 : 7, nja = baryon ? 8 : 6;
    // the unit-level cull bounds ([cb][jtiles] + [cb][rblocks] <= cb ((J + 1) / 2 + K) doubles) exist for 3+1D grids only -- the kernel's
    // own predicate (p.dim3), whatever K is; a 2+1D eta table of 241 rows must not pay 8 KB for them (two workgroups per CU need <= 80 KB each)
    const size_t bounds = dim3 ? (size_t)cb * ((J + 1) / 2 + K) : 0;
    // + one int2 per element of a unit record (rec doubles; 0 for the flat streams of variant 1)
    return sizeof(double) * ((size_t)nT * (1 + 2 * nspl) + (size_t)cb * (nka * K + nja * J) + bounds + (size_t)rec + 2) + sizeof(CellScal) * cb * 2;   // + 2: descriptor alignment; two CellScal buffers
}

hipError_t launch_prep(const PrepParams &p_in, hipStream_t stream)
{
    if (p_in.n_cells <= 0) return hipSuccess;
    PrepParams p = p_in;
    const int cb = prep_batch_cells(p.K);
    int nbatch = (p.n_cells + cb - 1) / cb;
    int grid = nbatch < 4096 ? nbatch : 4096;
    {   // dev switches (-DIS3D_DEV builds only, errors.h): IS3D_PREP_PAIR = 0 | 1 | 3, read per launch so that one process can alternate
        const char *e = dev_env("IS3D_PREP_PAIR");
        p.pair_writer = e ? atoi(e) : kPrepWriterDefault;
        if (p.pair_writer != 0 && p.pair_writer != 1 && p.pair_writer != 3) p.pair_writer = kPrepWriterDefault;
        const char *k = dev_env("IS3D_PREP_SKIP");
        p.dev_skip = k ? atoi(k) : 0;
        const char *w = dev_env("IS3D_PREP_W0");          // wave 0's share of the units while it runs the next batch's phase 1, in %
        p.w0_share = w ? atoi(w) : 0;
        if (p.w0_share < 1 || p.w0_share > 100) p.w0_share = p.dim3 ? kPrepW0Share3 : kPrepW0Share2;
    }
    size_t lds = prep_lds_bytes(p.spl.n, p.spl.nspl, p.J, p.K, p.baryon, p.tiled ? unit_rec_doubles(p.JT, p.R, p.baryon) : 0, p.dim3 ? 1 : 0);
    const bool nt = (p.dev_skip & 16) != 0;     // dev A/B (IS3D_PREP_SKIP bit 4): record stores non-temporal; default plain, see cf_prep
    if (nt) {
        if (cb == 16) hipLaunchKernelGGL((cf_prep<16, true>), dim3(grid), dim3(kPrepThreads), lds, stream, p);
        else hipLaunchKernelGGL((cf_prep<kPrepCB, true>), dim3(grid), dim3(kPrepThreads), lds, stream, p);
    } else {
        if (cb == 16) hipLaunchKernelGGL((cf_prep<16, false>), dim3(grid), dim3(kPrepThreads), lds, stream, p);
        else hipLaunchKernelGGL((cf_prep<kPrepCB, false>), dim3(grid), dim3(kPrepThreads), lds, stream, p);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// cf_main: shared task decode.  Workgroup = 4 waves = 4 consecutive lane-waves of the same
// (j tile, k tile, cell chunk) "stream"; the workgroups of one stream are placed on one XCD (blocks
// b and b+8 share an XCD under round-robin dispatch; speed only) so the stream is fetched into one L2.
// ------------------------------------------------------------------------------------------------
constexpr int kWPB = kWavesPerBlock;

struct Task {
    int l;            // lane slot
    int jt, kt, chunk;
    bool active;
};

__device__ __forceinline__ Task decode_task(const MainGeom &g)
{
    Task t;
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int grp = q % g.G;
    const int stream = (q / g.G) * 8 + xcd;
    const int wave = threadIdx.x >> 6;
    const int lw = grp * kWPB + wave;
    t.l = lw * 64 + (threadIdx.x & 63);
    t.active = (stream < g.NT) && (lw * 64 < g.Lpad);
    int s = stream;
    t.jt = s % g.jtiles; s /= g.jtiles;
    t.kt = s % g.ktiles; s /= g.ktiles;
    t.chunk = s;
    return t;
}

// ------------------------------------------------------------------------------------------------
// cf_main_direct (variant 1): one phi per wave-task, one exp per evaluation.  Simple reference kernel.
// 3+1D: KT accumulators (a tile of the y axis).  2+1D: one accumulator, loop over the eta table.
// ------------------------------------------------------------------------------------------------
template <bool CE, bool DIM3, bool OUTFLOW, bool REG, int KT>
__global__ void __launch_bounds__(256)
cf_main_direct(const double *__restrict__ S1, const double *__restrict__ S2, const double *__restrict__ S3,
               const double *__restrict__ lane_mT, const double *__restrict__ lane_pT,
               const double *__restrict__ lane_sign, double *__restrict__ partial, MainGeom g)
{
    const Task t = decode_task(g);
    if (!t.active) return;
    const int J = g.J, K = g.K;
    const int j = t.jt;
    const double mT = lane_mT[t.l], pT = lane_pT[t.l], sign = lane_sign[t.l];
    const double mT2 = mT * mT, mTpT = mT * pT, pT2 = pT * pT;
    int c0, c1;
    chunk_cells(g, t.chunk, c0, c1);
    const int k0 = DIM3 ? t.kt * KT : 0;

    double acc[KT];
#pragma unroll
    for (int kk = 0; kk < KT; kk++) acc[kk] = 0.0;

    for (int c = c0; c < c1; c++) {
        const double *r2 = S2 + ((int64_t)c * J + j) * kS2Rec;
        const double B = r2[0], Dp = r2[1], gamma = r2[2], kappa = r2[3];
        const double pTB = pT * B, pTD = pT * Dp, pT2g = pT2 * gamma;
        const double *r1 = S1 + (int64_t)c * K * kS1Rec;
        const double *r3 = S3 + ((int64_t)c * J + j) * K;
        if (DIM3) {
#pragma unroll
            for (int kk = 0; kk < KT; kk++) {
                const int k = k0 + kk;
                if (k < K) {
                    const double A = r1[k * kS1Rec + 0], Cp = r1[k * kS1Rec + 1], alpha = r1[k * kS1Rec + 2];
                    const double beta = r3[k];
                    double pds = __builtin_fma(mT, A, pTB);
                    double x = __builtin_fma(mT, Cp, -pTD);
                    double f; int n;
                    exp_core(-x, f, n);
                    double z = ldexp_fast(f, n);
                    double br = __builtin_fma(mT2, alpha, __builtin_fma(mTpT, beta, pT2g));
                    acc[kk] += eval_tail<CE, OUTFLOW, REG>(z, x, br, kappa, sign, pds);
                }
            }
        } else {
            // eta quadrature: KT independent partial sums for ILP
            for (int kb = 0; kb < K; kb += KT) {
#pragma unroll
                for (int kk = 0; kk < KT; kk++) {
                    const int k = kb + kk;
                    if (k < K) {
                        const double A = r1[k * kS1Rec + 0], Cp = r1[k * kS1Rec + 1], alpha = r1[k * kS1Rec + 2], W = r1[k * kS1Rec + 3];
                        const double beta = r3[k];
                        double pds = __builtin_fma(mT, A, pTB * W);
                        double x = __builtin_fma(mT, Cp, -pTD);
                        double f; int n;
                        exp_core(-x, f, n);
                        double z = ldexp_fast(f, n);
                        double br = __builtin_fma(mT2, alpha, __builtin_fma(mTpT, beta, pT2g));
                        acc[kk] += eval_tail<CE, OUTFLOW, REG>(z, x, br, kappa, sign, pds);
                    }
                }
            }
        }
    }

    const int64_t JKacc = (int64_t)J * g.Kacc;
    double *pp = partial + (int64_t)t.chunk * JKacc * g.Lpad;
    if (DIM3) {
#pragma unroll
        for (int kk = 0; kk < KT; kk++) {
            const int k = k0 + kk;
            if (k < K) {
                double *o = pp + ((int64_t)j * g.Kacc + k) * g.Lpad + t.l;
                *o = g.first_pass ? acc[kk] : (*o + acc[kk]);
            }
        }
    } else {
        double s = 0.0;
#pragma unroll
        for (int kk = 0; kk < KT; kk++) s += acc[kk];
        double *o = pp + (int64_t)j * g.Lpad + t.l;
        *o = g.first_pass ? s : (*o + s);
    }
}

// ------------------------------------------------------------------------------------------------
// cf_main_tile (variant 2, default): factorised exponential on a JT x KT tile of (phi, y) bins.
//
//   exp(-p.u/T) = exp(-(mT Cp_k - bmax)) * exp(pT Dp_j - bmax),   bmax = max_j pT Dp_j  (per lane, cell)
// One exponential per (cell, k) and one per (cell, j) instead of one per (cell, j, k); an evaluation
// costs a single multiply for z.  Both factors lie in (0, 1]: since p.u >= 0 for every j of the tile,
// mT Cp_k >= bmax, and a factor can only underflow when z itself is below 1e-307 (negligible).
//
// Per evaluation (14-moment): 12 fp64 VALU ops + v_rcp_f64:
//   z = E1*E2; d = fma(sign,z,1); r = rcp+1 Newton (2e-15); br = fma(mTpT,beta, a_k + g_j);
//   u = clamp01(fma(r, br/2, 1/2))  -- regulate_deltaf folded into the VOP3 clamp modifier: (1+df)/2 in [0,1];
//   acc += max(pds,0) * (z*r) * u    (the factor 2 is restored once, when the accumulators are stored).
// 3+1D: JT*KT accumulators.  2+1D: JT accumulators, loop over the whole eta table.
// ------------------------------------------------------------------------------------------------
// The tile's coefficients are staged through LDS: the 4 waves of a workgroup (4 different lane-waves,
// same tile and cell chunk) stream ONE contiguous run of unit records (cf_device.h) with coalesced
// 16-byte loads, double-buffered (global -> registers while the previous batch is being consumed ->
// LDS), one barrier per batch.  Operands then come from wave-uniform ds_read_b128 broadcasts that the
// compiler can hoist far ahead of their use, so no evaluation waits on memory.  (The first version
// used scalar loads; each s_load sat 4 instructions in front of its s_waitcnt and the kernel ran at
// ~60 % of its issue bound.)
template <bool CE, bool DIM3, bool OUTFLOW, bool REG, bool BARYON, int JT, int R, bool LAZY = false, bool DMA = true>
// The 14-moment 2+1D kernel on the 8 x 31 tile (8 accumulators, no staging registers since the direct-to-LDS copy) fits 168 VGPRs without
// a spill: three waves per SIMD (config 2: 31.5 -> 29.9 ms).  Everything else is a two-wave kernel (the Chapman-Enskog form of the same
// tile would spill 42 registers at 168), and the compiler is told not to trade the schedule for a third wave it cannot reach.
__global__ void __launch_bounds__(512)
__attribute__((amdgpu_waves_per_eu((!DIM3 && !BARYON && JT <= 8 && R <= 32 && DMA) ? 3 : 2, (!DIM3 && !BARYON && JT <= 8 && R <= 32 && DMA) ? 3 : 2)))
cf_main_tile(const double *__restrict__ TS, const double *__restrict__ lane_mT, const double *__restrict__ lane_pT,
             const double *__restrict__ lane_sign, const double *__restrict__ lane_b, double *__restrict__ partial,
             unsigned long long *__restrict__ stats, MainGeom g, const int32_t *__restrict__ lane_pe,
             const int32_t *__restrict__ lane_sub)
{
    // unit record layout (cf_device.h); BARYON records carry alpha_B after the header and L_k in every row
    constexpr int HDR = 4 * JT + (BARYON ? 2 : 0);
    constexpr int RS = BARYON ? 6 : 4;          // row scalars before the beta entries
    constexpr int REC = HDR + R * (RS + JT);
    constexpr int UB0 = (1536 / REC) > 0 ? (1536 / REC) : 1;  // units per batch (about 12 KB)
    constexpr int UB = (!DIM3 && R <= 32) ? ((UB0 + 3) / 4) * 4 : UB0;   // short 2+1D units: whole groups of 4 (unit-strided lanes, below)
    constexpr int BUF2 = UB * REC / 2;                       // double2 per buffer
    constexpr int NLD = (BUF2 + 127) / 128;                 // staging loads per thread for the smallest workgroup (2 waves)
    constexpr bool EARLY = LAZY;   // staging order, see the batch loop (eager tiles: 664 -> 666 ms, 2+1D 40.2 -> 42.0 ms with it)
    // evaluations per shared reciprocal (A/B on config 3 / config 2): 3+1D 8 x 7: 1 -> 685, 2 -> 651, 4 -> 637,
    // 8 -> 716 ms (spills); 2+1D 8 x 61: 1 -> 44.1, 2 -> 41.5, 4 -> 40.7, 8 -> 40.2 ms
    // (the Chapman-Enskog 8 x 31 tile takes four evaluations per reciprocal so that it, too, fits three waves per SIMD)
    constexpr int RB = DIM3 ? (JT % 4 == 0 ? 4 : (JT % 3 == 0 ? 3 : 2)) : ((JT % 8 == 0 && !(CE && !BARYON && R <= 32 && DMA)) ? 8 : 4);
    static_assert(JT % RB == 0, "the phi tile is a whole number of reciprocal batches");
    static_assert(REC % 2 == 0, "unit records must be 16-byte multiples (JT even)");
    // DMA (the default since round 2; DMA = false keeps the register-staged copy for A/B, variant 8): the next batch is staged by
    // direct-to-LDS loads -- no staging registers (28 VGPRs in 2+1D), no ds_write, nothing waits before the batch's closing
    // barrier: config 2 31.5 -> 29.9 ms, bitwise the same spectrum.  (With four evaluations per reciprocal the 2+1D kernel then
    // fits 146 VGPRs, three waves per SIMD: 30.5 ms -- the third wave is worth less than the eight-wide reciprocal batch.)
    constexpr int BUFP = DMA ? ((BUF2 * 16 + 1023) / 1024) * 64 : BUF2;   // DMA: whole 1-KiB pieces (64 double2 each), the last one over-reads
    __shared__ double2 lbuf[2][BUFP + (RS + JT) / 2 + 1];   // + one row of pad: the 2+1D row prefetch reads one row ahead

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int grp = q % g.G;
    const int stream = (q / g.G) * 8 + xcd;
    if (stream >= g.NT) return;  // uniform for the whole workgroup
    int sidx = stream;
    const int jt = sidx % g.jtiles; sidx /= g.jtiles;
    const int kt = sidx % g.ktiles; sidx /= g.ktiles;
    const int chunk = sidx;
    const int nthr = blockDim.x;                            // 128, 256 or 512: g.wpb lane-waves share the stream
    const int lw = grp * g.wpb + (tid >> 6);
    const bool wave_active = lw * 64 < g.Lpad;
    const int l = wave_active ? lw * 64 + (tid & 63) : 0;

    const int J = g.J, K = g.K;
    const double mT = lane_mT[l], pT = lane_pT[l], sign = lane_sign[l];
    // Unit-strided lanes (2+1D, g.split = S > 1): a surface with few momentum bins (3 species x 32 pT = 96 lanes) would leave a
    // quarter of its two waves idle.  Instead every bin gets S lane slots (bin, s), slot s takes the units u = s (mod S) of the
    // stream -- all units of a stream add into the same JT accumulators of a bin, so any partition of them is a partition of
    // the sum -- and 96 x 4 = 384 lanes are six full waves.  The operands of a lane then come from ITS unit: the LDS addresses
    // carry a per-lane offset (they are vector addresses anyway), the instruction stream is unchanged.  cf_finalize adds the S
    // slots of a bin in slot order.
    const int S = (!DIM3 && g.split > 1) ? g.split : 1;
    const int sub_off = (!DIM3 && lane_sub) ? lane_sub[l] * REC : 0;
    const double hs = REG ? 0.5 : 1.0;  // u = (1 + df) * hs
    const double mT2s = hs * mT * mT, mTpTs = hs * mT * pT, pT2s = hs * pT * pT;
    const double bq = BARYON ? lane_b[l] : 0.0;              // baryon number of the lane's species class
    const double hbmT = hs * bq * mT, hbpT = hs * bq * pT;
    int c0, c1;
    chunk_cells(g, chunk, c0, c1);
    const int n_units = (c1 - c0) * g.upc;
    const int s_tile = DIM3 ? (jt * g.ktiles + kt) : jt;
    const double2 *src = (const double2 *)(TS + (((int64_t)s_tile * g.n_cells + c0) * g.upc) * REC);
    const int nb = (n_units + UB - 1) / UB;

    constexpr int NACC = DIM3 ? JT * R : JT;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.0;

    int n_rows = 0, n_dead = 0, n_calls = 0;   // wave-uniform bookkeeping for the executed-work accounting of bench.py
    // Row culling threshold on earg = log E1.  Always: exp(earg) == +0 below -745.2.  With OUTFLOW && REG (g.zskip == 2) also
    // relative to the accumulators: every term of a row is pds w with 0 <= pds <= 1 (clamped, scaled stream), 0 <= u <= 1 and
    // z r <= 2 z <= 2 E1 (E2 <= 1; 1/(1 + sign z) <= 2 for z <= 1/2), the accumulators only grow, and fma(pds, w, acc) == acc
    // exactly whenever pds w < ulp(acc)/2.  So a row with 2 E1 < 2^-55 min(acc) for all 64 lanes changes no bit of any
    // accumulator and is skipped; min(acc) >= 2^(e-1) with e the frexp exponent of a (stale, hence still valid) minimum over
    // the lane's accumulators, refreshed once per batch.  On config 3 this culls 62 % of the wave-rows instead of 27 %.
    constexpr bool RELCULL = OUTFLOW && REG;
    double cull_thr = -745.2;
    // the lane's own bound on the scaled p.dsigma: pds <= max(mT/mTmax, pT/pTmax) < 2^pe (pe <= 0, from the plan), in place of pds <= 1
    const int pe = (RELCULL && lane_pe) ? lane_pe[l] : 0;
    auto process_unit = [&](const double *U) {
        if constexpr (!BARYON && DIM3 && JT >= 2) {
            // unit-level cull: earg_k = bmax - mT C'_k <= pT Dmax - mT Cmin for every row of the unit (both from cf_prep; the
            // roundings are monotone), so if that bound is below the threshold for the whole wave every row would be culled:
            // skip the unit's header work (JT exponentials) and its R row exponentials as well
            const double eu = __dsub_rn(__dmul_rn(pT, U[3]), __dmul_rn(mT, U[7]));   // the roundings of the row test, no contraction
            if (g.zskip && __all(eu < cull_thr)) { n_rows += R; n_dead += R; return; }
        }
        double pTB[JT], pTD[JT], pT2g[JT], E2[JT];
        double bmax = -1.0e300;
#pragma unroll
        for (int jj = 0; jj < JT; jj++) {
            pTB[jj] = pT * U[4 * jj + 0];
            pTD[jj] = pT * U[4 * jj + 1];
            pT2g[jj] = pT2s * U[4 * jj + 2];
            if (BARYON) pT2g[jj] = __builtin_fma(hbpT, U[4 * jj + 3], pT2g[jj]);   // + hs b pT L2_j
            bmax = __builtin_fmax(bmax, pTD[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < JT; jj++) E2[jj] = exp_p9(pTD[jj] - bmax);
        // Rows are software-pipelined by hand: the operands and the exponential of row r+1 are fetched / computed
        // before the evaluations of row r, so that neither LDS latency nor the 13-deep FMA chain of exp_core sits
        // at the head of a row.  Exact-zero culling: exp(earg) == +0 below -745.2; then z = 0 for every phi of the
        // tile and the row adds exactly +0 to its accumulators (what the reference computes as 1/(inf + sign) = 0).
        // When that holds for all 64 lanes of the wave the row's evaluations are skipped: bitwise the same result,
        // and on wide (y, eta) surfaces a quarter of all rows (high pT x large |y - eta|) go this way.
        const double baB = BARYON ? bq * U[4 * JT] : 0.0;   // b mu_B / T: f_eq = 1/(exp(x - b alpha_B) + sign)
        constexpr int RW = RS + JT;
        // LAZY: a Row carries only what the exponential of the NEXT row needs (Cp -> mTC, E1) plus the row's address, and the
        // other operands are read from LDS when the row is evaluated; otherwise the whole row is prefetched into registers.
        struct RowE { double v[RW]; double mTC, E1; bool live; };
        struct RowL { const double *v; double mTC, E1; bool live; };
        using Row = typename std::conditional<LAZY, RowL, RowE>::type;
        auto fetch = [&](Row &rw, const double *row) {
            if constexpr (LAZY) rw.v = row;
            else {
#pragma unroll
                for (int i = 0; i < RW; i++) rw.v[i] = row[i];
            }
            rw.mTC = mT * row[1];
            const double earg = BARYON ? (bmax - rw.mTC) + baB : bmax - rw.mTC;
            rw.live = !(g.zskip && __all(earg < cull_thr));
            n_rows += 1;
            n_dead += rw.live ? 0 : 1;
            rw.E1 = exp_p9(earg);
        };
        auto evals = [&](const Row &rw, int r) {
            const double mTA = mT * rw.v[0];
            const double mT2a = BARYON ? __builtin_fma(hbmT, rw.v[4], mT2s * rw.v[2]) : mT2s * rw.v[2];   // + hs b mT L_k
            const double W = rw.v[3], mTC = rw.mTC, E1 = rw.E1;
            // The reciprocals of RB evaluations share one v_rcp_f64 (rcp_batch, cf_math.h).  q = (1 + sign z) x lies in (1e-6, 2e9)
            // (cf_prep refuses p.u/T > 1e9), so a product of 8 neither overflows nor underflows.
#pragma unroll
            for (int j0 = 0; j0 < JT; j0 += RB) {
                double zz[RB], xx[RB], q[RB], inv[RB];
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    zz[i] = E1 * E2[j0 + i];
                    const double d = __builtin_fma(sign, zz[i], 1.0);
                    // Chapman-Enskog: df = br / ((1 + sign z) x) with the kappa x term already inside br (cf_prep), so one
                    // reciprocal Rc = 1/((1 + sign z) x) gives df = br Rc and 1/(1 + sign z) = Rc x.
                    if (CE) {
                        xx[i] = mTC - pTD[j0 + i];
                        q[i] = d * xx[i];
                    } else q[i] = d;
                }
                rcp_batch<RB>(q, inv);
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int jj = j0 + i;
                    const double beta = rw.v[RS + jj];
                    // |p.dsigma 2^-e| <= 1 (cf_device.h), so Theta(p.dsigma) p.dsigma is the clamp modifier of the instruction that forms it
                    double pds;
                    if (OUTFLOW) pds = DIM3 ? add_clamp01(mTA, pTB[jj]) : fma_clamp01(pTB[jj], W, mTA);
                    else pds = DIM3 ? (mTA + pTB[jj]) : __builtin_fma(pTB[jj], W, mTA);
                    const double br = __builtin_fma(mTpTs, beta, mT2a + pT2g[jj]);
                    const double dfr = inv[i];
                    const double rr = CE ? dfr * xx[i] : dfr;
                    const double u = REG ? fma_clamp01_half(dfr, br) : __builtin_fma(dfr, br, 1.0);
                    const double w = (zz[i] * rr) * u;
                    if (DIM3) acc[jj * R + r] = __builtin_fma(pds, w, acc[jj * R + r]);
                    else acc[jj] = __builtin_fma(pds, w, acc[jj]);
                }
            }
        };
        const double *rows = U + HDR;
        Row cur, nxt;
        if constexpr (DIM3 || R > 32) fetch(cur, rows);
        if (DIM3) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r + 1 < R) fetch(nxt, rows + (r + 1) * RW);
                if (cur.live) evals(cur, r);
                if (r + 1 < R) cur = nxt;
            }
        } else if constexpr (R <= 32) {
            // variant 7 (short units): no hand pipeline -- a row tests its liveness, then computes its exponential inside its own
            // block, where the dependent chain overlaps the E1-independent part of the row's evaluations (what made variant 6
            // faster than variant 5 in 3+1D)
#pragma clang loop unroll_count(2)
            for (int r = 0; r < R; r++) {
                Row rw;
                const double *row = rows + r * RW;
                if constexpr (LAZY) rw.v = row;
                else {
#pragma unroll
                    for (int i = 0; i < RW; i++) rw.v[i] = row[i];
                }
                rw.mTC = mT * row[1];
                const double earg = BARYON ? (bmax - rw.mTC) + baB : bmax - rw.mTC;
                n_rows += 1;
                if (g.zskip && __all(earg < cull_thr)) { n_dead += 1; continue; }
                rw.live = true;
                rw.E1 = exp_p9(earg);
                evals(rw, 0);
            }
        } else {
            // eta quadrature rows all feed the same JT accumulators: the loop stays ROLLED, two rows per trip with the
            // two Row registers swapping roles (code size: left to itself the compiler unrolls all R = 61 rows into 78 KB of
            // code, more than the instruction cache).  For even R the last fetch reads the head of the next unit (or the pad
            // behind the buffer); it is never evaluated.
#pragma clang loop unroll(disable)
            for (int r = 0; r + 1 < R; r += 2) {
                fetch(nxt, rows + (r + 1) * RW);
                if (cur.live) evals(cur, 0);
                fetch(cur, rows + (r + 2) * RW);
                if (nxt.live) evals(nxt, 0);
            }
            if (R & 1) {
                if (cur.live) evals(cur, 0);
            }
        }
    };

    // double-buffered staging: batch ib+1 is copied global -> LDS by every wave BEFORE it consumes batch ib (the other buffer
    // was released by the barrier that ended batch ib-1).  The copy's latency is covered by the other waves of the SIMD;
    // holding the batch in registers across the evaluations instead costs 4 NLD VGPRs that the 8 x 7 tile does not have.
    if constexpr (DMA) {
        // whole 1-KiB pieces, contiguous ranges per wave, four per address (as in cf_main_tile3e); the last piece of a batch and
        // the pieces of a short last batch over-read the stream (slack behind TS) into the pad / unused units of the buffer
        constexpr int NP = BUFP / 64;
        auto stage = [&](int ib, int buf) { stage_pieces<NP>((const char *)(src + (int64_t)ib * BUF2), lbuf[buf], tid, nthr); };
        // The lane constants must have ARRIVED before the batch loop: the compiler sinks the loads of restrict-qualified data to their first
        // use and waits for them there (s_waitcnt vmcnt(0) in every row of the unrolled loop) -- a wait that also covers the direct-to-LDS
        // loads of the next batch, which it does not know about, i.e. it would expose the staging latency in every batch
        asm volatile("" :: "v"(mT), "v"(pT), "v"(sign), "v"(mT2s), "v"(mTpTs), "v"(pT2s), "v"(hbmT), "v"(hbpT), "v"(bq), "v"(sub_off), "v"(pe) : "memory");
        if (nb > 0) {
            stage(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int ib = 0; ib < nb; ib++) {
                if (ib + 1 < nb) stage(ib + 1, (ib + 1) & 1);
                if (wave_active) {
                    const int nu = min(UB, n_units - ib * UB);
                    const double *base = (const double *)lbuf[ib & 1] + sub_off;
                    for (int u = 0; u < nu; u += S) { process_unit(base + u * REC); n_calls++; }
                    if (RELCULL && g.zskip == 2 && (((ib + 1) & ib) == 0 || (ib & 31) == 31)) {
                        double m = acc[0];
#pragma unroll
                        for (int i = 1; i < NACC; i++) m = __builtin_fmin(m, acc[i]);
                        const int e = __builtin_amdgcn_frexp_exp(m);
                        cull_thr = (m > 1.0e-290) ? __builtin_fmax(-745.2, (double)(e - 58 - pe) * 0.6931471805599453) : -745.2;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    } else if (nb > 0) {
        {
            const int n2 = min(UB, n_units) * (REC / 2);
#pragma unroll
            for (int t = 0; t < NLD; t++) {
                const int i = tid + t * nthr;
                if (i < n2) lbuf[0][i] = src[i];
            }
        }
        __syncthreads();
        for (int ib = 0; ib < nb; ib++) {
            const bool more = ib + 1 < nb;
            const int n2next = more ? min(UB, n_units - (ib + 1) * UB) * (REC / 2) : 0;
            const double2 *s2 = src + (int64_t)(ib + 1) * BUF2;
            if (EARLY) {
#pragma unroll
                for (int t = 0; t < NLD; t++) {
                    const int i = tid + t * nthr;
                    if (i < n2next) lbuf[(ib + 1) & 1][i] = s2[i];
                }
            }
            double2 pre[NLD];
            if (!EARLY) {
#pragma unroll
                for (int t = 0; t < NLD; t++) {
                    const int i = tid + t * nthr;
                    pre[t] = (i < n2next) ? s2[i] : double2{0.0, 0.0};
                }
            }
            if (wave_active) {
                const int nu = min(UB, n_units - ib * UB);
                const double *base = (const double *)lbuf[ib & 1] + sub_off;
                for (int u = 0; u < nu; u += S) { process_unit(base + u * REC); n_calls++; }   // nu is a multiple of S (plan)
                // the threshold follows log2 of the accumulators, which grow about linearly with the cells seen: refresh after batches
                // 0, 1, 3, 7, 15, ... and every 32nd
                if (RELCULL && g.zskip == 2 && (((ib + 1) & ib) == 0 || (ib & 31) == 31)) {
                    double m = acc[0];
#pragma unroll
                    for (int i = 1; i < NACC; i++) m = __builtin_fmin(m, acc[i]);
                    const int e = __builtin_amdgcn_frexp_exp(m);                     // m = f 2^e, f in [0.5, 1)
                    cull_thr = (m > 1.0e-290) ? __builtin_fmax(-745.2, (double)(e - 58 - pe) * 0.6931471805599453) : -745.2;
                }
            }
            if (!EARLY) {
#pragma unroll
                for (int t = 0; t < NLD; t++) {
                    const int i = tid + t * nthr;
                    if (i < n2next) lbuf[(ib + 1) & 1][i] = pre[t];
                }
            }
            __syncthreads();
        }
    }
    if (!wave_active) return;
    if ((tid & 63) == 0) {
        // the 2+1D loop fetches one row past each unit; those are not rows of the surface
        const int fetched = (DIM3 || (R & 1) || R <= 32) ? n_rows : n_rows - n_calls;
        atomicAdd(&stats[2], (unsigned long long)fetched);
        atomicAdd(&stats[3], (unsigned long long)min(n_dead, fetched));
    }

    const double unscale = REG ? 2.0 : 1.0;
    const int64_t JKacc = (int64_t)J * g.Kacc;
    double *pp = partial + (int64_t)chunk * JKacc * g.Lpad;
#pragma unroll
    for (int jj = 0; jj < JT; jj++) {
        const int j = jt * JT + jj;
        if (j < J) {
            if (DIM3) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int k = kt * R + r;
                    if (k < K) {
                        double *o = pp + ((int64_t)j * g.Kacc + k) * g.Lpad + l;
                        const double v = unscale * acc[jj * R + r];
                        *o = g.first_pass ? v : (*o + v);
                    }
                }
            } else {
                double *o = pp + (int64_t)j * g.Lpad + l;
                const double v = unscale * acc[jj];
                *o = g.first_pass ? v : (*o + v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// cf_main_tile3e (variant 5): cf_main_tile for 3+1D without baryon slots, with the phi-side exponentials read from the E2
// table stream (cf_device.h "TE") instead of being computed by every lane: the JT header exponentials per live unit
// (16 % of all issued VALU instructions of the variant-3 kernel on config 3) become JT/2 ds_read_b128.  Everything else --
// unit records, unit- and row-level culling, the shared reciprocals, the partial layout -- is cf_main_tile<.., LAZY = true>.
// LDS: two buffers of g.ub units, each unit REC doubles of record + npT*JT doubles of table (dynamic, sized by the launcher).
// ------------------------------------------------------------------------------------------------
// dev: cycle accounting of the PROF instantiation (IS3D_DEV_PROF=1, tools/gpu_ab.py): sums over waves of s_memtime intervals
//   [0] waves  [1] wave lifetime  [2] staging issue  [3] vmcnt + barrier waits  [4] dead units  [5] live units  [6] their headers + row tests
//   [7] dead units (count)  [8] live units (count)  [9] first-wave share of [3]  [10] threshold refresh  [11] vmcnt part of [3]  [12] prologue
__device__ unsigned long long g_prof3e[16];

// E2G (variant 10, round 5, developer build): the E2 tables do NOT go through LDS -- a lane fetches its own column (8 doubles per unit, plain global
// loads from the table stream through L2) one unit ahead into registers; the LDS batches hold records only (3.2 x as many units per batch).
// RAWH (round 5): the header values B_j, D'_j, gamma_j stay RAW in their registers and enter the evaluations as FMA operands -- p.dsigma = fma(pT, B_j, mT A)
// with the clamp, x = fma(-pT, D'_j, mT C'), the quadratic form's constant part fma(pT^2, gamma_j, mT^2 alpha) -- instead of being multiplied by the
// lane's pT / pT^2 once per live unit (24 multiplications per unit, two more per row in exchange: the per-evaluation count is the same, an FMA each).
// E2L (variant 12, round 5, developer build): the E2 tables are neither streamed nor staged -- the workgroup BUILDS the tables of a batch in LDS from the
// staged records' D'_j and Dmax (the expression of cf_prep's table writer, rounding for rounding: bitwise the same tables), one table region beside
// the two record buffers, one more barrier per batch.
template <bool CE, bool OUTFLOW, bool REG, int JT, int R, int MODE = 0, bool PROF = false, bool BARYON = false, bool E2G = false, bool RAWH = false, bool E2L = false>
__global__ void __launch_bounds__(512)
cf_main_tile3e(const double *__restrict__ TS, const double *__restrict__ TE, const double *__restrict__ lane_mT,
               const double *__restrict__ lane_pT, const double *__restrict__ lane_sign, const int32_t *__restrict__ lane_ipT,
               double *__restrict__ partial, unsigned long long *__restrict__ stats, MainGeom g, const int32_t *__restrict__ lane_pe,
               const double *__restrict__ lane_b, const double *__restrict__ cull_floor, const double *__restrict__ pTgrid = nullptr)
{
    // "B" records (include_baryon, cf_device.h): alpha_B and Dmax behind the header, {L_k, Cmin (row 0)} behind W in every row, L2_j in
    // the header's x; the lane's baryon number b enters the exponent (b alpha_B) and the b-linear part of df (cf_main_tile)
    constexpr int HDR = 4 * JT + (BARYON ? 2 : 0), RS = BARYON ? 6 : 4, RW = RS + JT;
    constexpr int REC = HDR + R * RW;
    constexpr int RB = JT % 4 == 0 ? 4 : (JT % 3 == 0 ? 3 : 2);
    static_assert(JT % RB == 0 && JT % 2 == 0, "phi tile: whole reciprocal batches, 16-byte records");
    // MODE 0 (variant 5): hand-pipelined rows; 1 (variant 6): the rows of a unit tested for liveness before their exponentials.
    // (Tried and dropped, round 2: the liveness of all rows of a BATCH of units in one pipelined pass before the batch, masks in
    // SGPRs -- 356.6 against 355.9 ms: the pass costs what it saves.)
    // (Tried and dropped, round 2: a 6 x 4 tile in 166 VGPRs, three waves per SIMD -- 408 against 353 ms, with culling off 835
    // against 691: the third wave does not pay for the exponentials and headers amortised over fewer evaluations.)
    // (Tried and dropped, round 2: a ring of three LDS buffers with one progress word per wave in place of the per-batch
    // barrier -- bitwise the same spectrum, 360 against 352 ms: with LDS for 12 units in all, a wave can run at most one
    // 4-unit batch ahead of its partner, and the polling costs more than that slack returns.  And the fine-grained form: 13
    // single-unit slots, wave w stages the units u = w (mod 2) whole, landed / consumed counts in LDS words, no barrier in the
    // loop -- 381 against 346 ms (stage 8.4 %, wait 9.5 % of the wave cycles): a wave can only know "all but my newest unit have
    // landed", two units after their issue instead of a batch later, and that wait plus the publish lag eat the slack.  Two
    // traps met on the way, both because the compiler does not know about the written-out direct-to-LDS loads: a release store
    // or acquire fence at workgroup scope, and a compiler-placed s_waitcnt vmcnt(n) for a lane constant whose load it had sunk
    // into the loop, each drain the staging pipeline every unit (405 ms) -- flags must be relaxed LDS accesses behind the
    // kernel's own waits, and the lane constants must be forced to arrive before the loop.  The three-buffer batch ring again
    // with relaxed flags: 342.8 against 335.8 ms, culling off 713.9 against 677.9.)
    // (Tried and dropped, round 2: one LDS round trip at the head of a unit -- test operands first, the header speculatively behind
    // them, votes while it arrives, dead units drop the reads: 352.1 against 347.9 ms.)
    constexpr bool ROWMASK = MODE >= 1;
    extern __shared__ double2 lds2[];

    const int tid = threadIdx.x;
    // block -> task, XCD-aware (blocks b and b + 8 share an XCD under round-robin dispatch; speed only): all workgroups that
    // read the E2 table of one (phi tile, cell chunk) pair -- its ktiles row-block streams x G lane-wave groups -- sit on ONE XCD,
    // next to each other in that XCD's block order, so the table (and each record stream) is fetched into one L2
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int grp = q % g.G;
    const int m = q / g.G;
    const int kt = m % g.ktiles;
    const int pair = (m / g.ktiles) * 8 + xcd;
    if (pair >= g.jtiles * g.nch_run) return;  // uniform for the whole workgroup
    const int jt = pair % g.jtiles;
    const int chunk = g.ch0 + pair / g.jtiles;
    const int nthr = blockDim.x;
    const int lw = grp * g.wpb + (tid >> 6);
    const bool wave_active = lw * 64 < g.Lpad;
    const int l = wave_active ? lw * 64 + (tid & 63) : 0;

    const int J = g.J, K = g.K;
    constexpr int TEREC = kE2Stride * JT;         // doubles of table per unit ([jj][kE2Stride], the first npT columns used)
    const int UB = g.ub;
    const double mT = lane_mT[l], pT = lane_pT[l], sign = lane_sign[l];
    const int tabrow = lane_ipT[l];               // the lane's column within a unit's table, stored [jj][ipT]: lanes of a wave read
                                                  // consecutive 8-byte words for consecutive pT indices (no LDS bank conflicts)
    const double hs = REG ? 0.5 : 1.0;
    const double mT2s = hs * mT * mT, mTpTs = hs * mT * pT, pT2s = hs * pT * pT;
    const double bq = BARYON ? lane_b[l] : 0.0;
    const double hbmT = hs * bq * mT, hbpT = hs * bq * pT;
    int c0, c1;
    chunk_cells(g, chunk, c0, c1);
    const int n_units = c1 - c0;                  // one unit per cell in 3+1D
    const int s_tile = jt * g.ktiles + kt;
    const double2 *src_ts = (const double2 *)(TS + ((int64_t)s_tile * g.n_cells + c0) * REC);
    const double2 *src_te = (const double2 *)(TE + ((int64_t)jt * g.n_cells + c0) * TEREC);
    const int nb = (n_units + UB - 1) / UB;

    constexpr int NACC = JT * R;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.0;

    int n_rows = 0, n_dead = 0;
    constexpr bool RELCULL = OUTFLOW && REG;      // accumulator-relative culling, see cf_main_tile
    // surface-relative cull: the threshold never falls below what the partial spectrum of the chunks that ran first allows (cf_cull_floor)
    const double thr_floor = (RELCULL && cull_floor && g.zskip == 2) ? cull_floor[(int64_t)s_tile * g.Lpad + l] : -745.2;
    double cull_thr = thr_floor;
    const int pe = (RELCULL && lane_pe) ? lane_pe[l] : 0;

    unsigned long long pf_stage = 0, pf_wait = 0, pf_dead = 0, pf_live = 0, pf_hdr = 0, pf_nd = 0, pf_nl = 0, pf_thr = 0, pf_t0 = 0, pf_u0 = 0, pf_vm = 0, pf_pro = 0, pf_mid = 0, pf_ts = 0;
    if constexpr (PROF) pf_t0 = clock64();
    double e2cur[E2G ? JT : 1], e2nxt[E2G ? JT : 1];              // E2G: the lane's table column of this unit / of the next one (in flight)
    const double *te_lane = TE + ((int64_t)jt * g.n_cells + c0) * TEREC + tabrow;
    auto process_unit = [&](const double *U, const double *tab) -> bool {
        // (both bounds in ONE LDS round trip: read before the test is formed, not under its g.zskip short-circuit)
        const double dmaxv = BARYON ? U[4 * JT + 1] : U[3], cminv = BARYON ? U[HDR + 5] : U[7];
        // ... and, in the same round trip, the operands of the row tests (MODE 1): a live unit then forms its row mask while its header
        // is arriving instead of paying two more round trips for the C'_k; a dead unit drops four reads
        double cn[R];
        if constexpr (ROWMASK) {
#pragma unroll
            for (int r = 0; r < R; r++) cn[r] = U[HDR + r * RW + 1];
            __builtin_amdgcn_sched_barrier(0);
        }
        const double bmax = __dmul_rn(pT, dmaxv);                                        // pT Dmax == max_j pT Dp_j (pT >= 0)
        const double baB = BARYON ? bq * U[4 * JT] : 0.0;                                // b mu_B / T: f_eq = 1/(exp(x - b alpha_B) + sign)
        {
            double eu = __dsub_rn(bmax, __dmul_rn(mT, cminv));                           // unit-level cull, as in cf_main_tile
            if (BARYON) eu += baB;
            const bool dead = __all(eu < cull_thr);
            if (g.zskip && dead) { n_rows += R; n_dead += R; return false; }
        }
        if constexpr (ROWMASK) {
            // live unit: the C'_k are wave-uniform and stay live through the rows (each row's exponential starts from them before the
            // row's own reads return): moved to SGPRs, no VGPR is free there
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(cn[r])), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(cn[r]));
                cn[r] = __hiloint2double((int)hi, (int)lo);
            }
        }
        double pTB[JT], pTD[JT], pT2g[JT], E2[JT];
#pragma unroll
        for (int jj = 0; jj < JT; jj++) {
            if constexpr (RAWH) {   // raw header values (the names keep the products' roles)
                pTB[jj] = U[4 * jj + 0];
                pTD[jj] = U[4 * jj + 1];
                pT2g[jj] = U[4 * jj + 2];
            } else {
                pTB[jj] = pT * U[4 * jj + 0];
                pTD[jj] = pT * U[4 * jj + 1];
                pT2g[jj] = pT2s * U[4 * jj + 2];
            }
            if (BARYON) pT2g[jj] = __builtin_fma(hbpT, U[4 * jj + 3], pT2g[jj]);       // + hs b pT L2_j
            E2[jj] = E2G ? e2cur[jj] : tab[jj * kE2Stride];
        }
        struct Row { const double *v; double mTC, E1; bool live; };
        auto fetch = [&](Row &rw, const double *row) {
            rw.v = row;
            rw.mTC = mT * row[1];
            const double earg = BARYON ? (bmax - rw.mTC) + baB : bmax - rw.mTC;
            rw.live = !(g.zskip && __all(earg < cull_thr));
            n_rows += 1;
            n_dead += rw.live ? 0 : 1;
            rw.E1 = exp_p9(earg);   // degree 9, one-fma reduction (cf_math.h): 7e-14, two instructions fewer per row
        };
        auto evals = [&](const Row &rw, int r) {
            const double mTA = mT * rw.v[0];
            const double mT2a = BARYON ? __builtin_fma(hbmT, rw.v[4], mT2s * rw.v[2]) : mT2s * rw.v[2];   // + hs b mT L_k
            const double mTC = rw.mTC, E1 = rw.E1;
#pragma unroll
            for (int j0 = 0; j0 < JT; j0 += RB) {
                // Chapman-Enskog: with v = z x the shared reciprocal is of q = (1 + sign z) x = x + sign v, and f_eq = z / (1 + sign z)
                // = v / q: v serves both (one multiplication fewer per evaluation than forming 1 + sign z, q and R x separately)
                double zv[RB], qq[RB], inv[RB];
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const double z = E1 * E2[j0 + i];
                    if (CE) {
                        const double x = RAWH ? __builtin_fma(-pT, pTD[j0 + i], mTC) : mTC - pTD[j0 + i];
                        zv[i] = z * x;
                        qq[i] = __builtin_fma(sign, zv[i], x);
                    } else {
                        zv[i] = z;
                        qq[i] = __builtin_fma(sign, z, 1.0);
                    }
                }
                rcp_batch<RB>(qq, inv);
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int jj = j0 + i;
                    const double beta = rw.v[RS + jj];
                    const double pds = RAWH ? (OUTFLOW ? fma_clamp01(pT, pTB[jj], mTA) : __builtin_fma(pT, pTB[jj], mTA))
                                            : (OUTFLOW ? add_clamp01(mTA, pTB[jj]) : (mTA + pTB[jj]));
                    const double br = RAWH ? __builtin_fma(mTpTs, beta, __builtin_fma(pT2s, pT2g[jj], mT2a)) : __builtin_fma(mTpTs, beta, mT2a + pT2g[jj]);
                    const double dfr = inv[i];
                    const double u = REG ? fma_clamp01_half(dfr, br) : __builtin_fma(dfr, br, 1.0);
                    const double w = (zv[i] * dfr) * u;
                    acc[jj * R + r] = __builtin_fma(pds, w, acc[jj * R + r]);
                }
            }
        };
        const double *rows = U + HDR;
        if constexpr (ROWMASK) {
            // Liveness of the R rows first (3 instructions a row), then only the live rows pay their exponential: a dead row inside
            // a live unit costs its test, not the 15-instruction exponential the row pipeline below computes for every row.
            // Within a live row the exponential's dependent chain overlaps the row's E1-independent work (x, p.dsigma, br).
            unsigned live = (1u << R) - 1u;
            if (g.zskip) {
                live = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double earg = BARYON ? (bmax - mT * cn[r]) + baB : bmax - mT * cn[r];
                    live |= __all(earg < cull_thr) ? 0u : (1u << r);
                }
            }
            n_rows += R;
            n_dead += R - __builtin_popcount(live);
            if constexpr (PROF) pf_hdr += clock64() - pf_u0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (live & (1u << r)) {
                    Row rw;
                    rw.v = rows + r * RW;
                    rw.mTC = mT * cn[r];   // C'_k is already here (read with the unit bounds): the exponential starts before the row's own reads return
                    rw.E1 = exp_p9(BARYON ? (bmax - rw.mTC) + baB : bmax - rw.mTC);   // degree 9, one-fma reduction (cf_math.h): 7e-14
                    rw.live = true;
                    evals(rw, r);
                }
            }
        } else {
            Row cur, nxt;
            fetch(cur, rows);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r + 1 < R) fetch(nxt, rows + (r + 1) * RW);
                if (cur.live) evals(cur, r);
                if (r + 1 < R) cur = nxt;
            }
        }
        return true;
    };

    // staging: records and tables of batch ib+1 go global -> LDS by direct-to-LDS loads (global_load_lds_dwordx4: 1 KiB per wave-
    // instruction, no staging registers, nothing waits until the barrier that ends batch ib), issued BEFORE batch ib is consumed.
    // LDS image of a buffer = the two contiguous global runs, the first padded to whole KiB: [UB records | pad][UB tables].
    // Whole 1-KiB pieces only, never predicated (a predicated piece is a v_cmp / saveexec / branch chain per load; the loop below is
    // scalar): the last record piece and the pieces of a short last batch over-read the run -- the next units of the stream, or
    // the slack the plan allocates behind TS and TE -- into the pad / the unused units of the buffer.
    const int TSP = (UB * REC * (int)sizeof(double) + 1023) & ~1023;          // bytes of the padded record part
    const int TEP = (E2G || E2L) ? 0 : ((UB * TEREC * (int)sizeof(double) + 1023) & ~1023);   // bytes of the padded table part
    const int BUFB = TSP + TEP;                                               // bytes per buffer
    const unsigned lane16 = (unsigned)(tid & 63) * 16u;
    // every wave of the workgroup takes a contiguous range of the pieces, four per address: the instruction's immediate offset
    // advances the global and the LDS address together, so a group of four costs one address and one M0 setup
    const int nw_ = nthr >> 6, wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np1 = TSP >> 10, np2 = TEP >> 10;
    const int p1lo = (np1 * wave_) / nw_, p1hi = (np1 * (wave_ + 1)) / nw_;
    const int p2lo = (np2 * wave_) / nw_, p2hi = (np2 * (wave_ + 1)) / nw_;
    auto pieces = [&](const char *g, char *l, int lo, int hi) {
        for (int p = lo; p < hi; p += 4) {
            const char *gp = g + p * 1024 + lane16;
            const int n = hi - p;
            // written-out instruction (glds16a, cf_math.h): with the builtin in flight the compiler's waitcnt pass drains the LDS
            // queue (lgkmcnt(0)) before every use of an LDS read in the whole kernel; this way it counts (692 -> 678 ms with culling off)
            const unsigned lp = __builtin_amdgcn_readfirstlane(lds_addr32(l + p * 1024));
            glds16a<0>(gp, lp);
            if (n > 1) glds16a<1024>(gp, lp);
            if (n > 2) glds16a<2048>(gp, lp);
            if (n > 3) glds16a<3072>(gp, lp);
        }
    };
    auto stage = [&](int ib, int buf) {
        char *dst = (char *)lds2 + (size_t)buf * BUFB;
        const char *s1 = (const char *)src_ts + (int64_t)ib * UB * REC * (int64_t)sizeof(double);
        const char *s2 = (const char *)src_te + (int64_t)ib * UB * TEREC * (int64_t)sizeof(double);
        pieces(s1, dst, p1lo, p1hi);
        if constexpr (PROF) pf_mid = clock64();
        pieces(s2, dst + TSP, p2lo, p2hi);
    };
    auto consume = [&](int ib, int buf) {
        const int nu = min(UB, n_units - ib * UB);
        const double *base = (const double *)((const char *)lds2 + (size_t)buf * BUFB);
        const double *tabs = E2L ? (const double *)((const char *)lds2 + 2 * (size_t)BUFB) + tabrow     // the one table region behind the two record buffers
                                 : (const double *)((const char *)base + TSP) + tabrow;
        for (int u = 0; u < nu; u++) {
            if constexpr (PROF) pf_u0 = clock64();
            if constexpr (E2G) {
                // take over the column requested during the previous unit, request the next unit's (clamped at the chunk's end)
#pragma unroll
                for (int jj = 0; jj < JT; jj++) e2cur[jj] = e2nxt[jj];
                const int un = min(ib * UB + u + 1, n_units - 1);
                const double *t = te_lane + (int64_t)un * TEREC;
#pragma unroll
                for (int jj = 0; jj < JT; jj++) e2nxt[jj] = t[jj * kE2Stride];
            }
            const bool lv = process_unit(base + u * REC, tabs + u * TEREC);
            if constexpr (PROF) {
                const unsigned long long d = clock64() - pf_u0;
                if (lv) { pf_live += d; pf_nl++; } else { pf_dead += d; pf_nd++; }
            }
        }
        unsigned long long pa = 0;
        if constexpr (PROF) pa = clock64();
        // the threshold follows log2 of the accumulators, which grow about linearly with the cells seen: refresh after batches
        // 0, 1, 3, 7, 15, ... and every 64th
        if (RELCULL && g.zskip == 2 && (((ib + 1) & ib) == 0 || (ib & 63) == 63)) {
            double m = acc[0];
#pragma unroll
            for (int i = 1; i < NACC; i++) m = __builtin_fmin(m, acc[i]);
            const int e = __builtin_amdgcn_frexp_exp(m);
            cull_thr = (m > 1.0e-290) ? __builtin_fmax(thr_floor, (double)(e - 58 - pe) * 0.6931471805599453) : thr_floor;
        }
        if constexpr (PROF) pf_thr += clock64() - pa;
    };
    {
        if (nb > 0) {
            stage(0, 0);
            if constexpr (E2G) {
#pragma unroll
                for (int jj = 0; jj < JT; jj++) e2nxt[jj] = te_lane[jj * kE2Stride];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if constexpr (PROF) pf_pro = clock64() - pf_t0;
            for (int ib = 0; ib < nb; ib++) {
                unsigned long long pa = 0;
                if constexpr (PROF) pa = clock64();
                if (ib + 1 < nb) stage(ib + 1, (ib + 1) & 1);
                if constexpr (PROF) { pf_stage += clock64() - pa; if (ib + 1 < nb) pf_ts += pf_mid - pa; }
                if constexpr (E2L) {
                    // the batch's tables from its records (landed at the previous barrier): lanes <-> (unit, jj, ipT), ipT = tid & 31 fixed per lane
                    const int nu = min(UB, n_units - ib * UB);
                    const double *recs = (const double *)((const char *)lds2 + (size_t)(ib & 1) * BUFB);
                    double *tb = (double *)((char *)lds2 + 2 * (size_t)BUFB);
                    const double pTl = pTgrid[min(tid & 31, g.npT - 1)];
                    for (int idx = tid; idx < nu * TEREC; idx += nthr) {
                        const int u = idx / TEREC, e = idx - u * TEREC, jj = e >> 5;
                        const double *Uh = recs + u * REC;
                        tb[idx] = exp_full(__dsub_rn(__dmul_rn(pTl, Uh[4 * jj + 1]), __dmul_rn(pTl, Uh[3])));
                    }
                    __syncthreads();
                }
                if (wave_active) consume(ib, ib & 1);
                if constexpr (PROF) pa = clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the direct-to-LDS loads of batch ib+1 have landed
                if constexpr (PROF) { const unsigned long long pb = clock64(); pf_vm += pb - pa; }
                __syncthreads();
                if constexpr (PROF) pf_wait += clock64() - pa;
            }
        }
    }
    if (!wave_active) return;
    if ((tid & 63) == 0) {
        atomicAdd(&stats[2], (unsigned long long)n_rows);
        atomicAdd(&stats[3], (unsigned long long)min(n_dead, n_rows));
        if constexpr (PROF) {
            atomicAdd(&g_prof3e[0], 1ull);
            atomicAdd(&g_prof3e[1], clock64() - pf_t0);
            atomicAdd(&g_prof3e[2], pf_stage);
            atomicAdd(&g_prof3e[3], pf_wait);
            atomicAdd(&g_prof3e[4], pf_dead);
            atomicAdd(&g_prof3e[5], pf_live);
            atomicAdd(&g_prof3e[6], pf_hdr);
            atomicAdd(&g_prof3e[7], pf_nd);
            atomicAdd(&g_prof3e[8], pf_nl);
            if ((tid >> 6) == 0) atomicAdd(&g_prof3e[9], pf_wait);
            atomicAdd(&g_prof3e[10], pf_thr);
            atomicAdd(&g_prof3e[11], pf_vm);
            atomicAdd(&g_prof3e[12], pf_pro);
            atomicAdd(&g_prof3e[13], pf_ts);
        }
    }
    const double unscale = REG ? 2.0 : 1.0;
    const int64_t JKacc = (int64_t)J * g.Kacc;
    double *pp = partial + (int64_t)chunk * JKacc * g.Lpad;
#pragma unroll
    for (int jj = 0; jj < JT; jj++) {
        const int j = jt * JT + jj;
        if (j < J) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int k = kt * R + r;
                if (k < K) {
                    double *o = pp + ((int64_t)j * g.Kacc + k) * g.Lpad + l;
                    const double v = unscale * acc[jj * R + r];
                    *o = g.first_pass ? v : (*o + v);
                }
            }
        }
    }
}

// LDS bytes per workgroup of cf_main_tile3e: the CU's 160 KB shared by its 8 waves (two per SIMD), 20 KB per wave of the workgroup
constexpr int kTile3eLdsPerWave = 20 * 1024;
static size_t tile3e_lds_bytes(int JT, int R, int ub, int nbuf = 2, int baryon = 0, int e2g = 0)   // nbuf buffers of [ub records, padded to whole KiB][ub tables]
{
    const size_t tsp = ((size_t)ub * unit_rec_doubles(JT, R, baryon) * sizeof(double) + 1023) & ~(size_t)1023;
    const size_t tep = e2g ? 0 : (((size_t)ub * kE2Stride * JT * sizeof(double) + 1023) & ~(size_t)1023);   // E2G / E2L: no staged tables
    return nbuf * (tsp + tep) + (e2g == 2 ? (size_t)ub * kE2Stride * JT * sizeof(double) : 0);                // E2L: one table region, built in place
}
int tile3e_units_per_batch(int JT, int R, int npT, int wpb, int baryon, int e2g)
{
    if (npT > kE2Stride) return 0;
    const size_t budget = (size_t)kTile3eLdsPerWave * (wpb > 0 ? wpb : 1);
    int ub = e2g ? 48 : 16;
    while (ub > 0 && tile3e_lds_bytes(JT, R, ub, 2, baryon, e2g) > budget) ub--;
    return ub;
}
// doubles of slack the plan allocates behind TS and TE: the unpredicated staging pieces of the last batch over-read up to one
// batch of units plus one piece
int tile3e_stream_slack_doubles(int JT, int R) { return 16 * (unit_rec_doubles(JT, R, 1) + kE2Stride * JT) + 128; }

// ------------------------------------------------------------------------------------------------
// cf_main_tile3s (variant 9, round 5; DEVELOPER BUILD ONLY -- measured and dropped): cf_main_tile3e<MODE 1> with the unit records on the SCALAR path.
// Every operand of a unit record is wave-uniform, yet in cf_main_tile3e it travels global -> LDS (direct-to-LDS pieces, ~200 cycles of issue each) ->
// ds_read_b128 -> VGPRs behind a per-batch barrier: 5.2 % of the wave cycles go to issuing the staging loads and 8.3 % to the barrier that pairs two
// waves of unequal work (profiles/r05_cycle_accounting_tile3e.log).  Here a workgroup is ONE wave that reads its record stream through the constant
// address space (s_load_dwordx2 ... x16 into SGPRs, the scalar cache in front of the XCD's L2) and feeds the values to the fp64 instructions as their one
// scalar source: no LDS image of the records, no staging instruction, no barrier, no readfirstlane.
//   * unit bounds {Dmax, Cmin} are loaded two units ahead, a unit's liveness is decided one unit ahead, and only a live unit's C'_k (row tests) and
//     E2 column (the lane's 8 table entries, plain global loads into VGPRs -- the table is the one lane-dependent stream) are fetched, one unit ahead;
//   * a row's scalars are consumed into VGPR values at the head of the row (m_T A, m_T C', the 8 quadratic forms) and the NEXT live row's are
//     requested right behind that: one 22-SGPR set, a whole row (~1100 cycles) of cover, one lgkmcnt(0) per row (scalar loads return out of order);
//   * the header (B_j, D'_j, gamma_j) is requested at the head of a live unit, in front of the row tests that cover most of its latency.
// tools/ubench_smem.hip (profiles/r05_ubench_smem.log) had measured the bare row loop fed this way against the same loop fed from LDS-resident rows:
// + 1 % (L2-resident streams) to + 3-6 % (HBM-resident), against the 13.5 % above.  THE KERNEL: 120.4 against 100.2 ms per 3e5 cells x 305 species
// (237.7 against 201.0 with culling off), spectrum equal to 1.1e-14 (profiles/r05_ab_tile3s.log).  Why, from the ISA:
//   * the SGPR file (102) cannot hold the row in flight (22) + gamma_j of the unit (16) + the next live unit's C'_k (14) + two sets of bounds (8) +
//     addresses and counters (~16) + the constants of the row exponential (26, which the register allocator keeps resident): gamma_j is spilled to the
//     lanes of a VGPR and comes back through 16 v_readlane_b32 in EVERY row -- VALU slots, 12 % of a row's ~130;
//   * a load from the constant address space is rematerialisable: left alone the compiler re-issues the gamma_j loads in every row with an
//     lgkmcnt(0) behind them, which also waits for the next row's request (the asm pin below stops that, and produces the spill above);
//   * 256 VGPRs are needed anyway (a second E2 column for the unit ahead replaces the row operands), two accumulators' worth spill to scratch.
// Holding gamma_j as lane products costs the 16 VGPRs that are not there; building the exponential's constants per use costs 26 SALU instructions per
// live row.  Each way out costs the few per cent the scheme could gain: dropped, kept in the developer build (kernel_variant = 9) for A/B.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) double cdouble;

template <bool CE, bool OUTFLOW, bool REG, int JT, int R>
__global__ void __launch_bounds__(512)   // launched with 64 threads; the bound keeps the register budget at two waves per SIMD (256 VGPRs)
cf_main_tile3s(const double *__restrict__ TS, const double *__restrict__ TE, const double *__restrict__ lane_mT,
               const double *__restrict__ lane_pT, const double *__restrict__ lane_sign, const int32_t *__restrict__ lane_ipT,
               double *__restrict__ partial, unsigned long long *__restrict__ stats, MainGeom g, const int32_t *__restrict__ lane_pe,
               const double *__restrict__ cull_floor)
{
    constexpr int HDR = 4 * JT, RS = 4, RW = RS + JT, REC = HDR + R * RW;
    constexpr int RB = JT % 4 == 0 ? 4 : (JT % 3 == 0 ? 3 : 2);
    constexpr int TEREC = kE2Stride * JT;
    static_assert(JT % RB == 0 && JT % 2 == 0, "phi tile: whole reciprocal batches");
    // block -> task as cf_main_tile3e with one-wave workgroups: the workgroups of one (phi tile, cell chunk) pair on ONE XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int grp = q % g.G;
    const int m = q / g.G;
    const int kt = m % g.ktiles;
    const int pair = (m / g.ktiles) * 8 + xcd;
    if (pair >= g.jtiles * g.nch_run) return;
    const int jt = pair % g.jtiles;
    const int chunk = g.ch0 + pair / g.jtiles;
    if (grp * 64 >= g.Lpad) return;
    const int l = grp * 64 + (int)threadIdx.x;

    const int J = g.J, K = g.K;
    const double mT = lane_mT[l], pT = lane_pT[l], sign = lane_sign[l];
    const int tabrow = lane_ipT[l];
    const double hs = REG ? 0.5 : 1.0;
    const double mT2s = hs * mT * mT, mTpTs = hs * mT * pT, pT2s = hs * pT * pT;
    int c0, c1;
    chunk_cells(g, chunk, c0, c1);
    const int n_units = c1 - c0;
    const int s_tile = jt * g.ktiles + kt;
    cdouble *S = (cdouble *)(TS + ((int64_t)s_tile * g.n_cells + c0) * REC);
    const double *te = TE + ((int64_t)jt * g.n_cells + c0) * TEREC + tabrow;

    constexpr int NACC = JT * R;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.0;
    int n_rows = 0, n_dead = 0;
    constexpr bool RELCULL = OUTFLOW && REG;
    const double thr_floor = (RELCULL && cull_floor && g.zskip == 2) ? cull_floor[(int64_t)s_tile * g.Lpad + l] : -745.2;
    double cull_thr = thr_floor;
    const int pe = (RELCULL && lane_pe) ? lane_pe[l] : 0;
    asm volatile("" :: "v"(mT), "v"(pT), "v"(sign), "v"(mT2s), "v"(mTpTs), "v"(pT2s), "v"(tabrow), "v"(cull_thr) : "memory");   // the lane constants have arrived

    auto unit_live = [&](double dmaxv, double cminv) -> bool {
        const double eu = __dsub_rn(__dmul_rn(pT, dmaxv), __dmul_rn(mT, cminv));       // the unit-level cull of cf_main_tile3e
        // (readfirstlane: the vote is wave-uniform by construction, this tells the compiler -- branches on it are scalar, values set under them stay in SGPRs)
        return !(g.zskip && __builtin_amdgcn_readfirstlane((int)__all(eu < cull_thr)));
    };
    struct RowS { double A, C, al, be[JT]; };
    auto load_row = [&](cdouble *U, int r) -> RowS {
        cdouble *p = U + HDR + r * RW;
        RowS o;
        o.A = p[0]; o.C = p[1]; o.al = p[2];
#pragma unroll
        for (int jj = 0; jj < JT; jj++) o.be[jj] = p[RS + jj];
        return o;
    };

    // SGPR budget (102): gamma_j of the unit 16, the row in flight 22, the next live unit's C'_k 14, bounds 8, addresses and counters ~16; the
    // exponential's constants are rematerialised where they do not fit.  VGPRs: 112 accumulators, 16 + 16 E2 columns (this unit, next live unit),
    // pT B_j and pT D'_j 32, the row's 8 quadratic forms 16, lane constants 14, the reciprocal batch ~36.
    double E2[JT], E2n[JT], cn[R];
    double dmax_cur = 0.0, dmax_nxt = 0.0, cmin_nxt = 0.0;
    bool live_cur = false, live_nxt = false;
    auto fetch_cn = [&](int u) {
        cdouble *U = S + (int64_t)u * REC;
#pragma unroll
        for (int r = 0; r < R; r++) cn[r] = U[HDR + r * RW + 1];
    };
    auto fetch_e2n = [&](int u) {
        const double *t = te + (int64_t)u * TEREC;
#pragma unroll
        for (int jj = 0; jj < JT; jj++) E2n[jj] = t[jj * kE2Stride];
    };
    // what is requested one unit ahead, issued where the SGPRs for it are free: the bounds of unit u + 2 and, if unit u + 1 is live, its C'_k and E2 column
    auto request_ahead = [&](int u) {
        if (u + 2 < n_units) { cdouble *U2 = S + (int64_t)(u + 2) * REC; dmax_nxt = U2[3]; cmin_nxt = U2[7]; }
        if (live_nxt) { fetch_cn(u + 1); fetch_e2n(u + 1); }
    };
    auto refresh_threshold = [&]() {
        double mn = acc[0];
#pragma unroll
        for (int i = 1; i < NACC; i++) mn = __builtin_fmin(mn, acc[i]);
        const int e = __builtin_amdgcn_frexp_exp(mn);
        cull_thr = (mn > 1.0e-290) ? __builtin_fmax(thr_floor, (double)(e - 58 - pe) * 0.6931471805599453) : thr_floor;
    };
    if (n_units > 0) {
        dmax_cur = S[3];
        const double cmin_cur = S[7];
        live_cur = unit_live(dmax_cur, cmin_cur);
        if (live_cur) {
            fetch_cn(0);
            const double *t = te;
#pragma unroll
            for (int jj = 0; jj < JT; jj++) E2[jj] = t[jj * kE2Stride];
        }
        if (n_units > 1) { cdouble *U1 = S + REC; dmax_nxt = U1[3]; cmin_nxt = U1[7]; }
    }
    for (int u = 0; u < n_units; u++) {
        cdouble *U = S + (int64_t)u * REC;
        // liveness of unit u + 1: its bounds were requested during unit u - 1
        const double dmax_u1 = dmax_nxt;
        live_nxt = (u + 1 < n_units) && unit_live(dmax_nxt, cmin_nxt);
        if (live_cur) {
            // header: B_j, D'_j become lane values, gamma_j stays scalar for the unit's rows; requested in front of the row tests, which cover most of the latency
            double hB[JT], hD[JT], hG[JT];
#pragma unroll
            for (int jj = 0; jj < JT; jj++) { hB[jj] = U[4 * jj + 0]; hD[jj] = U[4 * jj + 1]; hG[jj] = U[4 * jj + 2]; }
            // gamma_j must STAY in its SGPR pair through the rows: a load from the constant address space is rematerialisable, and the compiler otherwise
            // re-issues it in every row with an lgkmcnt(0) behind it -- which also waits for the next row's request
#pragma unroll
            for (int jj = 0; jj < JT; jj++) asm volatile("" : "+s"(hG[jj]));
            __builtin_amdgcn_sched_barrier(0);
            const double bmax = __dmul_rn(pT, dmax_cur);
            unsigned live = (1u << R) - 1u;
            if (g.zskip) {
                live = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double earg = bmax - mT * cn[r];
                    live |= __all(earg < cull_thr) ? 0u : (1u << r);
                }
            }
            live = __builtin_amdgcn_readfirstlane(live);
            n_rows += R;
            n_dead += R - __builtin_popcount(live);
            __builtin_amdgcn_sched_barrier(0);
            RowS ro = load_row(U, live ? __builtin_ctz(live) : 0);      // cn is dead from here on
            double pTB[JT], pTD[JT];
#pragma unroll
            for (int jj = 0; jj < JT; jj++) {
                pTB[jj] = pT * hB[jj];
                pTD[jj] = pT * hD[jj];
            }
            __builtin_amdgcn_sched_barrier(0);
            request_ahead(u);                                            // into the SGPRs the header and cn have left
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (live & (1u << r)) {
                    // the row's scalars become lane values at its head ...
                    const double mTC = mT * ro.C;
                    const double mTA = mT * ro.A;
                    const double mT2a = mT2s * ro.al;
                    double br[JT];
#pragma unroll
                    for (int jj = 0; jj < JT; jj++) br[jj] = __builtin_fma(mTpTs, ro.be[jj], __builtin_fma(pT2s, hG[jj], mT2a));
                    __builtin_amdgcn_sched_barrier(0);
                    // ... and the next live row's are requested into the same SGPRs, a whole row ahead of their use
                    const unsigned rest = live >> (r + 1);
                    if (rest) ro = load_row(U, r + 1 + __builtin_ctz(rest));
                    __builtin_amdgcn_sched_barrier(0);
                    const double E1 = exp_p9(bmax - mTC);
#pragma unroll
                    for (int j0 = 0; j0 < JT; j0 += RB) {
                        double zv[RB], qq[RB], inv[RB];
#pragma unroll
                        for (int i = 0; i < RB; i++) {
                            const double z = E1 * E2[j0 + i];
                            if (CE) {
                                const double x = mTC - pTD[j0 + i];
                                zv[i] = z * x;
                                qq[i] = __builtin_fma(sign, zv[i], x);
                            } else {
                                zv[i] = z;
                                qq[i] = __builtin_fma(sign, z, 1.0);
                            }
                        }
                        rcp_batch<RB>(qq, inv);
#pragma unroll
                        for (int i = 0; i < RB; i++) {
                            const int jj = j0 + i;
                            const double pds = OUTFLOW ? add_clamp01(mTA, pTB[jj]) : (mTA + pTB[jj]);
                            const double dfr = inv[i];
                            const double uu = REG ? fma_clamp01_half(dfr, br[jj]) : __builtin_fma(dfr, br[jj], 1.0);
                            const double w = (zv[i] * dfr) * uu;
                            acc[jj * R + r] = __builtin_fma(pds, w, acc[jj * R + r]);
                        }
                    }
                }
            }
        } else {
            n_rows += R;
            n_dead += R;
            request_ahead(u);
        }
        // the threshold follows log2 of the accumulators: refreshed on cf_main_tile3e's schedule (its 6-unit batches 0, 1, 3, 7, 15, ... and every 64th)
        if (RELCULL && g.zskip == 2 && (u % 6) == 5) {
            const int ib = u / 6;
            if (((ib + 1) & ib) == 0 || (ib & 63) == 63) refresh_threshold();
        }
        live_cur = live_nxt;
        dmax_cur = dmax_u1;
        if (live_nxt) {
#pragma unroll
            for (int jj = 0; jj < JT; jj++) E2[jj] = E2n[jj];
        }
    }

    if (threadIdx.x == 0) {
        atomicAdd(&stats[2], (unsigned long long)n_rows);
        atomicAdd(&stats[3], (unsigned long long)min(n_dead, n_rows));
    }
    const double unscale = REG ? 2.0 : 1.0;
    const int64_t JKacc = (int64_t)J * g.Kacc;
    double *pp = partial + (int64_t)chunk * JKacc * g.Lpad;
#pragma unroll
    for (int jj = 0; jj < JT; jj++) {
        const int j = jt * JT + jj;
        if (j < J) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int k = kt * R + r;
                if (k < K) {
                    double *o = pp + ((int64_t)j * g.Kacc + k) * g.Lpad + l;
                    const double v = unscale * acc[jj * R + r];
                    *o = g.first_pass ? v : (*o + v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// cf_finalize: out[sp + npart*(ipT + npT*(j + J*k))] (smooth_kernels.cpp:363)
//              (+)= prefactor * degeneracy[sp] * sum_chunks partial[chunk][j*Kacc + k][class(sp)*npT + ipT]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cf_finalize(const double *__restrict__ partial, const int *__restrict__ cls, const double *__restrict__ degeneracy,
            double *__restrict__ out, int64_t nout, int npart, int npT, int J, int Kacc, int Lpad, int nch,
            double prefactor, int accumulate, const unsigned long long *__restrict__ pds_bound, int split, int Lbins)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nout) return;
    const int sp = (int)(idx % npart);
    int64_t r = idx / npart;
    const int ipT = (int)(r % npT);
    r /= npT;
    const int j = (int)(r % J);
    const int k = (int)(r / J);
    const int l = cls[sp * npT + ipT];   // lane slot of (species, pT): lanes are sorted by mT, classes shared
    const int64_t stride = (int64_t)J * Kacc * Lpad;
    const double *p = partial + ((int64_t)j * Kacc + k) * Lpad + l;
    double s = 0.0;
    for (int ch = 0; ch < nch; ch++)
        for (int sl = 0; sl < split; sl++) s += p[ch * stride + (int64_t)sl * Lbins];   // unit-strided lanes: the bin's slots, in slot order
    double unscale;                            // the stream carried p.dsigma 2^-e: exact to put back
    (void)pds_scale(pds_bound, &unscale);
    const double v = (prefactor * degeneracy[sp]) * (s * unscale);
    out[idx] = accumulate ? (out[idx] + v) : v;
}

// Fixed-order sum over the cell chunks, lanes <-> threads: fully coalesced reads of the partials (the species scatter of
// cf_finalize gathers lanes and would read them at a fraction of the bandwidth).  In place: the thread that owns element
// (jk, lane) of chunk 0 is the only one that touches it.
__global__ void __launch_bounds__(256)
cf_reduce_chunks(double *__restrict__ partial, int64_t n_per_chunk, int nch)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_per_chunk) return;
    double s = 0.0;
    for (int ch = 0; ch < nch; ch++) s += partial[ch * n_per_chunk + i];
    partial[i] = s;
}

// Surface-relative cull (cf_main_tile3e): the chunks that ran first give a partial spectrum P_A <= the final one (every term is >= 0 under
// outflow && regulate_deltaf); a row whose every term is below 2^-58 of the smallest P_A accumulator of its (lane, tile) can change the final
// sum by less than 2^-57 of it per cell.  floor[s_tile][l] = the row-test threshold that follows from min over the tile of P_A, in the
// units of the kernel's own (cull_thr: natural log of the half-scaled accumulator, lane exponent pe removed).
__global__ void __launch_bounds__(256)
cf_cull_floor(const double *__restrict__ partial, int nA, int J, int K, int Kacc, int Lpad, int JT, int R, int jtiles, int ktiles,
              const int32_t *__restrict__ lane_pe, double unscale, double *__restrict__ floor_out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)jtiles * ktiles * Lpad) return;
    const int l = (int)(idx % Lpad), s_tile = (int)(idx / Lpad), jt = s_tile / ktiles, kt = s_tile - jt * ktiles;
    const int64_t per_chunk = (int64_t)J * Kacc * Lpad;
    double m = 1.0e300;
    for (int jj = 0; jj < JT; jj++) {
        const int j = jt * JT + jj;
        if (j >= J) break;
        for (int r = 0; r < R; r++) {
            const int k = kt * R + r;
            if (k >= K) break;
            double sum = 0.0;
            for (int c = 0; c < nA; c++) sum += partial[c * per_chunk + ((int64_t)j * Kacc + k) * Lpad + l];
            m = __builtin_fmin(m, sum);
        }
    }
    m /= unscale;
    const int e = __builtin_amdgcn_frexp_exp(m);
    const int pe = lane_pe ? lane_pe[l] : 0;
    floor_out[idx] = (m > 1.0e-290 && m < 1.0e299) ? __builtin_fmax(-745.2, (double)(e - 58 - pe) * 0.6931471805599453) : -745.2;
}

hipError_t launch_cull_floor(const double *partial, int nA, int J, int K, int Kacc, int Lpad, int JT, int R, int jtiles, int ktiles,
                             const int32_t *lane_pe, double unscale, double *floor_out, hipStream_t stream)
{
    const int64_t n = (int64_t)jtiles * ktiles * Lpad;
    hipLaunchKernelGGL(cf_cull_floor, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, nA, J, K, Kacc, Lpad, JT, R, jtiles, ktiles,
                       lane_pe, unscale, floor_out);
    return hipGetLastError();
}

hipError_t launch_finalize(double *partial, const int *cls, const double *degeneracy, double *out,
                           int64_t nout, int npart, int npT, int J, int Kacc, int Lpad, int nch, double prefactor,
                           int accumulate, const unsigned long long *pds_bound, hipStream_t stream, int split, int Lbins)
{
    if (nout <= 0) return hipSuccess;
    if (nch > 1) {
        const int64_t n_per_chunk = (int64_t)J * Kacc * Lpad;
        hipLaunchKernelGGL(cf_reduce_chunks, dim3((unsigned)((n_per_chunk + 255) / 256)), dim3(256), 0, stream, partial, n_per_chunk, nch);
    }
    int grid = (int)((nout + 255) / 256);
    hipLaunchKernelGGL(cf_finalize, dim3(grid), dim3(256), 0, stream, (const double *)partial, cls, degeneracy, out, nout, npart,
                       npT, J, Kacc, Lpad, 1, prefactor, accumulate, pds_bound, split < 1 ? 1 : split, Lbins);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// main-kernel dispatch
// ------------------------------------------------------------------------------------------------
template <bool CE, bool DIM3, bool OUTFLOW, bool REG, int KT>
static void launch_direct_t(const MainArgs &a, hipStream_t st)
{
    int grid = ((a.g.NT + 7) / 8) * 8 * a.g.G;
    hipLaunchKernelGGL((cf_main_direct<CE, DIM3, OUTFLOW, REG, KT>), dim3(grid), dim3(kWPB * 64), 0, st, a.S1, a.S2,
                       a.S3, a.lane_mT, a.lane_pT, a.lane_sign, a.partial, a.g);
}
template <bool CE, bool DIM3, bool OUTFLOW, bool REG, int JT, int R, bool LAZY = false, bool DMA = true>
static void launch_tile_t(const MainArgs &a, hipStream_t st)
{
    int grid = ((a.g.NT + 7) / 8) * 8 * a.g.G;
    if constexpr (!DMA) {   // variant 8: the register-staged copy of round 1, kept for A/B (without baryon slots)
        if (!a.g.baryon) {
            hipLaunchKernelGGL((cf_main_tile<CE, DIM3, OUTFLOW, REG, false, JT, R, LAZY, false>), dim3(grid), dim3(a.g.wpb * 64), 0, st, a.TS,
                               a.lane_mT, a.lane_pT, a.lane_sign, a.lane_b, a.partial, a.stats, a.g, a.lane_pe, a.lane_sub);
            return;
        }
    }
    if (a.g.baryon)
        hipLaunchKernelGGL((cf_main_tile<CE, DIM3, OUTFLOW, REG, true, JT, R, LAZY>), dim3(grid), dim3(a.g.wpb * 64), 0, st, a.TS,
                           a.lane_mT, a.lane_pT, a.lane_sign, a.lane_b, a.partial, a.stats, a.g, a.lane_pe, a.lane_sub);
    else
        hipLaunchKernelGGL((cf_main_tile<CE, DIM3, OUTFLOW, REG, false, JT, R, LAZY>), dim3(grid), dim3(a.g.wpb * 64), 0, st, a.TS,
                           a.lane_mT, a.lane_pT, a.lane_sign, a.lane_b, a.partial, a.stats, a.g, a.lane_pe, a.lane_sub);
}

// Kernel variants.  1: direct (flat streams).  2 (default), 3, 4: LDS-staged tile kernel, tile shapes for tuning.
//   variant : 3+1D (JT x R rows of y) / 2+1D (JT x R rows of the eta table per unit)
constexpr int kV1KT3 = 7, kV1KT2 = 4;
constexpr int kTileJT3[3] = {6, 8, 4}, kTileR3[3] = {7, 7, 7};
constexpr int kTileJT2[3] = {8, 12, 4}, kTileR2[3] = {61, 61, 61};
constexpr int kTile7JT = 8, kTile7R = 31;   // variant 7 (2+1D)

void main_tile_shape(int variant, int dim3, int *JT, int *KT)
{
    if (variant == 1) { *JT = 1; *KT = dim3 ? kV1KT3 : kV1KT2; return; }
    if (variant == 5 || variant == 6 || (variant >= 9 && variant <= 12)) variant = 3;   // same tile, E2 table stream
    if (variant == 7 || variant == 8) {
        if (!dim3) { *JT = kTile7JT; *KT = kTile7R; return; }
        variant = 3;
    }
    int i = (variant >= 2 && variant <= 4) ? variant - 2 : 0;
    *JT = dim3 ? kTileJT3[i] : kTileJT2[i];
    *KT = dim3 ? kTileR3[i] : kTileR2[i];
}

template <bool CE, bool OF, bool RG, int JT, int R, int MODE = 0, bool BARYON = false, bool E2G = false, bool RAWH = false, bool E2L = false>
static void launch_tile3e_t(const MainArgs &a_in, hipStream_t st)
{
    MainArgs a = a_in;
    if (a.g.nch_run <= 0) { a.g.ch0 = 0; a.g.nch_run = a.g.nch; }
    const int pairs = a.g.jtiles * a.g.nch_run;                               // (phi tile, cell chunk) pairs, dealt round-robin to the 8 XCDs
    const int grid = ((pairs + 7) / 8) * 8 * a.g.ktiles * a.g.G;
    const size_t lds = tile3e_lds_bytes(JT, R, a.g.ub, 2, BARYON ? 1 : 0, E2L ? 2 : E2G ? 1 : 0);
    if constexpr (E2G || RAWH || E2L) {
        hipLaunchKernelGGL((cf_main_tile3e<CE, OF, RG, JT, R, MODE, false, BARYON, E2G, RAWH, E2L>), dim3(grid), dim3(a.g.wpb * 64), lds, st, a.TS, a.TE, a.lane_mT,
                           a.lane_pT, a.lane_sign, a.lane_ipT, a.partial, a.stats, a.g, a.lane_pe, a.lane_b, a.cull_floor, a.pTgrid);
        return;
    }
    if constexpr (CE && OF && RG && MODE >= 1 && !BARYON) {
        // dev: the cycle-accounting instantiation, synchronous, counters to stderr
        static const bool prof = dev_env("IS3D_DEV_PROF") != nullptr;
        if (prof) {
            unsigned long long h[16] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof3e), h, sizeof h);
            hipLaunchKernelGGL((cf_main_tile3e<CE, OF, RG, JT, R, MODE, true>), dim3(grid), dim3(a.g.wpb * 64), lds, st, a.TS, a.TE, a.lane_mT,
                               a.lane_pT, a.lane_sign, a.lane_ipT, a.partial, a.stats, a.g, a.lane_pe, a.lane_b, a.cull_floor);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof3e), sizeof h);
            const double T = (double)h[1];
            fprintf(stderr, "[prof3e] waves %llu  cycles/wave %.3e  stage %.4f  wait %.4f (first wave of the workgroup %.4f)  dead units %.4f  live units %.4f "
                            "(headers + row tests %.4f)  thr %.4f  vmcnt part of wait %.4f  prologue %.4f  record part of stage %.4f | dead units %llu (%.0f cycles each)  live units %llu (%.0f cycles each)\n",
                    h[0], T / (double)h[0], h[2] / T, h[3] / T, h[9] / T, h[4] / T, h[5] / T, h[6] / T, h[10] / T, h[11] / T, h[12] / T, h[13] / T, h[7], h[7] ? (double)h[4] / h[7] : 0.0,
                    h[8], h[8] ? (double)h[5] / h[8] : 0.0);
            return;
        }
    }
    hipLaunchKernelGGL((cf_main_tile3e<CE, OF, RG, JT, R, MODE, false, BARYON>), dim3(grid), dim3(a.g.wpb * 64), lds, st, a.TS, a.TE, a.lane_mT,
                       a.lane_pT, a.lane_sign, a.lane_ipT, a.partial, a.stats, a.g, a.lane_pe, a.lane_b, a.cull_floor);
}

template <bool CE, bool DIM3, bool OF, bool RG>
static void launch_variant(int variant, const MainArgs &a, hipStream_t st)
{
    // The shipped library holds the kernels a default path of the plan reaches (cf_plan.cpp maps every other request onto them):
    //   3+1D: 6 -- cf_main_tile3e<MODE 1>, with or without baryon slots (pT grids of up to 32 values); 3 -- the 8 x 7 tile without the E2 stream
    //         (larger pT grids, no baryon slots); 2 -- the 6 x 7 tile (larger pT grids with baryon slots);   2+1D: 7 -- the 8 x 31 tile.
    // The other variants (1 direct, 2 / 4 other tile shapes, 5 hand-pipelined rows, 8 register-staged copy, 9 scalar path) are A/B forms of rounds
    // 1-5: developer build only (make DEV=1 -> is3d_amd/lib_dev), where their parity tests run (tests/test_gpu_devlib.py).
    if constexpr (DIM3) {
        if constexpr (kDevBuild) {
            // variant 5: the 8 x 7 tile with the E2 table stream, rows hand-pipelined (the plan only sets TE up for 3+1D)
            if (variant == 5 && a.TE && !a.g.baryon) { launch_tile3e_t<CE, OF, RG, kTileJT3[1], kTileR3[1]>(a, st); return; }
        }
        if (variant == 6 && a.TE && !a.g.baryon) { launch_tile3e_t<CE, OF, RG, kTileJT3[1], kTileR3[1], 1>(a, st); return; }
        if ((variant == 5 || variant == 6) && a.TE && a.g.baryon) { launch_tile3e_t<CE, OF, RG, kTileJT3[1], kTileR3[1], 1, true>(a, st); return; }
        if constexpr (kDevBuild) {
            // variant 10 (round 5): the E2 column straight from global memory into registers, records-only LDS batches
            if (variant == 10 && a.TE && !a.g.baryon) { launch_tile3e_t<CE, OF, RG, kTileJT3[1], kTileR3[1], 1, false, true>(a, st); return; }
            // variant 11 (round 5): raw header values as FMA operands
            if (variant == 11 && a.TE && !a.g.baryon) { launch_tile3e_t<CE, OF, RG, kTileJT3[1], kTileR3[1], 1, false, false, true>(a, st); return; }
            // variant 12 (round 5): the E2 tables built per workgroup in LDS from the staged records
            if (variant == 12 && a.TE && a.pTgrid && !a.g.baryon) { launch_tile3e_t<CE, OF, RG, kTileJT3[1], kTileR3[1], 1, false, false, false, true>(a, st); return; }
        }
        if constexpr (kDevBuild)   // measured and dropped (profiles/r05_ab_tile3s.log: 120.4 against 100.2 ms)
        if (variant == 9 && a.TE && !a.g.baryon && a.g.wpb == 1) {   // cf_main_tile3s: one-wave workgroups, no LDS
            MainArgs b = a;
            if (b.g.nch_run <= 0) { b.g.ch0 = 0; b.g.nch_run = b.g.nch; }
            const int pairs = b.g.jtiles * b.g.nch_run;
            const int grid = ((pairs + 7) / 8) * 8 * b.g.ktiles * b.g.G;
            hipLaunchKernelGGL((cf_main_tile3s<CE, OF, RG, kTileJT3[1], kTileR3[1]>), dim3(grid), dim3(64), 0, st, b.TS, b.TE, b.lane_mT, b.lane_pT, b.lane_sign,
                               b.lane_ipT, b.partial, b.stats, b.g, b.lane_pe, b.cull_floor);
            return;
        }
    }
    if (variant == 5 || variant == 6 || (variant >= 9 && variant <= 12)) variant = 3;
    if constexpr (!DIM3) {
        // variant 7: 2+1D, 8 x 31 tile: units short enough for four of them per LDS buffer, i.e. for unit-strided lanes with S = 4
        if (variant == 7 || !kDevBuild) { launch_tile_t<CE, false, OF, RG, kTile7JT, kTile7R>(a, st); return; }
        if constexpr (kDevBuild) {
            if (variant == 8) { launch_tile_t<CE, false, OF, RG, kTile7JT, kTile7R, false, false>(a, st); return; }
        }
    }
    if (variant == 7 || variant == 8) variant = 3;
    if constexpr (kDevBuild) {
        switch (variant) {
        case 1: launch_direct_t<CE, DIM3, OF, RG, (DIM3 ? kV1KT3 : kV1KT2)>(a, st); break;
        // 8 x 7 in 3+1D: its 56 accumulators leave no room to prefetch whole rows (54 VGPRs spill, 180 GB of scratch traffic per
        // config-3 launch): row operands are read when the row is evaluated (LAZY) -- 8 spills, and 0.7 % faster besides
        case 3: launch_tile_t<CE, DIM3, OF, RG, (DIM3 ? kTileJT3[1] : kTileJT2[1]), (DIM3 ? kTileR3[1] : kTileR2[1]), DIM3>(a, st); break;
        case 4: launch_tile_t<CE, DIM3, OF, RG, (DIM3 ? kTileJT3[2] : kTileJT2[2]), (DIM3 ? kTileR3[2] : kTileR2[2])>(a, st); break;
        default: launch_tile_t<CE, DIM3, OF, RG, (DIM3 ? kTileJT3[0] : kTileJT2[0]), (DIM3 ? kTileR3[0] : kTileR2[0])>(a, st); break;
        }
    } else if constexpr (DIM3) {
        if (variant == 3) launch_tile_t<CE, true, OF, RG, kTileJT3[1], kTileR3[1], true>(a, st);   // 8 x 7, rows read when evaluated (LAZY)
        else launch_tile_t<CE, true, OF, RG, kTileJT3[0], kTileR3[0]>(a, st);                      // 6 x 7 (baryon slots, pT grids of more than 32 values)
    }
}

template <bool CE, bool DIM3>
static void launch_flags(int variant, bool outflow, bool reg, const MainArgs &a, hipStream_t st)
{
    if (outflow && reg) launch_variant<CE, DIM3, true, true>(variant, a, st);
    else if (outflow && !reg) launch_variant<CE, DIM3, true, false>(variant, a, st);
    else if (!outflow && reg) launch_variant<CE, DIM3, false, true>(variant, a, st);
    else launch_variant<CE, DIM3, false, false>(variant, a, st);
}

// One idle wave per workgroup, 8 workgroups (one per XCD under round-robin dispatch): shader-clock ticks (s_memtime) per
// ref_ticks of the constant-rate counter (s_memrealtime).  s_sleep keeps the wave off the issue ports of its SIMD.
__global__ void __launch_bounds__(64) cf_clock_probe(unsigned long long ref_ticks, unsigned long long *__restrict__ out)
{
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = clock64();
    unsigned long long r1;
    do {
        __builtin_amdgcn_s_sleep(64);
        r1 = wall_clock64();
    } while (r1 - r0 < ref_ticks);
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Domain-error flags of an execute (status[0]: first cell outside the coefficient table, status[7]: first cell whose p.u/T can
// exceed 1e9) folded into the plan's sticky pair, so that a caller that passes status == NULL still learns of them at the next
// is3d_plan_check (the reference aborts on such a cell; neutralising it silently would return an incomplete spectrum).
__global__ void cf_fold_status(const unsigned long long *__restrict__ status, unsigned long long *__restrict__ sticky)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (status[0] < sticky[0]) sticky[0] = status[0];
        if (status[7] < sticky[1]) sticky[1] = status[7];
    }
}

hipError_t launch_fold_status(const unsigned long long *status, unsigned long long *sticky, hipStream_t st)
{
    hipLaunchKernelGGL(cf_fold_status, dim3(1), dim3(64), 0, st, status, sticky);
    return hipGetLastError();
}

hipError_t launch_clock_probe(unsigned long long ref_ticks, unsigned long long *out, hipStream_t st)
{
    hipLaunchKernelGGL(cf_clock_probe, dim3(8), dim3(64), 0, st, ref_ticks, out);
    return hipGetLastError();
}

hipError_t launch_main(int variant, int ce, int dim3, int outflow, int reg, const MainArgs &a, hipStream_t st)
{
    if (a.g.n_cells <= 0) return hipSuccess;
    if (ce && dim3) launch_flags<true, true>(variant, outflow, reg, a, st);
    else if (ce && !dim3) launch_flags<true, false>(variant, outflow, reg, a, st);
    else if (!ce && dim3) launch_flags<false, true>(variant, outflow, reg, a, st);
    else launch_flags<false, false>(variant, outflow, reg, a, st);
    return hipGetLastError();
}

const char *main_kernel_name(int variant)
{
    return variant == 1 ? "cf_main_direct" : variant == 9 ? "cf_main_tile3s" : ((variant == 5 || variant == 6 || variant == 10 || variant == 11 || variant == 12) ? "cf_main_tile3e" : "cf_main_tile");
}

// ------------------------------------------------------------------------------------------------
// Derived observables from the device-resident spectrum: what the reference's writers
// compute on the host before printing -- write_dN_twopipTdpTdy_toFile (emissionfunction.cpp:639-677),
// write_continuous_vn_toFile (:1053-1136), write_dN_dy_toFile (:729-772) -- so that a caller that only wants
// these does not pull 39 MB of spectrum through PCIe and text.
//   cf_obs_phi : thread <-> (species, pT, y): phi sums -> dN/(2 pi pT dpT dy) and v_1..v_7
//   cf_obs_dndy: thread <-> (species, y): the double sum of write_dN_dy_toFile, in its loop order (phi outer, pT inner)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cf_obs_phi(const double *__restrict__ dN, const double *__restrict__ phi_w, const double *__restrict__ coskphi,
           const double *__restrict__ sinkphi, double *__restrict__ spec2pi, double *__restrict__ vn, int npart, int npT,
           int J, int ny)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // sp fastest, then pT, then y: coalesced reads
    if (idx >= (int64_t)npart * npT * ny) return;
    const int sp = (int)(idx % npart);
    const int ipT = (int)((idx / npart) % npT);
    const int iy = (int)(idx / ((int64_t)npart * npT));
    double re[7], im[7], den = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 7; k++) { re[k] = 0.0; im[k] = 0.0; }
    for (int j = 0; j < J; j++) {
        const double v = dN[sp + (int64_t)npart * (ipT + (int64_t)npT * (j + (int64_t)J * iy))];
        const double w = phi_w[j];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            re[k] += coskphi[k * J + j] * w * v;     // cos((k+1) phi) * w * dN, as :1106-1107
            im[k] += sinkphi[k * J + j] * w * v;
        }
        den += w * v;
        s2 += w * v / (2.0 * M_PI);                  // :665
    }
    const int64_t o = ((int64_t)sp * ny + iy) * npT + ipT;
    if (spec2pi) spec2pi[o] = s2;
    if (vn) {
#pragma unroll
        for (int k = 0; k < 7; k++) {
            double x = sqrt(re[k] * re[k] + im[k] * im[k]) / den;
            if (den < 1.e-15) x = 0.0;               // :1121
            vn[o * 7 + k] = x;
        }
    }
}

__global__ void __launch_bounds__(256)
cf_obs_dndy(const double *__restrict__ dN, const double *__restrict__ phi_w, const double *__restrict__ pT_w,
            double *__restrict__ dndy, int npart, int npT, int J, int ny)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npart * ny) return;
    const int sp = idx % npart, iy = idx / npart;
    double s = 0.0;
    for (int j = 0; j < J; j++)
        for (int i = 0; i < npT; i++)
            s += phi_w[j] * pT_w[i] * dN[sp + (int64_t)npart * (i + (int64_t)npT * (j + (int64_t)J * iy))];   // :761
    dndy[(int64_t)sp * ny + iy] = s;
}

hipError_t launch_observables(const double *dN, const double *phi_w, const double *pT_w, const double *coskphi,
                              const double *sinkphi, double *dndy, double *spec2pi, double *vn, int npart, int npT, int J,
                              int ny, hipStream_t st)
{
    if (spec2pi || vn) {
        const int64_t n = (int64_t)npart * npT * ny;
        hipLaunchKernelGGL(cf_obs_phi, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dN, phi_w, coskphi, sinkphi, spec2pi,
                           vn, npart, npT, J, ny);
    }
    if (dndy) {
        const int n = npart * ny;
        hipLaunchKernelGGL(cf_obs_dndy, dim3((n + 255) / 256), dim3(256), 0, st, dN, phi_w, pT_w, dndy, npart, npT, J, ny);
    }
    return hipGetLastError();
}

}  // namespace is3d

// ------------------------------------------------------------------------------------------------
// Diagnostic: the elementary functions of cf_math.h, one evaluation per element (is3d_math_probe, include/is3d_amd.h).  The kernels'
// accuracy claims -- exp_full 1.4e-15, exp_p9 7e-14, the one-step square root 3e-15, rcp_nr1 2e-15 -- are checked against glibc / numpy
// through this entry in tests/test_gpu_math.py instead of being taken from the comments.
// ------------------------------------------------------------------------------------------------
namespace is3d {
__global__ void __launch_bounds__(256) cf_math_probe(int which, int64_t n, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r;
    switch (which) {
    case 0: r = exp_full(v); break;
    case 1: r = exp_p9(v); break;
    case 2: r = exp_p9_sat(v); break;
    case 3: r = exp_full_sat(v); break;
    case 4: r = sqrt_g1(v); break;
    case 5: r = sqrt_nr(v); break;
    case 6: r = rcp_nr1(v); break;
    case 8: r = exp_p9_scaled(v, kExpShift + 5.0); break;   // 32 exp_p9(v), bit for bit
    default: r = rcp_nr(v); break;
    }
    y[i] = r;
}

hipError_t launch_math_probe(int which, int64_t n, const double *x, double *y, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cf_math_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, which, n, x, y);
    return hipGetLastError();
}
}  // namespace is3d
