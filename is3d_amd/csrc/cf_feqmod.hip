// cf_feqmod.hip -- gfx950 kernels of the modified-equilibrium smooth path (df_mode 3, 4); see cf_feqmod.h for the
// factorisation.  Reference: EmissionFunctionArray::calculate_dN_ptdptdphidy_feqmod
// (/root/reference/src/cpp/emissionfunction_smooth_kernels.cpp:396-996) and its helpers Milne_Basis /
// Shear_Stress::boost_pimunu_to_lrf (viscous_correction.cpp:8-27, :99-115), GaussThermal (gaussThermal.cpp),
// does_feqmod_breakdown (emissionfunction.cpp:109-150), Deltaf_Data::cubic_spline (deltafReader.cpp:347-384).
//
//   cf_prep_feqmod    lanes <-> cells: per-cell A, A^-1, T_mod, renorm (df_mode 4), breakdown test (df_mode 3), the
//                     quadratic-form coefficients as tiled unit records, fallback records for flagged cells
//   cf_feqmod_renorm  df_mode 3: thread <-> (cell, species class): n_linear / n_mod by Gauss-Laguerre quadrature
//   cf_main_feqmod    lanes <-> (species class, pT), LDS-staged coefficient stream as in cf_main_tile; one sqrt and one
//                     exp per evaluation (the exponential does not factorise here); fp64 VALU bound
//   cf_feqmod_compact ordered list of flagged cells (single workgroup scan, deterministic)
//   cf_feqmod_linear  thread <-> (lane, phi, y): linearised delta-f for flagged cells / narrow rows, added to chunk 0
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>

#include "cf_feqmod.h"
#include "cf_launch.h"
#include "cf_math.h"
#include "errors.h"

namespace is3d {

constexpr int kFqCB = 3;           // cells per workgroup batch of the prep kernel with long rows (2+1D: 241 eta nodes; 3 since round 4: with the
                                   // beta-minimum table two workgroups still share a CU's LDS -- 4 cells, one workgroup per CU: 7.9 against 4.3 ms) ...
constexpr int kFqCB3 = 16;         // ... and with K <= 32 (3+1D): the serial per-cell phase 1 (a 3 x 3 inverse, the Jonah tables) runs on CB lanes of the
                                   // workgroup, so 4-cell batches made it the kernel: 10.3 ms per 1e6 cells
static int fq_batch_cells(int K) { return K > 32 ? kFqCB : kFqCB3; }
constexpr int kFqThreads = 256;

struct FqScal {
    double dat, dax, day, dan_tau, eta, eta_scale;
    double live;                   // 1: cell is evaluated with feqmod; 0: neutral in the stream (skipped, bad, breakdown)
    double rn;                     // df_mode 4: |renorm| folded into A_k, W_k; df_mode 3: 1
    double invTm2;                 // 1 / T_mod^2
    double Xt, tXn, Zt, tZn;       // a_k = (-Xt ch + tXn sh, 0, -Zt ch + tZn sh)
    double Xx, Xy, Yx, Yy;         // b_j = (Xx cos + Xy sin, Yx cos + Yy sin, 0)
    double Ai[9];                  // A^-1, row major
    double detA;
    double narrow;                 // 1: 3+1D cell with detA < 0.01: rows with |y - eta| < detA go to the linear kernel
    double alphaB_mod;             // include_baryon: alpha_B + Pi G / beta_Pi (:637), else 0
    double shiftc;                 // kExpShift + e_c: the cell's rows carry A_k, W_k times 2^-e_c (df_mode 4 with the outflow clamp, see cf_main_feqmod)
};

enum FbIdx { FB_DAT = 0, FB_DAX, FB_DAY, FB_DAN, FB_UT, FB_UX, FB_UY, FB_UN, FB_TAU, FB_ETA, FB_T,
             FB_PITT, FB_PITX, FB_PITY, FB_PITN, FB_PIXX, FB_PIXY, FB_PIXN, FB_PIYY, FB_PIYN, FB_PINN,
             FB_SHEAR, FB_CA, FB_CB, FB_DETA, FB_KIND,
             FB_ALPHAB, FB_B1, FB_BER, FB_IBV, FB_VT, FB_VX, FB_VY, FB_VN,   // include_baryon (df_mode 3): :838-850
             FB_END };
static_assert(FB_END <= kFbRec, "fallback record too small");

// GaussThermal integrands (gaussThermal.cpp; neq_int, J10_int, J20_int); chem = baryon * alpha_B
__device__ __forceinline__ double gt_neq(const double *root, const double *weight, int n, double mbar, double sign, double chem = 0.0)
{
    double s = 0.0;
    for (int k = 0; k < n; k++) {
        const double pbar = root[k], Ebar = sqrt(pbar * pbar + mbar * mbar);
        s += weight[k] * (pbar * exp(pbar) / (exp(Ebar - chem) + sign));
    }
    return s;
}
__device__ __forceinline__ double gt_J10(const double *root, const double *weight, int n, double mbar, double sign, double chem)
{
    double s = 0.0;
    for (int k = 0; k < n; k++) {
        const double pbar = root[k], Ebar = sqrt(pbar * pbar + mbar * mbar);
        const double qstat = exp(Ebar - chem) + sign;
        s += weight[k] * (pbar * exp(pbar + Ebar - chem) / (qstat * qstat));
    }
    return s;
}
__device__ __forceinline__ double gt_J20(const double *root, const double *weight, int n, double mbar, double sign, double chem = 0.0)
{
    double s = 0.0;
    for (int k = 0; k < n; k++) {
        const double pbar = root[k], Ebar = sqrt(pbar * pbar + mbar * mbar);
        const double qstat = exp(Ebar - chem) + sign;
        s += weight[k] * (Ebar * exp(pbar + Ebar - chem) / (qstat * qstat));
    }
    return s;
}

// betaf_jk = 2 (a'_k . b'_j) / T_mod^2 with the roundings written out: the record writer and the minima over a phi tile (the main kernel's
// lower bound on a row's X^2) must hold the SAME doubles
__device__ __forceinline__ double fq_beta(double ax, double ay, double az, double bx, double by, double bz, double invTm2)
{
    return (2.0 * __builtin_fma(az, bz, __builtin_fma(ay, by, ax * bx))) * invTm2;
}

template <int CB>
__global__ void __launch_bounds__(kFqThreads) cf_prep_feqmod(FqPrepParams p)
{
    extern __shared__ double lds[];
    const int nT = p.spl.n, J = p.J, K = p.K, nj = p.nj, ngl = p.ngl;
    double *sx = lds;                          // [nT]
    double *sy = sx + nT;                      // [3][nT]
    double *sc = sy + 3 * nT;                  // [3][nT]
    double *jx = sc + 3 * nT;                  // [5][nj]: x, lambda^2, z, c_lambda, c_z
    double *gl = jx + 5 * nj;                  // [4][ngl]
    FqScal *cs = (FqScal *)(gl + 4 * ngl);     // [CB]
    double *lk = (double *)(cs + CB);       // [6][CB][K]: A, alphaf, W, a'x, a'y, a'z
    double *lj = lk + 6 * CB * K;           // [5][CB][J]: B, gammaf, b'x, b'y, b'z
    double *l_bm = lj + 5 * CB * J;         // [CB][K][jtiles]: min over a phi tile of betaf_jk
    // bounds of the main kernel's unit-level cull and unit threshold, per (cell, row block): min alphaf, max |A|, max |W|; per (cell, row
    // block, phi tile): min betaf; per (cell, phi tile): min gammaf, max |B|
    double *l_ub = l_bm + CB * K * p.jtiles;   // [CB][3 rblocks + rblocks jtiles + 2 jtiles]
    // element descriptors of a unit record, one int2 per element, filled once per workgroup (as in cf_prep, cf_kernels.hip)
    int2 *desc = (int2 *)(l_ub + CB * (3 * p.rblocks + p.rblocks * p.jtiles + 2 * p.jtiles));
    const int CK = CB * K, CJ = CB * J;
    double *l_A = lk, *l_al = lk + CK, *l_W = lk + 2 * CK, *l_ax = lk + 3 * CK, *l_ay = lk + 4 * CK, *l_az = lk + 5 * CK;
    double *l_B = lj, *l_ga = lj + CJ, *l_bx = lj + 2 * CJ, *l_by = lj + 3 * CJ, *l_bz = lj + 4 * CJ;

    const int tid = threadIdx.x;
    for (int i = tid; i < nT; i += kFqThreads) {
        sx[i] = p.spl.x[i];
        for (int s = 0; s < 3; s++) { sy[s * nT + i] = p.spl.y[s][i]; sc[s * nT + i] = p.spl.c[s][i]; }
    }
    for (int i = tid; i < nj; i += kFqThreads) {
        jx[i] = p.jx[i]; jx[nj + i] = p.jl2[i]; jx[2 * nj + i] = p.jz[i]; jx[3 * nj + i] = p.jcl[i]; jx[4 * nj + i] = p.jcz[i];
    }
    for (int i = tid; i < 4 * ngl; i += kFqThreads) gl[i] = p.gl[i];
    // Record-element descriptors: every element of a unit record is a copy of one LDS double, an exact zero, or a betaf_jk, and WHICH
    // is the same for every unit -- decoded once here instead of per element with five divisions and a branch per field:
    //   x: bits 0-15 offset of the source in the LDS block | 16-18 index rule (0: + c J + j, 1: + c K + k, 2: + c ubs + 3 rb, 3: + c ubs + rb
    //      jtiles + jt, 4: + c ubs + 2 jt, 5: + c sizeof(FqScal)/8, 6: + (c K + k) jtiles + jt) | 20 betaf | 21 zero in a padding row | 22 zero
    //   y: jj (header entry / beta column) | r << 16 (row)
    const int ubs = 3 * p.rblocks + p.rblocks * p.jtiles + 2 * p.jtiles;   // doubles per cell in l_ub
    {
        const int JT = p.JT, R = p.R, HDR = 4 * JT, RWD = 4 + JT, REC = HDR + R * RWD;
        const int o_A = (int)(l_A - lds), o_al = (int)(l_al - lds), o_W = (int)(l_W - lds), o_B = (int)(l_B - lds), o_ga = (int)(l_ga - lds);
        const int o_bm = (int)(l_bm - lds), o_ub = (int)(l_ub - lds), o_aB = (int)(&cs[0].alphaB_mod - lds), o_sh = (int)(&cs[0].shiftc - lds);
        constexpr int ZERO = 1 << 22, PADZ = 1 << 21, BETA = 1 << 20;
        for (int e = tid; e < REC; e += kFqThreads) {
            int x = ZERO, y = 0;
            if (e < HDR) {
                const int jj = e >> 2, f = e & 3;
                y = jj;
                if (f == 0) x = o_B;
                else if (f == 1) x = o_ga;
                else if (e == 2) x = o_aB | (5 << 16);
                else if (JT >= 4 && jj < 4) {
                    // bounds of the main kernel's unit-level cull and unit threshold (phase 2d): min alphaf_k (e = 3), min_jk betaf_jk (6),
                    // max |A_k| (10), max |W_k| (11) over the unit's rows; min gammaf_j (7), max |B_j| (14) over its phi's
                    if (e == 3) x = o_ub | (2 << 16);
                    else if (e == 10) x = (o_ub + 1) | (2 << 16);
                    else if (e == 11) x = (o_ub + 2) | (2 << 16);
                    else if (e == 6) x = (o_ub + 3 * p.rblocks) | (3 << 16);
                    else if (e == 7) x = (o_ub + 3 * p.rblocks + p.rblocks * p.jtiles) | (4 << 16);
                    else if (e == 14) x = (o_ub + 3 * p.rblocks + p.rblocks * p.jtiles + 1) | (4 << 16);
                    else if (e == 15) x = o_sh | (5 << 16);   // kExpShift + e_c: the exponent of the cell's power-of-two scale (phase 2e)
                }
            } else {
                const int q = e - HDR, r = q / RWD, f = q - r * RWD;
                y = r << 16;
                // row scalars in the slot order of fq_row_slots (cf_feqmod.h): 3+1D {alphaf, min_j betaf, A, W}, 2+1D {A, alphaf, W, min_j betaf}
                const FqRowSlots sl = fq_row_slots(p.dim3 != 0);
                if (f == sl.A) x = o_A | (1 << 16) | PADZ;     // padding rows: p.dsigma = 0, the quadratic form of row K-1
                else if (f == sl.AL) x = o_al | (1 << 16);
                else if (f == sl.W) x = o_W | (1 << 16) | PADZ;
                else if (f == sl.BM) x = o_bm | (6 << 16);
                else { x = BETA; y |= f - 4; }
            }
            desc[e] = int2{x, y};
        }
    }
    __syncthreads();
    const double two_pi2_hbarC3 = 2.0 * M_PI * M_PI * (kHbarC * kHbarC * kHbarC);   // iS3D.h:11

    const int nbatch = (p.n_cells + CB - 1) / CB;
    for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const int cbase = batch * CB;
        const int ncb = min(CB, p.n_cells - cbase);

        // ---- phase 1: per-cell scalars (smooth_kernels.cpp:486-735) ----
        if (tid < ncb) {
            const int cell = cbase + tid;
            const int64_t gi = p.cell0 + cell;
            FqScal s;
            const double tau = p.cells.tau[gi], tau2 = tau * tau;
            const double dat = p.cells.dat[gi], dax = p.cells.dax[gi], day = p.cells.day[gi], dan = p.cells.dan[gi];
            const double ux = p.cells.ux[gi], uy = p.cells.uy[gi], un = p.cells.un[gi];
            const double ut = sqrt(1.0 + ux * ux + uy * uy + tau2 * un * un);
            const double udsigma = ut * dat + ux * dax + uy * day + un * dan;
            bool valid = udsigma > 0.0;                                               // :502
            if (!valid) atomicAdd(&p.status[1], 1ULL);
            const double T = p.cells.T[gi];
            double muB = 0.0, nB = 0.0, Vx = 0.0, Vy = 0.0, Vn = 0.0;
            if (valid && p.baryon && p.baryondiff) {                                  // :572-584
                muB = p.cells.muB[gi]; nB = p.cells.nB[gi];
                Vx = p.cells.Vx[gi]; Vy = p.cells.Vy[gi]; Vn = p.cells.Vn[gi];
            }
            double bl[5] = {0.0, 0.0, 0.0, 0.0, 0.0};                                 // F, G, betabulk, betaV, betapi / powers of T
            if (valid && p.baryon) {
                if (!bilinear5(p.bil, T, muB, bl)) {                                  // outside the (T, mu_B) table: exit(-1) in the reference
                    atomicMin(&p.status[0], (unsigned long long)gi);
                    valid = false;
                }
            } else if (valid && !(T >= sx[0] && T <= sx[nT - 1])) {                   // GSL domain error in the reference
                atomicMin(&p.status[0], (unsigned long long)gi);
                valid = false;
            }
            const double P = p.cells.P[gi], E = p.cells.E[gi];
            double bulkPi = (valid && p.include_bulk) ? p.cells.bulkPi[gi] : 0.0;
            if (valid && p.mode == 4) {                                               // :584-590
                if (bulkPi < -P) bulkPi = -(1.0 - 1.e-5) * P;
                else if (bulkPi / P > p.bp_max) bulkPi = P * (p.bp_max - 1.e-5);
                const double r = bulkPi / P;
                if (!(r >= jx[0] && r <= jx[nj - 1])) {                               // outside the Jonah table (P <= 0, NaN)
                    atomicMin(&p.status[0], (unsigned long long)gi);
                    valid = false;
                }
            }
            int flag = 0;
            s.narrow = 0.0;
            if (valid) {
                const double ut2 = ut * ut, ux2 = ux * ux, uy2 = uy * uy;
                const double uperp = sqrt(ux * ux + uy * uy), utperp = sqrt(1.0 + ux * ux + uy * uy);
                double pitt = 0, pitx = 0, pity = 0, pitn = 0, pixx = 0, pixy = 0, pixn = 0, piyy = 0, piyn = 0, pinn = 0;
                if (p.include_shear) {                                                // :531-545
                    pixx = p.cells.pixx[gi]; pixy = p.cells.pixy[gi]; pixn = p.cells.pixn[gi];
                    piyy = p.cells.piyy[gi]; piyn = p.cells.piyn[gi];
                    pinn = (pixx * (ux2 - ut2) + piyy * (uy2 - ut2) + 2.0 * (pixy * ux * uy + tau2 * un * (pixn * ux + piyn * uy))) / (tau2 * utperp * utperp);
                    pitn = (pixn * ux + piyn * uy + tau2 * pinn * un) / ut;
                    pity = (pixy * ux + piyy * uy + tau2 * piyn * un) / ut;
                    pitx = (pixx * ux + pixy * uy + tau2 * pixn * un) / ut;
                    pitt = (pitx * ux + pity * uy + tau2 * pitn * un) / ut;
                }
                // evaluate_df_coefficients -> cubic_spline (deltafReader.cpp:347-384)
                const double T4 = T * T * T * T;
                double F = 0.0, G = 0.0, betabulk = 0.0, betaV = 1.0, lambda = 0.0, z = 0.0, delta_lambda = 0.0, delta_z = 0.0;
                const double betapi = p.baryon ? bl[4] * T4 : spline_eval_lds(nT, sx, sy + 2 * nT, sc + 2 * nT, T) * T4;
                const double alphaB = muB / T;
                if (p.baryon) {                                                       // deltafReader.cpp:454-468 (case 2: case 3:)
                    F = bl[0] * T; G = bl[1]; betabulk = bl[2] * T4; betaV = bl[3] * T * T * T;
                } else if (p.mode == 3) {
                    F = spline_eval_lds(nT, sx, sy, sc, T) * T;
                    betabulk = spline_eval_lds(nT, sx, sy + nT, sc + nT, T) * T4;
                } else {
                    const double r = bulkPi / P;
                    const double lambda_squared = spline_eval_lds(nj, jx, jx + nj, jx + 3 * nj, r);
                    if (bulkPi < 0.0) lambda = -sqrt(lambda_squared);
                    else if (bulkPi > 0.0) lambda = sqrt(lambda_squared);
                    z = spline_eval_lds(nj, jx, jx + 2 * nj, jx + 4 * nj, r);
                    delta_lambda = bulkPi / (5.0 * betapi - 3.0 * P * (E + P) / E);
                    delta_z = -3.0 * delta_lambda * P / E;
                }
                // Milne_Basis (viscous_correction.cpp:8-27)
                const double sinhL = tau * un / utperp, coshL = ut / utperp;
                const double Xt = uperp * coshL, Zt = sinhL, Xn = uperp * sinhL / tau, Zn = coshL / tau;
                double Xx = 1.0, Yx = 0.0, Xy = 0.0, Yy = 1.0;
                if (uperp > 1.e-5) { Xx = utperp * ux / uperp; Yx = -uy / uperp; Xy = utperp * uy / uperp; Yy = ux / uperp; }
                // boost_pimunu_to_lrf (viscous_correction.cpp:99-115)
                const double pixx_LRF = pitt * Xt * Xt + pixx * Xx * Xx + piyy * Xy * Xy + tau2 * tau2 * pinn * Xn * Xn
                                      + 2.0 * (-Xt * (pitx * Xx + pity * Xy) + pixy * Xx * Xy + tau2 * Xn * (pixn * Xx + piyn * Xy - pitn * Xt));
                const double pixy_LRF = Yx * (-pitx * Xt + pixx * Xx + pixy * Xy + tau2 * pixn * Xn) + Yy * (-pity * Xt + pixy * Xx + piyy * Xy + tau2 * piyn * Xn);
                const double pixz_LRF = Zt * (pitt * Xt - pitx * Xx - pity * Xy - tau2 * pitn * Xn) - tau2 * Zn * (pitn * Xt - pixn * Xx - piyn * Xy - tau2 * pinn * Xn);
                const double piyy_LRF = pixx * Yx * Yx + 2.0 * pixy * Yx * Yy + piyy * Yy * Yy;
                const double piyz_LRF = -Zt * (pitx * Yx + pity * Yy) + tau2 * Zn * (pixn * Yx + piyn * Yy);
                const double pizz_LRF = -(pixx_LRF + piyy_LRF);
                double T_mod = T;
                if (p.mode == 3) T_mod = T + bulkPi * F / betabulk;                   // :627-631
                const double shear_mod = 0.5 / betapi;
                const double bulk_mod = (p.mode == 4) ? lambda : bulkPi / (3.0 * betabulk);
                const double Axx = 1.0 + pixx_LRF * shear_mod + bulk_mod, Axy = pixy_LRF * shear_mod, Axz = pixz_LRF * shear_mod;
                const double Ayy = 1.0 + piyy_LRF * shear_mod + bulk_mod, Ayz = piyz_LRF * shear_mod;
                const double Azz = 1.0 + pizz_LRF * shear_mod + bulk_mod;
                const double detA = Axx * (Ayy * Azz - Ayz * Ayz) - Axy * (Axy * Azz - Ayz * Axz) + Axz * (Axy * Ayz - Ayy * Axz);   // :668
                // cofactor inverse of the symmetric A (the reference: GSL LU + iterative refinement, :690-707, :915-926)
                const double c00 = Ayy * Azz - Ayz * Ayz, c01 = Ayz * Axz - Axy * Azz, c02 = Axy * Ayz - Ayy * Axz;
                const double dd = Axx * c00 + Axy * c01 + Axz * c02;
                s.Ai[0] = c00 / dd; s.Ai[1] = c01 / dd; s.Ai[2] = c02 / dd;
                s.Ai[3] = s.Ai[1]; s.Ai[4] = (Axx * Azz - Axz * Axz) / dd; s.Ai[5] = (Axy * Axz - Axx * Ayz) / dd;
                s.Ai[6] = s.Ai[2]; s.Ai[7] = s.Ai[5]; s.Ai[8] = (Axx * Ayy - Axy * Axy) / dd;
                // does_feqmod_breakdown (emissionfunction.cpp:109-150, fast = 0)
                bool breakdown = false;
                if (p.mode == 3) {
                    const double neq_fact = T * T * T / two_pi2_hbarC3, J20_fact = T * neq_fact;
                    const double mbar_pion0 = p.mass_pion0 / T;
                    const double neq_pion0 = neq_fact * gt_neq(gl, gl + ngl, ngl, mbar_pion0, -1.0);
                    const double J20_pion0 = J20_fact * gt_J20(gl + 2 * ngl, gl + 3 * ngl, ngl, mbar_pion0, -1.0);
                    const double dn_pion0 = bulkPi * (neq_pion0 + J20_pion0 * F / T / T) / betabulk;
                    if (detA <= p.detA_min || (neq_pion0 + dn_pion0) < 0.0) breakdown = true;
                }
                double eta_scale = 1.0;
                if (detA > p.detA_min && detA < 1.0 && !p.dim3) eta_scale = detA;     // :727-728
                double rn = 1.0;
                if (p.mode == 4) {                                                    // :761-777
                    if (p.include_bulk) rn = z;
                    if (p.dim3) rn /= detA;
                    rn = fabs(rn);
                    if (isnan(rn) || isinf(rn)) rn = 0.0;                             // :768-772: every species skipped
                }
                // Exponent range of the main kernel's exponential (exp_p9, |v| < 1.4e9): E_mod/T_mod <= (1 + ||A^-1||_F) E_LRF / T_mod with
                // E_LRF = p.u <= mTmax (u^tau + |tau u^eta|) cosh(max |y - eta|).  A cell beyond it -- a singular or nearly singular A that the
                // breakdown test did not catch, a vanishing T_mod -- is reported (status[7]) and left out; the reference's LU solve returns
                // inf / nan there.  df_mode 4 with a renormalisation that is nan or inf: the reference skips every species (:768-772) -- the cell
                // is neutral here too, silently, whatever its A^-1 holds.
                bool leave_out = (p.mode == 4 && !(rn > 0.0));
                if (!breakdown && !leave_out) {
                    double an = 0.0;
                    for (int i = 0; i < 9; i++) an += s.Ai[i] * s.Ai[i];
                    const double eta0 = p.dim3 ? p.cells.eta[gi] : 0.0;
                    const double xb = p.mTmax * (ut + fabs(tau * un)) * cosh(fmax(fabs(p.kmin - eta0), fabs(p.kmax - eta0))) * (1.0 + sqrt(an)) / fabs(T_mod);
                    if (!(xb < 1.0e9)) {
                        atomicMin(&p.status[7], (unsigned long long)gi);
                        leave_out = true;
                    }
                }
                if (leave_out) breakdown = false;
                const bool narrow = p.dim3 && !breakdown && !leave_out && detA < 0.01;   // :807-813
                s.dat = dat; s.dax = dax; s.day = day; s.dan_tau = dan / tau;
                s.eta = p.dim3 ? p.cells.eta[gi] : 0.0;
                s.eta_scale = eta_scale;
                s.live = (breakdown || leave_out) ? 0.0 : 1.0;
                s.rn = rn;
                s.invTm2 = 1.0 / (T_mod * T_mod);
                s.Xt = Xt; s.tXn = tau * Xn; s.Zt = Zt; s.tZn = tau * Zn;
                s.Xx = Xx; s.Xy = Xy; s.Yx = Yx; s.Yy = Yy;
                s.detA = detA;
                s.narrow = narrow ? 1.0 : 0.0;
                s.alphaB_mod = (p.mode == 3) ? alphaB + bulkPi * G / betabulk : alphaB;   // :632-637
                flag = breakdown ? 1 : (narrow ? 2 : 0);
                if (p.mode == 3) {
                    double *cr = p.CR + (int64_t)cell * kCrRec;
                    cr[0] = T; cr[1] = T_mod; cr[2] = bulkPi / betabulk; cr[3] = F; cr[4] = detA; cr[5] = s.live;
                    cr[6] = alphaB; cr[7] = G;
                }
                if (flag) {
                    double *fb = p.FB + (int64_t)cell * kFbRec;
                    fb[FB_DAT] = dat; fb[FB_DAX] = dax; fb[FB_DAY] = day; fb[FB_DAN] = dan;
                    fb[FB_UT] = ut; fb[FB_UX] = ux; fb[FB_UY] = uy; fb[FB_UN] = un; fb[FB_TAU] = tau; fb[FB_ETA] = s.eta; fb[FB_T] = T;
                    fb[FB_PITT] = pitt; fb[FB_PITX] = pitx; fb[FB_PITY] = pity; fb[FB_PITN] = pitn; fb[FB_PIXX] = pixx;
                    fb[FB_PIXY] = pixy; fb[FB_PIXN] = pixn; fb[FB_PIYY] = piyy; fb[FB_PIYN] = piyn; fb[FB_PINN] = pinn;
                    fb[FB_SHEAR] = 0.5 / (betapi * T);
                    if (p.mode == 3) {                                                // :833-858 with baryon terms off
                        fb[FB_CA] = (F / (T * T * betabulk)) * bulkPi;
                        fb[FB_CB] = (1.0 / (3.0 * T * betabulk)) * bulkPi;
                    } else {                                                          // :859-880
                        fb[FB_CA] = delta_z - 3.0 * delta_lambda;
                        fb[FB_CB] = delta_lambda / T;
                    }
                    fb[FB_DETA] = detA;
                    fb[FB_KIND] = (double)flag;
                    const double tau2V = tau2 * Vn;
                    fb[FB_ALPHAB] = alphaB;
                    fb[FB_B1] = (p.mode == 3 && p.baryon) ? (G / betabulk) * bulkPi : 0.0;        // bulk1_coeff Pi, :643, :849
                    fb[FB_BER] = nB / (E + P);                                                     // baryon_enthalpy_ratio, :583
                    fb[FB_IBV] = 1.0 / betaV;
                    fb[FB_VT] = (Vx * ux + Vy * uy + tau2V * un) / ut;                             // :580
                    fb[FB_VX] = Vx; fb[FB_VY] = Vy; fb[FB_VN] = Vn;
                }
            } else {
                s.live = 0.0;
                if (p.mode == 3) {
                    double *cr = p.CR + (int64_t)cell * kCrRec;
                    for (int i = 0; i < kCrRec; i++) cr[i] = 0.0;
                    cr[0] = 1.0; cr[1] = 1.0; cr[4] = 1.0;
                }
            }
            if (s.live == 0.0) {
                // neutral cell: p.dsigma == 0 for every momentum and (E_mod/T_mod)^2 = mT^2 stays finite
                s.dat = s.dax = s.day = s.dan_tau = 0.0;
                s.eta = 0.0; s.eta_scale = 1.0; s.rn = 0.0; s.invTm2 = 1.0;
                s.Xt = s.tXn = s.Zt = s.tZn = 0.0;
                s.Xx = s.Xy = s.Yx = s.Yy = 0.0;
                for (int i = 0; i < 9; i++) s.Ai[i] = 0.0;
                s.detA = 1.0; s.narrow = 0.0; s.alphaB_mod = 0.0;
            }
            p.flag[cell] = flag;
            cs[tid] = s;
        }
        __syncthreads();

        // ---- phase 2b: (cell, j) quantities ----
        for (int idx = tid; idx < ncb * J; idx += kFqThreads) {
            const int c = idx / J, j = idx - c * J;
            const FqScal &s = cs[c];
            const double cp = p.cosphi[j], sp = p.sinphi[j];
            const double b0 = s.Xx * cp + s.Xy * sp, b1 = s.Yx * cp + s.Yy * sp;
            const double bx = s.Ai[0] * b0 + s.Ai[1] * b1, by = s.Ai[3] * b0 + s.Ai[4] * b1, bz = s.Ai[6] * b0 + s.Ai[7] * b1;
            l_B[idx] = cp * s.dax + sp * s.day;
            l_ga[idx] = ((bx * bx + by * by + bz * bz) - 1.0) * s.invTm2;
            l_bx[idx] = bx; l_by[idx] = by; l_bz[idx] = bz;
        }
        __syncthreads();
        // ---- phase 2: (cell, k) quantities, and with a'_k in registers min_j betaf_jk over every phi tile: the main kernel's lower bound on a
        // row's X^2 (the SAME betaf_jk the writer stores: fq_beta) ----
        for (int idx = tid; idx < ncb * K; idx += kFqThreads) {
            const int c = idx / K, k = idx - c * K;
            const FqScal &s = cs[c];
            double dlt, w;
            if (p.dim3) { dlt = p.kgrid[k] - s.eta; w = 1.0; }                        // y - eta_cell
            else { dlt = 0.0 - s.eta_scale * p.kgrid[k]; w = p.kweight[k]; }          // y = 0, eta = eta_scale * node, :902-903
            double v = s.live;
            if (s.narrow != 0.0 && fabs(dlt) < s.detA) v = 0.0;                       // row goes to the linear kernel
            const double ch = cosh(dlt), sh = sinh(dlt);
            const double a0 = -s.Xt * ch + s.tXn * sh, a2 = -s.Zt * ch + s.tZn * sh;  // p_LRF = mT a + pT b, :913
            const double ax = s.Ai[0] * a0 + s.Ai[2] * a2, ay = s.Ai[3] * a0 + s.Ai[5] * a2, az = s.Ai[6] * a0 + s.Ai[8] * a2;
            l_A[idx] = (v * s.rn) * (w * ch * s.dat + sh * s.dan_tau);                // dsigma_eta outside the eta weight, :905
            l_W[idx] = (v * s.rn) * w;
            l_al[idx] = (1.0 + (ax * ax + ay * ay + az * az)) * s.invTm2;
            l_ax[idx] = ax; l_ay[idx] = ay; l_az[idx] = az;
            const double *pbx = l_bx + c * J, *pby = l_by + c * J, *pbz = l_bz + c * J;
            for (int jt = 0; jt < p.jtiles; jt++) {
                double m = 1.0e300;
                for (int q2 = 0; q2 < p.JT; q2++) {
                    const int j = min(jt * p.JT + q2, J - 1);
                    m = fmin(m, fq_beta(ax, ay, az, pbx[j], pby[j], pbz[j], s.invTm2));
                }
                l_bm[idx * p.jtiles + jt] = m;
            }
        }
        __syncthreads();
        // ---- phase 2d: the unit-level bounds ----
        {
            const int R = p.R, JT = p.JT;
            for (int idx = tid; idx < ncb * p.rblocks; idx += kFqThreads) {
                const int c = idx / p.rblocks, rb = idx - c * p.rblocks;
                double amin = 1.0e300, Amax = 0.0, Wmax = 0.0;
                for (int r = 0; r < R; r++) {
                    const int k = rb * R + r, kc = min(k, K - 1);
                    amin = fmin(amin, l_al[c * K + kc]);
                    if (k < K) { Amax = fmax(Amax, fabs(l_A[c * K + kc])); Wmax = fmax(Wmax, fabs(l_W[c * K + kc])); }
                }
                double *o = l_ub + c * ubs + 3 * rb;
                o[0] = amin; o[1] = Amax; o[2] = Wmax;
            }
            for (int idx = tid; idx < ncb * p.rblocks * p.jtiles; idx += kFqThreads) {
                const int c = idx / (p.rblocks * p.jtiles), rem = idx - c * (p.rblocks * p.jtiles), rb = rem / p.jtiles, jt = rem - rb * p.jtiles;
                double bmm = 1.0e300;
                for (int r = 0; r < R; r++) bmm = fmin(bmm, l_bm[(c * K + min(rb * R + r, K - 1)) * p.jtiles + jt]);
                l_ub[c * ubs + 3 * p.rblocks + rb * p.jtiles + jt] = bmm;
            }
            for (int idx = tid; idx < ncb * p.jtiles; idx += kFqThreads) {
                const int c = idx / p.jtiles, jt = idx - c * p.jtiles;
                double gmin = 1.0e300, Bmax = 0.0;
                for (int q2 = 0; q2 < JT; q2++) {
                    const int j2 = min(jt * JT + q2, J - 1);
                    gmin = fmin(gmin, l_ga[c * J + j2]);
                    Bmax = fmax(Bmax, fabs(l_B[c * J + j2]));
                }
                double *o = l_ub + c * ubs + 3 * p.rblocks + p.rblocks * p.jtiles + 2 * jt;
                o[0] = gmin; o[1] = Bmax;
            }
            __syncthreads();
            // ---- phase 2e: the cell's power-of-two scale.  |p.dsigma| <= mTmax max_k |A_k| + pTmax max_j |B_j| max_k |W_k| < 2^e_c for every lane, so
            // with A_k, W_k stored times 2^-e_c the main kernel's outflow clamp max(p.dsigma, 0) is the VOP3 clamp of the fma that forms
            // p.dsigma (one instruction fewer per evaluation); 2^e_c comes back inside the exponential, whose shift constant the header
            // carries as kExpShift + e_c.  Exact: powers of two.  df_mode 4 with outflow only (df_mode 3 multiplies |renorm| in the lanes);
            // the unit bounds of the header (max |A_k|, max |W_k|: the cull threshold) stay unscaled. ----
            if (tid < ncb) {
                int ec = 0;
                if (p.scale_rows) {
                    const double *ub = l_ub + tid * ubs;
                    double Amax = 0.0, Wmax = 0.0, Bmax = 0.0;
                    for (int rb = 0; rb < p.rblocks; rb++) { Amax = fmax(Amax, ub[3 * rb + 1]); Wmax = fmax(Wmax, ub[3 * rb + 2]); }
                    for (int jt = 0; jt < p.jtiles; jt++) Bmax = fmax(Bmax, ub[3 * p.rblocks + p.rblocks * p.jtiles + 2 * jt + 1]);
                    const double bound = (p.mTmax * Amax + p.pTmax * Bmax * Wmax) * 1.0000001;
                    if (bound >= 1.0 && bound < 1.0e300) ec = __builtin_amdgcn_frexp_exp(bound);   // bound < 2^ec
                }
                cs[tid].shiftc = kExpShift + (double)ec;
            }
            __syncthreads();
            if (p.scale_rows) {
                for (int idx = tid; idx < ncb * K; idx += kFqThreads) {
                    const double sc = ldexp_fast(1.0, -(int)(cs[idx / K].shiftc - kExpShift));
                    l_A[idx] *= sc; l_W[idx] *= sc;
                }
                __syncthreads();
            }
        }
        // ---- phase 3: unit records (layout of cf_device.h, slots as in cf_feqmod.h): one unit per wave at a time, in two kinds of trip.  Raw
        // trips, lanes <-> the header and the four scalars of every row: one descriptor read, one source read, no betaf arithmetic.  Beta trips,
        // lanes <-> (row, phi column): no descriptor, six reads and fq_beta.  (One loop over all elements paid for a descriptor decode AND a
        // betaf on every lane: 4.8 of the kernel's 6.9 ms in 3+1D, 2.9 of 4.3 ms in 2+1D.) ----
        {
            const int JT = p.JT, R = p.R;
            const int HDR = 4 * JT, RWD = 4 + JT, REC = HDR + R * RWD;
            const int NRAW = HDR + 4 * R, NBETA = R * JT;
            const int invJT = (1 << 20) / JT + 1;                       // (t * invJT) >> 20 == t / JT for t < 4096, JT <= 16
            const int wave = tid >> 6, lane = tid & 63;
            const int cSd = (int)(sizeof(FqScal) / sizeof(double));
            int n = 0;
            for (int c = 0; c < ncb; c++) {
                const int64_t cell = cbase + c;
                const double i2c = cs[c].invTm2;
                for (int jt = 0; jt < p.jtiles; jt++) {
                    for (int rb = 0; rb < p.rblocks; rb++, n++) {
                        if ((n & (kFqThreads / 64 - 1)) != wave) continue;
                        int64_t unit;
                        if (p.dim3) unit = (int64_t)(jt * p.rblocks + rb) * p.n_cells + cell;
                        else unit = ((int64_t)jt * p.n_cells + cell) * p.rblocks + rb;
                        double *o = p.TS + unit * REC;
                        const int cJ = c * J, cK = c * K;
                        for (int t = lane; t < NRAW; t += 64) {
                            int e = t;
                            if (t >= HDR) { const int q = t - HDR; e = HDR + (q >> 2) * RWD + (q & 3); }
                            const int2 d = desc[e];
                            const int jj = d.y & 0xffff, r = d.y >> 16;
                            const int jcl = min(jt * JT + jj, J - 1);
                            const int k = rb * R + r, kcl = min(k, K - 1);
                            const int msel = (d.x >> 16) & 7;
                            const int add = msel == 0 ? cJ + jcl : msel == 1 ? cK + kcl : msel == 2 ? c * ubs + 3 * rb
                                          : msel == 3 ? c * ubs + rb * p.jtiles + jt : msel == 4 ? c * ubs + 2 * jt : msel == 5 ? c * cSd
                                          : (cK + kcl) * p.jtiles + jt;
                            const bool zero = ((d.x >> 22) & 1) | (((d.x >> 21) & 1) & (k >= K));
                            const double raw = lds[zero ? 0 : (d.x & 0xffff) + add];
                            o[e] = zero ? 0.0 : raw;
                        }
                        for (int t = lane; t < NBETA; t += 64) {
                            const int r = (t * invJT) >> 20, jj = t - r * JT;
                            const int kcl = min(rb * R + r, K - 1), jcl = min(jt * JT + jj, J - 1);
                            // betaf_jk = 2 (a'_k . b'_j) / T_mod^2 (padding rows and columns repeat the last real one)
                            o[HDR + r * RWD + 4 + jj] = fq_beta(l_ax[cK + kcl], l_ay[cK + kcl], l_az[cK + kcl], l_bx[cJ + jcl], l_by[cJ + jcl], l_bz[cJ + jcl], i2c);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

size_t prep_feqmod_lds_bytes(int nT, int nj, int ngl, int J, int K, int jtiles, int rblocks, int rec)
{
    const int cb = fq_batch_cells(K);
    const size_t bounds = (size_t)cb * (3 * (size_t)rblocks + (size_t)rblocks * jtiles + 2 * (size_t)jtiles);
    return sizeof(double) * ((size_t)nT * 7 + (size_t)nj * 5 + (size_t)ngl * 4 + (size_t)cb * (6 * K + 5 * J + K * jtiles) + bounds + (size_t)rec) + sizeof(FqScal) * cb;   // + one int2 per record element
}

hipError_t launch_prep_feqmod(const FqPrepParams &p, hipStream_t st)
{
    if (p.n_cells <= 0) return hipSuccess;
    const int cb = fq_batch_cells(p.K);
    const int nbatch = (p.n_cells + cb - 1) / cb;
    const int grid = nbatch < 4096 ? nbatch : 4096;
    const size_t lds = prep_feqmod_lds_bytes(p.spl.n, p.nj, p.ngl, p.J, p.K, p.jtiles, p.rblocks, 4 * p.JT + p.R * (4 + p.JT));
    if (cb == kFqCB3) hipLaunchKernelGGL(cf_prep_feqmod<kFqCB3>, dim3(grid), dim3(kFqThreads), lds, st, p);
    else hipLaunchKernelGGL(cf_prep_feqmod<kFqCB>, dim3(grid), dim3(kFqThreads), lds, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// df_mode 3 renormalisation, smooth_kernels.cpp:747-777: renorm = n_linear / n_mod per (cell, species); the degeneracy
// cancels in the ratio, so one value per (mass, sign) class serves all its species.
// ------------------------------------------------------------------------------------------------
constexpr int kGlMax = 256;   // n_gla <= 256 (checked by the plan)
__global__ void __launch_bounds__(256)
cf_feqmod_renorm(const double *__restrict__ CR, const double *__restrict__ gl, int ngl, const double *__restrict__ cls_mass,
                 const double *__restrict__ cls_sign, const double *__restrict__ cls_baryon, int ncls, int n_cells, int include_bulk,
                 int dim3, double *__restrict__ RN)
{
    // node constants once per workgroup: w p e^p (alpha = 1 nodes: neq_int, J10_int) and w e^p (alpha = 2 nodes: J20_int) -- the
    // integrands of gt_neq / gt_J10 / gt_J20 above with the node-only exponential taken out (one exp per node and integral
    // instead of two; 25 -> 17 ms per 1e6 cells x 75 classes)
    __shared__ double l_p1[kGlMax], l_c1[kGlMax], l_p2[kGlMax], l_c2[kGlMax];
    for (int k = threadIdx.x; k < ngl; k += blockDim.x) {
        const double p1 = gl[k], p2 = gl[2 * ngl + k];
        l_p1[k] = p1 * p1; l_c1[k] = gl[ngl + k] * (p1 * exp(p1));
        l_p2[k] = p2 * p2; l_c2[k] = gl[3 * ngl + k] * exp(p2);
    }
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_cells * ncls) return;
    const int cell = (int)(idx / ncls), c = (int)(idx - (int64_t)cell * ncls);
    const double *cr = CR + (int64_t)cell * kCrRec;
    double renorm = 0.0;
    if (cr[5] != 0.0) {
        const double T = cr[0], T_mod = cr[1], dn_fact = cr[2], F = cr[3], detA = cr[4];
        renorm = 1.0;
        if (include_bulk) {
            const double two_pi2_hbarC3 = 2.0 * M_PI * M_PI * (kHbarC * kHbarC * kHbarC);
            const double neq_fact = T * T * T / two_pi2_hbarC3, J20_fact = T * neq_fact;
            const double nmod_fact = T_mod * T_mod * T_mod / two_pi2_hbarC3;
            const double mass = cls_mass[c], sign = cls_sign[c];
            const double baryon = cls_baryon ? cls_baryon[c] : 0.0;
            const double alphaB = cr[6], G = cr[7], alphaB_mod = alphaB + dn_fact * G;   // :637
            const double chem = baryon * alphaB, chem_mod = baryon * alphaB_mod;
            const double mbar = mass / T, mbar_mod = mass / T_mod;
            const double mb2 = mbar * mbar, mm2 = mbar_mod * mbar_mod;
            double s_neq = 0.0, s_n10 = 0.0, s_j20 = 0.0, s_mod = 0.0;
            for (int k = 0; k < ngl; k++) {
                const double c1 = l_c1[k];
                // (exp_full / sqrt_nr of cf_math.h instead of libm: 1e-15 relative, arguments of order 1..1e3, inf beyond 709 like exp();
                // the node constants above stay libm, once per workgroup)
                // The quotients are reciprocals by v_rcp_f64 + two Newton steps (1e-16); an exponential that overflowed is held at 1e300
                // so that 1/q stays a number: such a node adds < 1e-300 of its weight instead of exactly 0.
                const double e = __builtin_fmin(exp_full(sqrt_nr(l_p1[k] + mb2) - chem), 1.0e300), r = rcp_nr(e + sign);
                s_neq = __builtin_fma(c1, r, s_neq);
                if (baryon != 0.0) s_n10 = __builtin_fma(c1, (e * r) * r, s_n10);
                s_mod = __builtin_fma(c1, rcp_nr(__builtin_fmin(exp_full(sqrt_nr(l_p1[k] + mm2) - chem_mod), 1.0e300) + sign), s_mod);
                const double E2 = sqrt_nr(l_p2[k] + mb2), e2 = __builtin_fmin(exp_full(E2 - chem), 1.0e300), r2 = rcp_nr(e2 + sign);
                s_j20 = __builtin_fma(l_c2[k], E2 * ((e2 * r2) * r2), s_j20);
            }
            const double neq = neq_fact * s_neq;
            const double N10 = (baryon != 0.0) ? baryon * neq_fact * s_n10 : 0.0;     // N10_fact = neq_fact, :717
            const double J20 = J20_fact * s_j20;
            const double n_linear = neq + dn_fact * (neq + N10 * G + J20 * F / T / T);   // :760
            const double n_mod = nmod_fact * s_mod;
            renorm = n_linear / n_mod;
        }
        if (isnan(renorm) || isinf(renorm)) renorm = 0.0;                             // :768-772: species skipped in this cell
        else {
            if (dim3) renorm /= detA;
            renorm = fabs(renorm);
        }
    }
    RN[idx] = renorm;
}

hipError_t launch_feqmod_renorm(const double *CR, const double *gl, int ngl, const double *cls_mass, const double *cls_sign,
                                const double *cls_baryon, int ncls, int n_cells, int include_bulk, int is_dim3, double *RN,
                                hipStream_t st)
{
    if (n_cells <= 0) return hipSuccess;
    const int64_t n = (int64_t)n_cells * ncls;
    hipLaunchKernelGGL(cf_feqmod_renorm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, CR, gl, ngl, cls_mass, cls_sign,
                       cls_baryon, ncls, n_cells, include_bulk, is_dim3, RN);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// cf_main_feqmod: same task decomposition, LDS staging and partial layout as cf_main_tile (cf_kernels.hip).
// Per evaluation: X2 = fma(mT pT, betaf, mT^2 alphaf + pT^2 gammaf); X = sqrt (v_rsq_f64 + 6 FMA); z = exp(-X) (17 ops);
// f = z / (1 + sign z) (v_rcp_f64 + 3); acc += max(pds, 0) f.  ~36 fp64 VALU instructions, nothing amortised over the
// tile except the row and column products.  Rows whose X exceeds 745.2 for every lane and phi of the tile (z = +0
// exactly) are culled as in cf_main_tile.
// ------------------------------------------------------------------------------------------------
// dev: cycle accounting of the PROF instantiation (-DIS3D_DEV builds, IS3D_DEV_PROF=1; tools/gpu_ab.py): sums over waves of s_memtime intervals
//   [0] waves  [1] wave lifetime  [2] staging issue  [3] vmcnt + barrier waits  [4] dead units  [5] live units  [6] their bounds + row tests
//   [7] dead units (count)  [8] live units (count)  [9] first-wave share of [3]  [10] threshold refresh  [11] vmcnt part of [3]  [12] prologue
//   [13] live rows evaluated (count)  [14] cycles inside the evaluated rows
__device__ unsigned long long g_prof_fq[16];

// ROWS: how the R rows of a 3+1D unit are walked.
//   0  software pipeline (row r+1's twelve operands fetched before row r's evaluations), every row forms its own threshold and lower bound:
//      the round-1 form, kept for 2+1D (61-row units) and as the A/B reference (kernel_variant 5)
//   1  (kernel_variant 6; A/B) liveness of all rows first, from ONE LDS round trip at the head of the unit that also brings the unit-level
//      bounds: a row's test is {alphaf_k, min_j betaf_jk} (one 16-byte read) against the UNIT's threshold -- 4 instructions and no further read
//      for a dead row instead of 6 reads and ~17 instructions; rows that pass are tested again against their own (tighter) threshold when their
//      operands are there, so the set of culled rows is exactly that of ROWS = 0 (same status counters, bitwise the same spectrum as ROWS = 0)
//   2  (default, 3+1D) as 1 without the second, row-level test: 0.4 points fewer rows culled (60.0 against 60.4 % on the config-3 surface), ten
//      instructions fewer per evaluated row -- 499 against 541 (ROWS = 1) and 558 ms (ROWS = 0).  zero_skip on / off stay bitwise identical.
//   3  (default, 2+1D; kernel_variant 7) the rolled row loop of ROWS = 0 with the test of ROWS = 2: the UNIT's threshold (max |A_k|, max |W_k|,
//      max |B_j| in free header slots) against fma(mT pT, min_j betaf_jk, mT^2 alphaf_k + min_j pT^2 gammaf_j) -- the row carries min_j betaf_jk
//      in its fourth slot.  Two instructions and a vote per row; ROWS = 0 formed a threshold per row (~9 instructions) and the minimum over the
//      row's JT X^2 (JT more): 3.4 of its 30.7 instructions per evaluation.
// LDSD: doubles of unit records per LDS buffer (two buffers per workgroup).  1536 (13 units of the 8 x 7 tile: 24 KB per workgroup) for the
// two-wave workgroups that share a stream; 1100 (9 units, 18 KB) for ONE-wave workgroups (g.wpb == 1), eight of which must fit a CU's 160 KB:
// a wave that stages its own stream has no barrier partner to wait for -- on the config-3 surface 9.7 % of all wave cycles were spent at the
// per-batch barrier (the two lane-waves of a workgroup cull differently and drift apart), against 0.8 % issuing the staging loads that double.
// (Measured and dropped, round 4: the 6 x 7 tile at 166 VGPRs with one-wave workgroups of 7-unit batches = THREE waves per SIMD -- 487.1 against
// 488.0 ms for the 8 x 7 tile at two, with culling off 1114 against 1083: the third wave returns what the smaller tile's amortisation costs.
// profiles/r04_ab_feqmod_3w.log)
template <bool DIM3, bool OUTFLOW, bool MODE3, int JT, int R, bool BARYON = false, int ROWS = 0, bool PROF = false, int LDSD = 1536>
__global__ void __launch_bounds__(512)
cf_main_feqmod(const double *__restrict__ TS, const double *__restrict__ lane_mT, const double *__restrict__ lane_pT,
               const double *__restrict__ lane_sign, const double *__restrict__ RN, const int32_t *__restrict__ lane_cls,
               int ncls, double *__restrict__ partial, unsigned long long *__restrict__ stats, MainGeom g,
               const double *__restrict__ lane_b, const int32_t *__restrict__ lane_sub)
{
    constexpr int HDR = 4 * JT;
    constexpr int RW = 4 + JT;
    constexpr int REC = HDR + R * RW;
    constexpr int UB = (LDSD / REC) > 0 ? (LDSD / REC) : 1;
    constexpr int BUF2 = UB * REC / 2;
    constexpr int NLD = (BUF2 + 127) / 128;
    // evaluations per v_rcp_f64: 8 where the tile allows (3+1D since round 4: 466.6 -> 463.0 ms on the config-3 surface, culling off 1047.6 -> 1039.5,
    // 224 VGPRs; the two A/B row walks keep 4 -- the pipelined one spills with 8, and kernel_variant 5 and 6 stay bitwise equal to each other)
    constexpr int RB = (DIM3 && ROWS < 2) ? (JT % 4 == 0 ? 4 : (JT % 3 == 0 ? 3 : 2)) : (JT % 8 == 0 ? 8 : (JT % 4 == 0 ? 4 : (JT % 3 == 0 ? 3 : 2)));
    static_assert(REC % 2 == 0 && JT % RB == 0, "unit records must be 16-byte multiples (JT even)");
    static_assert(ROWS == 0 || (DIM3 && JT >= 4 && ROWS <= 2) || (!DIM3 && JT >= 4 && ROWS == 3), "the row tests against the unit threshold need the bounds in the header");
    constexpr bool MASK = ROWS == 1 || ROWS == 2;     // 3+1D: liveness of all rows of a unit first
    constexpr FqRowSlots SL = fq_row_slots(DIM3);
    constexpr int BUFP = ((BUF2 * 16 + 1023) / 1024) * 64;   // the batch as whole 1-KiB staging pieces (64 double2 each)
    __shared__ double2 lbuf[2][BUFP + RW / 2 + 1];
    (void)NLD;

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int grp = q % g.G;
    const int stream = (q / g.G) * 8 + xcd;
    if (stream >= g.NT) return;
    int sidx = stream;
    const int jt = sidx % g.jtiles; sidx /= g.jtiles;
    const int kt = sidx % g.ktiles; sidx /= g.ktiles;
    const int chunk = sidx;
    const int nthr = blockDim.x;
    const int lw = grp * g.wpb + (tid >> 6);
    const bool wave_active = lw * 64 < g.Lpad;
    const int l = wave_active ? lw * 64 + (tid & 63) : 0;

    const int J = g.J, K = g.K;
    const double mT = lane_mT[l], pT = lane_pT[l], sign = lane_sign[l];
    const double mT2 = mT * mT, mTpT = mT * pT, pT2 = pT * pT;
    const double bq = BARYON ? lane_b[l] : 0.0;
    // Unit-strided lanes (2+1D, g.split = S > 1; cf_main_tile has the same): a surface with few momentum bins gives every bin S lane
    // slots, slot s takes the units u = s (mod S) of the stream -- all units of a stream add into the same JT accumulators of a
    // bin -- and cf_finalize adds a bin's slots in slot order.  S divides the units per cell, so the S units of a step belong to
    // one cell (one renormalisation factor); the row votes then span lanes on different units and skip a row only when it is
    // dead for every one of them.
    const int S = (!DIM3 && g.split > 1) ? g.split : 1;
    const int sub_off = (!DIM3 && g.split > 1 && lane_sub) ? lane_sub[l] * REC : 0;
    int c0, c1;
    chunk_cells(g, chunk, c0, c1);
    const int n_units = (c1 - c0) * g.upc;
    const int s_tile = DIM3 ? (jt * g.ktiles + kt) : jt;
    const double2 *src = (const double2 *)(TS + (((int64_t)s_tile * g.n_cells + c0) * g.upc) * REC);
    const int nb = (n_units + UB - 1) / UB;
    const double *rn_col = MODE3 ? RN + lane_cls[l] : nullptr;

    constexpr int NACC = DIM3 ? JT * R : JT;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.0;

    int n_rows = 0, n_dead = 0;
    unsigned long long pf_stage = 0, pf_wait = 0, pf_dead = 0, pf_live = 0, pf_hdr = 0, pf_nd = 0, pf_nl = 0, pf_thr = 0, pf_t0 = 0, pf_u0 = 0, pf_vm = 0, pf_pro = 0,
                       pf_nr = 0, pf_rows = 0;
    if constexpr (PROF) pf_t0 = clock64();
    // Accumulator-relative row cull (with the outflow clamp, g.zskip == 2), as in cf_main_tile: every term of a row is
    // pds w with 0 <= pds <= pmax = |mT A_k| + max_j |pT B_j| |W_k| and w = z/(1 + sign z) <= 2 z for z <= 1/2,
    // z = e^(cm - X) <= e^(cm - sqrt(x2lb)); the accumulators only grow and fma(pds, w, acc) == acc when pds w < ulp(acc)/2.
    // With pmax < 2^ep and min(acc) >= 2^(acc_e - 1): X > cm + (ep - acc_e + 58) ln 2 leaves every accumulator unchanged.
    constexpr bool RELCULL = OUTFLOW;
    // Outflow clamp by the VOP3 clamp modifier (df_mode 4): cf_prep_feqmod stores a cell's A_k, W_k times 2^-e_c with 2^e_c above every lane's
    // |p.dsigma| (phase 2e), so max(p.dsigma, 0) 2^-e_c is the clamp to [0, 1] of the fma that forms it -- one instruction fewer per
    // evaluation than fma + v_max_f64; the factor 2^e_c comes back in z' = 2^e_c exp(-X) (exp_p9_scaled: the header carries the shift
    // constant), with sign 2^-e_c in 1 + sign z.  All exact: the spectrum's bits are those of the unscaled form.
    constexpr bool CLAMP = OUTFLOW && !MODE3;
    int acc_e = -100000;   // frexp exponent of a (stale) minimum over the lane's accumulators; refreshed once per batch
    auto process_unit = [&](const double *U, double rn) -> bool {
        double pTB[JT], pT2g[JT];
        const double rpT = MODE3 ? rn * pT : pT, rmT = MODE3 ? rn * mT : mT;
        const double cm = BARYON ? bq * U[2] : 0.0;   // chem_mod = baryon * alpha_B,mod (:742): f = |renorm| / (exp(E_mod/T_mod - chem_mod) + sign)
        // exact-zero culling: exp(cm - X) == +0 needs X > 745.25 + cm
        const double xcut = BARYON ? 745.25 + __builtin_fmax(cm, 0.0) : 745.25;
        const double x2cut = BARYON ? xcut * xcut : 555400.0;
        // threshold on X^2 from an upper bound pmax of the row's (or the unit's) p.dsigma: see RELCULL above
        auto x2_threshold = [&](double pmax, int eadd = 0) {   // eadd: e_c when pmax is formed from the cell's scaled rows
            const int de = __builtin_amdgcn_frexp_exp(pmax) + eadd - acc_e + 58;
            const double xc = cm + (double)(de > 1 ? de : 1) * 0.6931471805599453;
            const double xcp = __builtin_fmax(xc, 0.0);
            return __builtin_fmin(x2cut, xcp * xcp);
        };
        double x2c_u = x2cut;
        double ral[MASK ? R : 1], rbm[MASK ? R : 1];
        double gmin_u = 0.0;
        if constexpr (DIM3 && JT >= 4) {
            // Unit-level cull (3+1D; bounds from cf_prep_feqmod in free header slots): X^2 >= mT^2 min_k alphaf + mT pT min_jk betaf + pT^2
            // min_j gammaf for every evaluation of the unit, and the threshold of every row is at most the one formed with max_k |A_k|,
            // max_k |W_k| and max_j |B_j| (all roundings monotone): a unit that fails here would have every row culled below, so its
            // header work and its R row fetches are skipped -- bitwise the same spectrum
            if constexpr (MASK) {
                // ... and with them, in the same LDS round trip, the operands of the R row tests
                const double u_al = U[3], u_be = U[6], u_ga = U[7], u_A = U[10], u_W = U[11], u_B = U[14];
#pragma unroll
                for (int r = 0; r < R; r++) { ral[r] = U[HDR + r * RW + SL.AL]; rbm[r] = U[HDR + r * RW + SL.BM]; }
                __builtin_amdgcn_sched_barrier(0);
                gmin_u = pT2 * u_ga;                   // == min_j pT^2 gammaf_j (pT^2 >= 0, monotone rounding)
                if (g.zskip) {
                    if (RELCULL && g.zskip == 2) x2c_u = x2_threshold(__builtin_fma(__builtin_fabs(rpT * u_B), u_W, __builtin_fabs(rmT * u_A)));
                    const double x2lb_u = __builtin_fma(mTpT, u_be, mT2 * u_al + gmin_u);
                    if (__all(x2lb_u > x2c_u)) { n_rows += R; n_dead += R; return false; }
                }
            } else if (g.zskip) {
                if (RELCULL && g.zskip == 2) x2c_u = x2_threshold(__builtin_fma(__builtin_fabs(rpT * U[14]), U[11], __builtin_fabs(rmT * U[10])));
                const double x2lb_u = __builtin_fma(mTpT, U[6], mT2 * U[3] + pT2 * U[7]);
                if (__all(x2lb_u > x2c_u)) { n_rows += R; n_dead += R; return false; }
            }
        }
        if constexpr (ROWS == 3) {
            // 2+1D: the unit's threshold and min_j pT^2 gammaf_j from the bounds cf_prep_feqmod left in the header; no unit-level cull (a unit is
            // 31 of a cell's 241 eta rows: its row tests are 2 % of a live unit's work)
            const double u_ga = U[7], u_A = U[10], u_W = U[11], u_B = U[14];
            gmin_u = pT2 * u_ga;
            if (RELCULL && g.zskip == 2) x2c_u = x2_threshold(__builtin_fma(__builtin_fabs(rpT * u_B), u_W, __builtin_fabs(rmT * u_A)));
        }
        double shiftc = kExpShift, signc = sign;
        int eu = 0;
        if constexpr (CLAMP) {
            shiftc = U[15];
            eu = (int)(shiftc - kExpShift);
            signc = ldexp_fast(sign, -eu);
        }
        unsigned live = (1u << (MASK ? R : 1)) - 1u;
        if constexpr (MASK) {
            if (g.zskip) {
                live = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double x2lb = __builtin_fma(mTpT, rbm[r], mT2 * ral[r] + gmin_u);
                    live |= __all(x2lb > x2c_u) ? 0u : (1u << r);
                }
            }
            n_rows += R;
            n_dead += R - __builtin_popcount(live);
            if constexpr (PROF) pf_hdr += clock64() - pf_u0;
        }
        double g_min = 1.0e300, pb_max = 0.0;
#pragma unroll
        for (int jj = 0; jj < JT; jj++) {
            pTB[jj] = rpT * U[4 * jj + 0];
            pT2g[jj] = pT2 * U[4 * jj + 1];
            if (DIM3 && ROWS == 0) g_min = __builtin_fmin(g_min, pT2g[jj]);
            if (RELCULL && ROWS < 2) pb_max = __builtin_fmax(pb_max, __builtin_fabs(pTB[jj]));
        }
        if (MASK) g_min = gmin_u;
        struct Row { double v[RW]; };
        auto fetch = [&](Row &rw, const double *row) {
#pragma unroll
            for (int i = 0; i < RW; i++) rw.v[i] = row[i];
        };
        auto evals = [&](const Row &rw, int r) {
            const double mTA = rmT * rw.v[SL.A];
            const double a = mT2 * rw.v[SL.AL];
            const double W = rw.v[SL.W];
            if (!MASK) n_rows += 1;
            double X2[JT];
            double x2c = x2cut;                                                     // X > 745.25 (+ cm): exp(cm - X) == +0
            if (RELCULL && ROWS < 2 && g.zskip == 2) x2c = x2_threshold(__builtin_fma(pb_max, __builtin_fabs(W), __builtin_fabs(mTA)), eu);
            if constexpr (DIM3) {
                // X^2_j >= mT^2 alphaf_k + mT pT min_j betaf_jk + pT^2 min_j gammaf_j (mT pT >= 0; the row carries min_j betaf_jk):
                // two instructions per row instead of a minimum per evaluation, and a culled row forms no X^2 at all
                if constexpr (ROWS != 2) {
                    const double x2lb = __builtin_fma(mTpT, rw.v[SL.BM], a + g_min);
                    if (g.zskip && __all(x2lb > x2c)) { n_dead += 1; return; }
                }
#pragma unroll
                for (int jj = 0; jj < JT; jj++) X2[jj] = __builtin_fma(mTpT, rw.v[4 + jj], a + pT2g[jj]);
            } else if constexpr (ROWS == 3) {
                // X^2_j >= mT^2 alphaf_k + mT pT min_j betaf_jk + min_j pT^2 gammaf_j, against the unit's threshold (mT pT >= 0, monotone roundings)
                const double x2lb = __builtin_fma(mTpT, rw.v[SL.BM], a + gmin_u);
                if (g.zskip && __all(x2lb > x2c_u)) { n_dead += 1; return; }
#pragma unroll
                for (int jj = 0; jj < JT; jj++) X2[jj] = __builtin_fma(mTpT, rw.v[4 + jj], a + pT2g[jj]);
            } else {
                // 2+1D, the round-1 form (A/B): a threshold per row and the minimum over the row's X^2
                double x2min = 1.0e300;
#pragma unroll
                for (int jj = 0; jj < JT; jj++) {
                    X2[jj] = __builtin_fma(mTpT, rw.v[4 + jj], a + pT2g[jj]);
                    x2min = __builtin_fmin(x2min, X2[jj]);
                }
                if (g.zskip && __all(x2min > x2c)) { n_dead += 1; return; }
            }
            if constexpr (PROF) pf_nr++;
            // the reciprocals of RB evaluations share one v_rcp_f64 (rcp_batch, cf_math.h): d = 1 + sign z lies in (1e-3, 2]
#pragma unroll
            for (int j0 = 0; j0 < JT; j0 += RB) {
                double zz[RB], d[RB], inv[RB];
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const double X = sqrt_g1(X2[j0 + i]);   // 3e-15 relative: e^-X moves by X * 3e-15
                    // degree 9 + one-fma reduction (cf_math.h): 7e-14; cf_prep_feqmod keeps X = |A^-1 p|/T_mod below 1e9 (status[7])
                    zz[i] = CLAMP ? exp_p9_scaled(-X, shiftc) : exp_p9(BARYON ? cm - X : -X);
                    d[i] = __builtin_fma(signc, zz[i], 1.0);
                }
                rcp_batch<RB>(d, inv);
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int jj = j0 + i;
                    double pds;
                    if constexpr (CLAMP) pds = fma_clamp01(pTB[jj], W, mTA);
                    else {
                        pds = __builtin_fma(pTB[jj], W, mTA);
                        if (OUTFLOW) pds = __builtin_fmax(pds, 0.0);
                    }
                    const double w = zz[i] * inv[i];
                    if (DIM3) acc[jj * R + r] = __builtin_fma(pds, w, acc[jj * R + r]);
                    else acc[jj] = __builtin_fma(pds, w, acc[jj]);
                }
            }
        };
        const double *rows = U + HDR;
        if constexpr (MASK) {
            unsigned long long pr = 0;
            if constexpr (PROF) pr = clock64();
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (live & (1u << r)) {
                    Row rw;
                    fetch(rw, rows + r * RW);
                    evals(rw, r);
                }
            }
            if constexpr (PROF) pf_rows += clock64() - pr;
        } else if (DIM3) {
            Row cur, nxt;
            fetch(cur, rows);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r + 1 < R) fetch(nxt, rows + (r + 1) * RW);
                evals(cur, r);
                if (r + 1 < R) cur = nxt;
            }
        } else {
            // rolled, two rows per trip (see cf_main_tile); for even R the last fetch reads the head of the next unit or the pad
            Row cur, nxt;
            fetch(cur, rows);
#pragma clang loop unroll(disable)
            for (int r = 0; r + 1 < R; r += 2) {
                fetch(nxt, rows + (r + 1) * RW);
                evals(cur, 0);
                fetch(cur, rows + (r + 2) * RW);
                evals(nxt, 0);
            }
            if (R & 1) evals(cur, 0);
        }
        return true;
    };

    // staging: the next batch by direct-to-LDS loads (stage_pieces, cf_math.h), issued before the current batch is consumed
    auto stage = [&](int ib, int buf) { stage_pieces<BUFP / 64>((const char *)(src + (int64_t)ib * BUF2), lbuf[buf], tid, nthr); };
    // The lane constants must have ARRIVED before the batch loop: the compiler sinks the loads of restrict-qualified data to their first
    // use and waits for them there (s_waitcnt vmcnt(0) in every row of the unrolled loop) -- a wait that also covers the direct-to-LDS
    // loads of the next batch, which it does not know about, i.e. it would expose the staging latency in every batch
    asm volatile("" :: "v"(mT), "v"(pT), "v"(sign), "v"(mT2), "v"(mTpT), "v"(pT2), "v"(bq), "v"(sub_off) : "memory");
    // df_mode 3: the renormalisation factor of a (cell, class) is a per-lane global load.  The compiler waits for it with
    // s_waitcnt vmcnt(0), which also drains the direct-to-LDS loads of the next batch it does not know about, so no such load may be
    // consumed while a batch is young: the factors of a batch's first two steps are loaded at the end of the batch before (they arrive
    // under its closing wait), step u > 0 loads the factor of step u + 1 while it runs.
    auto rn_at = [&](int u) { return rn_col[(int64_t)(c0 + min(u, n_units - 1) / g.upc) * ncls]; };
    double rn = 1.0, rn1 = 1.0;
    if (MODE3 && nb > 0) { rn = rn_at(0); rn1 = rn_at(S); }
    if (nb > 0) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE3) asm volatile("" :: "v"(rn), "v"(rn1));   // a use: the compiler's own wait for the two loads goes here
        __syncthreads();
        if constexpr (PROF) pf_pro = clock64() - pf_t0;
        for (int ib = 0; ib < nb; ib++) {
            unsigned long long pa = 0;
            if constexpr (PROF) pa = clock64();
            if (ib + 1 < nb) stage(ib + 1, (ib + 1) & 1);
            if constexpr (PROF) pf_stage += clock64() - pa;
            if (wave_active) {
                const int nu = min(UB, n_units - ib * UB);
                const double *base = (const double *)lbuf[ib & 1] + sub_off;
                const int u0 = ib * UB;
                for (int u = 0; u < nu; u += S) {      // nu is a multiple of S (plan)
                    double rn_next = rn1;
                    if (MODE3 && u > 0) rn_next = rn_at(u0 + u + S);
                    if constexpr (PROF) pf_u0 = clock64();
                    const bool lv = process_unit(base + u * REC, rn);
                    if constexpr (PROF) {
                        const unsigned long long d = clock64() - pf_u0;
                        if (lv) { pf_live += d; pf_nl++; } else { pf_dead += d; pf_nd++; }
                    }
                    rn = rn_next;
                }
                if (MODE3) rn1 = rn_at(u0 + nu + S);
                if constexpr (PROF) pa = clock64();
                if (RELCULL && g.zskip == 2 && (((ib + 1) & ib) == 0 || (ib & 31) == 31)) {   // as in cf_main_tile: log2(acc) moves slowly
                    double m = acc[0];
#pragma unroll
                    for (int i = 1; i < NACC; i++) m = __builtin_fmin(m, acc[i]);
                    acc_e = (m > 1.0e-290) ? __builtin_amdgcn_frexp_exp(m) : -100000;
                }
                if constexpr (PROF) pf_thr += clock64() - pa;
            }
            if constexpr (PROF) pa = clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (PROF) pf_vm += clock64() - pa;
            if (MODE3) asm volatile("" :: "v"(rn), "v"(rn1));
            __syncthreads();
            if constexpr (PROF) pf_wait += clock64() - pa;
        }
    }
    if (!wave_active) return;
    if ((tid & 63) == 0) {
        atomicAdd(&stats[2], (unsigned long long)n_rows);
        atomicAdd(&stats[3], (unsigned long long)n_dead);
        if constexpr (PROF) {
            atomicAdd(&g_prof_fq[0], 1ull);
            atomicAdd(&g_prof_fq[1], clock64() - pf_t0);
            atomicAdd(&g_prof_fq[2], pf_stage);
            atomicAdd(&g_prof_fq[3], pf_wait);
            atomicAdd(&g_prof_fq[4], pf_dead);
            atomicAdd(&g_prof_fq[5], pf_live);
            atomicAdd(&g_prof_fq[6], pf_hdr);
            atomicAdd(&g_prof_fq[7], pf_nd);
            atomicAdd(&g_prof_fq[8], pf_nl);
            if ((tid >> 6) == 0) atomicAdd(&g_prof_fq[9], pf_wait);
            atomicAdd(&g_prof_fq[10], pf_thr);
            atomicAdd(&g_prof_fq[11], pf_vm);
            atomicAdd(&g_prof_fq[12], pf_pro);
            atomicAdd(&g_prof_fq[13], pf_nr);
            atomicAdd(&g_prof_fq[14], pf_rows);
        }
    }

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
                        *o = g.first_pass ? acc[jj * R + r] : (*o + acc[jj * R + r]);
                    }
                }
            } else {
                double *o = pp + (int64_t)j * g.Lpad + l;
                *o = g.first_pass ? acc[jj] : (*o + acc[jj]);
            }
        }
    }
}

template <bool DIM3, bool OF, bool M3, int JT, int R, int ROWS, int LDSD>
static void launch_fq_l(const FqMainArgs &a, hipStream_t st)
{
    const int grid = ((a.g.NT + 7) / 8) * 8 * a.g.G;
    if constexpr (M3) {
        if (a.lane_b) {   // include_baryon (df_mode 3 only)
            hipLaunchKernelGGL((cf_main_feqmod<DIM3, OF, M3, JT, R, true, ROWS, false, LDSD>), dim3(grid), dim3(a.g.wpb * 64), 0, st, a.TS, a.lane_mT,
                               a.lane_pT, a.lane_sign, a.RN, a.lane_cls, a.ncls, a.partial, a.stats, a.g, a.lane_b, a.lane_sub);
            return;
        }
    }
#ifdef IS3D_DEV
    if constexpr (DIM3 && OF && !M3 && JT == 8) {
        // dev: the cycle-accounting instantiation (df_mode 4, 8 x 7), synchronous, counters to stderr
        if (dev_env("IS3D_DEV_PROF")) {
            unsigned long long h[16] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_fq), h, sizeof h);
            hipLaunchKernelGGL((cf_main_feqmod<DIM3, OF, M3, JT, R, false, ROWS, true, LDSD>), dim3(grid), dim3(a.g.wpb * 64), 0, st, a.TS, a.lane_mT, a.lane_pT,
                               a.lane_sign, a.RN, a.lane_cls, a.ncls, a.partial, a.stats, a.g, a.lane_b, a.lane_sub);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof_fq), sizeof h);
            const double T = (double)h[1];
            fprintf(stderr, "[prof_fq rows=%d waves/workgroup=%d culling=%s] waves %llu  cycles/wave %.3e  stage %.4f  wait %.4f (first wave of the workgroup %.4f)  dead units %.4f  "
                            "live units %.4f (bounds + row tests %.4f, evaluated rows %.4f)  thr %.4f  vmcnt part of wait %.4f  prologue %.4f | dead units %llu (%.0f cycles each)  "
                            "live units %llu (%.0f cycles each)  rows evaluated %llu (%.0f cycles each)\n",
                    ROWS, a.g.wpb, a.g.zskip == 2 ? "on" : a.g.zskip == 1 ? "exact zeros" : "off", h[0], T / (double)h[0], h[2] / T, h[3] / T, h[9] / T, h[4] / T, h[5] / T,
                    h[6] / T, h[14] / T, h[10] / T, h[11] / T, h[12] / T, h[7], h[7] ? (double)h[4] / h[7] : 0.0, h[8], h[8] ? (double)h[5] / h[8] : 0.0, h[13],
                    h[13] ? (double)h[14] / h[13] : 0.0);
            return;
        }
    }
#endif
    hipLaunchKernelGGL((cf_main_feqmod<DIM3, OF, M3, JT, R, false, ROWS, false, LDSD>), dim3(grid), dim3(a.g.wpb * 64), 0, st, a.TS, a.lane_mT, a.lane_pT,
                       a.lane_sign, a.RN, a.lane_cls, a.ncls, a.partial, a.stats, a.g, a.lane_b, a.lane_sub);
}

template <bool DIM3, bool OF, bool M3, int JT, int R, int ROWS = 0>
static void launch_fq_t(const FqMainArgs &a, hipStream_t st)
{
    if constexpr (!DIM3 && ROWS == 3) {
        // 2+1D, 8 x 31: four units per LDS buffer (4 x 404 doubles), one per lane slot of a bin under unit-strided lanes
        launch_fq_l<DIM3, OF, M3, JT, R, ROWS, 4 * (4 * JT + R * (4 + JT))>(a, st);
    } else {
        if constexpr (DIM3 && JT == 8 && ROWS != 0) {
            // one-wave workgroups (3+1D, 8 x 7): 9-unit LDS batches so that eight workgroups fit a CU, no barrier partner
            if (a.g.wpb == 1) { launch_fq_l<DIM3, OF, M3, JT, R, ROWS, 1100>(a, st); return; }
        }
        launch_fq_l<DIM3, OF, M3, JT, R, ROWS, 1536>(a, st);
    }
}

template <bool DIM3, bool OF, bool M3>
static void launch_fq_variant(int variant, const FqMainArgs &a, hipStream_t st)
{
    // tile shapes of main_tile_shape (cf_kernels.hip): variants 2, 3, 4 (3+1D: rows walked by the row mask from the unit threshold, ROWS = 2:
    // 499 against 558 ms on the config-3 surface, profiles/r04_ab_feqmod.log); variants 5 and 6 are the 8 x 7 tile of variant 3 with ROWS = 0 (the
    // pipelined round-1 form) resp. ROWS = 1 (row mask + the exact per-row thresholds: the culled set of ROWS = 0, 541 ms) for A/B
    // The shipped library holds the defaults (3+1D: 3, 2+1D: 7; the plan maps every other request onto them); the other forms are developer-build A/B.
    if constexpr (!kDevBuild) {
        if constexpr (DIM3) launch_fq_t<DIM3, OF, M3, 8, 7, 2>(a, st);
        else launch_fq_t<DIM3, OF, M3, 8, 31, 3>(a, st);
    } else if constexpr (DIM3) {
        switch (variant) {
        case 3: launch_fq_t<DIM3, OF, M3, 8, 7, 2>(a, st); break;
        case 5: launch_fq_t<DIM3, OF, M3, 8, 7, 0>(a, st); break;
        case 6: launch_fq_t<DIM3, OF, M3, 8, 7, 1>(a, st); break;
        case 4: launch_fq_t<DIM3, OF, M3, 4, 7, 2>(a, st); break;
        default: launch_fq_t<DIM3, OF, M3, 6, 7, 2>(a, st); break;
        }
    } else {
        switch (variant) {
        case 7: launch_fq_t<DIM3, OF, M3, 8, 31, 3>(a, st); break;
        case 3: launch_fq_t<DIM3, OF, M3, 12, 61>(a, st); break;
        case 4: launch_fq_t<DIM3, OF, M3, 4, 61>(a, st); break;
        default: launch_fq_t<DIM3, OF, M3, 8, 61>(a, st); break;
        }
    }
}

hipError_t launch_main_feqmod(int variant, int dim3, int outflow, int mode3, int baryon, const FqMainArgs &a, hipStream_t st)
{
    if (a.g.n_cells <= 0) return hipSuccess;
    if ((baryon != 0) != (a.lane_b != nullptr) || (baryon && !mode3)) return hipErrorInvalidValue;
    const int sel = (dim3 ? 4 : 0) | (outflow ? 2 : 0) | (mode3 ? 1 : 0);
    switch (sel) {
    case 0: launch_fq_variant<false, false, false>(variant, a, st); break;
    case 1: launch_fq_variant<false, false, true>(variant, a, st); break;
    case 2: launch_fq_variant<false, true, false>(variant, a, st); break;
    case 3: launch_fq_variant<false, true, true>(variant, a, st); break;
    case 4: launch_fq_variant<true, false, false>(variant, a, st); break;
    case 5: launch_fq_variant<true, false, true>(variant, a, st); break;
    case 6: launch_fq_variant<true, true, false>(variant, a, st); break;
    default: launch_fq_variant<true, true, true>(variant, a, st); break;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Ordered compaction of the flagged cells: one workgroup scans the flags in index order, so the list (and with it the
// summation order of cf_feqmod_linear) is the same on every run.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
cf_feqmod_compact(const int32_t *__restrict__ flag, int n, int32_t *__restrict__ list, int32_t *__restrict__ count,
                  unsigned long long *__restrict__ status)
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int off = 0, n1 = 0, n2 = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int f = (i < n) ? flag[i] : 0;
        const bool m = f != 0;
        n1 += (f == 1);
        n2 += (f == 2);
        const unsigned long long bal = __ballot(m);
        const int pre = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int qv = 0; qv < 16; qv++) {
            const int sq = wsum[qv];
            if (qv < w) woff += sq;
            tot += sq;
        }
        if (m) list[off + woff + pre] = i;
        off += tot;
        __syncthreads();
    }
    if (tid == 0) *count = off;
    if (n1) atomicAdd(&status[4], (unsigned long long)n1);
    if (n2) atomicAdd(&status[5], (unsigned long long)n2);
}

hipError_t launch_feqmod_compact(const int32_t *flag, int n, int32_t *list, int32_t *count, unsigned long long *status,
                                 hipStream_t st)
{
    hipLaunchKernelGGL(cf_feqmod_compact, dim3(1), dim3(1024), 0, st, flag, n, list, count, status);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Linearised delta-f for the flagged cells, in the reference's form (smooth_kernels.cpp:822-884): rare by construction
// (breakdown cells, the |y - eta| < detA window of nearly singular cells), so written for clarity: libm exp/cosh/sinh,
// one thread per (lane, phi, y) looping over the list; the per-cell record arrives through wave-uniform loads.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cf_feqmod_linear(FqLinearArgs a)
{
    const int n = *a.count;
    if (n == 0) return;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)a.Lpad * a.J * a.Kacc;
    if (idx >= total) return;
    const int l = (int)(idx % a.Lpad);
    const int jk = (int)(idx / a.Lpad);
    if (l >= a.Lbins) return;   // unit-strided lanes: the fallback cells go to a bin's first slot only (padding lanes add nothing either way)
    const int j = jk / a.Kacc, k = jk - j * a.Kacc;
    const double mT = a.lane_mT[l], pT = a.lane_pT[l], sign = a.lane_sign[l], mass = a.lane_mass[l];
    const double baryon = a.lane_b ? a.lane_b[l] : 0.0;
    const double mass2 = mass * mass;
    const double px = pT * a.cosphi[j], py = pT * a.sinphi[j];
    const double y = a.dim3 ? a.kgrid[k] : 0.0;
    const int n_eta = a.dim3 ? 1 : a.K;
    double total_sum = 0.0;
    for (int e = 0; e < n; e++) {
        const double *fb = a.FB + (int64_t)a.list[e] * kFbRec;
        const double tau = fb[FB_TAU], tau2 = tau * tau, T = fb[FB_T];
        const double mT_over_tau = mT / tau;
        const bool narrow_only = fb[FB_KIND] == 2.0;
        double sum = 0.0;
        for (int ie = 0; ie < n_eta; ie++) {
            const double eta = a.dim3 ? fb[FB_ETA] : a.kgrid[ie];
            const double eta_weight = a.dim3 ? 1.0 : a.kweight[ie];
            if (narrow_only && !(fabs(y - eta) < fb[FB_DETA])) continue;             // :807-813
            const double pt = mT * cosh(y - eta), pn = mT_over_tau * sinh(y - eta), tau2_pn = tau2 * pn;
            const double pdotdsigma = eta_weight * (pt * fb[FB_DAT] + px * fb[FB_DAX] + py * fb[FB_DAY]) + pn * fb[FB_DAN];   // :828
            if (a.outflow && pdotdsigma <= 0.0) continue;
            const double pdotu = pt * fb[FB_UT] - px * fb[FB_UX] - py * fb[FB_UY] - tau2_pn * fb[FB_UN];
            const double pimunu_pmu_pnu = fb[FB_PITT] * pt * pt + fb[FB_PIXX] * px * px + fb[FB_PIYY] * py * py + fb[FB_PINN] * tau2_pn * tau2_pn
                + 2.0 * (-(fb[FB_PITX] * px + fb[FB_PITY] * py) * pt + fb[FB_PIXY] * px * py + tau2_pn * (fb[FB_PIXN] * px + fb[FB_PIYN] * py - fb[FB_PITN] * pt));
            const double chem = (a.mode == 3) ? baryon * fb[FB_ALPHAB] : 0.0;                                               // :741, :838
            const double feq = 1.0 / (exp(pdotu / T - chem) + sign), feqbar = 1.0 - sign * feq;
            const double df_shear = fb[FB_SHEAR] * pimunu_pmu_pnu / pdotu;
            double df;
            if (a.mode == 3) {                                                                                               // :833-858
                const double Vmu_pmu = fb[FB_VT] * pt - fb[FB_VX] * px - fb[FB_VY] * py - fb[FB_VN] * tau2_pn;
                const double df_bulk = fb[FB_CA] * pdotu + fb[FB_B1] * baryon + fb[FB_CB] * (pdotu - mass2 / pdotu);
                const double df_diff = (fb[FB_BER] - baryon / pdotu) * Vmu_pmu * fb[FB_IBV];
                df = feqbar * (df_shear + df_bulk + df_diff);
            }
            else df = feqbar * df_shear + fb[FB_CA] + feqbar * fb[FB_CB] * (pdotu - mass2 / pdotu);                         // :859-880
            if (a.regulate) df = fmax(-1.0, fmin(df, 1.0));
            sum += pdotdsigma * (feq * (1.0 + df));
        }
        total_sum += sum;
    }
    a.partial[idx] += total_sum;
}

hipError_t launch_feqmod_linear(const FqLinearArgs &a, hipStream_t st)
{
    const int64_t total = (int64_t)a.Lpad * a.J * a.Kacc;
    hipLaunchKernelGGL(cf_feqmod_linear, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace is3d
