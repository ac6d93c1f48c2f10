// cf_vah.hip -- anisotropic-hydro (VAH, P_L matching) smooth Cooper-Frye kernel on the device: BASELINE config 5, SURVEY.md 8f
// rank 4 (second half).  Device path of EmissionFunctionArray::calculate_dN_pTdpTdphidy_VAH_PL
// (/root/reference/src/cpp/emissionfunction_smooth_kernels.cpp:2140-2393), which the reference itself never calls
// (emissionfunction.cpp:1650-1654): restated from the source text, with the per-cell coefficients c0..c4 as inputs, as in the
// method's own signature.
//
//   f_a = 1/(exp(E_a/Lambda) + sign),  E_a^2 = (p.u)^2 + xi_L (p.z)^2,  xi_L = 1/alpha_L^2 - 1
//   df/(f_a fbar_a) = c3 (p.z)(p.W) + c4 pi_perp^{mu nu} p_mu p_nu + (c0 m^2 + c1 (p.z)^2 + c2 (p.u)^2) Pi
// With p.u = mT C_k - pT D_j, p.z = mT Z_k, p.W = mT V1_k - pT V2_j both the exponent and delta-f are quadratic forms in the
// lane constants (mT, pT) with wave-uniform coefficients:
//   (E_a/Lambda)^2 = mT^2 ax_k + mT pT bx_jk + pT^2 gx_j          df/(f_a fbar_a) = mT^2 ad_k + mT pT bd_jk + pT^2 gd_j
// i.e. the exponent of the modified-equilibrium kernel (one sqrt + one exp per evaluation, cf_feqmod.hip) combined with the
// delta-f of the tile kernel (cf_kernels.hip).  Unit record (JT phi's x R rows): header jj {B_j, gx_j, gd_j, 0}; row r
// {A_k, ax_k, ad_k, W_k, bx_{j0..,k}, bd_{j0..,k}}.  No outflow cut and no skipped cells on this path; in 2+1D the eta
// weights are the table weights times the node spacing (:2180-2188).
//
// Round 3.  (i) The 3+1D default is cf_main_vah3 on "F" records: the exponent factored, (E_a/Lambda)^2 = (mT c_k - pT d_j)^2 + mT^2 e_k, on the
// 8 x 7 tile with lower-bound row / unit culls and the shorter exponential (see the comment at the kernel); the kernel above stays for
// 2+1D and as kernel_variant 2.  (ii) The coefficients can come from the reference's tables instead of the cell arrays: cf_vah_coeffs is
// the per-cell bilinear of the CUDA tree's loader (src/cuda/deltafReader.cu:224-278; src/cpp has none).  (iii) The host side is a
// device-resident plan (is3d_vah_plan_*): is3d_smooth_spectra_vah[_df] = create + upload + execute + download.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/is3d_amd.h"
#include "cf_device.h"
#include "cf_launch.h"
#include "cf_math.h"
#include "errors.h"

namespace is3d {

struct VahPtrs {
    const double *tau, *eta, *ux, *uy, *un, *dat, *dax, *day, *dan;
    const double *pitt, *pitx, *pity, *pitn, *pixx, *pixy, *pixn, *piyy, *piyn, *pinn;
    const double *bulkPi, *Wx, *Wy, *Lambda, *aL, *c0, *c1, *c2, *c3, *c4;
};

struct VahPrepParams {
    VahPtrs cells;
    int32_t n_cells, J, K, dim3, include_bulk, include_shear;
    const double *cosphi, *sinphi, *kgrid, *kweight;
    int32_t JT, R, jtiles, rblocks;
    double *TS;
    // "F" records (3+1D, kernel variant 3; layout at cf_main_vah3): the exponent travels factored, (E_a/Lambda)^2 = (mT c_k - pT d_j)^2 + mT^2 e_k
    int32_t fact;
    int64_t cell0;                      // index of the pass's first cell in the caller's arrays (domain report)
    double mTmax, pTmax;                // largest lane mT, pT: bound of E_a/Lambda for the exponential's domain (exp_p9: < 1.4e9)
    unsigned long long *status;         // [7] min cell whose E_a/Lambda can exceed 1e9
};

constexpr int kVahCB = 4;      // cells per workgroup batch of the prep kernel with long rows (2+1D eta table) ...
constexpr int kVahCB3 = 16;    // ... and with K <= 32 (3+1D): the serial per-cell phase runs on CB lanes of the workgroup (cf_prep, cf_prep_feqmod alike)
static inline int vah_batch_cells(int K) { return K > 32 ? kVahCB : kVahCB3; }
constexpr int kVahThreads = 256;

struct VahScal {
    double dat, dax, day, dan_tau, eta, ut, ux, uy, tau_un, zt, tzn;
    double pitt, pitx, pity, tpitn, pixx, pixy, tpixn, piyy, tpiyn, t2pinn;
    double Wt, Wx, Wy, tWn, invL2, xi, Pi, c0, c1, c2, c3, c4, invL;
};

template <int CB>
__global__ void __launch_bounds__(kVahThreads) cf_prep_vah(VahPrepParams p)
{
    extern __shared__ double lds[];
    const int J = p.J, K = p.K;
    VahScal *cs = (VahScal *)lds;                     // [CB]
    double *lk = (double *)(cs + CB);             // [10][CB][K]: A, ax, ad, W, ch, sh, C, Z, V1, e (F records in 2+1D)
    double *lj = lk + 10 * CB * K;                // [7][CB][J]: B, gx, gd, D, V2, E, F
    double *l_dmx = lj + 7 * CB * J;              // [CB][jtiles] F records: max_j d_j of a phi tile
    const int CK = CB * K, CJ = CB * J;
    double *l_A = lk, *l_ax = lk + CK, *l_ad = lk + 2 * CK, *l_W = lk + 3 * CK, *l_ch = lk + 4 * CK, *l_sh = lk + 5 * CK;
    double *l_C = lk + 6 * CK, *l_Z = lk + 7 * CK, *l_V1 = lk + 8 * CK;
    double *l_B = lj, *l_gx = lj + CJ, *l_gd = lj + 2 * CJ, *l_D = lj + 3 * CJ, *l_V2 = lj + 4 * CJ, *l_E = lj + 5 * CJ, *l_F = lj + 6 * CJ;
    // F records: the slots of ax_k and gx_j carry the factored exponent's c_k = C_k / Lambda and d_j = D_j / Lambda; e_k = xi Z_k^2 / Lambda^2 has its
    // own array (2+1D rows keep their eta weight W_k beside it; in 3+1D W_k = 1 is not stored)
    double *l_ck = l_ax, *l_ek = lk + 9 * CK, *l_dj = l_gx;
    const int tid = threadIdx.x;
    const int nbatch = (p.n_cells + CB - 1) / CB;
    for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const int cbase = batch * CB;
        const int ncb = min(CB, p.n_cells - cbase);
        if (tid < ncb) {                                                          // :2208-2256
            const int64_t gi = cbase + tid;
            VahScal s;
            const double tau = p.cells.tau[gi], tau2 = tau * tau;
            const double ux = p.cells.ux[gi], uy = p.cells.uy[gi], un = p.cells.un[gi];
            const double ut = sqrt(1.0 + ux * ux + uy * uy + tau2 * un * un);
            const double u0 = sqrt(1.0 + ux * ux + uy * uy);
            s.zt = tau * un / u0;
            s.tzn = tau * (ut / (u0 * tau));                                        // tau * zn
            s.dat = p.cells.dat[gi]; s.dax = p.cells.dax[gi]; s.day = p.cells.day[gi]; s.dan_tau = p.cells.dan[gi] / tau;
            s.eta = p.dim3 ? p.cells.eta[gi] : 0.0;
            s.ut = ut; s.ux = ux; s.uy = uy; s.tau_un = tau * un;
            s.pitt = p.cells.pitt[gi]; s.pitx = p.cells.pitx[gi]; s.pity = p.cells.pity[gi]; s.tpitn = tau * p.cells.pitn[gi];
            s.pixx = p.cells.pixx[gi]; s.pixy = p.cells.pixy[gi]; s.tpixn = tau * p.cells.pixn[gi];
            s.piyy = p.cells.piyy[gi]; s.tpiyn = tau * p.cells.piyn[gi]; s.t2pinn = tau2 * p.cells.pinn[gi];
            const double Wx = p.cells.Wx[gi], Wy = p.cells.Wy[gi];
            const double Wt = (ux * Wx + uy * Wy) * ut / (u0 * u0);
            s.Wt = Wt; s.Wx = Wx; s.Wy = Wy; s.tWn = tau * (Wt * un / ut);
            const double Lambda = p.cells.Lambda[gi], aL = p.cells.aL[gi];
            s.invL2 = 1.0 / (Lambda * Lambda);
            s.invL = 1.0 / Lambda;
            s.xi = 1.0 / (aL * aL) - 1.0;
            s.Pi = p.include_bulk ? p.cells.bulkPi[gi] : 0.0;
            s.c0 = p.cells.c0[gi]; s.c1 = p.cells.c1[gi]; s.c2 = p.cells.c2[gi];
            s.c3 = p.include_shear ? p.cells.c3[gi] : 0.0;
            s.c4 = p.include_shear ? p.cells.c4[gi] : 0.0;
            cs[tid] = s;
        }
        __syncthreads();
        for (int idx = tid; idx < ncb * K; idx += kVahThreads) {
            const int c = idx / K, k = idx - c * K;
            const VahScal &s = cs[c];
            double dlt, w;
            if (p.dim3) { dlt = p.kgrid[k] - s.eta; w = 1.0; }
            else { dlt = 0.0 - p.kgrid[k]; w = p.kweight[k]; }                      // host passes eta_w * delta_eta
            const double ch = cosh(dlt), sh = sinh(dlt);
            const double C = ch * s.ut - sh * s.tau_un;                             // p.u = mT C - pT D
            const double Z = ch * s.zt - sh * s.tzn;                                // p.z = mT Z
            const double V1 = ch * s.Wt - sh * s.tWn;                               // p.W = mT V1 - pT V2
            const double Q0 = s.pitt * ch * ch + s.t2pinn * sh * sh - 2.0 * s.tpitn * ch * sh;
            l_A[idx] = w * (ch * s.dat + sh * s.dan_tau);
            if (p.fact) {
                l_ck[idx] = C * s.invL;
                l_ek[idx] = s.xi * Z * Z * s.invL2;
                l_W[idx] = w;
            } else {
                l_W[idx] = w;
                l_ax[idx] = (C * C + s.xi * Z * Z) * s.invL2;
            }
            l_ad[idx] = s.c3 * Z * V1 + s.c4 * Q0 + s.Pi * (s.c0 + s.c1 * Z * Z + s.c2 * C * C);
            l_ch[idx] = ch; l_sh[idx] = sh; l_C[idx] = C; l_Z[idx] = Z; l_V1[idx] = V1;
        }
        for (int idx = tid; idx < ncb * J; idx += kVahThreads) {
            const int c = idx / J, j = idx - c * J;
            const VahScal &s = cs[c];
            const double cp = p.cosphi[j], sp = p.sinphi[j];
            const double D = cp * s.ux + sp * s.uy;
            const double Q2 = s.pixx * cp * cp + s.piyy * sp * sp + 2.0 * s.pixy * cp * sp;
            l_B[idx] = cp * s.dax + sp * s.day;
            if (p.fact) l_dj[idx] = D * s.invL;
            else l_gx[idx] = D * D * s.invL2;
            l_gd[idx] = s.c4 * Q2 + s.Pi * (s.c2 * D * D - s.c0);
            l_D[idx] = D;
            l_V2[idx] = cp * s.Wx + sp * s.Wy;
            l_E[idx] = -2.0 * (s.pitx * cp + s.pity * sp);
            l_F[idx] = 2.0 * (s.tpixn * cp + s.tpiyn * sp);
        }
        __syncthreads();
        if (tid < ncb) {
            // domain of the main kernels' exponential (exp_p9: |v| < 1.4e9): E_a/Lambda <= (mTmax max_k sqrt(C_k^2 + |xi| Z_k^2) + pTmax max_j |D_j|) / Lambda
            const VahScal &s = cs[tid];
            double ck = 0.0, dj = 0.0;
            for (int k = 0; k < K; k++) ck = fmax(ck, sqrt(l_C[tid * K + k] * l_C[tid * K + k] + fabs(s.xi) * l_Z[tid * K + k] * l_Z[tid * K + k]));
            for (int j = 0; j < J; j++) dj = fmax(dj, fabs(l_D[tid * J + j]));
            if (!((p.mTmax * ck + p.pTmax * dj) * fabs(s.invL) < 1.0e9)) atomicMin(&p.status[7], (unsigned long long)(p.cell0 + cbase + tid));
        }
        if (p.fact) {
            // ---- F records (cf_main_vah3): header jj {B_j, d_j, gd_j, x}, x of jj = 0, 1, 2 = min_k c_k, min_k e_k over the unit's rows and
            // max_j d_j over its phi tile (the bounds of the unit- and row-level culls); row r {A_k, c_k, ad_k, e_k[, W_k, 0], bd_{j0..,k}}
            // (3+1D: four row scalars; 2+1D: six -- the eta weight travels with the row, rows stay 16-byte multiples).
            // The writer (round 4; it was a flat index decoded with four integer divisions and a dozen branches per element: 22 of the 85 ms of a
            // 2+1D step): WHICH quantity an element of a record is does not depend on the unit, so it is decoded once per workgroup into a
            // descriptor table; the tile / row-block minima are formed once per (cell, tile) / (cell, row block); then one unit per wave at a time,
            // lanes <-> elements, units walked so that a wave's consecutive records are adjacent in the stream.
            const int JT = p.JT, R = p.R;
            const int RS = p.dim3 ? 4 : 6;
            const int HDR = 4 * JT, RWD = RS + JT, REC = HDR + R * RWD;
            double *l_ckmin = l_dmx + CB * p.jtiles;          // [CB][rblocks]
            double *l_ekmin = l_ckmin + CB * p.rblocks;       // [CB][rblocks]
            // descriptor of a record element: x = bits 0-19 offset of its source array in the LDS block | 20-22 index rule (0: + c J + j, 1: + c K + k,
            // 2: + c rblocks + rb, 3: + c jtiles + jt) | 24 beta element | 25 zero in a padding row | 26 always zero;  y = jj | r << 8.
            // Branch-free per element: ONE read of the selected source, the beta form evaluated beside it (six more reads) and selected by a flag -- a
            // switch over the thirteen kinds ran seven exec-masked bodies with an LDS round trip each per trip (13.5 ms per 1e5 cells in 2+1D).
            int2 *desc = (int2 *)(l_ekmin + CB * p.rblocks);  // [REC]
            constexpr int BETA = 1 << 24, PADZ = 1 << 25, ZERO = 1 << 26;
            if (batch == (int)blockIdx.x) {                   // first batch of this workgroup: the table
                const int o_B = (int)(l_B - lds), o_dj = (int)(l_dj - lds), o_gd = (int)(l_gd - lds), o_ckm = (int)(l_ckmin - lds), o_ekm = (int)(l_ekmin - lds);
                const int o_dmx = (int)(l_dmx - lds), o_A = (int)(l_A - lds), o_ck = (int)(l_ck - lds), o_ad = (int)(l_ad - lds), o_ek = (int)(l_ek - lds), o_W = (int)(l_W - lds);
                for (int e = tid; e < REC; e += kVahThreads) {
                    int x, y;
                    if (e < HDR) {
                        const int jj = e >> 2, f = e & 3;
                        y = jj;
                        x = f == 0 ? o_B : f == 1 ? o_dj : f == 2 ? o_gd : jj == 0 ? (o_ckm | (2 << 20)) : jj == 1 ? (o_ekm | (2 << 20)) : jj == 2 ? (o_dmx | (3 << 20)) : ZERO;
                    } else {
                        const int q = e - HDR, r = q / RWD, f = q - r * RWD;
                        y = r << 8;
                        if (f == 0) x = o_A | (1 << 20) | PADZ;
                        else if (f == 1) x = o_ck | (1 << 20);
                        else if (f == 2) x = o_ad | (1 << 20);
                        else if (f == 3) x = o_ek | (1 << 20);
                        else if (f < RS) x = f == 4 ? (o_W | (1 << 20) | PADZ) : ZERO;
                        else { x = BETA; y |= f - RS; }
                    }
                    desc[e] = int2{x, y};
                }
            }
            for (int idx = tid; idx < ncb * p.jtiles; idx += kVahThreads) {
                const int c = idx / p.jtiles, jt = idx - c * p.jtiles;
                double v = -1.0e300;
                for (int q2 = 0; q2 < JT; q2++) v = fmax(v, l_dj[c * J + min(jt * JT + q2, J - 1)]);
                l_dmx[c * p.jtiles + jt] = v;
            }
            for (int idx = tid; idx < ncb * p.rblocks; idx += kVahThreads) {
                const int c = idx / p.rblocks, rb = idx - c * p.rblocks;
                double a = 1.0e300, b2 = 1.0e300;
                for (int r = 0; r < R; r++) {
                    const int kc = min(rb * R + r, K - 1);
                    a = fmin(a, l_ck[c * K + kc]);
                    b2 = fmin(b2, l_ek[c * K + kc]);
                }
                l_ckmin[c * p.rblocks + rb] = a;
                l_ekmin[c * p.rblocks + rb] = b2;
            }
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63, nwaves = kVahThreads / 64;
            const int n_units = ncb * p.jtiles * p.rblocks;
            const int NRAW = HDR + RS * R, NBETA = R * JT;
            const int invRS = (1 << 20) / RS + 1, invJT = (1 << 20) / JT + 1;   // (t * inv) >> 20 == t / RS resp. t / JT for t < 4096
            for (int u = wave; u < n_units; u += nwaves) {
                int c, jt, rb;
                if (p.dim3) { c = u % ncb; const int t = u / ncb; rb = t % p.rblocks; jt = t / p.rblocks; }      // cells innermost: adjacent records of one stream
                else { rb = u % p.rblocks; const int t = u / p.rblocks; c = t % ncb; jt = t / ncb; }             // row blocks innermost
                const VahScal &s = cs[c];
                const int64_t cell = cbase + c;
                const int64_t unit = p.dim3 ? (int64_t)(jt * p.rblocks + rb) * p.n_cells + cell : ((int64_t)jt * p.n_cells + cell) * p.rblocks + rb;
                double *o = p.TS + unit * REC;
                const int cJ = c * J, cK = c * K;
                const double c4 = s.c4, c3 = s.c3, pc2 = 2.0 * s.Pi * s.c2;
                // two kinds of trip (as in cf_prep_feqmod): raw trips, lanes <-> the header and the RS scalars of every row (one descriptor read, one
                // source read); beta trips, lanes <-> (row, phi column): no descriptor, nine reads and the bilinear form.  One loop over all
                // elements evaluated the form on every lane.
                for (int t = lane; t < NRAW; t += 64) {
                    int e = t;
                    if (t >= HDR) { const int q = t - HDR, r = (q * invRS) >> 20; e = HDR + r * RWD + (q - r * RS); }
                    const int2 d = desc[e];
                    const int jj = d.y & 0xff, r = d.y >> 8;
                    const int j = cJ + min(jt * JT + jj, J - 1);
                    const int k = rb * R + r, kc = cK + min(k, K - 1);
                    const int rule = (d.x >> 20) & 7;
                    const int add = rule == 0 ? j : rule == 1 ? kc : rule == 2 ? c * p.rblocks + rb : c * p.jtiles + jt;
                    const bool zero = ((d.x >> 26) & 1) | (((d.x >> 25) & 1) & (k >= K));   // padding rows: the forms of row K-1 with p.dsigma = 0
                    const double raw = lds[zero ? 0 : (d.x & 0xfffff) + add];
                    o[e] = zero ? 0.0 : raw;
                }
                for (int t = lane; t < NBETA; t += 64) {
                    const int r = (t * invJT) >> 20, jj = t - r * JT;
                    const int j = cJ + min(jt * JT + jj, J - 1), kc = cK + min(rb * R + r, K - 1);
                    const double X = l_E[j] * l_ch[kc] + l_F[j] * l_sh[kc];
                    o[HDR + r * RWD + RS + jj] = c4 * X - c3 * l_Z[kc] * l_V2[j] - pc2 * l_C[kc] * l_D[j];
                }
            }
        } else {
            const int JT = p.JT, R = p.R;
            const int HDR = 4 * JT, RWD = 4 + 2 * JT, REC = HDR + R * RWD;
            const int units_per_cell = p.jtiles * p.rblocks, per_cell = units_per_cell * REC;
            for (int idx = tid; idx < ncb * per_cell; idx += kVahThreads) {
                const int c = idx / per_cell;
                const int rem = idx - c * per_cell;
                const int ut = rem / REC, e = rem - ut * REC;
                const int jt = ut / p.rblocks, rb = ut - jt * p.rblocks;
                const VahScal &s = cs[c];
                double v = 0.0;
                if (e < HDR) {
                    const int jj = e >> 2, f = e & 3;
                    const int j = min(jt * JT + jj, J - 1);
                    if (f == 0) v = l_B[c * J + j];
                    else if (f == 1) v = l_gx[c * J + j];
                    else if (f == 2) v = l_gd[c * J + j];
                } else {
                    const int q = e - HDR, r = q / RWD, f = q - r * RWD;
                    const int k = rb * R + r, kc = min(k, K - 1);     // padding rows: the forms of row K-1 with p.dsigma = 0
                    if (f == 0) v = (k < K) ? l_A[c * K + kc] : 0.0;
                    else if (f == 1) v = l_ax[c * K + kc];
                    else if (f == 2) v = l_ad[c * K + kc];
                    else if (f == 3) v = (k < K) ? l_W[c * K + kc] : 0.0;
                    else if (f < 4 + JT) {
                        const int j = min(jt * JT + (f - 4), J - 1);
                        v = -2.0 * l_C[c * K + kc] * l_D[c * J + j] * s.invL2;
                    } else {
                        const int j = min(jt * JT + (f - 4 - JT), J - 1);
                        const double X = l_E[c * J + j] * l_ch[c * K + kc] + l_F[c * J + j] * l_sh[c * K + kc];
                        v = s.c4 * X - s.c3 * l_Z[c * K + kc] * l_V2[c * J + j] - 2.0 * s.Pi * s.c2 * l_C[c * K + kc] * l_D[c * J + j];
                    }
                }
                const int64_t cell = cbase + c;
                int64_t unit;
                if (p.dim3) unit = (int64_t)ut * p.n_cells + cell;
                else unit = ((int64_t)jt * p.n_cells + cell) * p.rblocks + rb;
                p.TS[unit * REC + e] = v;
            }
        }
        __syncthreads();
    }
}

// Same task decomposition, LDS staging and partial layout as cf_main_tile / cf_main_feqmod.
template <bool DIM3, bool REG, int JT, int R>
__global__ void __launch_bounds__(512)
cf_main_vah(const double *__restrict__ TS, const double *__restrict__ lane_mT, const double *__restrict__ lane_pT,
            const double *__restrict__ lane_sign, double *__restrict__ partial, unsigned long long *__restrict__ stats, MainGeom g)
{
    constexpr int HDR = 4 * JT;
    constexpr int RW = 4 + 2 * JT;
    constexpr int REC = HDR + R * RW;
    constexpr int UB = (1536 / REC) > 0 ? (1536 / REC) : 1;
    constexpr int BUF2 = UB * REC / 2;
    constexpr int NLD = (BUF2 + 127) / 128;
    constexpr int RB = DIM3 ? (JT % 4 == 0 ? 4 : (JT % 3 == 0 ? 3 : 2)) : (JT % 8 == 0 ? 8 : 4);   // as in cf_main_tile
    static_assert(REC % 2 == 0 && JT % RB == 0, "unit records must be 16-byte multiples");
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
    const double hs = REG ? 0.5 : 1.0;   // u = (1 + fbar df) * hs, clamped to [0, 1] by the VOP3 clamp modifier when REG
    const double mT2s = hs * mT2, mTpTs = hs * mTpT, pT2s = hs * pT2;
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
    int n_rows = 0, n_dead = 0;

    auto process_unit = [&](const double *U) {
        double pTB[JT], gx[JT], gd[JT];
#pragma unroll
        for (int jj = 0; jj < JT; jj++) {
            pTB[jj] = pT * U[4 * jj + 0];
            gx[jj] = pT2 * U[4 * jj + 1];
            gd[jj] = pT2s * U[4 * jj + 2];
        }
        struct Row { double v[RW]; };
        auto fetch = [&](Row &rw, const double *row) {
#pragma unroll
            for (int i = 0; i < RW; i++) rw.v[i] = row[i];
        };
        auto evals = [&](const Row &rw, int r) {
            const double mTA = mT * rw.v[0], ax = mT2 * rw.v[1], ad = mT2s * rw.v[2], W = rw.v[3];
            double X2[JT];
            double x2min = 1.0e300;
#pragma unroll
            for (int jj = 0; jj < JT; jj++) {
                X2[jj] = __builtin_fma(mTpT, rw.v[4 + jj], ax + gx[jj]);
                x2min = __builtin_fmin(x2min, X2[jj]);
            }
            n_rows += 1;
            if (g.zskip && __all(x2min > 555400.0)) { n_dead += 1; return; }   // E_a/Lambda > 745.25: f_a == +0 for the whole wave-row
#pragma unroll
            for (int j0 = 0; j0 < JT; j0 += RB) {   // one v_rcp_f64 per RB evaluations (rcp_batch, cf_math.h)
                double zz[RB], d[RB], inv[RB];
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const double X = sqrt_g1(X2[j0 + i]);
                    zz[i] = DIM3 ? exp_full_sat(-X) : exp_p9(-X);   // 3+1D here is the round-1 kernel kept AS IT WAS for A/B; 2+1D takes the shorter exponential (cf_math.h); cf_prep_vah keeps X below 1e9
                    d[i] = __builtin_fma(sign, zz[i], 1.0);
                }
                rcp_batch<RB>(d, inv);                                                         // fbar_a
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int jj = j0 + i;
                    const double rr = inv[i];
                    const double br = __builtin_fma(mTpTs, rw.v[4 + JT + jj], ad + gd[jj]);   // hs * df/(f_a fbar_a)
                    const double u = REG ? fma_clamp01_half(rr, br) : __builtin_fma(rr, br, 1.0);
                    const double pds = __builtin_fma(pTB[jj], W, mTA);
                    const double w = (zz[i] * rr) * u;
                    if (DIM3) acc[jj * R + r] = __builtin_fma(pds, w, acc[jj * R + r]);
                    else acc[jj] = __builtin_fma(pds, w, acc[jj]);
                }
            }
        };
        const double *rows = U + HDR;
        Row cur, nxt;
        fetch(cur, rows);
        if (DIM3) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r + 1 < R) fetch(nxt, rows + (r + 1) * RW);
                evals(cur, r);
                if (r + 1 < R) cur = nxt;
            }
        } else {
            // rolled, two rows per trip (see cf_main_tile)
#pragma clang loop unroll(disable)
            for (int r = 0; r + 1 < R; r += 2) {
                fetch(nxt, rows + (r + 1) * RW);
                evals(cur, 0);
                fetch(cur, rows + (r + 2) * RW);
                evals(nxt, 0);
            }
            if (R & 1) evals(cur, 0);
        }
    };

    // staging: the next batch by direct-to-LDS loads (stage_pieces, cf_math.h), issued before the current batch is consumed
    auto stage = [&](int ib, int buf) { stage_pieces<BUFP / 64>((const char *)(src + (int64_t)ib * BUF2), lbuf[buf], tid, nthr); };
    // The lane constants must have ARRIVED before the batch loop: the compiler sinks the loads of restrict-qualified data to their first
    // use and waits for them there (s_waitcnt vmcnt(0) in every row of the unrolled loop) -- a wait that also covers the direct-to-LDS
    // loads of the next batch, which it does not know about, i.e. it would expose the staging latency in every batch
    asm volatile("" :: "v"(mT), "v"(pT), "v"(sign), "v"(mT2s), "v"(mTpTs), "v"(pT2s), "v"(mT2), "v"(mTpT), "v"(pT2) : "memory");
    if (nb > 0) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ib = 0; ib < nb; ib++) {
            if (ib + 1 < nb) stage(ib + 1, (ib + 1) & 1);
            if (wave_active) {
                const int nu = min(UB, n_units - ib * UB);
                const double *base = (const double *)lbuf[ib & 1];
                for (int u = 0; u < nu; u++) process_unit(base + u * REC);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (!wave_active) return;
    if ((tid & 63) == 0) {
        atomicAdd(&stats[2], (unsigned long long)n_rows);
        atomicAdd(&stats[3], (unsigned long long)n_dead);
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
                    if (k < K) pp[((int64_t)j * g.Kacc + k) * g.Lpad + l] = unscale * acc[jj * R + r];
                }
            } else {
                pp[(int64_t)j * g.Lpad + l] = unscale * acc[jj];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// cf_main_vah3 (3+1D, kernel variant 3, the default): the round-1 kernel above brought up to the toolbox of cf_main_tile3e /
// cf_main_feqmod.
//   * "F" unit records, REC = 4 JT + R (4 + JT) doubles (116 for the 8 x 7 tile, as the delta-f tile's): the exponent travels FACTORED,
//       (E_a/Lambda)^2 = (p.u/Lambda)^2 + xi (p.z/Lambda)^2 = (mT c_k - pT d_j)^2 + mT^2 e_k,
//     so a row carries {A_k, c_k, ad_k, e_k, bd_jk...} instead of {A_k, ax_k, ad_k, W_k, bx_jk..., bd_jk...}: 12 instead of 20 doubles through
//     LDS and registers per row (W_k = 1 in 3+1D), which is what lets the 8 x 7 tile fit 256 VGPRs with four evaluations per v_rcp_f64
//     (the 6 x 7 tile shared one among three).  p.u = mT C_k - pT D_j is formed as the difference it is in the reference (:2290) rather
//     than through the expanded quadratic form: the same instruction count (one add, one fma) and no cancellation between squares.
//   * a row is tested BEFORE its evaluations from a lower bound: p.u/Lambda >= mT c_k - pT max_j d_j >= 0 (u timelike, pT >= 0, monotone
//     roundings), X^2 >= lb^2 + mT^2 e_k: three instructions per row instead of a minimum per evaluation, and a dead row forms no X^2;
//     a unit whose bound (min_k c_k, min_k e_k in free header slots) is dead skips its header and its R row fetches.  The rule is the
//     exact-zero rule of the round-1 kernel (E_a/Lambda > 745.25: f_a == +0): bitwise the same spectrum with zero_skip on or off.
//   * exp_p9 (cf_math.h): one fma for the range reduction, degree 9 -- two instructions fewer per evaluation, 7e-14 relative.
//     cf_prep_vah refuses cells whose E_a/Lambda could exceed 1e9 (status[7]) so that the shift-trick conversion is in its domain.
//   * 2+1D (round 4; DIM3 = false): the same factored records with the eta weight kept in the row -- {A_k, c_k, ad_k, e_k, W_k, 0, bd_jk...}, 14
//     doubles per row of the 8-wide tile against 20 for the round-1 layout -- on the 8 x 31 tile of config 2 with unit-strided lanes (g.split = S
//     lane slots per momentum bin, slot s takes the units u = s (mod S) of the stream; cf_finalize adds a bin's slots in slot order): pi / K / p
//     are 96 bins = 1.5 waves, 384 slots are six full ones.  Eight evaluations share a v_rcp_f64 (the 8 accumulators of a 2+1D lane leave
//     the registers for it).  The row and unit tests are the same lower bounds; with strided lanes a vote spans lanes on different units.
// ------------------------------------------------------------------------------------------------
template <bool DIM3, bool REG, int JT, int R, int LDSD = 1536>
__global__ void __launch_bounds__(512)
cf_main_vah3(const double *__restrict__ TS, const double *__restrict__ lane_mT, const double *__restrict__ lane_pT,
             const double *__restrict__ lane_sign, double *__restrict__ partial, unsigned long long *__restrict__ stats, MainGeom g,
             const int32_t *__restrict__ lane_sub)
{
    constexpr int HDR = 4 * JT;
    constexpr int RS = DIM3 ? 4 : 6;
    constexpr int RW = RS + JT;
    constexpr int REC = HDR + R * RW;
    // 3+1D: LDSD doubles of records per LDS buffer -- 1536 (13 units) for two-wave workgroups, 1100 (9 units, 18 KB per workgroup) for the one-wave
    // workgroups of g.wpb == 1 (no barrier partner; cf_main_feqmod has the same); 2+1D: four units per buffer (S in {1, 2, 4} divides it)
    constexpr int UB = DIM3 ? ((LDSD / REC) > 0 ? (LDSD / REC) : 1) : 4;
    constexpr int BUF2 = UB * REC / 2;
    constexpr int RB = DIM3 ? (JT % 4 == 0 ? 4 : (JT % 3 == 0 ? 3 : 2)) : (JT % 8 == 0 ? 8 : 4);
    static_assert(REC % 2 == 0 && JT % RB == 0 && JT >= 3, "unit records must be 16-byte multiples; the cull bounds sit in header slots jj = 0, 1, 2");
    constexpr int BUFP = ((BUF2 * 16 + 1023) / 1024) * 64;   // the batch as whole 1-KiB staging pieces (64 double2 each)
    __shared__ double2 lbuf[2][BUFP + RW / 2 + 1];

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
    const double mT2 = mT * mT;
    const double hs = REG ? 0.5 : 1.0;   // u = (1 + fbar df) * hs, clamped to [0, 1] by the VOP3 clamp modifier when REG
    const double mT2s = hs * mT2, mTpTs = hs * (mT * pT), pT2s = hs * (pT * pT);
    const int S = (!DIM3 && g.split > 1) ? g.split : 1;
    const int sub_off = (!DIM3 && g.split > 1 && lane_sub) ? lane_sub[l] * REC : 0;
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
    int n_rows = 0, n_dead = 0;
    constexpr double X2CUT = 555400.0;   // E_a/Lambda > 745.25: exp(-E_a/Lambda) == +0

    auto process_unit = [&](const double *U) {
        // max_j d_j is wave-uniform and lives through the unit's rows (every row test uses it): kept in an SGPR pair and fed to the row test's
        // fma as its one scalar operand -- as the product pT * dmax it held a VGPR pair through the unit, and the 3+1D 8 x 7 tile then kept its
        // 56th accumulator in a scratch slot (scratch_load / v_fmac / scratch_store in every unit: tools/count_isa.py `scratch_in_loop`)
        double dmax_s;
        {
            const double dm = U[11];
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(dm)), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(dm));
            dmax_s = __hiloint2double((int)hi, (int)lo);
        }
        if (g.zskip) {   // unit-level cull: every row of the unit would fail its own test below
            const double lbu = __builtin_fmax(__builtin_fma(mT, U[3], -(pT * dmax_s)), 0.0);
            const double x2lb_u = __builtin_fma(lbu, lbu, mT2 * U[7]);
            if (__all(x2lb_u > X2CUT)) { n_rows += R; n_dead += R; return; }
        }
        double pTB[JT], pTd[JT], gd[JT];
#pragma unroll
        for (int jj = 0; jj < JT; jj++) {
            pTB[jj] = pT * U[4 * jj + 0];
            pTd[jj] = pT * U[4 * jj + 1];
            gd[jj] = pT2s * U[4 * jj + 2];
        }
        struct Row { double v[RW]; };
        auto fetch = [&](Row &rw, const double *row) {
#pragma unroll
            for (int i = 0; i < RW; i++) rw.v[i] = row[i];
        };
        auto evals = [&](const Row &rw, int r) {
            const double mTc = mT * rw.v[1], mT2e = mT2 * rw.v[3];
            n_rows += 1;
            const double lb = __builtin_fmax(__builtin_fma(-pT, dmax_s, mTc), 0.0);   // one rounding less than mTc - pT dmax: the bound moves by an ulp, the cut has 1.6e-4 of slack (745.25 against 745.14)
            const double x2lb = __builtin_fma(lb, lb, mT2e);
            if (g.zskip && __all(x2lb > X2CUT)) { n_dead += 1; return; }
            const double mTA = mT * rw.v[0], ad = mT2s * rw.v[2];
#pragma unroll
            for (int j0 = 0; j0 < JT; j0 += RB) {   // one v_rcp_f64 per RB evaluations (rcp_batch, cf_math.h)
                double zz[RB], d[RB], inv[RB];
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const double t = mTc - pTd[j0 + i];                      // p.u / Lambda
                    const double X = sqrt_g1(__builtin_fma(t, t, mT2e));     // E_a / Lambda
                    zz[i] = exp_p9(-X);
                    d[i] = __builtin_fma(sign, zz[i], 1.0);
                }
                rcp_batch<RB>(d, inv);                                                         // fbar_a
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int jj = j0 + i;
                    const double rr = inv[i];
                    const double br = __builtin_fma(mTpTs, rw.v[RS + jj], ad + gd[jj]);        // hs * df/(f_a fbar_a)
                    const double u = REG ? fma_clamp01_half(rr, br) : __builtin_fma(rr, br, 1.0);
                    const double pds = DIM3 ? pTB[jj] + mTA : __builtin_fma(pTB[jj], rw.v[4], mTA);   // W_k = 1 in 3+1D
                    const double w = (zz[i] * rr) * u;
                    if (DIM3) acc[jj * R + r] = __builtin_fma(pds, w, acc[jj * R + r]);
                    else acc[jj] = __builtin_fma(pds, w, acc[jj]);
                }
            }
        };
        const double *rows = U + HDR;
        Row cur, nxt;
        fetch(cur, rows);
        if (DIM3) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r + 1 < R) fetch(nxt, rows + (r + 1) * RW);
                evals(cur, r);
                if (r + 1 < R) cur = nxt;
            }
        } else {
            // rolled, two rows per trip (see cf_main_tile); for even R the last fetch reads the head of the next unit or the pad
#pragma clang loop unroll(disable)
            for (int r = 0; r + 1 < R; r += 2) {
                fetch(nxt, rows + (r + 1) * RW);
                evals(cur, 0);
                fetch(cur, rows + (r + 2) * RW);
                evals(nxt, 0);
            }
            if (R & 1) evals(cur, 0);
        }
    };

    // staging: the next batch by direct-to-LDS loads (stage_pieces, cf_math.h), issued before the current batch is consumed
    auto stage = [&](int ib, int buf) { stage_pieces<BUFP / 64>((const char *)(src + (int64_t)ib * BUF2), lbuf[buf], tid, nthr); };
    // the lane constants must have ARRIVED before the batch loop (see cf_main_vah)
    asm volatile("" :: "v"(mT), "v"(pT), "v"(sign), "v"(mT2s), "v"(mTpTs), "v"(pT2s), "v"(mT2), "v"(sub_off) : "memory");
    if (nb > 0) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ib = 0; ib < nb; ib++) {
            if (ib + 1 < nb) stage(ib + 1, (ib + 1) & 1);
            if (wave_active) {
                const int nu = min(UB, n_units - ib * UB);
                const double *base = (const double *)lbuf[ib & 1] + sub_off;
                for (int u = 0; u < nu; u += S) process_unit(base + u * REC);   // nu is a multiple of S (the plan: S divides the units per cell and UB)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (!wave_active) return;
    if ((tid & 63) == 0) {
        atomicAdd(&stats[2], (unsigned long long)n_rows);
        atomicAdd(&stats[3], (unsigned long long)n_dead);
    }
    const double unscale = REG ? 2.0 : 1.0;
    double *pp = partial + (int64_t)chunk * J * g.Kacc * g.Lpad;
#pragma unroll
    for (int jj = 0; jj < JT; jj++) {
        const int j = jt * JT + jj;
        if (j < J) {
            if (DIM3) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int k = kt * R + r;
                    if (k < K) pp[((int64_t)j * g.Kacc + k) * g.Lpad + l] = unscale * acc[jj * R + r];
                }
            } else {
                pp[(int64_t)j * g.Lpad + l] = unscale * acc[jj];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Per-cell 14-moment coefficients from the (Lambda, alpha_L) tables: src/cuda/deltafReader.cu:224-278.  The reference scans
// i2 (alpha_L) outer, i1 (Lambda) inner for the first pair with i1 > 0 && Lambda < L[i1] && i2 > 0 && aL < aL[i2]: the two conditions
// are independent, so that is the first i2 >= 1 with aL < aL[i2] and the first i1 >= 1 with Lambda < L[i1] (ascending nodes: an
// upper bound, at least 1 -- below the first node the bilinear form extrapolates, as in the reference).  No such node: the reference
// leaves the cell's c0..c4 unset; here the cell is reported (status[0] = lowest index) and gets zeros.  The arithmetic is the
// reference's expression with every rounding written out (no contraction), so that the values agree with a host evaluation bit for bit.
// ------------------------------------------------------------------------------------------------
struct VahCoefArgs {
    int64_t n, cell0;                   // cells in this launch; index of the first one in the caller's arrays (for the report)
    const double *Lambda, *aL;          // per cell: GeV, 1
    int32_t nL, naL;
    const double *L, *aLg, *tab[5];     // nodes (fm^-1, 1) and tables [naL][nL]
    double *out[5];
    unsigned long long *status;         // [0] min bad cell
};

__device__ __forceinline__ int first_node_above(const double *g, int n, double x)   // first i >= 1 with x < g[i], or n
{
    int lo = 1, hi = n;                 // invariant: every i in [1, lo) has !(x < g[i])
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x < g[mid]) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

__global__ void __launch_bounds__(256) cf_vah_coeffs(VahCoefArgs a)
{
    // every product and sum below is its own rounding, as in the reference's host code: the pragma applies to the operators written
    // in THIS block (the __dmul_rn / __dadd_rn wrappers are inlined plain operators and would still be fused into FMAs)
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double aL = a.aL[i];
    const double Lam = a.Lambda[i] / kHbarC;                              // :228
    const int i1 = first_node_above(a.L, a.nL, Lam), i2 = first_node_above(a.aLg, a.naL, aL);
    if (i1 >= a.nL || i2 >= a.naL) {                                      // includes NaN: every comparison fails
#pragma unroll
        for (int k = 0; k < 5; k++) a.out[k][i] = 0.0;
        atomicMin(a.status, (unsigned long long)(a.cell0 + i));
        return;
    }
    const double L1 = a.L[i1 - 1], L2 = a.L[i1], A1 = a.aLg[i2 - 1], A2 = a.aLg[i2];
    const double dL2 = L2 - Lam, dL1 = Lam - L1, dA2 = A2 - aL, dA1 = aL - A1;
    const double den = (A2 - A1) * (L2 - L1);
    const double hbarC3 = (kHbarC * kHbarC) * kHbarC;                     // :219
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const double *t = a.tab[k];
        const double c00 = t[(i2 - 1) * a.nL + i1 - 1], c10 = t[(i2 - 1) * a.nL + i1], c01 = t[i2 * a.nL + i1 - 1], c11 = t[i2 * a.nL + i1];
        const double p00 = c00 * dL2, p10 = c10 * dL1, p01 = c01 * dL2, p11 = c11 * dL1;
        const double lo = p00 + p10;                                      // (c[i1-1][i2-1] (L2 - L) + c[i1][i2-1] (L - L1))
        const double hi = p01 + p11;
        const double q0 = lo * dA2, q1 = hi * dA1;
        const double v = (q0 + q1) / den;
        a.out[k][i] = v / hbarC3;                                         // :262-266
    }
}

}  // namespace is3d

namespace {

#define VAH_TRY(expr)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return is3d::set_error(IS3D_ENODEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevMem {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t alloc(size_t b)
    {
        release();
        bytes = b;
        if (!b) return hipSuccess;
        is3d::count_resource(1);
        return hipMalloc(&p, b);
    }
    template <class T>
    hipError_t upload(const std::vector<T> &h)
    {
        hipError_t e = alloc(h.size() * sizeof(T));
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t upload(const double *h, size_t n)
    {
        hipError_t e = alloc(n * sizeof(double));
        if (e != hipSuccess || !n) return e;
        return hipMemcpy(p, h, n * sizeof(double), hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    ~DevMem() { release(); }
    template <class T> T *as() const { return (T *)p; }
};

constexpr int kJT3 = 6, kR3 = 7, kJT2 = 8, kR2 = 61;   // the round-1 kernel's tile shapes (kernel variant 2; 2+1D always)
constexpr int kJT3F = 8, kR3F = 7;                       // cf_main_vah3 (3+1D default, kernel variant 3)
constexpr int kJT2F = 8, kR2F = 31;                      // cf_main_vah3 in 2+1D (default since round 4): config 2's tile, four units per LDS buffer

template <bool DIM3>
void launch_vah3(bool reg, const double *TS, const double *mT, const double *pT, const double *sg, double *partial, unsigned long long *stats,
                 const is3d::MainGeom &g, const int32_t *lane_sub, hipStream_t st)
{
    const int grid = ((g.NT + 7) / 8) * 8 * g.G;
    constexpr int JT = DIM3 ? kJT3F : kJT2F, R = DIM3 ? kR3F : kR2F;
    if constexpr (DIM3) {
        if (g.wpb == 1) {   // one-wave workgroups: 9-unit LDS batches, eight workgroups per CU
            if (reg) hipLaunchKernelGGL((is3d::cf_main_vah3<DIM3, true, JT, R, 1100>), dim3(grid), dim3(64), 0, st, TS, mT, pT, sg, partial, stats, g, lane_sub);
            else hipLaunchKernelGGL((is3d::cf_main_vah3<DIM3, false, JT, R, 1100>), dim3(grid), dim3(64), 0, st, TS, mT, pT, sg, partial, stats, g, lane_sub);
            return;
        }
    }
    if (reg) hipLaunchKernelGGL((is3d::cf_main_vah3<DIM3, true, JT, R>), dim3(grid), dim3(g.wpb * 64), 0, st, TS, mT, pT, sg, partial, stats, g, lane_sub);
    else hipLaunchKernelGGL((is3d::cf_main_vah3<DIM3, false, JT, R>), dim3(grid), dim3(g.wpb * 64), 0, st, TS, mT, pT, sg, partial, stats, g, lane_sub);
}

template <bool DIM3>
void launch_vah(bool reg, const double *TS, const double *mT, const double *pT, const double *sg, double *partial, unsigned long long *stats,
                const is3d::MainGeom &g, hipStream_t st)
{
    const int grid = ((g.NT + 7) / 8) * 8 * g.G;
    constexpr int JT = DIM3 ? kJT3 : kJT2, R = DIM3 ? kR3 : kR2;
    if (reg) hipLaunchKernelGGL((is3d::cf_main_vah<DIM3, true, JT, R>), dim3(grid), dim3(g.wpb * 64), 0, st, TS, mT, pT, sg, partial, stats, g);
    else hipLaunchKernelGGL((is3d::cf_main_vah<DIM3, false, JT, R>), dim3(grid), dim3(g.wpb * 64), 0, st, TS, mT, pT, sg, partial, stats, g);
}

int check_tables(const is3d_vah_df_tables *t)
{
    using is3d::set_error;
    if (t->n_L < 2 || t->n_aL < 2 || !t->L || !t->aL || !t->c0 || !t->c1 || !t->c2 || !t->c3 || !t->c4)
        return set_error(IS3D_EINVAL, "VAH coefficient tables need >= 2 x 2 nodes and all five tables");
    for (int i = 1; i < t->n_L; i++)
        if (!(t->L[i] > t->L[i - 1])) return set_error(IS3D_EINVAL, "VAH coefficient tables: Lambda nodes must ascend");
    for (int i = 1; i < t->n_aL; i++)
        if (!(t->aL[i] > t->aL[i - 1])) return set_error(IS3D_EINVAL, "VAH coefficient tables: alpha_L nodes must ascend");
    return IS3D_OK;
}

// device copy of the coefficient tables
struct TabDev {
    DevMem L, aL, c;
    int nL = 0, naL = 0;
    int upload(const is3d_vah_df_tables *t)
    {
        nL = t->n_L; naL = t->n_aL;
        const size_t n = (size_t)nL * naL;
        std::vector<double> all(5 * n);
        const double *src[5] = {t->c0, t->c1, t->c2, t->c3, t->c4};
        for (int k = 0; k < 5; k++) memcpy(all.data() + k * n, src[k], n * sizeof(double));
        VAH_TRY(L.upload(t->L, (size_t)nL));
        VAH_TRY(aL.upload(t->aL, (size_t)naL));
        VAH_TRY(c.upload(all));
        return IS3D_OK;
    }
    void fill(is3d::VahCoefArgs &a) const
    {
        a.nL = nL; a.naL = naL; a.L = L.as<double>(); a.aLg = aL.as<double>();
        for (int k = 0; k < 5; k++) a.tab[k] = c.as<double>() + (size_t)k * nL * naL;
    }
};

}  // namespace

struct is3d_vah_plan {
    is3d_options o{};
    int device = 0;
    bool three_d = true, tables = false, timing = false;
    bool fact = false;            // F records + cf_main_vah3 (3+1D, kernel variant 0 / 3)
    double mTmax = 0.0, pTmax = 0.0;
    int npart = 0, npT = 0, J = 0, K = 0, Kacc = 1, ncls = 0, Lpad = 0;
    int Lbins = 0, split = 1;      // momentum bins (classes x pT); lane slots per bin (unit-strided lanes, 2+1D F records)
    int JT = 0, R = 0, jtiles = 0, rblocks = 0, ktiles = 0, upc = 0, REC = 0, wpb = 4, nch = 1;
    int64_t nout = 0, max_cells = 0, pass_cells = 0;
    size_t lds_prep = 0;
    DevMem d_mT, d_pT, d_sg, d_lane, d_deg, d_cos, d_sin, d_kg, d_kw, d_TS, d_partial, d_coef, d_status, d_lane_sub;
    TabDev tab;
    std::vector<hipEvent_t> ev;   // [pass][0..2]: start, after coefficients + prep, after main; last: after finalize
    int last_passes = 0;

    ~is3d_vah_plan()
    {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

extern "C" int is3d_vah_plan_create(is3d_vah_plan **out, const is3d_species *sp, const is3d_grid *gr, const is3d_vah_df_tables *tab,
                                    const is3d_options *o, int64_t max_cells)
{
    using is3d::set_error;
    if (!out) return set_error(IS3D_EINVAL, "null plan pointer");
    *out = nullptr;
    if (!sp || !gr || !o) return set_error(IS3D_EINVAL, "null argument");
    if (o->dimension != 2 && o->dimension != 3) return set_error(IS3D_EINVAL, "dimension must be 2 or 3 (got %d)", o->dimension);
    if (sp->n < 1 || !sp->mass || !sp->sign || !sp->degeneracy) return set_error(IS3D_EINVAL, "empty species list");
    if (gr->n_pT < 1 || gr->n_phi < 1 || !gr->pT || !gr->phi) return set_error(IS3D_EINVAL, "empty pT/phi grid");
    const bool three_d = o->dimension == 3;
    if (three_d && (gr->n_y < 1 || !gr->y)) return set_error(IS3D_EINVAL, "dimension 3 needs a y grid");
    if (!three_d && (gr->n_eta < 2 || !gr->eta || !gr->eta_w)) return set_error(IS3D_EINVAL, "dimension 2 needs an eta table of >= 2 nodes");
    if (tab)
        if (int rc = check_tables(tab)) return rc;
    if (max_cells < 1) max_cells = 1;
    if (max_cells > 0x7fff0000LL) return set_error(IS3D_EINVAL, "max_cells out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");
    std::unique_ptr<is3d_vah_plan> P(new is3d_vah_plan);
    is3d::count_resource(0);
    P->o = *o;
    if (o->device >= 0) VAH_TRY(hipSetDevice(o->device));
    VAH_TRY(hipGetDevice(&P->device));
    P->three_d = three_d;
    P->max_cells = max_cells;

    // ---- species classes and lanes sorted by mT (as cf_plan.cpp) ----
    const int npart = sp->n, npT = gr->n_pT, J = gr->n_phi, K = three_d ? gr->n_y : gr->n_eta, Kacc = three_d ? K : 1;
    P->npart = npart; P->npT = npT; P->J = J; P->K = K; P->Kacc = Kacc;
    P->nout = (int64_t)npart * npT * J * Kacc;
    std::vector<int> cls(npart);
    std::vector<double> cmass, csign;
    for (int s = 0; s < npart; s++) {
        int found = -1;
        if (o->collapse_species != 2)
            for (size_t c = 0; c < cmass.size(); c++)
                if (cmass[c] == sp->mass[s] && csign[c] == sp->sign[s]) { found = (int)c; break; }
        if (found < 0) { found = (int)cmass.size(); cmass.push_back(sp->mass[s]); csign.push_back(sp->sign[s]); }
        cls[s] = found;
    }
    if (o->kernel_variant != 0 && o->kernel_variant != 2 && o->kernel_variant != 3)
        return set_error(IS3D_EINVAL, "VAH kernel_variant %d: 0 (default), 2 (round-1 kernel: 6 x 7 tile in 3+1D, 8 x 61 in 2+1D) or 3 (factored exponent: "
                         "8 x 7 tile in 3+1D, 8 x 31 with unit-strided lanes in 2+1D)", o->kernel_variant);
    // kernel_variant 2 (the round-1 kernel, expanded quadratic form) exists in the developer build for A/B; the shipped library runs cf_main_vah3
    P->fact = !(is3d::kDevBuild && o->kernel_variant == 2);
    P->JT = three_d ? (P->fact ? kJT3F : kJT3) : (P->fact ? kJT2F : kJT2); P->R = three_d ? (P->fact ? kR3F : kR3) : (P->fact ? kR2F : kR2);
    P->jtiles = (J + P->JT - 1) / P->JT; P->rblocks = (K + P->R - 1) / P->R;
    const int ncls = (int)cmass.size(), Lbins = ncls * npT;
    // unit-strided lanes (2+1D F records, as cf_plan.cpp's variant 7): S lane slots per momentum bin when that fills the waves better -- S in
    // {1, 2, 4} must divide the units per cell (the eta table's row blocks) and the four units of an LDS buffer
    int split = 1;
    if (!three_d && P->fact) {
        double best = 1.0 - (double)Lbins / (double)(((Lbins + 63) / 64) * 64);
        for (int S : {2, 4}) {
            if (P->rblocks % S) continue;
            const int tot = Lbins * S, pad = ((tot + 127) / 128) * 128;   // whole 2-wave workgroups
            const double waste = 1.0 - (double)tot / (double)pad;
            if (waste < best - 0.05) { best = waste; split = S; }
        }
    }
    P->Lbins = Lbins; P->split = split;
    const int L = Lbins * split, Lpad = ((L + 63) / 64) * 64;
    P->ncls = ncls; P->Lpad = Lpad;
    std::vector<double> mT(Lpad, 1.0), pT(Lpad, 0.0), sg(Lpad, 1.0), mT_nat(Lbins);
    std::vector<int32_t> lsub(Lpad, 0);
    std::vector<int> order(Lbins), slot_of(Lbins), lane_sp((size_t)npart * npT);
    for (int c = 0; c < ncls; c++)
        for (int i = 0; i < npT; i++) { mT_nat[c * npT + i] = std::sqrt(cmass[c] * cmass[c] + gr->pT[i] * gr->pT[i]); order[c * npT + i] = c * npT + i; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return mT_nat[a] < mT_nat[b]; });
    for (int s = 0; s < Lbins; s++) {
        const int nat = order[s];
        slot_of[nat] = s;
        for (int sl = 0; sl < split; sl++) {   // slot (bin s, sub sl) = sl * Lbins + s (cf_finalize adds a bin's slots in this order)
            const int t = sl * Lbins + s;
            mT[t] = mT_nat[nat]; pT[t] = gr->pT[nat % npT]; sg[t] = csign[nat / npT]; lsub[t] = sl;
        }
    }
    for (int s = 0; s < npart; s++)
        for (int i = 0; i < npT; i++) lane_sp[(size_t)s * npT + i] = slot_of[cls[s] * npT + i];
    VAH_TRY(P->d_lane_sub.upload(lsub));
    std::vector<double> cosphi(J), sinphi(J), kgrid(K), kweight(K, 1.0), deg(sp->degeneracy, sp->degeneracy + npart);
    for (int j = 0; j < J; j++) { cosphi[j] = std::cos(gr->phi[j]); sinphi[j] = std::sin(gr->phi[j]); }
    for (int k = 0; k < K; k++) {
        kgrid[k] = three_d ? gr->y[k] : gr->eta[k];
        if (!three_d) kweight[k] = gr->eta_w[k] * (gr->eta[1] - gr->eta[0]);          // :2178-2187
    }
    VAH_TRY(P->d_mT.upload(mT)); VAH_TRY(P->d_pT.upload(pT)); VAH_TRY(P->d_sg.upload(sg)); VAH_TRY(P->d_lane.upload(lane_sp)); VAH_TRY(P->d_deg.upload(deg));
    VAH_TRY(P->d_cos.upload(cosphi)); VAH_TRY(P->d_sin.upload(sinphi)); VAH_TRY(P->d_kg.upload(kgrid)); VAH_TRY(P->d_kw.upload(kweight));
    if (tab) {
        if (int rc = P->tab.upload(tab)) return rc;
        P->tables = true;
        VAH_TRY(P->d_coef.alloc(sizeof(double) * 5 * (size_t)max_cells));
    }
    for (int s2 = 0; s2 < L; s2++) { P->mTmax = std::max(P->mTmax, mT[s2]); P->pTmax = std::max(P->pTmax, pT[s2]); }
    P->ktiles = three_d ? P->rblocks : 1; P->upc = three_d ? 1 : P->rblocks;
    P->REC = P->fact ? 4 * P->JT + P->R * ((three_d ? 4 : 6) + P->JT) : 4 * P->JT + P->R * (4 + 2 * P->JT);
    P->lds_prep = sizeof(is3d::VahScal) * is3d::vah_batch_cells(K) + sizeof(double) * ((size_t)is3d::vah_batch_cells(K) * (10 * K + 8 * J + 2 * P->rblocks) + (size_t)P->REC);   // + [CB][jtiles <= J] tile maxima, [CB][2 rblocks] row-block minima and the REC-entry descriptor table of the F records
    if (P->lds_prep > 160 * 1024) return set_error(IS3D_EINVAL, "grids too large for the prep kernel's LDS staging");
    // passes over the cell axis bounded by the workspace (default 16 GB), chunks as in cf_plan.cpp
    const size_t bytes_per_cell = sizeof(double) * (size_t)P->jtiles * P->rblocks * P->REC;
    const int64_t ws = is3d::default_stream_cap_bytes(o->workspace_bytes);   // cf_launch.h: the rule of cf_plan.cpp, its clamp beside the partial slab included
    P->pass_cells = std::max<int64_t>(1, std::min<int64_t>(max_cells, ws / (int64_t)bytes_per_cell));
    const int lane_waves = Lpad / 64;
    int best = 1 << 30;
    for (int w : {8, 4, 2}) { const int waste = ((lane_waves + w - 1) / w) * w - lane_waves; if (waste < best) { best = waste; P->wpb = w; } }
    if (o->waves_per_group == 2 || o->waves_per_group == 4 || o->waves_per_group == 8) P->wpb = o->waves_per_group;
    // cf_main_vah3, 3+1D: one-wave workgroups by default (no barrier partner; 962.2 against 974.6 ms on the config-5 surface, bitwise the same spectrum:
    // profiles/r04_ab_vah_wpb.log); waves_per_group = 2 keeps the pair
    if ((o->waves_per_group == 1 || o->waves_per_group == 0) && P->fact && three_d) P->wpb = 1;
    // chunk count as in cf_plan.cpp: ~24 rounds of the chip, chunks of at most ~1152 cells (the streams of a chunk stay near the XCD's L2), partials <= 12 GiB
    const int64_t tasks_per_chunk = (int64_t)lane_waves * P->jtiles * P->ktiles;
    int64_t nch = o->cell_chunks > 0 ? o->cell_chunks : std::max<int64_t>((24LL * 4096 + tasks_per_chunk - 1) / tasks_per_chunk, (P->pass_cells + 1151) / 1152);
    nch = std::min<int64_t>(nch, std::max<int64_t>(1, P->pass_cells / 64));
    nch = std::max<int64_t>(1, std::min<int64_t>(nch, ((int64_t)(o->cell_chunks > 0 ? 32 : 12) << 30) / ((int64_t)J * Kacc * Lpad * 8)));
    P->nch = (int)nch;
    for (int which = 0; which < 2; which++) {
        const size_t bytes = which == 0 ? (size_t)P->pass_cells * bytes_per_cell + 64 * 1024   // + slack: the staging pieces of the last batch over-read the stream
                                        : (size_t)(nch + is3d::kTaperExtra /* the tapered tail of the partition, below */) * J * Kacc * Lpad * sizeof(double);
        const hipError_t e = (which == 0 ? P->d_TS : P->d_partial).alloc(bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            size_t fr = 0, to = 0;
            (void)hipMemGetInfo(&fr, &to);
            return set_error(IS3D_ENOMEM, "out of device memory allocating %s (%.2f GB; %.2f GB free of %.2f GB): %lld cells per pass x %zu B of streams, "
                             "%d partial slabs -- lower opts.workspace_bytes (more passes) or opts.cell_chunks",
                             which == 0 ? "the unit-record stream" : "the per-chunk partial spectra", bytes / 1e9, fr / 1e9, to / 1e9, (long long)P->pass_cells,
                             bytes_per_cell, (int)(nch + is3d::kTaperExtra));
        }
        if (e != hipSuccess) return set_error(IS3D_ENODEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    VAH_TRY(P->d_status.alloc(8 * sizeof(unsigned long long)));
    *out = P.release();
    return IS3D_OK;
}

extern "C" int64_t is3d_vah_plan_output_size(const is3d_vah_plan *P) { return P ? P->nout : 0; }
extern "C" int64_t is3d_vah_plan_workspace_bytes(const is3d_vah_plan *P) { return P ? (int64_t)(P->d_TS.bytes + P->d_partial.bytes + P->d_coef.bytes) : 0; }
extern "C" int is3d_vah_plan_set_timing(is3d_vah_plan *P, int32_t enable)
{
    if (!P) return is3d::set_error(IS3D_EINVAL, "null plan");
    P->timing = enable != 0;
    return IS3D_OK;
}
extern "C" int is3d_vah_plan_tile_shape(const is3d_vah_plan *P, int32_t *JT, int32_t *R)
{
    if (!P || !JT || !R) return is3d::set_error(IS3D_EINVAL, "null argument");
    *JT = P->JT; *R = P->R;
    return IS3D_OK;
}
extern "C" const char *is3d_vah_plan_main_kernel_name(const is3d_vah_plan *P) { return (P && P->fact) ? "cf_main_vah3" : "cf_main_vah"; }
extern "C" void is3d_vah_plan_destroy(is3d_vah_plan *P)
{
    if (!P) return;
    (void)hipSetDevice(P->device);
    delete P;
}

extern "C" int is3d_vah_plan_execute(is3d_vah_plan *P, const is3d_vah_cells *cells, double *dN_out, void *hip_stream, is3d_status *status)
{
    using is3d::set_error;
    if (!P || !cells || !dN_out) return set_error(IS3D_EINVAL, "null argument");
    if (status) { memset(status, 0, sizeof *status); status->bad_cell = -1; status->n_classes = P->ncls; }
    const int64_t n = cells->n_cells;
    if (n < 0 || n > P->max_cells) return set_error(IS3D_EINVAL, "n_cells = %lld exceeds the plan's max_cells = %lld", (long long)n, (long long)P->max_cells);
    const double *src[30] = {cells->tau, cells->eta, cells->ux, cells->uy, cells->un, cells->dat, cells->dax, cells->day, cells->dan, cells->T,
                             cells->pitt, cells->pitx, cells->pity, cells->pitn, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn,
                             cells->pinn, cells->bulkPi, cells->Wx, cells->Wy, cells->Lambda, cells->aL, cells->c0, cells->c1, cells->c2,
                             cells->c3, cells->c4};
    if (n > 0)
        for (int a = 0; a < 30; a++)
            if (!src[a] && !(a == 1 && !P->three_d) && a != 9 && !(a >= 25 && P->tables))
                return set_error(IS3D_EINVAL, "a required VAH cell array is NULL (index %d)", a);
    hipStream_t st = (hipStream_t)hip_stream;
    VAH_TRY(hipSetDevice(P->device));
    const is3d_options &o = P->o;
    const int npasses = n == 0 ? 0 : (int)((n + P->pass_cells - 1) / P->pass_cells);
    if (P->timing)
        while (P->ev.size() < (size_t)npasses * 3 + 1) {
            hipEvent_t e;
            VAH_TRY(hipEventCreate(&e));
            P->ev.push_back(e);
        }
    P->last_passes = npasses;
    unsigned long long init[8] = {~0ULL, 0, 0, 0, 0, 0, 0, ~0ULL};
    unsigned long long *d_st = P->d_status.as<unsigned long long>();
    VAH_TRY(hipMemcpyAsync(d_st, init, sizeof init, hipMemcpyHostToDevice, st));
    if (n == 0) {
        if (!o.accumulate) VAH_TRY(hipMemsetAsync(dN_out, 0, (size_t)P->nout * sizeof(double), st));
    }
    int nch = (int)std::max<int64_t>(1, std::min<int64_t>(P->nch, std::min<int64_t>(n, P->pass_cells) / 64));
    // tapered tail of the cell partition as in cf_plan.cpp (cf_device.h: chunk_cells, chunk_taper)
    const int nch_small = is3d::chunk_taper(std::min<int64_t>(n, P->pass_cells), o.cell_chunks, nch);
    for (int pass = 0; pass < npasses; pass++) {
        const int64_t c0 = (int64_t)pass * P->pass_cells;
        const int32_t nc = (int32_t)std::min<int64_t>(P->pass_cells, n - c0);
        const double *q[30];
        for (int a = 0; a < 30; a++) q[a] = src[a] ? src[a] + c0 : nullptr;
        if (P->timing) VAH_TRY(hipEventRecord(P->ev[pass * 3 + 0], st));
        if (P->tables) {
            is3d::VahCoefArgs ca{};
            ca.n = nc; ca.Lambda = q[23]; ca.aL = q[24];
            P->tab.fill(ca);
            for (int k = 0; k < 5; k++) { ca.out[k] = P->d_coef.as<double>() + (size_t)k * P->max_cells + c0; q[25 + k] = ca.out[k]; }
            ca.status = d_st;
            ca.cell0 = c0;
            hipLaunchKernelGGL(is3d::cf_vah_coeffs, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, ca);
            VAH_TRY(hipGetLastError());
        }
        is3d::VahPrepParams pp{};
        pp.cells = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[10], q[11], q[12], q[13], q[14], q[15], q[16], q[17], q[18], q[19],
                    q[20], q[21], q[22], q[23], q[24], q[25], q[26], q[27], q[28], q[29]};
        pp.n_cells = nc; pp.J = P->J; pp.K = P->K; pp.dim3 = P->three_d;
        pp.include_bulk = o.include_bulk_deltaf != 0; pp.include_shear = o.include_shear_deltaf != 0;
        pp.cosphi = P->d_cos.as<double>(); pp.sinphi = P->d_sin.as<double>(); pp.kgrid = P->d_kg.as<double>(); pp.kweight = P->d_kw.as<double>();
        pp.JT = P->JT; pp.R = P->R; pp.jtiles = P->jtiles; pp.rblocks = P->rblocks; pp.TS = P->d_TS.as<double>();
        pp.fact = P->fact; pp.cell0 = c0; pp.mTmax = P->mTmax; pp.pTmax = P->pTmax; pp.status = d_st;
        const int cb = is3d::vah_batch_cells(P->K);
        const int nbatch = (nc + cb - 1) / cb;
        if (cb == is3d::kVahCB3) hipLaunchKernelGGL(is3d::cf_prep_vah<is3d::kVahCB3>, dim3(std::min(nbatch, 4096)), dim3(is3d::kVahThreads), P->lds_prep, st, pp);
        else hipLaunchKernelGGL(is3d::cf_prep_vah<is3d::kVahCB>, dim3(std::min(nbatch, 4096)), dim3(is3d::kVahThreads), P->lds_prep, st, pp);
        VAH_TRY(hipGetLastError());
        if (P->timing) VAH_TRY(hipEventRecord(P->ev[pass * 3 + 1], st));
        is3d::MainGeom g{};
        g.n_cells = nc; g.J = P->J; g.K = P->K; g.Lpad = P->Lpad; g.wpb = P->wpb; g.G = (P->Lpad / 64 + P->wpb - 1) / P->wpb;
        g.jtiles = P->jtiles; g.ktiles = P->ktiles; g.nch = nch; g.nch_small = nch_small; g.NT = P->jtiles * P->ktiles * nch; g.Kacc = P->Kacc;
        g.first_pass = 1; g.upc = P->upc; g.zskip = (o.zero_skip != 2); g.baryon = 0; g.split = P->split;
        if (P->fact && P->three_d) launch_vah3<true>(o.regulate_deltaf != 0, P->d_TS.as<double>(), P->d_mT.as<double>(), P->d_pT.as<double>(), P->d_sg.as<double>(), P->d_partial.as<double>(), d_st, g, nullptr, st);
        else if (P->fact) launch_vah3<false>(o.regulate_deltaf != 0, P->d_TS.as<double>(), P->d_mT.as<double>(), P->d_pT.as<double>(), P->d_sg.as<double>(), P->d_partial.as<double>(), d_st, g, P->d_lane_sub.as<int32_t>(), st);
#ifdef IS3D_DEV   // the round-1 kernel (kernel_variant 2): developer build only
        else if (P->three_d) launch_vah<true>(o.regulate_deltaf != 0, P->d_TS.as<double>(), P->d_mT.as<double>(), P->d_pT.as<double>(), P->d_sg.as<double>(), P->d_partial.as<double>(), d_st, g, st);
        else launch_vah<false>(o.regulate_deltaf != 0, P->d_TS.as<double>(), P->d_mT.as<double>(), P->d_pT.as<double>(), P->d_sg.as<double>(), P->d_partial.as<double>(), d_st, g, st);
#endif
        VAH_TRY(hipGetLastError());
        if (P->timing) VAH_TRY(hipEventRecord(P->ev[pass * 3 + 2], st));
        // each pass finalises into the output (accumulating after the first): the partial slots are rewritten per pass
        const double prefactor = 1.0 / (8.0 * (M_PI * M_PI * M_PI)) / is3d::kHbarC / is3d::kHbarC / is3d::kHbarC;   // :2147
        VAH_TRY(is3d::launch_finalize(P->d_partial.as<double>(), P->d_lane.as<int>(), P->d_deg.as<double>(), dN_out, P->nout, P->npart, P->npT, P->J,
                                      P->Kacc, P->Lpad, nch, prefactor, (pass > 0 || o.accumulate) ? 1 : 0, nullptr, st, P->split, P->Lbins));
    }
    if (P->timing && npasses) VAH_TRY(hipEventRecord(P->ev[npasses * 3], st));
    if (status) {
        unsigned long long h[8];
        VAH_TRY(hipMemcpyAsync(h, d_st, sizeof h, hipMemcpyDeviceToHost, st));
        VAH_TRY(hipStreamSynchronize(st));
        status->n_passes = npasses;
        status->kernel_variant = P->fact ? 3 : 2;
        status->n_wave_rows = (int64_t)h[2];
        status->n_wave_rows_culled = (int64_t)h[3];
        if (h[7] != ~0ULL && (h[0] == ~0ULL || h[7] < h[0])) {
            status->bad_cell = (int64_t)h[7];
            status->code = IS3D_EDOMAIN;
            return set_error(IS3D_EDOMAIN, "cell %lld: E_a/Lambda can exceed 1e9 for the momentum grid (flow, Lambda or alpha_L outside the kernel's "
                             "exponent range; the reference's exp() overflows to inf there)", (long long)status->bad_cell);
        }
        if (h[0] != ~0ULL) {
            status->bad_cell = (int64_t)h[0];
            status->code = IS3D_EDOMAIN;
            return set_error(IS3D_EDOMAIN, "cell %lld: (Lambda, alpha_L) beyond the last node of the VAH coefficient tables (the reference leaves the "
                             "cell's c0..c4 unset there, src/cuda/deltafReader.cu:237-276)", (long long)status->bad_cell);
        }
    }
    return IS3D_OK;
}

extern "C" int is3d_vah_plan_timings(is3d_vah_plan *P, is3d_status *status)
{
    if (!P || !status) return is3d::set_error(IS3D_EINVAL, "null argument");
    status->ms_prep = status->ms_main = status->ms_finalize = 0.0;
    if (!P->timing || P->last_passes == 0) return IS3D_OK;
    VAH_TRY(hipSetDevice(P->device));
    VAH_TRY(hipEventSynchronize(P->ev[P->last_passes * 3]));
    for (int pass = 0; pass < P->last_passes; pass++) {
        float a = 0, b = 0, c = 0;
        VAH_TRY(hipEventElapsedTime(&a, P->ev[pass * 3 + 0], P->ev[pass * 3 + 1]));
        VAH_TRY(hipEventElapsedTime(&b, P->ev[pass * 3 + 1], P->ev[pass * 3 + 2]));
        VAH_TRY(hipEventElapsedTime(&c, P->ev[pass * 3 + 2], P->ev[pass * 3 + 3]));   // the next pass's start, or the closing event
        status->ms_prep += a; status->ms_main += b; status->ms_finalize += c;
    }
    status->n_passes = P->last_passes;
    status->kernel_variant = P->fact ? 3 : 2;
    status->n_classes = P->ncls;
    return IS3D_OK;
}

extern "C" int is3d_vah_coefficients(const is3d_vah_df_tables *tab, int64_t n, const double *Lambda, const double *aL, double *c0, double *c1,
                                     double *c2, double *c3, double *c4, int32_t device, int64_t *bad_cell)
{
    using is3d::set_error;
    if (bad_cell) *bad_cell = -1;
    if (!tab || n < 0 || (n > 0 && (!Lambda || !aL || !c0 || !c1 || !c2 || !c3 || !c4))) return set_error(IS3D_EINVAL, "null argument");
    if (int rc = check_tables(tab)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");
    if (n == 0) return IS3D_OK;
    if (device >= 0) VAH_TRY(hipSetDevice(device));
    TabDev td;
    if (int rc = td.upload(tab)) return rc;
    DevMem d_in, d_out, d_st;
    VAH_TRY(d_in.alloc(sizeof(double) * 2 * (size_t)n));
    VAH_TRY(d_out.alloc(sizeof(double) * 5 * (size_t)n));
    VAH_TRY(d_st.alloc(sizeof(unsigned long long)));
    const unsigned long long init = ~0ULL;
    VAH_TRY(hipMemcpy(d_st.p, &init, sizeof init, hipMemcpyHostToDevice));
    VAH_TRY(hipMemcpy(d_in.p, Lambda, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    VAH_TRY(hipMemcpy(d_in.as<double>() + n, aL, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    is3d::VahCoefArgs ca{};
    ca.n = n; ca.Lambda = d_in.as<double>(); ca.aL = d_in.as<double>() + n;
    td.fill(ca);
    for (int k = 0; k < 5; k++) ca.out[k] = d_out.as<double>() + (size_t)k * n;
    ca.status = d_st.as<unsigned long long>();
    hipLaunchKernelGGL(is3d::cf_vah_coeffs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, ca);
    VAH_TRY(hipGetLastError());
    double *dst[5] = {c0, c1, c2, c3, c4};
    for (int k = 0; k < 5; k++) VAH_TRY(hipMemcpy(dst[k], ca.out[k], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    unsigned long long bad = ~0ULL;
    VAH_TRY(hipMemcpy(&bad, d_st.p, sizeof bad, hipMemcpyDeviceToHost));
    if (bad != ~0ULL) {
        if (bad_cell) *bad_cell = (int64_t)bad;
        return set_error(IS3D_EDOMAIN, "cell %llu: (Lambda, alpha_L) beyond the last node of the VAH coefficient tables (the reference leaves the "
                         "cell's c0..c4 unset there, src/cuda/deltafReader.cu:237-276)", bad);
    }
    return IS3D_OK;
}

extern "C" int is3d_smooth_spectra_vah_df(const is3d_vah_cells *cells, const is3d_species *sp, const is3d_grid *gr, const is3d_vah_df_tables *tab,
                                          const is3d_options *o, double *dN_out, is3d_status *status)
{
    using is3d::set_error;
    if (!cells || !sp || !gr || !o || !dN_out) return set_error(IS3D_EINVAL, "null argument");
    if (status) { memset(status, 0, sizeof *status); status->bad_cell = -1; }
    const int64_t n = cells->n_cells;
    if (n < 0 || n > 0x7fff0000LL) return set_error(IS3D_EINVAL, "n_cells out of range");
    is3d_vah_plan *P = nullptr;
    if (int rc = is3d_vah_plan_create(&P, sp, gr, tab, o, std::max<int64_t>(n, 1))) return rc;
    struct Guard { is3d_vah_plan *p; ~Guard() { is3d_vah_plan_destroy(p); } } guard{P};
    (void)is3d_vah_plan_set_timing(P, 1);
    const double *src[30] = {cells->tau, cells->eta, cells->ux, cells->uy, cells->un, cells->dat, cells->dax, cells->day, cells->dan, cells->T,
                             cells->pitt, cells->pitx, cells->pity, cells->pitn, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn,
                             cells->pinn, cells->bulkPi, cells->Wx, cells->Wy, cells->Lambda, cells->aL, cells->c0, cells->c1, cells->c2,
                             cells->c3, cells->c4};
    if (n > 0)
        for (int a = 0; a < 30; a++)
            if (!src[a] && !(a == 1 && o->dimension == 2) && a != 9 && !(a >= 25 && tab)) return set_error(IS3D_EINVAL, "a required VAH cell array is NULL (index %d)", a);
    DevMem d_cell, d_out;
    VAH_TRY(d_cell.alloc(sizeof(double) * 30 * (size_t)std::max<int64_t>(n, 1)));
    VAH_TRY(d_out.alloc(sizeof(double) * (size_t)P->nout));
    const double *dp[30];
    for (int a = 0; a < 30; a++) {
        dp[a] = nullptr;
        if (src[a] && a != 9 && !(a >= 25 && tab) && n > 0) {
            VAH_TRY(hipMemcpyAsync(d_cell.as<double>() + (size_t)a * n, src[a], (size_t)n * sizeof(double), hipMemcpyHostToDevice, nullptr));
            dp[a] = d_cell.as<double>() + (size_t)a * n;
        }
    }
    if (o->accumulate) VAH_TRY(hipMemcpyAsync(d_out.p, dN_out, (size_t)P->nout * sizeof(double), hipMemcpyHostToDevice, nullptr));
    is3d_vah_cells dc{};
    dc.n_cells = n;
    dc.tau = dp[0]; dc.eta = dp[1]; dc.ux = dp[2]; dc.uy = dp[3]; dc.un = dp[4]; dc.dat = dp[5]; dc.dax = dp[6]; dc.day = dp[7]; dc.dan = dp[8];
    dc.T = nullptr; dc.pitt = dp[10]; dc.pitx = dp[11]; dc.pity = dp[12]; dc.pitn = dp[13]; dc.pixx = dp[14]; dc.pixy = dp[15]; dc.pixn = dp[16];
    dc.piyy = dp[17]; dc.piyn = dp[18]; dc.pinn = dp[19]; dc.bulkPi = dp[20]; dc.Wx = dp[21]; dc.Wy = dp[22]; dc.Lambda = dp[23]; dc.aL = dp[24];
    dc.c0 = dp[25]; dc.c1 = dp[26]; dc.c2 = dp[27]; dc.c3 = dp[28]; dc.c4 = dp[29];
    is3d_status st{};
    const int rc = is3d_vah_plan_execute(P, &dc, d_out.as<double>(), nullptr, &st);
    if (rc) { if (status) *status = st; return rc; }
    VAH_TRY(hipMemcpy(dN_out, d_out.p, (size_t)P->nout * sizeof(double), hipMemcpyDeviceToHost));
    is3d_status t{};
    (void)is3d_vah_plan_timings(P, &t);
    st.ms_prep = t.ms_prep; st.ms_main = t.ms_main; st.ms_finalize = t.ms_finalize;
    st.code = IS3D_OK;
    if (status) *status = st;
    return IS3D_OK;
}

extern "C" int is3d_smooth_spectra_vah(const is3d_vah_cells *cells, const is3d_species *sp, const is3d_grid *gr, const is3d_options *o,
                                       double *dN_out, is3d_status *status)
{
    return is3d_smooth_spectra_vah_df(cells, sp, gr, nullptr, o, dN_out, status);
}
