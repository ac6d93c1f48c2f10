// cf_sampler.hip -- particle sampler (operation = 2) on the device: SURVEY.md 8f rank 4.
//
// Device path of EmissionFunctionArray::sample_dN_pTdpTdphidy
// (/root/reference/src/cpp/emissionfunction_sampling_kernels.cpp:833-1225) for viscous hydro, df_mode 1-4 (linear delta-f with
// its viscous weight; modified equilibrium with momentum rescaling), regular and "fast" mode, include_baryon = 1 in regular
// mode for df_mode 1-3 (chemical potential, bulk1 / diffusion terms, diffusion in the momentum rescaling); helpers
// max_particle_number / fast_max_particle_number (:239-359), sample_momentum (:456-617), compute_df_weight (:361-453),
// rescale_momentum (:619-650), does_feqmod_breakdown (emissionfunction.cpp:109-150), Milne_Basis / Surface_Element_Vector /
// boost_pimunu_to_lrf (viscous_correction.cpp:8-115), boost_pLRF_to_lab_frame (emissionfunction.cpp:40-51).
//
// The reference walks the cells serially and feeds five std::default_random_engine streams through implementation-defined
// std:: distributions; that order dependence cannot (and need not) be reproduced.  Here every (cell, event, stream) owns a
// counter-based Philox4x32-10 sequence (Salmon et al., SC'11): key = the 64-bit seed, counter = (block, stream, global cell
// index, event); streams 0..4 = hadron number, species, momentum, keep test, rapidity (the roles of the reference's five
// engines, :846-850); u = ((a >> 5) 2^26 + (b >> 6)) 2^-53 from two consecutive 32-bit outputs; Poisson numbers by
// sequential-search inversion in chunks of mean <= 256; species and the heavy-hadron K mixture by inversion of the cumulative
// weights; cos(theta) = 2u - 1.  The hadrons of a cell therefore depend on nothing but (seed, global cell index, event): any
// launch geometry, any cell sharding over GPUs and any event batching give the same particle list.
//
//   cf_sampler_density  thread <-> (cell, species class): 32-point Gauss-Laguerre equilibrium density integral
//   cf_sampler_cells    thread <-> cell: LRF basis, dsigma and pi^{mu nu} in the LRF, delta-f coefficients, mean hadron number
//   cf_sampler_poisson  thread <-> (event, cell): the Poisson number only.  A few per cent of the pairs emit anything, but nearly
//                       every wave holds one that does: sampling right here would drag 64 lanes through the rejection loops
//                       of one.  hipCUB compacts the emitting pairs into a dense, ordered list instead.
//   cf_sampler_run      thread <-> emitting (event, cell) pair: species, momentum rejection loop, viscous and flux weights, keep
//                       test; pass 1 counts, an exclusive scan turns counts into offsets, pass 2 replays the same streams and
//                       writes the particles -- the list comes out ordered by (event, cell, draw), no atomics, no sort.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/is3d_amd.h"
#include "cf_device.h"
#include "cf_math.h"
#include "errors.h"
#include "jonah.h"
#include "spline.h"

namespace is3d {

struct SamplerCell {
    double live, breakdown;
    double T_mod, shear_mod, bulk_mod, z;          // df_mode 3, 4
    double tau, x, y, eta, ut, ux, uy, un, T;
    double Xt, Xx, Xy, Xn, Yx, Yy, Zt, Zn;
    double dst, dsx, dsy, dsz, ds_max;
    double pixx, pixy, pixz, piyy, piyz, pizz;     // LRF
    double bulkPi, dn_tot, dn_sum, neq_fact;
    double c0, c2, F, betabulk, betapi, shear14;
    // include_baryon: alpha_B (:962), alpha_B,mod (:1022), n_B/(E+P), T/betaV (:1026), V^i in the LRF (boost_Vmu_to_lrf), c1 c3 c4 | G betaV
    double alphaB, alphaB_mod, ber, diff_mod, Vx, Vy, Vz, c1, c3, c4, G, betaV;
};

struct SamplerSpecies {       // device arrays, length npart / ncls
    const double *mass, *sign, *degeneracy;
    const int32_t *cls;
    const double *cls_mass, *cls_sign;
    int32_t npart, ncls;
    const double *baryon, *cls_baryon;   // include_baryon, else NULL (the baryon number is part of the class key then)
};

struct SamplerParams {
    CellPtrs cells;
    const double *x, *y;
    int64_t n_cells, first_cell;
    int32_t dim3, df_mode, include_bulk, include_shear;
    int32_t baryon, baryondiff;  // include_baryon; && include_baryondiff_deltaf: mu_B, n_B, V^mu are read (:953-964)
    BilinearDev bil;             // baryon: c0..c4 (df_mode 1) | F G betabulk betaV betapi (df_mode 2, 3) on the (mu_B, T) grid
    SplineDev spl;              // 14-moment: c0, c2; Chapman-Enskog: F, betabulk, betapi
    int32_t ngl;
    const double *gl;           // [4][ngl]: root1, weight1, root2, weight2 (alpha = 2 only for df_mode 3)
    int32_t fast;               // species densities at the surface-average temperature (host arrays eqd, bkd)
    const double *eqd, *bkd;    // [npart] Equilibrium_Density, Bulk_Density (deltafReader.cpp:536-650)
    double T_sw, F_avg, betabulk_avg;   // fast breakdown test (emissionfunction.cpp:114-119)
    int32_t nj;                 // Jonah tables (df_mode 4): abscissa, lambda^2, z and their spline c's
    const double *jx, *jl2, *jz, *jcl, *jcz;
    double bp_max, detA_min, mass_pion0;
    double y_max;
    uint64_t seed;
    unsigned long long *status; // [0] min bad cell, [1] skipped, [2] momentum samples, [3] acceptances, [4] hadrons drawn, [5] breakdown cells
    double *cdf;                // [ceil(npart / kCdfBlock)][n_cells]: the running sum of a cell's species weights after every kCdfBlock-th species, as
                                // cf_sampler_cells adds them (NULL: df_mode 3)
};

// ---- Philox4x32-10 streams ----
struct Rng {
    uint32_t k0, k1, stream, cell, event, blk, buf[4];
    int pos;
    __device__ void init(uint64_t seed, uint32_t s, uint32_t c, uint32_t e)
    {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        stream = s; cell = c; event = e; blk = 0; pos = 4;
    }
    __device__ double uniform()
    {
        if (pos >= 4) {
            uint32_t c0 = blk++, c1 = stream, c2 = cell, c3 = event, a = k0, b = k1;
#pragma unroll
            for (int r = 0; r < 10; r++) {
                const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
                const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
                c0 = hi1 ^ c1 ^ a; c1 = lo1; c2 = hi0 ^ c3 ^ b; c3 = lo0;
                a += 0x9E3779B9u; b += 0xBB67AE85u;
            }
            buf[0] = c0; buf[1] = c1; buf[2] = c2; buf[3] = c3;
            pos = 0;
        }
        const uint32_t a = buf[pos], b = buf[pos + 1];
        pos += 2;
        return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
    }
    __device__ long poisson(double mean)
    {
        long N = 0;
        double remaining = mean;
        while (remaining > 0.0) {
            const double l = remaining < 256.0 ? remaining : 256.0;
            remaining -= l;
            const double u = uniform();
            double p = exp(-l), F = p;
            long k = 0;
            while (u >= F && k < 4096) { k++; p *= l / (double)k; F += p; }
            N += k;
        }
        return N;
    }
};

// GaussThermal(neq_int | J10_int | J20_int, ...) (gaussThermal.cpp); chem = baryon * alpha_B
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

// GT[cell][class] = the n_eq integral; GT2 (df_mode 3, regular mode) = the J20 integral of n_linear, GT3 (with include_baryon) its
// J10 integral; muB_fo != NULL (include_baryon && include_baryondiff_deltaf): chem = baryon mu_B / T
// The integrands are those of gt_neq / gt_J20 / gt_J10 above with the node-only factors taken out and tabulated once per workgroup
// (c1 = w p e^p for the alpha = 1 nodes, c2 = w e^p for the alpha = 2 nodes: one exponential per node and integral instead of two), and with
// exp_full / sqrt_nr / rcp_nr of cf_math.h (1e-15 relative; arguments of order 1 .. 1e3) in place of libm's exp, sqrt and the division -- as
// cf_feqmod_renorm does since round 3: 3.3 -> ~1 ms per 1e6 cells x 75 classes.  An exponential that overflowed is held at 1e300 so that its
// node adds < 1e-300 of its weight instead of a division by inf.
constexpr int kSmpGlMax = 256;   // n_gla <= 256 (is3d_sampler_plan_create checks)
__global__ void __launch_bounds__(256)
cf_sampler_density(const double *__restrict__ T_fo, const double *__restrict__ muB_fo, int64_t n_cells, SamplerSpecies sp,
                   const double *__restrict__ gl, int ngl, double *__restrict__ GT, double *__restrict__ GT2, double *__restrict__ GT3)
{
    __shared__ double l_p1[kSmpGlMax], l_c1[kSmpGlMax], l_p2[kSmpGlMax], l_c2[kSmpGlMax];
    for (int k = threadIdx.x; k < ngl; k += blockDim.x) {
        const double p1 = gl[k];
        l_p1[k] = p1 * p1; l_c1[k] = gl[ngl + k] * (p1 * exp(p1));
        if (GT2) { const double p2 = gl[2 * ngl + k]; l_p2[k] = p2 * p2; l_c2[k] = gl[3 * ngl + k] * exp(p2); }
    }
    __syncthreads();
    // class-major, GT[class][cell]: the lanes of a wave are consecutive cells of ONE class -- T_fo and the stores coalesce, the class constants
    // are wave-uniform -- and the readers (thread <-> cell) read a class's value of consecutive cells as one run
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_cells * sp.ncls) return;
    const int c = (int)(idx / n_cells);
    const int64_t cell = idx - (int64_t)c * n_cells;
    const double T = T_fo[cell];
    const double mbar = sp.cls_mass[c] / T, mb2 = mbar * mbar, sign = sp.cls_sign[c];
    const double alphaB = muB_fo ? muB_fo[cell] / T : 0.0;
    const double chem = sp.cls_baryon ? sp.cls_baryon[c] * alphaB : 0.0;
    double s_neq = 0.0, s_j10 = 0.0, s_j20 = 0.0;
    for (int k = 0; k < ngl; k++) {
        const double e = __builtin_fmin(exp_full(sqrt_nr(l_p1[k] + mb2) - chem), 1.0e300), r = rcp_nr(e + sign);
        s_neq = __builtin_fma(l_c1[k], r, s_neq);
        if (GT3) s_j10 = __builtin_fma(l_c1[k], (e * r) * r, s_j10);
    }
    GT[idx] = s_neq;
    if (GT3) GT3[idx] = s_j10;
    if (GT2) {
        for (int k = 0; k < ngl; k++) {
            const double E2 = sqrt_nr(l_p2[k] + mb2), e2 = __builtin_fmin(exp_full(E2 - chem), 1.0e300), r2 = rcp_nr(e2 + sign);
            s_j20 = __builtin_fma(l_c2[k], E2 * ((e2 * r2) * r2), s_j20);
        }
        GT2[idx] = s_j20;
    }
}

// mean-number weight of species ip in a cell: fast_max_particle_number (:239-280) / max_particle_number (:282-359)
// gt, gt2, gt3: the cell's column of the class-major integral tables (GT + cell), element of class k at [k * p.n_cells]
// The running sums are kept for every kCdfBlock-th species only (round 5: 39 planes instead of 305 -- cf_sampler_cells was bound by these stores, 2.4 GB
// per 1e6 cells): a hadron's species is the bisection of the block sums followed by the producer's own additions inside one block, continued from the stored
// sum in front of it -- the same doubles in the same order, so the same species as the bisection of all 305 sums.
constexpr int kCdfBlock = 8;

__device__ __forceinline__ double species_dn(const SamplerParams &p, const SamplerSpecies &sp, const SamplerCell &c, const double *gt,
                                             const double *gt2, const double *gt3, int ip)
{
    const bool linear = p.df_mode <= 2 || c.breakdown != 0.0;
    if (p.fast) {
        if (linear) return 2.0 * p.eqd[ip];
        if (p.df_mode == 3) return p.eqd[ip] + c.bulkPi * p.bkd[ip];
        return c.z * p.eqd[ip];
    }
    const int64_t kk = (int64_t)sp.cls[ip] * p.n_cells;
    const double equilibrium_density = c.neq_fact * sp.degeneracy[ip] * gt[kk];
    if (linear) return 2.0 * equilibrium_density;
    if (p.df_mode == 3) {
        const double J20 = (c.T * c.neq_fact) * sp.degeneracy[ip] * gt2[kk];
        double bJ10G = 0.0;                                                             // baryon * J10 * G, :319-325
        if (gt3) bJ10G = sp.baryon[ip] * (c.neq_fact * sp.degeneracy[ip] * gt3[kk]) * c.G;
        const double bulk_density = (equilibrium_density + bJ10G + (J20 * c.F / c.T / c.T)) / c.betabulk;
        return equilibrium_density + c.bulkPi * bulk_density;
    }
    return c.z * equilibrium_density;
}

// The cell's n_eq integrals are summed over the SPECIES in list order (the order the reference adds them in).  With GT[cell][class] that was 305
// gathers per lane from rows 600 B apart -- 3.0 of the kernel's 3.1 ms (64 cache lines per load instruction); staged through LDS (38 KB per 64
// cells: four waves per CU) 1.75 ms; with the class-major tables a lane's read of class k is word `cell` of row k: consecutive lanes, one run,
// no LDS, full occupancy.  Same values, same order of additions.
__global__ void __launch_bounds__(128)
cf_sampler_cells(SamplerParams p, SamplerSpecies sp, const double *__restrict__ GT, const double *__restrict__ GT2,
                 const double *__restrict__ GT3, SamplerCell *__restrict__ out)
{
    const int64_t ic = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ic >= p.n_cells) return;
    SamplerCell c;
    memset(&c, 0, sizeof c);
    const double tau = p.cells.tau[ic], tau2 = tau * tau;
    const double dat = p.cells.dat[ic], dax = p.cells.dax[ic], day = p.cells.day[ic], dan = p.cells.dan[ic];
    const double ux = p.cells.ux[ic], uy = p.cells.uy[ic], un = p.cells.un[ic];
    const double ut = sqrt(1.0 + ux * ux + uy * uy + tau2 * un * un);
    const double udsigma = ut * dat + ux * dax + uy * day + un * dan;
    if (udsigma <= 0.0) {                                                          // :899
        atomicAdd(&p.status[1], 1ULL);
        out[ic] = c;
        return;
    }
    const double T = p.cells.T[ic], P = p.cells.P[ic], E = p.cells.E[ic];
    double muB = 0.0, nB = 0.0, Vt = 0.0, Vx = 0.0, Vy = 0.0, Vn = 0.0;               // :942-964
    if (p.baryon && p.baryondiff) {
        muB = p.cells.muB[ic]; nB = p.cells.nB[ic]; Vx = p.cells.Vx[ic]; Vy = p.cells.Vy[ic]; Vn = p.cells.Vn[ic];
        Vt = (Vx * ux + Vy * uy + tau2 * Vn * un) / ut;
        c.alphaB = muB / T;
        c.ber = nB / (E + P);
    }
    double bl[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (p.baryon ? !bilinear5(p.bil, T, muB, bl) : !(T >= p.spl.x[0] && T <= p.spl.x[p.spl.n - 1])) {   // exit(-1) / GSL domain error in the reference
        atomicMin(&p.status[0], (unsigned long long)(p.first_cell + ic));
        memset(&c, 0, sizeof c);
        out[ic] = c;
        return;
    }
    const double ut2 = ut * ut, ux2 = ux * ux, uy2 = uy * uy;
    const double uperp = sqrt(ux * ux + uy * uy), utperp = sqrt(1.0 + ux * ux + uy * uy);
    double pitt = 0, pitx = 0, pity = 0, pitn = 0, pixx = 0, pixy = 0, pixn = 0, piyy = 0, piyn = 0, pinn = 0;
    if (p.include_shear) {                                                          // :922-934
        pixx = p.cells.pixx[ic]; pixy = p.cells.pixy[ic]; pixn = p.cells.pixn[ic]; piyy = p.cells.piyy[ic]; piyn = p.cells.piyn[ic];
        pinn = (pixx * (ux2 - ut2) + piyy * (uy2 - ut2) + 2.0 * (pixy * ux * uy + tau2 * un * (pixn * ux + piyn * uy))) / (tau2 * utperp * utperp);
        pitn = (pixn * ux + piyn * uy + tau2 * pinn * un) / ut;
        pity = (pixy * ux + piyy * uy + tau2 * piyn * un) / ut;
        pitx = (pixx * ux + pixy * uy + tau2 * pixn * un) / ut;
        pitt = (pitx * ux + pity * uy + tau2 * pitn * un) / ut;
    }
    double bulkPi = p.include_bulk ? p.cells.bulkPi[ic] : 0.0;
    const double T4 = T * T * T * T;
    const int nT = p.spl.n;
    double lambda = 0.0;
    if (p.baryon) {                                                                 // deltafReader.cpp:436-468 (bilinear branch)
        const double T3 = T * T * T;
        if (p.df_mode == 1) { c.c0 = bl[0] / T4; c.c1 = bl[1] / T3; c.c2 = bl[2] / T4; c.c3 = bl[3] / T4; c.c4 = bl[4] / (T4 * T); }
        else { c.F = bl[0] * T; c.G = bl[1]; c.betabulk = bl[2] * T4; c.betaV = bl[3] * T3; c.betapi = bl[4] * T4; }
    } else if (p.df_mode == 1) {                                                    // deltafReader.cpp:337-344
        c.c0 = spline_eval_lds(nT, p.spl.x, p.spl.y[0], p.spl.c[0], T) / T4;
        c.c2 = spline_eval_lds(nT, p.spl.x, p.spl.y[1], p.spl.c[1], T) / T4;
    } else if (p.df_mode <= 3) {                                                    // :352-358
        c.F = spline_eval_lds(nT, p.spl.x, p.spl.y[0], p.spl.c[0], T) * T;
        c.betabulk = spline_eval_lds(nT, p.spl.x, p.spl.y[1], p.spl.c[1], T) * T4;
        c.betapi = spline_eval_lds(nT, p.spl.x, p.spl.y[2], p.spl.c[2], T) * T4;
    } else {                                                                        // :966-972, deltafReader.cpp:364-384
        if (bulkPi <= -P) bulkPi = -(1.0 - 1.e-5) * P;
        else if (bulkPi / P >= p.bp_max) bulkPi = P * (p.bp_max - 1.e-5);
        const double r = bulkPi / P;
        if (!(r >= p.jx[0] && r <= p.jx[p.nj - 1])) {
            atomicMin(&p.status[0], (unsigned long long)(p.first_cell + ic));
            out[ic] = c;
            return;
        }
        const double lambda_squared = spline_eval_lds(p.nj, p.jx, p.jl2, p.jcl, r);
        if (bulkPi < 0.0) lambda = -sqrt(lambda_squared);
        else if (bulkPi > 0.0) lambda = sqrt(lambda_squared);
        c.z = spline_eval_lds(p.nj, p.jx, p.jz, p.jcz, r);
        c.betapi = spline_eval_lds(nT, p.spl.x, p.spl.y[2], p.spl.c[2], T) * T4;
    }
    c.bulkPi = bulkPi;
    c.shear14 = 2.0 * T * T * (E + P);
    // Milne_Basis (viscous_correction.cpp:8-27)
    const double sinhL = tau * un / utperp, coshL = ut / utperp;
    c.Xt = uperp * coshL; c.Zt = sinhL; c.Xn = uperp * sinhL / tau; c.Zn = coshL / tau;
    c.Xx = 1.0; c.Yx = 0.0; c.Xy = 0.0; c.Yy = 1.0;
    if (uperp > 1.e-5) { c.Xx = utperp * ux / uperp; c.Yx = -uy / uperp; c.Xy = utperp * uy / uperp; c.Yy = ux / uperp; }
    const double Xt = c.Xt, Xx = c.Xx, Xy = c.Xy, Xn = c.Xn, Yx = c.Yx, Yy = c.Yy, Zt = c.Zt, Zn = c.Zn;
    // boost_dsigma_to_lrf, compute_dsigma_magnitude (:69-86)
    c.dst = dat * ut + dax * ux + day * uy + dan * un;
    c.dsx = -(dat * Xt + dax * Xx + day * Xy + dan * Xn);
    c.dsy = -(dax * Yx + day * Yy);
    c.dsz = -(dat * Zt + dan * Zn);
    c.ds_max = fabs(c.dst) + sqrt(c.dsx * c.dsx + c.dsy * c.dsy + c.dsz * c.dsz);
    // boost_pimunu_to_lrf (:99-115)
    c.pixx = pitt * Xt * Xt + pixx * Xx * Xx + piyy * Xy * Xy + tau2 * tau2 * pinn * Xn * Xn
           + 2.0 * (-Xt * (pitx * Xx + pity * Xy) + pixy * Xx * Xy + tau2 * Xn * (pixn * Xx + piyn * Xy - pitn * Xt));
    c.pixy = Yx * (-pitx * Xt + pixx * Xx + pixy * Xy + tau2 * pixn * Xn) + Yy * (-pity * Xt + pixy * Xx + piyy * Xy + tau2 * piyn * Xn);
    c.pixz = Zt * (pitt * Xt - pitx * Xx - pity * Xy - tau2 * pitn * Xn) - tau2 * Zn * (pitn * Xt - pixn * Xx - piyn * Xy - tau2 * pinn * Xn);
    c.piyy = pixx * Yx * Yx + 2.0 * pixy * Yx * Yy + piyy * Yy * Yy;
    c.piyz = -Zt * (pitx * Yx + pity * Yy) + tau2 * Zn * (pixn * Yx + piyn * Yy);
    c.pizz = -(c.pixx + c.piyy);
    // boost_Vmu_to_lrf (viscous_correction.cpp:161-173)
    c.Vx = -Vt * Xt + Vx * Xx + Vy * Xy + tau2 * Vn * Xn;
    c.Vy = Vx * Yx + Vy * Yy;
    c.Vz = -Vt * Zt + tau2 * Vn * Zn;
    c.tau = tau; c.x = p.x ? p.x[ic] : 0.0; c.y = p.y ? p.y[ic] : 0.0;
    c.eta = p.dim3 ? p.cells.eta[ic] : 0.0;
    c.ut = ut; c.ux = ux; c.uy = uy; c.un = un; c.T = T;
    // max_particle_number, df_mode 1 / 2: 2 n_eq per species (:282-303); total mean number of the cell (:1077)
    const double two_pi2_hbarC3 = 2.0 * M_PI * M_PI * (kHbarC * kHbarC * kHbarC);
    c.neq_fact = T * T * T / two_pi2_hbarC3;
    // modified temperature and rescaling coefficients (:1017-1036), detA and the breakdown test (:1038)
    c.T_mod = T;
    c.alphaB_mod = c.alphaB;
    if (p.df_mode == 3) {
        c.T_mod = T + bulkPi * c.F / c.betabulk; c.shear_mod = 0.5 / c.betapi; c.bulk_mod = bulkPi / (3.0 * c.betabulk);
        c.alphaB_mod = c.alphaB + bulkPi * c.G / c.betabulk;
        c.diff_mod = p.baryon ? T / c.betaV : 0.0;
    }
    else if (p.df_mode == 4) { c.shear_mod = 0.5 / c.betapi; c.bulk_mod = lambda; }
    if (p.df_mode == 3) {
        const double Axx = 1.0 + c.pixx * c.shear_mod + c.bulk_mod, Axy = c.pixy * c.shear_mod, Axz = c.pixz * c.shear_mod;
        const double Ayy = 1.0 + c.piyy * c.shear_mod + c.bulk_mod, Ayz = c.piyz * c.shear_mod, Azz = 1.0 + c.pizz * c.shear_mod + c.bulk_mod;
        const double detA = Axx * (Ayy * Azz - Ayz * Ayz) - Axy * (Axy * Azz - Ayz * Axz) + Axz * (Axy * Ayz - Ayy * Axz);
        double Tb = T, Fb = c.F, bbb = c.betabulk;
        if (p.fast) { Tb = p.T_sw; Fb = p.F_avg; bbb = p.betabulk_avg; }
        const double nf = Tb * Tb * Tb / two_pi2_hbarC3, Jf = Tb * nf, mbar_pion0 = p.mass_pion0 / Tb;
        const double neq_pion0 = nf * gt_neq(p.gl, p.gl + p.ngl, p.ngl, mbar_pion0, -1.0);
        const double J20_pion0 = Jf * gt_J20(p.gl + 2 * p.ngl, p.gl + 3 * p.ngl, p.ngl, mbar_pion0, -1.0);
        const double dn_pion0 = bulkPi * (neq_pion0 + J20_pion0 * Fb / Tb / Tb) / bbb;
        if (detA <= p.detA_min || (neq_pion0 + dn_pion0) < 0.0) { c.breakdown = 1.0; atomicAdd(&p.status[5], 1ULL); }
    }
    const double *gt = GT + ic, *gt2 = GT2 ? GT2 + ic : nullptr, *gt3 = GT3 ? GT3 + ic : nullptr;
    // the running sums are kept (species-major, so that the lanes of a wave -- consecutive cells -- store adjacent words): the sampling kernels
    // then find a hadron's species by bisection of exactly these sums instead of re-adding up to 305 gathered weights per hadron
    double dn = 0.0;
    if (p.cdf) {
        // (the stores never alias the tables read: said so, and the loop unrolled, so that four species' loads are in flight per trip)
        double *__restrict__ cdf = p.cdf + ic;
        const double *__restrict__ gtr = gt;
#pragma unroll 4
        for (int ip = 0; ip < sp.npart; ip++) {
            dn += species_dn(p, sp, c, gtr, gt2, gt3, ip);
            if ((ip & (kCdfBlock - 1)) == kCdfBlock - 1 || ip == sp.npart - 1) cdf[(int64_t)(ip / kCdfBlock) * p.n_cells] = dn;
        }
    } else {
        for (int ip = 0; ip < sp.npart; ip++) dn += species_dn(p, sp, c, gt, gt2, gt3, ip);
    }
    c.dn_sum = dn;
    c.dn_tot = dn * (2.0 * p.y_max * c.ds_max);
    c.live = (c.dn_tot > 0.0) ? 1.0 : 0.0;                                          // :1079
    out[ic] = c;
}

// :172-196
__device__ __forceinline__ double pion_thermal_weight_max(double x)
{
    const double x2 = x * x, x3 = x2 * x, x4 = x3 * x;
    const double max = (143206.88623164667 - 95956.76008684626 * x - 21341.937407169076 * x2 + 14388.446116867359 * x3 - 6083.775788504437 * x4) /
                       (-0.3541350577684533 + 143218.69233952634 * x - 24516.803600065778 * x2 - 115811.59391199696 * x3 + 35814.36403387459 * x4);
    return 1.00001 * max;
}

struct LrfMom { double E, px, py, pz; };

// sample_momentum (:456-617); chem = baryon * alpha_B enters the heavy-hadron weight only (:588)
__device__ LrfMom sample_momentum(Rng &g, long &acceptances, long &samples, double mass, double sign, double T, double chem)
{
    const double two_pi = 2.0 * M_PI;
    const double mbar = mass / T, mbar_squared = mbar * mbar;
    double pbar, Ebar, phi_over_2pi, costheta;
    if (mbar < 1.008) {
        double weq_max = 1.0;
        if (mbar < 0.8554 && sign == -1.0) weq_max = pion_thermal_weight_max(mbar);
        for (;;) {
            samples += 1;
            const double r1 = 1.0 - g.uniform(), r2 = 1.0 - g.uniform(), r3 = 1.0 - g.uniform();
            const double l1 = log(r1), l2 = log(r2), l3 = log(r3);
            const double l1_plus_l2 = l1 + l2;
            pbar = -(l1 + l2 + l3);
            Ebar = sqrt(pbar * pbar + mbar_squared);
            phi_over_2pi = l1_plus_l2 * l1_plus_l2 / (pbar * pbar);
            costheta = (l1 - l2) / l1_plus_l2;
            const double weight = 1.0 / (exp(Ebar) + sign) / weq_max / (r1 * r2 * r3);
            if (g.uniform() < weight) break;
        }
    } else {
        const double K0 = mbar_squared, K1 = 2.0 * mbar, K2 = 2.0, Ksum = K0 + K1 + K2;
        double kbar;
        for (;;) {
            samples += 1;
            const double uk = g.uniform() * Ksum;
            if (uk < K0) {
                kbar = -log(1.0 - g.uniform());
                phi_over_2pi = g.uniform();
                costheta = 2.0 * g.uniform() - 1.0;
            } else if (uk < K0 + K1) {
                const double l1 = log(1.0 - g.uniform()), l2 = log(1.0 - g.uniform());
                kbar = -(l1 + l2);
                phi_over_2pi = -l1 / kbar;
                costheta = 2.0 * g.uniform() - 1.0;
            } else {
                const double l1 = log(1.0 - g.uniform()), l2 = log(1.0 - g.uniform()), l3 = log(1.0 - g.uniform());
                const double l1_plus_l2 = l1 + l2;
                kbar = -(l1 + l2 + l3);
                phi_over_2pi = l1_plus_l2 * l1_plus_l2 / (kbar * kbar);
                costheta = (l1 - l2) / l1_plus_l2;
            }
            Ebar = kbar + mbar;
            pbar = sqrt(Ebar * Ebar - mbar_squared);
            const double exponent = exp(Ebar - chem);
            const double weight = pbar / Ebar * exponent / (exponent + sign);
            if (g.uniform() < weight) break;
        }
    }
    acceptances += 1;
    const double E = Ebar * T, pm = pbar * T, phi = phi_over_2pi * two_pi;
    const double sintheta = sqrt(1.0 - costheta * costheta);
    LrfMom q = {E, pm * sintheta * cos(phi), pm * sintheta * sin(phi), pm * costheta};
    return q;
}

// stream 0 of every (event, cell) pair of the batch: the Poisson number of hadrons (std::poisson_distribution(dn_tot), :1085-1090)
__global__ void __launch_bounds__(256)
cf_sampler_poisson(SamplerParams p, const SamplerCell *__restrict__ cellrec, int event0, int n_events, int32_t *__restrict__ n_drawn,
                   uint8_t *__restrict__ emits)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_events * p.n_cells) return;
    const int64_t ic = idx % p.n_cells;
    long N = 0;
    if (cellrec[ic].live != 0.0) {
        Rng g;
        g.init(p.seed, 0, (uint32_t)(p.first_cell + ic), (uint32_t)(event0 + (int)(idx / p.n_cells)));
        N = g.poisson(cellrec[ic].dn_tot);
    }
    n_drawn[idx] = (int32_t)(N < 0x7fffffffL ? N : 0x7fffffffL);
    emits[idx] = N > 0;
}

// one emitting (event, cell) pair; tallies = {momentum samples, acceptances, hadrons drawn} of this thread (count pass)
template <bool FILL>
__device__ __forceinline__ void sampler_thread(const SamplerParams &p, const SamplerSpecies &sp, const SamplerCell *__restrict__ cellrec,
                                               const double *__restrict__ GT, const double *__restrict__ GT2,
                                               const double *__restrict__ GT3, int event0, const int32_t *__restrict__ active, int64_t n_active, const int32_t *__restrict__ n_drawn,
                                               int64_t *__restrict__ counts, const int64_t *__restrict__ offsets, int64_t base,
                                               is3d_particle *__restrict__ particles, int64_t capacity, unsigned long long (&tally)[3])
{
    const int64_t ia = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // position in the list of emitting (event, cell) pairs
    if (ia >= n_active) return;
    if (FILL && offsets[ia + 1] == offsets[ia]) return;                    // pass 1 kept nothing here: nothing to replay
    const int64_t idx = active[ia];                                         // event-major: (event - event0) * n_cells + cell
    const int ievent = event0 + (int)(idx / p.n_cells);
    const int64_t ic = idx % p.n_cells;
    const SamplerCell &c = cellrec[ic];
    const uint32_t gcell = (uint32_t)(p.first_cell + ic);
    const long N_hadrons = n_drawn[idx];                                    // cf_sampler_poisson (stream 0)
    Rng g_type, g_momentum, g_keep, g_rapidity;
    g_type.init(p.seed, 1, gcell, (uint32_t)ievent);
    g_momentum.init(p.seed, 2, gcell, (uint32_t)ievent);
    g_keep.init(p.seed, 3, gcell, (uint32_t)ievent);
    g_rapidity.init(p.seed, 4, gcell, (uint32_t)ievent);
    const double *gt = GT + ic, *gt2 = GT2 ? GT2 + ic : nullptr, *gt3 = GT3 ? GT3 + ic : nullptr;
    const bool linear = p.df_mode <= 2 || c.breakdown != 0.0;
    const double sinheta = sinh(c.eta), cosheta = sqrt(1.0 + sinheta * sinheta);   // :888-889
    long kept = 0, samples = 0, acceptances = 0;
    int64_t slot = FILL ? base + offsets[ia] : 0;
    for (long ih = 0; ih < N_hadrons; ih++) {
        const double ut_ = g_type.uniform() * c.dn_sum;
        int chosen = sp.npart - 1;
        if (p.cdf) {
            // first species whose running sum exceeds ut_ (the last one if none does): bisection of the sums cf_sampler_cells stored -- the
            // weights are >= 0 for df_mode 1, 2, 4, so the sums are non-decreasing and this IS the linear inversion below, in 9 reads
            // (round 5) ... of the sums at the block ends: the first block whose end sum exceeds ut_ (the last block if none does), then the producer's
            // additions inside it, continued from the sum stored in front of it
            int lo = 0, hi = (sp.npart + kCdfBlock - 1) / kCdfBlock - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ut_ < p.cdf[(int64_t)mid * p.n_cells + ic]) hi = mid;
                else lo = mid + 1;
            }
            double cum = lo ? p.cdf[(int64_t)(lo - 1) * p.n_cells + ic] : 0.0;
            const int ip1 = min((lo + 1) * kCdfBlock, sp.npart);
            chosen = ip1 - 1;
            for (int ip = lo * kCdfBlock; ip < ip1; ip++) {
                cum += species_dn(p, sp, c, gt, gt2, gt3, ip);
                if (ut_ < cum) { chosen = ip; break; }
            }
        } else {
            double cum = 0.0;
            for (int ip = 0; ip < sp.npart; ip++) {
                cum += species_dn(p, sp, c, gt, gt2, gt3, ip);
                if (ut_ < cum) { chosen = ip; break; }
            }
        }
        const double mass = sp.mass[chosen], mass_squared = mass * mass, sign = sp.sign[chosen];
        const double baryon = sp.baryon ? sp.baryon[chosen] : 0.0;
        LrfMom q;
        double w_visc = 1.0;
        if (linear) {                                                               // :1100-1110, switch_to_linear_df
            const double chem = baryon * c.alphaB;
            q = sample_momentum(g_momentum, acceptances, samples, mass, sign, c.T, chem);
            // compute_df_weight (:361-453); df_mode 3 takes the Chapman-Enskog branch
            const double pimunu_pmu_pnu = q.px * q.px * c.pixx + q.py * q.py * c.piyy + q.pz * q.pz * c.pizz
                                        + 2.0 * (q.px * q.py * c.pixy + q.px * q.pz * c.pixz + q.py * q.pz * c.piyz);
            const double Vmu_pmu = -(q.px * c.Vx + q.py * c.Vy + q.pz * c.Vz);        // :384
            const double feqbar = 1.0 - sign / (exp(q.E / c.T - chem) + sign);
            double df_tot;
            if (p.df_mode == 1) {
                const double df_shear = pimunu_pmu_pnu / c.shear14;
                const double df_bulk = ((c.c0 - c.c2) * mass_squared + (baryon * c.c1 + (4.0 * c.c2 - c.c0) * q.E) * q.E) * c.bulkPi;
                const double df_diff = (baryon * c.c3 + c.c4 * q.E) * Vmu_pmu;
                df_tot = feqbar * (df_shear + df_bulk + df_diff);
            } else {
                const double betaV = p.baryon ? c.betaV : 1.0;
                const double df_shear = pimunu_pmu_pnu / (2.0 * q.E * c.betapi * c.T);
                const double df_bulk = (baryon * c.G + c.F * q.E / c.T / c.T + (q.E - mass_squared / q.E) / (3.0 * c.T)) * c.bulkPi / c.betabulk;
                const double df_diff = (c.ber - baryon / q.E) * Vmu_pmu / betaV;
                df_tot = feqbar * (df_shear + df_bulk + df_diff);
            }
            df_tot = fmax(-1.0, fmin(df_tot, 1.0));
            w_visc = (1.0 + df_tot) / 2.0;
        } else {                                                                    // :1112-1131 + rescale_momentum (:619-650)
            const bool mike = p.df_mode == 3;
            const LrfMom m = sample_momentum(g_momentum, acceptances, samples, mass, sign, c.T_mod, mike ? baryon * c.alphaB_mod : 0.0);
            const double diff_mod = c.diff_mod * (m.E * c.ber + (mike ? baryon : 0.0));   // :638
            q.px = (1.0 + c.bulk_mod) * m.px + c.shear_mod * (c.pixx * m.px + c.pixy * m.py + c.pixz * m.pz) + diff_mod * c.Vx;
            q.py = (1.0 + c.bulk_mod) * m.py + c.shear_mod * (c.pixy * m.px + c.piyy * m.py + c.piyz * m.pz) + diff_mod * c.Vy;
            q.pz = (1.0 + c.bulk_mod) * m.pz + c.shear_mod * (c.pixz * m.px + c.piyz * m.py + c.pizz * m.pz) + diff_mod * c.Vz;
            q.E = sqrt(mass_squared + q.px * q.px + q.py * q.py + q.pz * q.pz);
        }
        // boost_pLRF_to_lab_frame (emissionfunction.cpp:40-51)
        const double ptau = q.E * c.ut + q.px * c.Xt + q.pz * c.Zt;
        const double plx = q.E * c.ux + q.px * c.Xx + q.py * c.Yx;
        const double ply = q.E * c.uy + q.px * c.Xy + q.py * c.Yy;
        const double pn = q.E * c.un + q.px * c.Xn + q.pz * c.Zn;
        const double w_flux = fmax(0.0, q.E * c.dst - q.px * c.dsx - q.py * c.dsy - q.pz * c.dsz) / (q.E * c.ds_max);   // :1148
        if (!(g_keep.uniform() < (w_flux * w_visc))) continue;
        double Elab, pz, eta = c.eta, sh = sinheta, ch = cosheta;
        if (!p.dim3) {                                                              // :1168-1186
            const double yp = p.y_max * (2.0 * g_rapidity.uniform() - 1.0);
            const double sinhy = sinh(yp), coshy = sqrt(1.0 + sinhy * sinhy);
            const double tau_pn = c.tau * pn, mT = sqrt(mass_squared + plx * plx + ply * ply);
            sh = (ptau * sinhy - tau_pn * coshy) / mT;
            eta = asinh(sh);
            ch = sqrt(1.0 + sh * sh);
            pz = mT * sinhy;
            Elab = mT * coshy;
        } else {
            pz = c.tau * pn * ch + ptau * sh;
            Elab = sqrt(mass_squared + plx * plx + ply * ply + pz * pz);
        }
        if (FILL && slot < capacity) {
            is3d_particle o;
            o.cell = p.first_cell + ic; o.event = ievent; o.species = chosen;
            o.tau = c.tau; o.x = c.x; o.y = c.y; o.eta = eta; o.t = c.tau * ch; o.z = c.tau * sh;
            o.E = Elab; o.px = plx; o.py = ply; o.pz = pz;
            particles[slot] = o;
        }
        slot++;
        kept++;
    }
    if (!FILL) {
        counts[ia] = kept;
        tally[0] = (unsigned long long)samples; tally[1] = (unsigned long long)acceptances; tally[2] = (unsigned long long)N_hadrons;
    }
}

template <bool FILL>
__global__ void __launch_bounds__(128)
cf_sampler_run(SamplerParams p, SamplerSpecies sp, const SamplerCell *__restrict__ cellrec, const double *__restrict__ GT,
               const double *__restrict__ GT2, const double *__restrict__ GT3, int event0, const int32_t *__restrict__ active, int64_t n_active,
               const int32_t *__restrict__ n_drawn, int64_t *__restrict__ counts, const int64_t *__restrict__ offsets, int64_t base,
               is3d_particle *__restrict__ particles, int64_t capacity)
{
    unsigned long long tally[3] = {0ULL, 0ULL, 0ULL};
    sampler_thread<FILL>(p, sp, cellrec, GT, GT2, GT3, event0, active, n_active, n_drawn, counts, offsets, base, particles, capacity, tally);
    if (!FILL) {
        // the run-wide tallies: one global atomic per counter and workgroup instead of three per sampling thread
        __shared__ unsigned long long blk[3];
        if (threadIdx.x < 3) blk[threadIdx.x] = 0ULL;
        __syncthreads();
        for (int k = 0; k < 3; k++)
            if (tally[k]) atomicAdd(&blk[k], tally[k]);
        __syncthreads();
        if (threadIdx.x < 3 && blk[threadIdx.x]) atomicAdd(&p.status[2 + threadIdx.x], blk[threadIdx.x]);
    }
}

}  // namespace is3d

// ------------------------------------------------------------------------------------------------
// host entry
// ------------------------------------------------------------------------------------------------
namespace {

#define SMP_TRY(expr)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return is3d::set_error(IS3D_ENODEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevMem {
    void *p = nullptr;
    hipError_t alloc(size_t bytes)
    {
        release();
        if (!bytes) return hipSuccess;
        is3d::count_resource(1);
        return hipMalloc(&p, bytes);
    }
    template <class T>
    hipError_t upload(const std::vector<T> &h)
    {
        hipError_t e = alloc(h.size() * sizeof(T));
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    ~DevMem() { release(); }
    template <class T> T *as() const { return (T *)p; }
};

}  // namespace

struct is3d_sampler_plan {
    int device = 0;
    int64_t max_cells = 0;
    is3d_options o{};
    bool three_d = true;
    int ngla = 0, ncls = 0;
    is3d::SamplerParams p{};
    is3d::SamplerSpecies sp{};
    DevMem d_mass, d_sign, d_deg, d_cls, d_cmass, d_csign, d_gl, d_splx, d_sply[3], d_splc[3];
    DevMem d_bar, d_cbar, d_bilT, d_bilB, d_biltab[5];
    DevMem d_jonah, d_eqd, d_bkd;
    DevMem d_status, d_GT, d_GT2, d_GT3, d_rec, d_counts, d_offsets, d_scan_tmp;
    DevMem d_drawn, d_emits, d_active, d_nactive, d_cdf;
    int64_t cap_cells = 0, cap_bt = 0;      // what the workspaces above were sized for
    size_t tmp_bytes = 0;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

extern "C" int is3d_sampler_plan_create(is3d_sampler_plan **out, const is3d_species *species, const is3d_df_tables *df,
                                        const is3d_sampler_inputs *in, const is3d_options *opts, int64_t max_cells)
{
    using is3d::set_error;
    if (!out || !species || !df || !in || !opts) return set_error(IS3D_EINVAL, "null argument");
    *out = nullptr;
    if (max_cells < 1) max_cells = 1;
    if (opts->dimension != 2 && opts->dimension != 3) return set_error(IS3D_EINVAL, "dimension must be 2 or 3 (got %d)", opts->dimension);
    if (opts->df_mode < 1 || opts->df_mode > 4) return set_error(IS3D_EINVAL, "the sampler takes df_mode 1, 2, 3 or 4 (got %d)", opts->df_mode);
    const is3d_feqmod_tables *fq = in->feqmod;
    const bool need_alpha2 = opts->df_mode == 3 || (in->fast && opts->df_mode == 2);
    if ((opts->df_mode >= 3 || need_alpha2) && !fq) return set_error(IS3D_EINVAL, "df_mode 3 / 4 (and fast mode with df_mode 2) need in->feqmod");
    if (fq && (fq->n_gla != in->n_gla || !fq->root2 || !fq->weight2)) return set_error(IS3D_EINVAL, "in->feqmod: Gauss-Laguerre alpha = 2 nodes missing or of another size");
    if (opts->df_mode == 4 && (fq->n_pdg < 1 || !fq->pdg_mass || !fq->pdg_degeneracy || !fq->pdg_sign || !(fq->T_avg > 0.0)))
        return set_error(IS3D_EINVAL, "df_mode 4 needs the full PDG list and the surface-averaged temperature");
    if (in->fast && !(in->T_avg > 0.0)) return set_error(IS3D_EINVAL, "fast = 1 needs the surface-averaged temperature");
    const bool baryon = opts->include_baryon != 0, baryondiff = baryon && opts->include_baryondiff_deltaf != 0;
    if (baryon) {
        if (opts->df_mode == 4)   // deltafReader.cpp:470-474: "Jonah df doesn't work for nonzero muB. Exiting.."
            return set_error(IS3D_EINVAL, "df_mode 4 does not work with include_baryon = 1 (the reference exits there too)");
        if (!species->baryon) return set_error(IS3D_EINVAL, "include_baryon = 1 needs the species' baryon numbers");
        if (df->n_muB < 2 || !df->muB) return set_error(IS3D_EINVAL, "include_baryon = 1 needs the full (T, muB) coefficient tables");
        for (int i = 1; i < df->n_muB; i++)
            if (!(df->muB[i] > df->muB[i - 1])) return set_error(IS3D_EINVAL, "coefficient table muB values must ascend");
        if (opts->df_mode == 1 && (!df->c0 || !df->c1 || !df->c2 || !df->c3 || !df->c4)) return set_error(IS3D_EINVAL, "include_baryon = 1, df_mode 1 needs c0..c4 tables");
        if (opts->df_mode != 1 && (!df->F || !df->G || !df->betabulk || !df->betaV || !df->betapi))
            return set_error(IS3D_EINVAL, "include_baryon = 1, df_mode 2 / 3 need F, G, betabulk, betaV, betapi tables");
    }
    if (species->n < 1 || !species->mass || !species->sign || !species->degeneracy) return set_error(IS3D_EINVAL, "empty species list");
    for (int s = 0; s < species->n; s++)
        if (!(species->mass[s] > 0.0)) return set_error(IS3D_EINVAL, "species %d has mass 0: photons cannot be sampled with this method (reference: exit, sampling_kernels.cpp:478-482)", s);
    if (in->n_gla < 1 || in->n_gla > is3d::kSmpGlMax || !in->root1 || !in->weight1)
        return set_error(IS3D_EINVAL, "the sampler needs the Gauss-Laguerre roots and weights for alpha = 1 (1 to %d nodes)", is3d::kSmpGlMax);
    if (df->n_T < 3 || !df->T) return set_error(IS3D_EINVAL, "coefficient table needs >= 3 temperatures");
    if (opts->df_mode == 1 && (!df->c0 || !df->c2)) return set_error(IS3D_EINVAL, "df_mode 1 needs c0 and c2 tables");
    if ((opts->df_mode == 2 || opts->df_mode == 3) && (!df->F || !df->betabulk || !df->betapi))
        return set_error(IS3D_EINVAL, "df_mode 2 / 3 need F, betabulk, betapi tables");
    if (opts->df_mode == 4 && !df->betapi) return set_error(IS3D_EINVAL, "df_mode 4 needs the betapi table");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");
    std::unique_ptr<is3d_sampler_plan> P(new is3d_sampler_plan);
    is3d::count_resource(0);
    if (opts->device >= 0) SMP_TRY(hipSetDevice(opts->device));
    SMP_TRY(hipGetDevice(&P->device));
    P->max_cells = max_cells;
    P->o = *opts;
    P->three_d = opts->dimension == 3;
    P->ngla = in->n_gla;
    const bool three_d = P->three_d;
    auto &d_mass = P->d_mass; auto &d_sign = P->d_sign; auto &d_deg = P->d_deg; auto &d_cls = P->d_cls; auto &d_cmass = P->d_cmass; auto &d_csign = P->d_csign;
    auto &d_gl = P->d_gl; auto &d_splx = P->d_splx; auto &d_sply = P->d_sply; auto &d_splc = P->d_splc;
    auto &d_bar = P->d_bar; auto &d_cbar = P->d_cbar; auto &d_bilT = P->d_bilT; auto &d_bilB = P->d_bilB; auto &d_biltab = P->d_biltab;
    auto &d_jonah = P->d_jonah; auto &d_eqd = P->d_eqd; auto &d_bkd = P->d_bkd;
    is3d::SamplerParams &p = P->p;
    is3d::SamplerSpecies &sp = P->sp;

    // ---- species classes (mass, sign[, baryon number]): the density integral is per class ----
    const int npart = species->n;
    std::vector<int32_t> cls(npart);
    std::vector<double> cmass, csign, cbar;
    for (int s = 0; s < npart; s++) {
        int found = -1;
        const double bs = baryon ? species->baryon[s] : 0.0;
        for (size_t c = 0; c < cmass.size(); c++)
            if (cmass[c] == species->mass[s] && csign[c] == species->sign[s] && cbar[c] == bs) { found = (int)c; break; }
        if (found < 0) { found = (int)cmass.size(); cmass.push_back(species->mass[s]); csign.push_back(species->sign[s]); cbar.push_back(bs); }
        cls[s] = found;
    }
    const int ncls = (int)cmass.size();
    SMP_TRY(d_mass.upload(std::vector<double>(species->mass, species->mass + npart)));
    SMP_TRY(d_sign.upload(std::vector<double>(species->sign, species->sign + npart)));
    SMP_TRY(d_deg.upload(std::vector<double>(species->degeneracy, species->degeneracy + npart)));
    SMP_TRY(d_cls.upload(cls));
    SMP_TRY(d_cmass.upload(cmass));
    SMP_TRY(d_csign.upload(csign));
    std::vector<double> gl((size_t)4 * in->n_gla, 0.0);
    for (int k = 0; k < in->n_gla; k++) {
        gl[k] = in->root1[k]; gl[in->n_gla + k] = in->weight1[k];
        if (fq) { gl[2 * in->n_gla + k] = fq->root2[k]; gl[3 * in->n_gla + k] = fq->weight2[k]; }
    }
    SMP_TRY(d_gl.upload(gl));
    if (baryon) {
        SMP_TRY(d_bar.upload(std::vector<double>(species->baryon, species->baryon + npart)));
        SMP_TRY(d_cbar.upload(cbar));
    }
    sp = is3d::SamplerSpecies{d_mass.as<double>(), d_sign.as<double>(), d_deg.as<double>(), d_cls.as<int32_t>(),
                              d_cmass.as<double>(), d_csign.as<double>(), npart, ncls, d_bar.as<double>(), d_cbar.as<double>()};
    P->ncls = ncls;
    // ---- splines (deltafReader.cpp:300-322) ----
    std::vector<double> xs(df->T, df->T + df->n_T);
    for (int i = 1; i < df->n_T; i++)
        if (!(xs[i] > xs[i - 1])) return set_error(IS3D_EINVAL, "coefficient table temperatures must ascend");
    SMP_TRY(d_splx.upload(xs));
    const double *tabs[3] = {nullptr, nullptr, nullptr};
    int nspl;
    if (opts->df_mode == 1) { tabs[0] = df->c0; tabs[1] = df->c2; nspl = 2; }
    else if (opts->df_mode == 4) { tabs[0] = tabs[1] = tabs[2] = df->betapi; nspl = 3; }   // only slot 2 (betapi) is read
    else { tabs[0] = df->F; tabs[1] = df->betabulk; tabs[2] = df->betapi; nspl = 3; }
    p.spl.n = df->n_T; p.spl.x = d_splx.as<double>(); p.spl.nspl = nspl;
    for (int s = 0; s < nspl; s++) {
        std::vector<double> ys(tabs[s], tabs[s] + df->n_T), cc;
        if (!is3d::natural_cspline_init(xs, ys, cc)) return set_error(IS3D_EINVAL, "spline construction failed");
        SMP_TRY(d_sply[s].upload(ys));
        SMP_TRY(d_splc[s].upload(cc));
        p.spl.y[s] = d_sply[s].as<double>();
        p.spl.c[s] = d_splc[s].as<double>();
    }
    if (baryon) {   // full (mu_B, T) grids for the bilinear branch (deltafReader.cpp:412-484)
        const double *t5[5];
        if (opts->df_mode == 1) { t5[0] = df->c0; t5[1] = df->c1; t5[2] = df->c2; t5[3] = df->c3; t5[4] = df->c4; }
        else { t5[0] = df->F; t5[1] = df->G; t5[2] = df->betabulk; t5[3] = df->betaV; t5[4] = df->betapi; }
        SMP_TRY(d_bilT.upload(xs));
        SMP_TRY(d_bilB.upload(std::vector<double>(df->muB, df->muB + df->n_muB)));
        p.bil.nT = df->n_T; p.bil.nB = df->n_muB;
        p.bil.swap = opts->reference_bilinear_indexing != 0;
        p.bil.T = d_bilT.as<double>(); p.bil.muB = d_bilB.as<double>();
        for (int k = 0; k < 5; k++) {
            SMP_TRY(d_biltab[k].upload(std::vector<double>(t5[k], t5[k] + (size_t)df->n_T * df->n_muB)));
            p.bil.tab[k] = d_biltab[k].as<double>();
        }
    }
    p.baryon = baryon; p.baryondiff = baryondiff;
    p.dim3 = three_d; p.df_mode = opts->df_mode;
    p.include_bulk = opts->include_bulk_deltaf != 0; p.include_shear = opts->include_shear_deltaf != 0;
    p.ngl = in->n_gla; p.gl = d_gl.as<double>();
    // ---- df_mode 4: lambda(Pi/P), z(Pi/P) tables (deltafReader.cpp:222-297) ----
    if (opts->df_mode == 4) {
        std::vector<double> bp, l2, zz, cl, cz, jon;
        is3d::jonah_tables(fq, bp, l2, zz, p.bp_max);
        if (!is3d::natural_cspline_init(bp, l2, cl) || !is3d::natural_cspline_init(bp, zz, cz))
            return set_error(IS3D_EINVAL, "df_mode 4: bulkPi/Peq(lambda) is not ascending at T_avg = %.6g GeV (GSL would abort here)", fq->T_avg);
        p.nj = (int)bp.size();
        for (const auto *v : {&bp, &l2, &zz, &cl, &cz}) jon.insert(jon.end(), v->begin(), v->end());
        SMP_TRY(d_jonah.upload(jon));
        p.jx = d_jonah.as<double>(); p.jl2 = p.jx + p.nj; p.jz = p.jx + 2 * p.nj; p.jcl = p.jx + 3 * p.nj; p.jcz = p.jx + 4 * p.nj;
    }
    if (fq) { p.detA_min = fq->deta_min; p.mass_pion0 = fq->mass_pion0; }
    // ---- fast mode: Deltaf_Data::compute_particle_densities at the average temperature (deltafReader.cpp:536-650) ----
    p.fast = in->fast != 0;
    if (p.fast) {
        const double two_pi2_hbarC3 = 2.0 * std::pow(M_PI, 2) * std::pow(is3d::kHbarC, 3);
        auto spline_at = [&](int slot, double Tq) {   // gsl_spline_eval on the host copies of the spline slot
            std::vector<double> ys(tabs[slot], tabs[slot] + df->n_T), cc;
            (void)is3d::natural_cspline_init(xs, ys, cc);
            int lo = 0, hi = df->n_T - 1;
            while (hi > lo + 1) { int i = (hi + lo) >> 1; if (xs[i] > Tq) hi = i; else lo = i; }
            const double dx = xs[lo + 1] - xs[lo], dy = ys[lo + 1] - ys[lo], delx = Tq - xs[lo];
            const double b_i = (dy / dx) - dx * (cc[lo + 1] + 2.0 * cc[lo]) / 3.0, d_i = (cc[lo + 1] - cc[lo]) / (3.0 * dx);
            return ys[lo] + delx * (b_i + delx * (cc[lo] + delx * d_i));
        };
        auto in_table = [&](double Tq) { return Tq >= xs.front() && Tq <= xs.back(); };
        const double T = in->T_avg, Tsw = in->T_avg_switch > 0.0 ? in->T_avg_switch : in->T_avg;
        if (!in_table(T) || !in_table(Tsw)) return set_error(IS3D_EDOMAIN, "fast = 1: the average temperature %.6g GeV is outside the coefficient table", T);
        // include_baryon: Deltaf_Data::bilinear_interpolation at (T, muB_avg) on the host (deltafReader.cpp:412-484; cf_math.h::bilinear5 on the
        // host tables, same indexing option as the kernels)
        auto bilinear_at = [&](double Tq, double Bq, double (&v)[5]) -> bool {
            is3d::BilinearDev hb{};
            hb.nT = df->n_T; hb.nB = df->n_muB; hb.T = df->T; hb.muB = df->muB;
            hb.tab[0] = df->F; hb.tab[1] = df->G; hb.tab[2] = df->betabulk; hb.tab[3] = df->betaV; hb.tab[4] = df->betapi;
            hb.swap = opts->reference_bilinear_indexing != 0;
            return is3d::bilinear5(hb, Tq, Bq, v);
        };
        const double muB_avg = baryon ? in->muB_avg : 0.0, alphaB_avg = muB_avg / T;   // deltafReader.cpp:545-551
        double F = 0.0, G = 0.0, betabulk = 1.0;
        if (baryon && opts->df_mode != 1) {
            double bl[5];
            if (!bilinear_at(T, muB_avg, bl)) return set_error(IS3D_EDOMAIN, "fast = 1: the average (T, muB) = (%.6g, %.6g) GeV is outside the coefficient table", T, muB_avg);
            F = bl[0] * T; G = bl[1]; betabulk = bl[2] * (T * T * T * T);
        } else if (baryon) {   // df_mode 1: only the table range is checked (the 14-moment coefficients do not enter the fast densities used here)
            const double dB = std::fabs(df->muB[1] - df->muB[0]);
            const int iBL = (int)std::floor((muB_avg - df->muB[0]) / dB);
            if (!(iBL >= 0 && iBL + 1 < df->n_muB)) return set_error(IS3D_EDOMAIN, "fast = 1: the average muB = %.6g GeV is outside the coefficient table", muB_avg);
        } else if (opts->df_mode == 2 || opts->df_mode == 3) { F = spline_at(0, T) * T; betabulk = spline_at(1, T) * T * T * T * T; }
        std::vector<double> eqd(npart, 0.0), bkd(npart, 0.0);
        for (int ip = 0; ip < npart; ip++) {
            const double mbar = species->mass[ip] / T, sign = species->sign[ip];
            const double bnum = baryon ? species->baryon[ip] : 0.0, chem = bnum * alphaB_avg;
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int k = 0; k < in->n_gla; k++) {
                const double pbar = in->root1[k], Ebar = std::sqrt(pbar * pbar + mbar * mbar);
                s1 += in->weight1[k] * (pbar * std::exp(pbar) / (std::exp(Ebar - chem) + sign));
            }
            const double neq = species->degeneracy[ip] * std::pow(T, 3) / two_pi2_hbarC3 * s1;
            eqd[ip] = neq;
            if (opts->df_mode == 2 || opts->df_mode == 3) {
                for (int k = 0; k < in->n_gla; k++) {
                    const double pbar = fq->root2[k], Ebar = std::sqrt(pbar * pbar + mbar * mbar), qstat = std::exp(Ebar - chem) + sign;
                    s2 += fq->weight2[k] * (Ebar * std::exp(pbar + Ebar - chem) / (qstat * qstat));
                }
                for (int k = 0; k < in->n_gla; k++) {
                    const double pbar = in->root1[k], Ebar = std::sqrt(pbar * pbar + mbar * mbar), qstat = std::exp(Ebar - chem) + sign;
                    s3 += in->weight1[k] * (pbar * std::exp(pbar + Ebar - chem) / (qstat * qstat));
                }
                const double J10 = species->degeneracy[ip] * std::pow(T, 3) / two_pi2_hbarC3 * s3;
                const double J20 = species->degeneracy[ip] * std::pow(T, 4) / two_pi2_hbarC3 * s2;
                bkd[ip] = (neq + (bnum * J10 * G) + (J20 * F / std::pow(T, 2))) / betabulk;
            }
        }
        SMP_TRY(d_eqd.upload(eqd));
        SMP_TRY(d_bkd.upload(bkd));
        p.eqd = d_eqd.as<double>(); p.bkd = d_bkd.as<double>();
        p.T_sw = Tsw;
        if (opts->df_mode == 3 && baryon) {                                            // :862-867 at (Tavg with T_switch, muBavg)
            double bl[5];
            if (!bilinear_at(Tsw, muB_avg, bl)) return set_error(IS3D_EDOMAIN, "fast = 1: (T_switch, muB_avg) = (%.6g, %.6g) GeV is outside the coefficient table", Tsw, muB_avg);
            p.F_avg = bl[0] * Tsw; p.betabulk_avg = bl[2] * (Tsw * Tsw * Tsw * Tsw);
        } else if (opts->df_mode == 3) { p.F_avg = spline_at(0, Tsw) * Tsw; p.betabulk_avg = spline_at(1, Tsw) * Tsw * Tsw * Tsw * Tsw; }
    }
    p.y_max = three_d ? 0.5 : in->y_cut;                                              // :837-838
    SMP_TRY(P->d_status.alloc(8 * sizeof(unsigned long long)));
    p.status = P->d_status.as<unsigned long long>();
    for (auto &e : P->ev) SMP_TRY(hipEventCreate(&e));
    *out = P.release();
    return IS3D_OK;
}

namespace {
// cell-array checks shared by the host and the device entry (pointers are only tested for NULL here)
int check_cells(const is3d_cells *cells, const is3d_options *o, int64_t first_cell)
{
    using is3d::set_error;
    const int64_t n = cells->n_cells;
    if (n < 0 || n + first_cell > 0xffffffffLL) return set_error(IS3D_EINVAL, "cell indices must fit 32 bits for the counter-based streams");
    const bool three_d = o->dimension == 3;
    const bool baryondiff = o->include_baryon != 0 && o->include_baryondiff_deltaf != 0;
    if (n > 0) {
        if (!cells->tau || !cells->dat || !cells->dax || !cells->day || !cells->dan || !cells->ux || !cells->uy || !cells->un ||
            !cells->T || !cells->P || !cells->E || (three_d && !cells->eta))
            return set_error(IS3D_EINVAL, "a required cell array is NULL");
        if (o->include_shear_deltaf && (!cells->pixx || !cells->pixy || !cells->pixn || !cells->piyy || !cells->piyn))
            return set_error(IS3D_EINVAL, "include_shear_deltaf needs pixx, pixy, pixn, piyy, piyn");
        if (o->include_bulk_deltaf && !cells->bulkPi) return set_error(IS3D_EINVAL, "include_bulk_deltaf needs bulkPi");
        if (baryondiff && (!cells->muB || !cells->nB || !cells->Vx || !cells->Vy || !cells->Vn))
            return set_error(IS3D_EINVAL, "include_baryon && include_baryondiff_deltaf need muB, nB, Vx, Vy, Vn");
    }
    return IS3D_OK;
}
// which of the 23 cell arrays (CellPtrs order) the kernels read under these options
bool cell_array_needed(int a, const is3d_options *o)
{
    const bool three_d = o->dimension == 3;
    const bool baryondiff = o->include_baryon != 0 && o->include_baryondiff_deltaf != 0;
    return a < 12 ? (a != 1 || three_d) : (a < 17 ? o->include_shear_deltaf != 0 : (a == 17 ? o->include_bulk_deltaf != 0 : baryondiff));
}
}  // namespace

extern "C" void is3d_sampler_plan_destroy(is3d_sampler_plan *P)
{
    if (!P) return;
    (void)hipSetDevice(P->device);
    for (auto &e : P->ev)
        if (e) (void)hipEventDestroy(e);
    delete P;
}

// cells_dev: DEVICE arrays; x_dev, y_dev: DEVICE arrays or NULL; particles_dev: DEVICE buffer of `capacity` entries or NULL (count only)
extern "C" int is3d_sampler_plan_execute(is3d_sampler_plan *P, const is3d_cells *cells, const double *x_dev, const double *y_dev, int32_t n_events,
                                         uint64_t seed, int64_t first_cell, int32_t batch_events, is3d_particle *particles_dev, int64_t capacity,
                                         int64_t *n_particles, is3d_sampler_stats *stats)
{
    using is3d::set_error;
    if (!P || !cells || !n_particles) return set_error(IS3D_EINVAL, "null argument");
    *n_particles = 0;
    if (stats) memset(stats, 0, sizeof *stats);
    if (n_events < 1) return set_error(IS3D_EINVAL, "n_events must be >= 1");
    if (particles_dev == nullptr) capacity = 0;
    if (int rc = check_cells(cells, &P->o, first_cell)) return rc;
    const int64_t n = cells->n_cells;
    if (n > P->max_cells) return set_error(IS3D_EINVAL, "%lld cells, the sampler plan was created for %lld", (long long)n, (long long)P->max_cells);
    SMP_TRY(hipSetDevice(P->device));
    if (n == 0) return IS3D_OK;
    const int ncls = P->ncls;
    is3d::SamplerParams &p = P->p;
    const is3d::SamplerSpecies &sp = P->sp;
    const double *src[23] = {cells->tau, cells->eta, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un,
                             cells->T, cells->P, cells->E, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn, cells->bulkPi,
                             cells->muB, cells->nB, cells->Vx, cells->Vy, cells->Vn};
    const double *dptr[23];
    for (int a = 0; a < 23; a++) dptr[a] = (src[a] && cell_array_needed(a, &P->o)) ? src[a] : nullptr;
    p.cells = {dptr[0], dptr[1], dptr[2], dptr[3], dptr[4], dptr[5], dptr[6], dptr[7], dptr[8], dptr[9], dptr[10], dptr[11],
               dptr[12], dptr[13], dptr[14], dptr[15], dptr[16], dptr[17], dptr[18], dptr[19], dptr[20], dptr[21], dptr[22]};
    p.x = x_dev; p.y = y_dev;
    p.n_cells = n; p.first_cell = first_cell;
    p.seed = seed;
    p.cdf = nullptr;           // set below, once the workspaces exist
    hipEvent_t *ev = P->ev;
    unsigned long long init[8] = {~0ULL, 0, 0, 0, 0, 0, 0, 0};
    SMP_TRY(hipMemcpyAsync(P->d_status.p, init, sizeof init, hipMemcpyHostToDevice, nullptr));
    // ---- workspaces: sized by the largest (cells, event batch) seen so far; a second execute of the same shape allocates nothing ----
    const int64_t max_threads = (int64_t)1 << 25;
    int eb = (int)std::max<int64_t>(1, std::min<int64_t>(n_events, max_threads / n));
    if (batch_events > 0) eb = std::min(eb, batch_events);
    const int64_t bt = (int64_t)eb * n;
    if (n > P->cap_cells) {
        SMP_TRY(P->d_GT.alloc((size_t)n * ncls * sizeof(double)));
        if (P->o.df_mode == 3 && !p.fast) SMP_TRY(P->d_GT2.alloc((size_t)n * ncls * sizeof(double)));
        if (P->o.df_mode == 3 && p.baryon) SMP_TRY(P->d_GT3.alloc((size_t)n * ncls * sizeof(double)));
        SMP_TRY(P->d_rec.alloc((size_t)n * sizeof(is3d::SamplerCell)));
        // running sums of the species weights per cell, one per block of 8 species (312 B per cell for 305 species): df_mode 3's weights may be negative (n_eq + Pi dn_bulk),
        // its sums are not monotone and the species is found by the linear inversion there
        if (P->o.df_mode != 3) SMP_TRY(P->d_cdf.alloc((size_t)n * ((sp.npart + is3d::kCdfBlock - 1) / is3d::kCdfBlock) * sizeof(double)));
        P->cap_cells = n;
    }
    if (bt > P->cap_bt) {
        SMP_TRY(P->d_drawn.alloc((size_t)bt * sizeof(int32_t)));
        SMP_TRY(P->d_emits.alloc((size_t)bt));
        SMP_TRY(P->d_active.alloc((size_t)bt * sizeof(int32_t)));
        SMP_TRY(P->d_nactive.alloc(sizeof(int32_t)));
        SMP_TRY(P->d_counts.alloc((size_t)(bt + 1) * sizeof(int64_t)));
        SMP_TRY(P->d_offsets.alloc((size_t)(bt + 1) * sizeof(int64_t)));
        size_t tb = 0, tmp2 = 0;
        SMP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, P->d_counts.as<int64_t>(), P->d_offsets.as<int64_t>(), (int)(bt + 1), nullptr));
        SMP_TRY(hipcub::DeviceSelect::Flagged(nullptr, tmp2, hipcub::CountingInputIterator<int32_t>(0), P->d_emits.as<uint8_t>(), P->d_active.as<int32_t>(),
                                              P->d_nactive.as<int32_t>(), (int)bt, nullptr));
        P->tmp_bytes = std::max(tb, tmp2);
        SMP_TRY(P->d_scan_tmp.alloc(P->tmp_bytes));
        P->cap_bt = bt;
    }
    p.cdf = P->d_cdf.as<double>();
    DevMem &d_GT = P->d_GT, &d_GT2 = P->d_GT2, &d_GT3 = P->d_GT3, &d_rec = P->d_rec, &d_counts = P->d_counts, &d_offsets = P->d_offsets, &d_scan_tmp = P->d_scan_tmp;
    DevMem &d_drawn = P->d_drawn, &d_emits = P->d_emits, &d_active = P->d_active, &d_nactive = P->d_nactive;
    size_t tmp_bytes = P->tmp_bytes;
    SMP_TRY(hipEventRecord(ev[1], nullptr));
    {
        const int64_t tot = n * ncls;
        hipLaunchKernelGGL(is3d::cf_sampler_density, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, nullptr, dptr[9], dptr[18], n, sp,
                           P->d_gl.as<double>(), P->ngla, d_GT.as<double>(), d_GT2.as<double>(), d_GT3.as<double>());
        SMP_TRY(hipEventRecord(ev[6], nullptr));
        hipLaunchKernelGGL(is3d::cf_sampler_cells, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, nullptr, p, sp, d_GT.as<double>(),
                           d_GT2.as<double>(), d_GT3.as<double>(), d_rec.as<is3d::SamplerCell>());
        SMP_TRY(hipGetLastError());
    }
    SMP_TRY(hipEventRecord(ev[2], nullptr));
    // ---- events in batches of <= 2^25 (event, cell) threads: count, scan, fill ----
    int64_t base = 0;
    double ms_count = 0.0, ms_fill = 0.0, ms_poisson = 0.0;
    for (int e0 = 0; e0 < n_events; e0 += eb) {
        const int ne = std::min(eb, n_events - e0);
        const int64_t nt = (int64_t)ne * n;
        SMP_TRY(hipEventRecord(ev[5], nullptr));
        // Poisson numbers of all pairs, then the ordered list of the pairs that emit
        hipLaunchKernelGGL(is3d::cf_sampler_poisson, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, nullptr, p, d_rec.as<is3d::SamplerCell>(), e0, ne,
                           d_drawn.as<int32_t>(), d_emits.as<uint8_t>());
        SMP_TRY(hipGetLastError());
        SMP_TRY(hipcub::DeviceSelect::Flagged(d_scan_tmp.p, tmp_bytes, hipcub::CountingInputIterator<int32_t>(0), d_emits.as<uint8_t>(),
                                              d_active.as<int32_t>(), d_nactive.as<int32_t>(), (int)nt, nullptr));
        SMP_TRY(hipEventRecord(ev[7], nullptr));
        int32_t n_active = 0;
        SMP_TRY(hipMemcpy(&n_active, d_nactive.p, sizeof(int32_t), hipMemcpyDeviceToHost));
        int64_t batch_total = 0;
        if (n_active > 0) {
            const unsigned grid = (unsigned)(((int64_t)n_active + 127) / 128);
            SMP_TRY(hipMemsetAsync(d_counts.as<int64_t>() + n_active, 0, sizeof(int64_t), nullptr));
            hipLaunchKernelGGL((is3d::cf_sampler_run<false>), dim3(grid), dim3(128), 0, nullptr, p, sp, d_rec.as<is3d::SamplerCell>(),
                               d_GT.as<double>(), d_GT2.as<double>(), d_GT3.as<double>(), e0, d_active.as<int32_t>(), (int64_t)n_active,
                               d_drawn.as<int32_t>(), d_counts.as<int64_t>(), (const int64_t *)nullptr, (int64_t)0, (is3d_particle *)nullptr, (int64_t)0);
            SMP_TRY(hipGetLastError());
            // element n_active of the scan (counts[n_active] = 0) is the batch total
            SMP_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp.p, tmp_bytes, d_counts.as<int64_t>(), d_offsets.as<int64_t>(), n_active + 1, nullptr));
            SMP_TRY(hipEventRecord(ev[3], nullptr));
            SMP_TRY(hipMemcpy(&batch_total, d_offsets.as<int64_t>() + n_active, sizeof(int64_t), hipMemcpyDeviceToHost));
            if (capacity > 0 && base < capacity && batch_total > 0) {
                hipLaunchKernelGGL((is3d::cf_sampler_run<true>), dim3(grid), dim3(128), 0, nullptr, p, sp, d_rec.as<is3d::SamplerCell>(),
                                   d_GT.as<double>(), d_GT2.as<double>(), d_GT3.as<double>(), e0, d_active.as<int32_t>(), (int64_t)n_active,
                                   d_drawn.as<int32_t>(), (int64_t *)nullptr, d_offsets.as<int64_t>(), base, particles_dev, capacity);
                SMP_TRY(hipGetLastError());
            }
        } else {
            SMP_TRY(hipEventRecord(ev[3], nullptr));
        }
        SMP_TRY(hipEventRecord(ev[4], nullptr));
        SMP_TRY(hipEventSynchronize(ev[4]));
        float a = 0, b = 0, c = 0;
        SMP_TRY(hipEventElapsedTime(&a, ev[5], ev[3]));
        SMP_TRY(hipEventElapsedTime(&b, ev[3], ev[4]));
        SMP_TRY(hipEventElapsedTime(&c, ev[5], ev[7]));
        ms_count += a; ms_fill += b; ms_poisson += c;
        base += batch_total;
    }
    unsigned long long h[8];
    SMP_TRY(hipMemcpy(h, P->d_status.p, sizeof h, hipMemcpyDeviceToHost));
    *n_particles = base;
    if (stats) {
        float b = 0, dn = 0;
        (void)hipEventElapsedTime(&b, ev[1], ev[2]);
        (void)hipEventElapsedTime(&dn, ev[1], ev[6]);
        stats->ms_h2d = 0.0;
        stats->ms_prep = b;
        stats->ms_density = dn;
        stats->ms_poisson = ms_poisson;
        stats->n_cells_skipped = (int64_t)h[1];
        stats->n_momentum_samples = (int64_t)h[2];
        stats->n_acceptances = (int64_t)h[3];
        stats->n_hadrons_drawn = (int64_t)h[4];
        stats->n_cells_breakdown = (int64_t)h[5];
        stats->n_classes = ncls;
        stats->ms_count = ms_count; stats->ms_fill = ms_fill;
    }
    if (h[0] != ~0ULL)
        return set_error(IS3D_EDOMAIN, "cell %lld: T%s outside the coefficient table (the reference aborts in gsl_spline_eval here)", (long long)h[0],
                         P->o.df_mode == 4 ? " (or bulkPi/P)" : (p.baryon ? " or (T, muB)" : ""));
    if (particles_dev && base > capacity)
        return set_error(IS3D_ENOMEM, "%lld particles sampled but the caller's buffer holds %lld (call with particles = NULL for the count)",
                         (long long)base, (long long)capacity);
    return IS3D_OK;
}

// the host-pointer entry: plan + upload + execute + download (the particle list is the plan's, bit for bit)
extern "C" int is3d_sample_particles(const is3d_cells *cells, const is3d_species *species, const is3d_df_tables *df,
                                     const is3d_sampler_inputs *in, const is3d_options *opts, is3d_particle *particles,
                                     int64_t capacity, int64_t *n_particles, is3d_sampler_stats *stats)
{
    using is3d::set_error;
    if (!cells || !species || !df || !in || !opts || !n_particles) return set_error(IS3D_EINVAL, "null argument");
    *n_particles = 0;
    if (stats) memset(stats, 0, sizeof *stats);
    if (in->n_events < 1) {
        // (argument checks of the plan first, so that a bad option is reported before a bad event count -- as the one-shot entry always did)
        is3d_sampler_plan *chk = nullptr;
        const int rc0 = is3d_sampler_plan_create(&chk, species, df, in, opts, 1);
        is3d_sampler_plan_destroy(chk);
        if (rc0 && rc0 != IS3D_ENODEVICE) return rc0;
        return set_error(IS3D_EINVAL, "n_events must be >= 1");
    }
    if (particles == nullptr) capacity = 0;
    is3d_sampler_plan *P = nullptr;
    if (int rc = is3d_sampler_plan_create(&P, species, df, in, opts, std::max<int64_t>(cells->n_cells, 1))) return rc;
    struct Guard { is3d_sampler_plan *p; ~Guard() { is3d_sampler_plan_destroy(p); } } guard{P};
    if (int rc = check_cells(cells, opts, in->first_cell)) return rc;
    const int64_t n = cells->n_cells;
    if (n == 0) return IS3D_OK;
    const double *src[23] = {cells->tau, cells->eta, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un,
                             cells->T, cells->P, cells->E, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn, cells->bulkPi,
                             cells->muB, cells->nB, cells->Vx, cells->Vy, cells->Vn};
    DevMem d_cell[23], d_x, d_y, d_particles;
    const double *dptr[23];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    SMP_TRY(hipEventCreate(&e0));
    SMP_TRY(hipEventCreate(&e1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{e0, e1};
    SMP_TRY(hipEventRecord(e0, nullptr));
    for (int a = 0; a < 23; a++) {
        dptr[a] = nullptr;
        if (src[a] && cell_array_needed(a, opts)) {
            SMP_TRY(d_cell[a].alloc((size_t)n * sizeof(double)));
            SMP_TRY(hipMemcpyAsync(d_cell[a].p, src[a], (size_t)n * sizeof(double), hipMemcpyHostToDevice, nullptr));
            dptr[a] = d_cell[a].as<double>();
        }
    }
    if (in->x) { SMP_TRY(d_x.alloc((size_t)n * sizeof(double))); SMP_TRY(hipMemcpyAsync(d_x.p, in->x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, nullptr)); }
    if (in->y) { SMP_TRY(d_y.alloc((size_t)n * sizeof(double))); SMP_TRY(hipMemcpyAsync(d_y.p, in->y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, nullptr)); }
    SMP_TRY(hipEventRecord(e1, nullptr));
    is3d_cells dc{};
    dc.n_cells = n;
    dc.tau = dptr[0]; dc.eta = dptr[1]; dc.dat = dptr[2]; dc.dax = dptr[3]; dc.day = dptr[4]; dc.dan = dptr[5]; dc.ux = dptr[6]; dc.uy = dptr[7]; dc.un = dptr[8];
    dc.T = dptr[9]; dc.P = dptr[10]; dc.E = dptr[11]; dc.pixx = dptr[12]; dc.pixy = dptr[13]; dc.pixn = dptr[14]; dc.piyy = dptr[15]; dc.piyn = dptr[16];
    dc.bulkPi = dptr[17]; dc.muB = dptr[18]; dc.nB = dptr[19]; dc.Vx = dptr[20]; dc.Vy = dptr[21]; dc.Vn = dptr[22];
    if (capacity > 0) SMP_TRY(d_particles.alloc((size_t)capacity * sizeof(is3d_particle)));
    int64_t total = 0;
    const int rc = is3d_sampler_plan_execute(P, &dc, d_x.as<double>(), d_y.as<double>(), in->n_events, in->seed, in->first_cell, in->batch_events,
                                             d_particles.as<is3d_particle>(), capacity, &total, stats);
    *n_particles = total;
    if (stats) {
        float a = 0;
        (void)hipEventElapsedTime(&a, e0, e1);
        stats->ms_h2d = a;
    }
    if (rc && rc != IS3D_ENOMEM) return rc;
    const std::string kept = rc ? is3d_last_error() : "";
    const int64_t ncopy = std::min<int64_t>(total, capacity);
    if (ncopy > 0) SMP_TRY(hipMemcpy(particles, d_particles.p, (size_t)ncopy * sizeof(is3d_particle), hipMemcpyDeviceToHost));
    if (rc) return set_error(rc, "%s", kept.c_str());
    return IS3D_OK;
}
