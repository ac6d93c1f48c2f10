// cf_yield.hip -- is3d_total_yield: the mean particle yield of the surface that sizes an oversampled run.
//
// Replaces EmissionFunctionArray::calculate_total_yield (/root/reference/src/cpp/emissionfunction_sampling_kernels.cpp:653-830,
// call sites emissionfunction.cpp:1527, :1591) with estimate_mean_particle_number (:200-236) and the species densities of
// Deltaf_Data::compute_particle_densities (deltafReader.cpp:536-650).  The reference loops serially over cells x species; nothing
// in a cell's term depends on the species except the three density arrays, so
//   Ntot = sum_cells [ ds_time (S_eq + bulkPi S_bulk) - ds_space V.dsigma S_diff ]        df_mode 1-3
//        = sum_cells   ds_time z(bulkPi / P) S_eq                                          df_mode 4 (breakdown test is `false`, emissionfunction.cpp:138-146)
// with S_* = the sums of the density arrays over the chosen species (host, once) -- a per-cell weight and one device reduction:
// cf_yield_cells (thread <-> cell, fixed-shape tree per block) writes one partial per block, the host adds them in block order
// (deterministic).  HBM bound: 8-16 cell arrays read once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/is3d_amd.h"
#include "cf_device.h"
#include "cf_math.h"
#include "errors.h"
#include "jonah.h"
#include "spline.h"

namespace is3d {

struct YieldParams {
    CellPtrs cells;
    int64_t n_cells;
    int32_t df_mode, include_bulk, baryon, baryondiff;
    double S_eq, S_bulk, S_diff;
    // domain of evaluate_df_coefficients per cell (:761): spline range, or the (T, mu_B) grid of the bilinear branch
    double T_lo, T_hi, dT, B_lo, dB;
    int32_t nT, nB, swap;   // swap: opts.reference_bilinear_indexing (cf_math.h::bilinear5)
    // df_mode 4: z(bulkPi / P) spline
    int32_t nj;
    const double *jx, *jz, *jcz;
    double bp_max;
    double *partial;                 // [gridDim.x]
    unsigned long long *status;      // [0] min bad cell
};

__global__ void __launch_bounds__(256) cf_yield_cells(YieldParams p)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t ic = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ic < p.n_cells; ic += (int64_t)gridDim.x * blockDim.x) {
        const double tau = p.cells.tau[ic], tau2 = tau * tau;
        const double dat = p.cells.dat[ic], dax = p.cells.dax[ic], day = p.cells.day[ic], dan = p.cells.dan[ic];
        const double ux = p.cells.ux[ic], uy = p.cells.uy[ic], un = p.cells.un[ic];
        const double ut = sqrt(1.0 + ux * ux + uy * uy + tau2 * un * un);
        const double udsigma = ut * dat + ux * dax + uy * day + un * dan;
        if (udsigma <= 0.0) continue;                                                  // :689
        const double T = p.cells.T[ic], P = p.cells.P[ic];
        double bulkPi = p.include_bulk ? p.cells.bulkPi[ic] : 0.0;                     // :722-724
        double muB = 0.0, Vdsigma = 0.0;
        if (p.baryon && p.baryondiff) {                                                // :736-748
            muB = p.cells.muB[ic];
            const double Vx = p.cells.Vx[ic], Vy = p.cells.Vy[ic], Vn = p.cells.Vn[ic];
            const double Vt = (Vx * ux + Vy * uy + tau2 * Vn * un) / ut;
            Vdsigma = Vt * dat + Vx * dax + Vy * day + Vn * dan;
        }
        bool bad;
        if (p.baryon) {                                                                // bilinear_interpolation's range test, deltafReader.cpp:417-427
            const int iTL = (int)floor((T - p.T_lo) / p.dT), iBL = (int)floor((muB - p.B_lo) / p.dB);
            bad = !(iTL >= 0 && iTL + 1 < p.nT) || !(iBL >= 0 && iBL + 1 < p.nB) || (p.swap && !(iTL + 1 < p.nB && iBL + 1 < p.nT));
        } else bad = !(T >= p.T_lo && T <= p.T_hi);                                    // gsl_spline_eval domain
        double z = 0.0;
        if (p.df_mode == 4 && !bad) {                                                  // :752-758, deltafReader.cpp:364-377
            if (bulkPi <= -P) bulkPi = -(1.0 - 1.e-5) * P;
            else if (bulkPi / P >= p.bp_max) bulkPi = P * (p.bp_max - 1.e-5);
            const double r = bulkPi / P;
            if (!(r >= p.jx[0] && r <= p.jx[p.nj - 1])) bad = true;
            else z = spline_eval_lds(p.nj, p.jx, p.jz, p.jcz, r);
        }
        if (bad) { atomicMin(&p.status[0], (unsigned long long)ic); continue; }
        // Milne_Basis, Surface_Element_Vector::boost_dsigma_to_lrf (viscous_correction.cpp:8-27, :69-86)
        const double uperp = sqrt(ux * ux + uy * uy), utperp = sqrt(1.0 + ux * ux + uy * uy);
        const double sinhL = tau * un / utperp, coshL = ut / utperp;
        const double Xt = uperp * coshL, Zt = sinhL, Xn = uperp * sinhL / tau, Zn = coshL / tau;
        double Xx = 1.0, Yx = 0.0, Xy = 0.0, Yy = 1.0;
        if (uperp > 1.e-5) { Xx = utperp * ux / uperp; Yx = -uy / uperp; Xy = utperp * uy / uperp; Yy = ux / uperp; }
        const double ds_time = dat * ut + dax * ux + day * uy + dan * un;
        const double dsx = -(dat * Xt + dax * Xx + day * Xy + dan * Xn);
        const double dsy = -(dax * Yx + day * Yy);
        const double dsz = -(dat * Zt + dan * Zn);
        const double ds_space = sqrt(dsx * dsx + dsy * dsy + dsz * dsz);
        if (p.df_mode <= 3) acc += ds_time * (p.S_eq + bulkPi * p.S_bulk) - ds_space * Vdsigma * p.S_diff;   // :200-211
        else acc += ds_time * z * p.S_eq;                                                                    // :219
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.partial[blockIdx.x] = red[0];
}

}  // namespace is3d

namespace {

#define YLD_TRY(expr)                                                                                            \
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
    hipError_t upload(const void *h, size_t bytes)
    {
        hipError_t e = alloc(bytes);
        if (e != hipSuccess || !bytes) return e;
        return hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
    ~DevMem() { release(); }
    template <class T> T *as() const { return (T *)p; }
};

// gsl_spline_eval on a host table (the reference evaluates the df coefficients at the surface averages once per run)
bool host_spline_at(const std::vector<double> &xs, const double *tab, double xq, double *out)
{
    const int n = (int)xs.size();
    if (!(xq >= xs.front() && xq <= xs.back())) return false;
    std::vector<double> ys(tab, tab + n), cc;
    if (!is3d::natural_cspline_init(xs, ys, cc)) return false;
    int lo = 0, hi = n - 1;
    while (hi > lo + 1) { int i = (hi + lo) >> 1; if (xs[i] > xq) hi = i; else lo = i; }
    const double dx = xs[lo + 1] - xs[lo], dy = ys[lo + 1] - ys[lo], delx = xq - xs[lo];
    const double b_i = (dy / dx) - dx * (cc[lo + 1] + 2.0 * cc[lo]) / 3.0, d_i = (cc[lo + 1] - cc[lo]) / (3.0 * dx);
    *out = ys[lo] + delx * (b_i + delx * (cc[lo] + delx * d_i));
    return true;
}

// GaussThermal (gaussThermal.cpp) with the integrand given as a callable of (pbar, Ebar, e = exp(pbar + Ebar - chem), qstat)
template <class F>
double gauss_thermal(const double *root, const double *weight, int n, double mbar, double chem, double sign, F f)
{
    double s = 0.0;
    for (int k = 0; k < n; k++) {
        const double pbar = root[k], Ebar = std::sqrt(pbar * pbar + mbar * mbar), qstat = std::exp(Ebar - chem) + sign;
        s += weight[k] * f(pbar, Ebar, qstat, chem);
    }
    return s;
}

}  // namespace

extern "C" int is3d_total_yield(const is3d_cells *cells, const is3d_species *species, const is3d_df_tables *df,
                                const is3d_sampler_inputs *in, const is3d_yield_inputs *avg, const is3d_options *opts,
                                double *mean_yield, double *densities)
{
    using is3d::set_error;
    if (!cells || !species || !df || !in || !avg || !opts || !mean_yield) return set_error(IS3D_EINVAL, "null argument");
    *mean_yield = 0.0;
    const int mode = opts->df_mode;
    if (opts->dimension != 2 && opts->dimension != 3) return set_error(IS3D_EINVAL, "dimension must be 2 or 3 (got %d)", opts->dimension);
    if (mode < 1 || mode > 4) return set_error(IS3D_EINVAL, "df_mode must be 1, 2, 3 or 4 (got %d)", mode);
    const bool baryon = opts->include_baryon != 0, baryondiff = baryon && opts->include_baryondiff_deltaf != 0;
    if (baryon && mode == 4) return set_error(IS3D_EINVAL, "df_mode 4 does not work with include_baryon = 1 (the reference exits there too)");
    const is3d_feqmod_tables *fq = in->feqmod;
    if (in->n_gla < 1 || !in->root1 || !in->weight1) return set_error(IS3D_EINVAL, "the yield needs the Gauss-Laguerre roots and weights for alpha = 1");
    if (mode <= 3 && (!fq || fq->n_gla != in->n_gla || !fq->root2 || !fq->weight2))
        return set_error(IS3D_EINVAL, "df_mode 1-3: in->feqmod must carry the Gauss-Laguerre alpha = 2 nodes (J20 of the bulk densities)");
    if (mode == 1 && (!avg->root3 || !avg->weight3)) return set_error(IS3D_EINVAL, "df_mode 1 needs the Gauss-Laguerre alpha = 3 nodes (J30, J31)");
    if (mode == 4 && (!fq || fq->n_pdg < 1 || !fq->pdg_mass || !fq->pdg_degeneracy || !fq->pdg_sign || !(fq->T_avg > 0.0) || !fq->root2 || !fq->weight2))
        return set_error(IS3D_EINVAL, "df_mode 4 needs the full PDG list, the alpha = 2 nodes and the surface-averaged temperature");
    if (species->n < 1 || !species->mass || !species->sign || !species->degeneracy) return set_error(IS3D_EINVAL, "empty species list");
    if (baryon && !species->baryon) return set_error(IS3D_EINVAL, "include_baryon = 1 needs the species' baryon numbers");
    if (df->n_T < 3 || !df->T) return set_error(IS3D_EINVAL, "coefficient table needs >= 3 temperatures");
    if (baryon && (df->n_muB < 2 || !df->muB)) return set_error(IS3D_EINVAL, "include_baryon = 1 needs the full (T, muB) coefficient tables");
    if (!(avg->T > 0.0)) return set_error(IS3D_EINVAL, "the surface-averaged temperature must be positive");
    const int64_t n = cells->n_cells;
    if (n < 0) return set_error(IS3D_EINVAL, "n_cells < 0");
    if (n > 0) {
        if (!cells->tau || !cells->dat || !cells->dax || !cells->day || !cells->dan || !cells->ux || !cells->uy || !cells->un || !cells->T || !cells->P)
            return set_error(IS3D_EINVAL, "a required cell array is NULL");
        if (opts->include_bulk_deltaf && !cells->bulkPi) return set_error(IS3D_EINVAL, "include_bulk_deltaf needs bulkPi");
        if (baryondiff && (!cells->muB || !cells->Vx || !cells->Vy || !cells->Vn)) return set_error(IS3D_EINVAL, "include_baryondiff_deltaf needs muB, Vx, Vy, Vn");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");
    if (opts->device >= 0) YLD_TRY(hipSetDevice(opts->device));

    // ---- Deltaf_Data::compute_particle_densities at the averages (deltafReader.cpp:536-650) ----
    const int npart = species->n;
    const double two_pi2_hbarC3 = 2.0 * std::pow(M_PI, 2) * std::pow(is3d::kHbarC, 3);
    // the averages as Plasma::load_thermodynamic_averages reads them, muB and nB included whatever include_baryon says (deltafReader.cpp:541-552:
    // alphaB = muB_avg / T enters every density integrand through the species' own baryon number; the readers of modes 0 and 1 write muB_avg = 0
    // without include_baryon, the MUSIC / hic-eventgen readers always carry it)
    const double T = avg->T, E = avg->E, P = avg->P, muB = avg->muB, nB = avg->nB;
    const std::vector<double> xs(df->T, df->T + df->n_T);
    for (int i = 1; i < df->n_T; i++)
        if (!(xs[i] > xs[i - 1])) return set_error(IS3D_EINVAL, "coefficient table temperatures must ascend");
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, F = 0, G = 0, betabulk = 1, betaV = 1;
    if (baryon) {   // bilinear_interpolation (deltafReader.cpp:412-484), intended [imuB][iT] indexing
        const double *t5[5];
        if (mode == 1) { t5[0] = df->c0; t5[1] = df->c1; t5[2] = df->c2; t5[3] = df->c3; t5[4] = df->c4; }
        else { t5[0] = df->F; t5[1] = df->G; t5[2] = df->betabulk; t5[3] = df->betaV; t5[4] = df->betapi; }
        for (int k = 0; k < 5; k++)
            if (!t5[k]) return set_error(IS3D_EINVAL, "include_baryon = 1 needs the five (T, muB) tables of the df_mode");
        is3d::BilinearDev hb{};
        hb.nT = df->n_T; hb.nB = df->n_muB; hb.T = df->T; hb.muB = df->muB;
        for (int k = 0; k < 5; k++) hb.tab[k] = t5[k];
        hb.swap = opts->reference_bilinear_indexing != 0;
        double v[5];
        if (!is3d::bilinear5(hb, T, muB, v))
            return set_error(IS3D_EDOMAIN, "the surface averages (T, muB) = (%.6g, %.6g) GeV are outside the coefficient table", T, muB);
        const double T3 = T * T * T, T4 = T3 * T;
        if (mode == 1) { c0 = v[0] / T4; c1 = v[1] / T3; c2 = v[2] / T4; c3 = v[3] / T4; c4 = v[4] / (T4 * T); }
        else { F = v[0] * T; G = v[1]; betabulk = v[2] * T4; betaV = v[3] * T3; }
    } else {        // cubic_spline (:325-395)
        const double T4 = T * T * T * T;
        double v = 0.0;
        bool ok = true;
        if (mode == 1) {
            if (!df->c0 || !df->c2) return set_error(IS3D_EINVAL, "df_mode 1 needs c0 and c2 tables");
            ok = host_spline_at(xs, df->c0, T, &v); c0 = v / T4;
            ok = ok && host_spline_at(xs, df->c2, T, &v); c2 = v / T4;
        } else if (mode <= 3) {
            if (!df->F || !df->betabulk) return set_error(IS3D_EINVAL, "df_mode 2 / 3 need F and betabulk tables");
            ok = host_spline_at(xs, df->F, T, &v); F = v * T;
            ok = ok && host_spline_at(xs, df->betabulk, T, &v); betabulk = v * T4;
        } else {
            if (!df->betapi) return set_error(IS3D_EINVAL, "df_mode 4 needs the betapi table");
            ok = host_spline_at(xs, df->betapi, T, &v);
        }
        if (!ok) return set_error(IS3D_EDOMAIN, "the surface-averaged temperature %.6g GeV is outside the coefficient table", T);
    }
    const double alphaB = muB / T, baryon_enthalpy_ratio = nB / (E + P);              // :551-552
    std::vector<double> eqd(npart, 0.0), bkd(npart, 0.0), dfd(npart, 0.0);
    const int ng = in->n_gla;
    const double *r1 = in->root1, *w1 = in->weight1, *r2 = fq ? fq->root2 : nullptr, *w2 = fq ? fq->weight2 : nullptr;
    auto neq_i = [](double pbar, double, double qstat, double) { return pbar * std::exp(pbar) / qstat; };
    auto J10_i = [](double pbar, double Ebar, double qstat, double chem) { return pbar * std::exp(pbar + Ebar - chem) / (qstat * qstat); };
    auto J11_i = [](double pbar, double Ebar, double qstat, double chem) { return pbar * pbar * pbar / (Ebar * Ebar) * std::exp(pbar + Ebar - chem) / (qstat * qstat); };
    auto J20_i = [](double pbar, double Ebar, double qstat, double chem) { return Ebar * std::exp(pbar + Ebar - chem) / (qstat * qstat); };
    auto J30_i = [](double pbar, double Ebar, double qstat, double chem) { return Ebar * Ebar / pbar * std::exp(pbar + Ebar - chem) / (qstat * qstat); };
    auto J31_i = [](double pbar, double Ebar, double qstat, double chem) { return pbar * std::exp(pbar + Ebar - chem) / (qstat * qstat); };
    for (int i = 0; i < npart; i++) {
        const double mass = species->mass[i], g = species->degeneracy[i], sign = species->sign[i];
        // the particle's own baryon number, whatever include_baryon says (deltafReader.cpp:575); alpha_B = muB_avg / T of the averages file
        const double b = species->baryon ? species->baryon[i] : 0.0, mbar = mass / T, chem = b * alphaB;
        const double f3 = g * std::pow(T, 3) / two_pi2_hbarC3, f4 = g * std::pow(T, 4) / two_pi2_hbarC3, f5 = g * std::pow(T, 5) / two_pi2_hbarC3;
        const double neq = f3 * gauss_thermal(r1, w1, ng, mbar, chem, sign, neq_i);
        double dn_bulk = 0.0, dn_diff = 0.0;
        if (mode == 1) {                                                              // :587-612
            const double J10 = f3 * gauss_thermal(r1, w1, ng, mbar, chem, sign, J10_i);
            const double J20 = f4 * gauss_thermal(r2, w2, ng, mbar, chem, sign, J20_i);
            const double J30 = f5 * gauss_thermal(avg->root3, avg->weight3, ng, mbar, chem, sign, J30_i);
            const double J31 = f5 / 3.0 * gauss_thermal(avg->root3, avg->weight3, ng, mbar, chem, sign, J31_i);
            dn_bulk = ((c0 - c2) * mass * mass * J10 + c1 * b * J20 + (4.0 * c2 - c0) * J30);
            dn_diff = b * c3 * neq * T + c4 * J31;
        } else if (mode <= 3) {                                                       // :613-632
            const double J10 = f3 * gauss_thermal(r1, w1, ng, mbar, chem, sign, J10_i);
            const double J11 = f3 / 3.0 * gauss_thermal(r1, w1, ng, mbar, chem, sign, J11_i);
            const double J20 = f4 * gauss_thermal(r2, w2, ng, mbar, chem, sign, J20_i);
            dn_bulk = (neq + (b * J10 * G) + (J20 * F / std::pow(T, 2))) / betabulk;
            dn_diff = (neq * T * baryon_enthalpy_ratio - b * J11) / betaV;
        }
        eqd[i] = neq; bkd[i] = dn_bulk; dfd[i] = dn_diff;
    }
    if (densities)
        for (int i = 0; i < npart; i++) { densities[i] = eqd[i]; densities[npart + i] = bkd[i]; densities[2 * npart + i] = dfd[i]; }
    is3d::YieldParams p{};
    for (int i = 0; i < npart; i++) { p.S_eq += eqd[i]; p.S_bulk += bkd[i]; p.S_diff += dfd[i]; }
    if (n == 0) return IS3D_OK;

    // ---- per-cell weights on the device ----
    DevMem d_cell[16], d_jx, d_jz, d_jcz, d_partial, d_status;
    const double *src[16] = {cells->tau, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un, cells->T, cells->P,
                             opts->include_bulk_deltaf ? cells->bulkPi : nullptr, baryondiff ? cells->muB : nullptr,
                             baryondiff ? cells->Vx : nullptr, baryondiff ? cells->Vy : nullptr, baryondiff ? cells->Vn : nullptr, nullptr};
    const double *dp[16];
    for (int a = 0; a < 16; a++) {
        dp[a] = nullptr;
        if (src[a]) { YLD_TRY(d_cell[a].upload(src[a], (size_t)n * sizeof(double))); dp[a] = d_cell[a].as<double>(); }
    }
    p.cells.tau = dp[0]; p.cells.dat = dp[1]; p.cells.dax = dp[2]; p.cells.day = dp[3]; p.cells.dan = dp[4];
    p.cells.ux = dp[5]; p.cells.uy = dp[6]; p.cells.un = dp[7]; p.cells.T = dp[8]; p.cells.P = dp[9]; p.cells.bulkPi = dp[10];
    p.cells.muB = dp[11]; p.cells.Vx = dp[12]; p.cells.Vy = dp[13]; p.cells.Vn = dp[14];
    p.n_cells = n; p.df_mode = mode; p.include_bulk = opts->include_bulk_deltaf != 0; p.baryon = baryon; p.baryondiff = baryondiff;
    p.T_lo = xs.front(); p.T_hi = xs.back(); p.dT = std::fabs(xs[1] - xs[0]); p.nT = df->n_T;
    if (baryon) { p.B_lo = df->muB[0]; p.dB = std::fabs(df->muB[1] - df->muB[0]); p.nB = df->n_muB; p.swap = opts->reference_bilinear_indexing != 0; }
    if (mode == 4) {
        std::vector<double> bp, l2, zz, cz;
        is3d::jonah_tables(fq, bp, l2, zz, p.bp_max);
        if (!is3d::natural_cspline_init(bp, zz, cz))
            return set_error(IS3D_EINVAL, "df_mode 4: bulkPi/Peq(lambda) is not ascending at T_avg = %.6g GeV (GSL would abort here)", fq->T_avg);
        p.nj = (int)bp.size();
        YLD_TRY(d_jx.upload(bp.data(), bp.size() * sizeof(double)));
        YLD_TRY(d_jz.upload(zz.data(), zz.size() * sizeof(double)));
        YLD_TRY(d_jcz.upload(cz.data(), cz.size() * sizeof(double)));
        p.jx = d_jx.as<double>(); p.jz = d_jz.as<double>(); p.jcz = d_jcz.as<double>();
    }
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 1024);
    YLD_TRY(d_partial.alloc((size_t)grid * sizeof(double)));
    unsigned long long st0 = ~0ULL;
    YLD_TRY(d_status.upload(&st0, sizeof st0));
    p.partial = d_partial.as<double>(); p.status = d_status.as<unsigned long long>();
    hipLaunchKernelGGL(is3d::cf_yield_cells, dim3(grid), dim3(256), 0, nullptr, p);
    YLD_TRY(hipGetLastError());
    std::vector<double> part(grid);
    YLD_TRY(hipMemcpy(part.data(), d_partial.p, (size_t)grid * sizeof(double), hipMemcpyDeviceToHost));
    YLD_TRY(hipMemcpy(&st0, d_status.p, sizeof st0, hipMemcpyDeviceToHost));
    if (st0 != ~0ULL)
        return set_error(IS3D_EDOMAIN, "cell %llu: T%s outside the coefficient table (the reference aborts in evaluate_df_coefficients here)", st0,
                         mode == 4 ? " (or bulkPi/P)" : (baryon ? " or muB" : ""));
    double Ntot = 0.0;
    for (int b = 0; b < grid; b++) Ntot += part[b];
    if (opts->dimension == 2) Ntot *= (2.0 * in->y_cut);                               // :822-826
    *mean_yield = Ntot;
    return IS3D_OK;
}
