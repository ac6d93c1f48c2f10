// cf_plan.cpp -- plan object and C ABI of the device path (include/is3d_amd.h).
//
// Host-side counterpart of EmissionFunctionArray's packing/dispatch for the smooth path
// (/root/reference/src/cpp/emissionfunction.cpp:1282-1307 species scalars, :1503-1521 dispatch):
// groups species into (mass, sign) classes, builds the lane table, the natural cubic splines of
// Deltaf_Data::construct_cubic_splines (deltafReader.cpp:300-322) and drives prep -> main -> finalize.
// There is no CPU compute path in this file: without a HIP device every entry fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/is3d_amd.h"
#include "cf_feqmod.h"
#include "cf_launch.h"
#include "errors.h"
#include "jonah.h"
#include "spline.h"

#define fail is3d::set_error

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(IS3D_ENODEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (!count) return hipSuccess;
        is3d::count_resource(1);
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    hipError_t upload(const std::vector<T> &h)
    {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

}  // namespace

struct is3d_plan {
    is3d_options opts{};
    int device = 0;
    int npart = 0, ncls = 0, npT = 0, J = 0, K = 0, Kacc = 0, ny_eff = 0;
    int L = 0, Lpad = 0;      // L: lane slots in use (bins x split), Lpad: padded to whole waves
    int Lbins = 0, split = 1; // momentum bins (classes x pT); lane slots per bin (unit-strided lanes, 2+1D variant 7)
    DevBuf<int32_t> d_lane_sub;
    int variant = 2, JT = 1, KT = 1, jtiles = 1, ktiles = 1;
    bool dim3 = true, ce = false;
    int64_t max_cells = 0, pass_cells = 0, nout = 0;
    int nch_max = 1;
    size_t bytes_per_cell = 0;
    double prefactor = 0;
    double mTmax = 0, pTmax = 0;   // over the lane table: bound of |p.dsigma| for the stream's power-of-two scale
    double kmin = 0, kmax = 0, gw2d = 1;   // y range (3+1D) / max_k w_k cosh(eta_k) (2+1D) for the same bound

    DevBuf<double> d_mT, d_pT, d_sign, d_lane_b, d_degeneracy, d_cosphi, d_sinphi, d_kgrid, d_kweight, d_kch, d_ksh;
    DevBuf<double> d_bilT, d_bilB, d_biltab[5];
    DevBuf<double> d_coskphi, d_sinkphi, d_phiw, d_pTw;   // derived observables
    is3d::BilinearDev bil{};
    bool baryon = false, baryondiff = false;
    DevBuf<int> d_cls;
    DevBuf<int32_t> d_lane_pe;      // per lane: max(mT/mTmax, pT/pTmax) < 2^pe
    DevBuf<double> d_cull_floor;    // zero_skip 3 (cf_main_tile3e): [jtiles * ktiles][Lpad] threshold floors from the chunks that ran first
    DevBuf<double> d_splx, d_sply[3], d_splc[3];
    DevBuf<double> d_S1, d_S2, d_S3, d_TS, d_partial;
    DevBuf<double> d_TE, d_pTgrid;   // variant 5: E2 table stream (cf_device.h), the pT grid for cf_prep
    DevBuf<int32_t> d_lane_ipT;      // variant 5: lane -> index of its pT
    bool e2tab = false;
    int ub3e = 0;
    int rblocks = 1, upc = 1;   // row blocks of the tiled stream; units per cell within a stream
    int wpb = 4;                // waves per workgroup of the main kernel
    DevBuf<unsigned long long> d_status, d_sticky;   // d_sticky: {min bad cell, min fast cell} over the executes since the last is3d_plan_check
    is3d::SplineDev spl{};

    // modified equilibrium (df_mode 3, 4)
    bool feqmod = false;
    int nj = 0, ngl = 0;
    double bp_max = 0.0, detA_min = 0.0, mass_pion0 = 0.0;
    DevBuf<double> d_gl, d_jonah, d_cls_mass, d_cls_sign, d_cls_baryon, d_lane_mass, d_RN, d_CR, d_FB;
    DevBuf<int32_t> d_lane_cls, d_flag, d_list, d_count;

    bool timing = false;
    std::vector<hipEvent_t> ev_list;  // [pass][0..3]: start, after prep, after main; last: after finalize
    int last_passes = 0;
    int64_t workspace = 0;

    ~is3d_plan()
    {
        for (hipEvent_t e : ev_list)
            if (e) (void)hipEventDestroy(e);
    }
};

static int validate(const is3d_species *sp, const is3d_grid *g, const is3d_df_tables *df, const is3d_feqmod_tables *fq,
                    const is3d_options *o)
{
    if (!sp || !g || !df || !o) return fail(IS3D_EINVAL, "null argument");
    if (o->dimension != 2 && o->dimension != 3) return fail(IS3D_EINVAL, "dimension must be 2 or 3 (got %d)", o->dimension);
    if (fq) {
        if (o->df_mode != 3 && o->df_mode != 4) return fail(IS3D_EINVAL, "the feqmod entries take df_mode 3 or 4 (got %d)", o->df_mode);
        if (o->include_baryon && o->df_mode == 4)   // the reference: "Jonah df doesn't work for nonzero muB. Exiting..", deltafReader.cpp:470-474
            return fail(IS3D_EINVAL, "df_mode 4 does not work with include_baryon = 1 (the reference exits there too)");
        if (o->kernel_variant == 1) return fail(IS3D_EINVAL, "df_mode 3/4 runs on the tile kernel only (kernel_variant 0, 2-4)");
        if (fq->n_gla < 1 || fq->n_gla > 256 || !fq->root1 || !fq->weight1 || !fq->root2 || !fq->weight2)
            return fail(IS3D_EINVAL, "df_mode 3/4 needs the Gauss-Laguerre roots and weights for alpha = 1, 2");
        if (!df->betapi) return fail(IS3D_EINVAL, "df_mode 3/4 needs the betapi table");
        if (o->df_mode == 3 && (!df->F || !df->betabulk)) return fail(IS3D_EINVAL, "df_mode 3 needs the F and betabulk tables");
        if (o->df_mode == 4 && (fq->n_pdg < 1 || !fq->pdg_mass || !fq->pdg_degeneracy || !fq->pdg_sign || !(fq->T_avg > 0.0)))
            return fail(IS3D_EINVAL, "df_mode 4 needs the full PDG list and the surface-averaged temperature");
    } else if (o->df_mode != 1 && o->df_mode != 2)
        return fail(IS3D_EINVAL, "df_mode must be 1 (14-moment) or 2 (Chapman-Enskog) on this entry (got %d); 3 and 4 go through is3d_*_feqmod", o->df_mode);
    if (o->include_baryon) {
        if (o->kernel_variant == 1) return fail(IS3D_EINVAL, "include_baryon = 1 runs on the tile kernel only (kernel_variant 2-4)");
        if (!sp->baryon) return fail(IS3D_EINVAL, "include_baryon = 1 needs the species' baryon numbers");
        if (df->n_muB < 2 || !df->muB) return fail(IS3D_EINVAL, "include_baryon = 1 needs the full (T, muB) coefficient tables");
        for (int i = 1; i < df->n_muB; i++)
            if (!(df->muB[i] > df->muB[i - 1])) return fail(IS3D_EINVAL, "coefficient table muB values must ascend");
        if (o->df_mode == 1 && (!df->c1 || !df->c3 || !df->c4)) return fail(IS3D_EINVAL, "include_baryon = 1, df_mode 1 needs c0..c4 tables");
        if ((o->df_mode == 2 || o->df_mode == 3) && (!df->G || !df->betaV)) return fail(IS3D_EINVAL, "include_baryon = 1, df_mode 2 needs F, G, betabulk, betaV, betapi tables");
    }
    if (sp->n < 1 || !sp->mass || !sp->sign || !sp->degeneracy) return fail(IS3D_EINVAL, "empty species list");
    if (g->n_pT < 1 || g->n_phi < 1 || !g->pT || !g->phi) return fail(IS3D_EINVAL, "empty pT/phi grid");
    if (o->dimension == 3 && (g->n_y < 1 || !g->y)) return fail(IS3D_EINVAL, "dimension 3 needs a y grid");
    if (o->dimension == 2 && (g->n_eta < 1 || !g->eta || !g->eta_w)) return fail(IS3D_EINVAL, "dimension 2 needs an eta table");
    if (df->n_T < 3 || !df->T) return fail(IS3D_EINVAL, "coefficient table needs >= 3 temperatures");
    if (o->df_mode == 1 && (!df->c0 || !df->c2)) return fail(IS3D_EINVAL, "df_mode 1 needs c0 and c2 tables");
    if (o->df_mode == 2 && (!df->F || !df->betabulk || !df->betapi)) return fail(IS3D_EINVAL, "df_mode 2 needs F, betabulk, betapi tables");
    for (int i = 1; i < df->n_T; i++)
        if (!(df->T[i] > df->T[i - 1])) return fail(IS3D_EINVAL, "coefficient table temperatures must ascend");
    return IS3D_OK;
}

extern "C" const char *is3d_version(void) { return is3d::kDevBuild ? "is3d_amd 0.1 (gfx950) DEV BUILD" : "is3d_amd 0.1 (gfx950)"; }
extern "C" int is3d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int plan_create_impl(is3d_plan **out, const is3d_species *sp, const is3d_grid *g, const is3d_df_tables *df,
                            const is3d_feqmod_tables *fq, const is3d_options *o, int64_t max_cells)
{
    if (!out) return fail(IS3D_EINVAL, "null plan pointer");
    *out = nullptr;
    int rc = validate(sp, g, df, fq, o);
    if (rc) return rc;
    if (max_cells < 1) max_cells = 1;
    if (max_cells > (int64_t)1 << 40) return fail(IS3D_EINVAL, "max_cells too large");
    if (is3d_device_count() < 1) return fail(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");

    std::unique_ptr<is3d_plan> P(new is3d_plan);
    is3d::count_resource(0);
    P->opts = *o;
    if (o->device >= 0) HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipGetDevice(&P->device));
    P->dim3 = (o->dimension == 3);
    P->ce = (o->df_mode == 2);
    P->feqmod = fq != nullptr;
    P->npart = sp->n;
    P->npT = g->n_pT;
    P->J = g->n_phi;
    P->K = P->dim3 ? g->n_y : g->n_eta;
    P->Kacc = P->dim3 ? P->K : 1;
    P->ny_eff = P->Kacc;
    P->nout = (int64_t)P->npart * P->npT * P->J * P->ny_eff;
    P->prefactor = std::pow(2.0 * M_PI * is3d::kHbarC, -3);  // smooth_kernels.cpp:36
    // Default kernel per mode (A/B on MI355X):
    //   3+1D delta-f without baryon slots, pT grid <= 32: variant 6 -- 8 x 7 tile, phi-side exponentials from the E2 table stream,
    //     rows tested for liveness before their exponential (config 3: 354 ms against 404 ms for variant 3);
    //   3+1D otherwise: 8 x 7 (modified equilibrium: 1005 vs 1028 ms for 6 x 7), 6 x 7 with the baryon slots;
    //   2+1D delta-f: 8 x 61; with few momentum bins variant 7 -- 8 x 31 tile with unit-strided lanes, when that fills the waves
    //     (config 2, 96 bins: 31.4 ms against 38.7 ms).
    auto split_for = [&](int n_bins) {   // lane slots per bin for variant 7: S in {1, 2, 4} dividing the units per cell
        int JT7 = 0, R7 = 0, S_best = 1;
        is3d::main_tile_shape(7, 0, &JT7, &R7);
        const int rblocks7 = (g->n_eta + R7 - 1) / R7;
        double best = 1.0 - (double)n_bins / (double)(((n_bins + 63) / 64) * 64);
        for (int S : {2, 4}) {
            if (rblocks7 % S) continue;
            const int tot = n_bins * S, pad = ((tot + 127) / 128) * 128;   // whole 2-wave workgroups
            const double waste = 1.0 - (double)tot / (double)pad;
            if (waste < best - 0.05) { best = waste; S_best = S; }
        }
        return S_best;
    };
    const bool plain3 = o->dimension == 3 && !o->include_baryon;
    // (modified equilibrium in 3+1D: 8 x 7 with or without the baryon slots -- its records do not grow with them; 247.1 against 260.5 ms for 6 x 7 per 3e5 cells)
    int default_variant = (plain3 || (fq && o->dimension == 3)) ? 3 : 2;
    if (!fq && o->dimension == 3 && g->n_pT <= is3d::kE2Stride) default_variant = 6;   // with or without baryon slots
    P->variant = (o->kernel_variant >= 1 && o->kernel_variant <= 12) ? o->kernel_variant : default_variant;
    // The shipped library holds the kernels the defaults reach (cf_kernels.hip::launch_variant, cf_feqmod.hip): any other request runs the default --
    // status.kernel_variant says which one ran.  One explicit choice is honoured: the 8 x 7 tile without the E2 table stream (3) for a 3+1D delta-f
    // surface without baryon slots, which is also the default for pT grids of more than 32 values.  The A/B forms of rounds 1-5 (1, 2, 4, 5, 8, 9; the
    // modified-equilibrium row walks 5, 6 and 61-row tiles) exist in the developer build (make DEV=1), where tests/test_gpu_devlib.py runs their parity tests.
    const bool ab_variants = is3d::kDevBuild;
    if (!ab_variants && !(!fq && plain3 && P->variant == 3)) P->variant = default_variant;
    // modified equilibrium in 2+1D: variant 7 (8 x 31 tile, unit-strided lanes, rows tested against the unit's threshold from the beta minimum
    // they carry) is the default since round 4; variants 2-4 keep the round-1 row walk on the 61-row tiles for A/B
    if (fq && o->dimension == 2 && !(ab_variants && o->kernel_variant >= 2 && o->kernel_variant <= 4)) P->variant = 7;
    if ((P->variant == 7 || P->variant == 8) && o->dimension == 3) P->variant = default_variant;
    if (P->variant == 8 && fq) P->variant = 7;
    if (P->variant == 8 && o->include_baryon) P->variant = 7;   // variant 8 = variant 7 with the register-staged copy (A/B), without baryon slots only   // unit-strided lanes: the 2+1D delta-f tile kernel
    const bool e2ok = o->dimension == 3 && !fq && g->n_pT <= is3d::kE2Stride;   // the E2 table stream exists for the 3+1D delta-f kernels
    // (modified equilibrium in 3+1D: variants 5 and 6 are A/B forms of its 8 x 7 kernel -- rows pipelined as in round 1 / row mask with the exact
    // per-row thresholds, cf_feqmod.hip -- without baryon slots)
    const bool fq56 = fq && plain3;
    // variant 9 (round 5, developer build): cf_main_tile3s, the E2-table kernel with its unit records on the scalar path -- 3+1D delta-f without baryon slots only
    if (P->variant >= 9 && P->variant <= 12 && !(e2ok && plain3 && is3d::kDevBuild)) P->variant = default_variant;   // measured and dropped: the developer build keeps them for A/B
    if ((P->variant == 5 || P->variant == 6) && !e2ok && !fq56) P->variant = (fq || !plain3) ? default_variant : 3;
    P->e2tab = (P->variant == 5 || P->variant == 6 || (P->variant >= 9 && P->variant <= 12)) && e2ok;

    // ---- species classes: identical (mass, sign) => identical integrand up to the degeneracy ----
    std::vector<int> cls(P->npart);
    // (with include_baryon the baryon number enters f_eq and delta-f, so it is part of the class key)
    P->baryon = o->include_baryon != 0;
    P->baryondiff = P->baryon && o->include_baryondiff_deltaf != 0;
    std::vector<double> cmass, csign, cbar;
    const bool collapse = (o->collapse_species != 2);
    for (int s = 0; s < P->npart; s++) {
        int found = -1;
        const double bs = P->baryon ? sp->baryon[s] : 0.0;
        if (collapse)
            for (size_t c = 0; c < cmass.size(); c++)
                if (cmass[c] == sp->mass[s] && csign[c] == sp->sign[s] && cbar[c] == bs) { found = (int)c; break; }
        if (found < 0) { found = (int)cmass.size(); cmass.push_back(sp->mass[s]); csign.push_back(sp->sign[s]); cbar.push_back(bs); }
        cls[s] = found;
    }
    P->ncls = (int)cmass.size();
    P->Lbins = P->ncls * P->npT;
    // unit-strided lanes (variant 7, 2+1D): S lane slots per bin so that the slots fill whole waves (96 bins: 128 slots = 25 %
    // idle lanes with S = 1, 384 = 6 full waves with S = 4); S must divide the units per cell and the units per LDS batch (4)
    P->split = 1;
    // default in 2+1D: the 8 x 31 tile, with or without extra lane slots (305 species, one slot per bin: 96.0 against 101.0 ms for 8 x 61 per 2e4 cells)
    if (!fq && o->dimension == 2 && !(ab_variants && o->kernel_variant >= 1 && o->kernel_variant <= 8)) P->variant = 7;   // (9 is 3+1D only)
    if (P->variant == 7 || P->variant == 8) P->split = split_for(P->Lbins);
    if (fq && o->dimension == 2 && P->variant != 7) {
        // modified equilibrium, 2+1D, the 61-row tiles (A/B): the same lane slots on the kernel's own tile -- S = 2 when it divides the units per cell and
        // the units per LDS batch (cf_main_feqmod stages 1536 / REC units) and fills the two-wave workgroups better
        int JTf = 0, Rf = 0;
        is3d::main_tile_shape(P->variant, 0, &JTf, &Rf);
        const int recf = 4 * JTf + Rf * (4 + JTf), ubf = std::max(1, 1536 / recf), rblocksf = (g->n_eta + Rf - 1) / Rf;
        const int nb = P->Lbins;
        const double waste1 = 1.0 - (double)nb / (double)(((nb + 63) / 64) * 64);
        const double waste2 = 1.0 - (double)(2 * nb) / (double)(((2 * nb + 63) / 64) * 64);
        if (rblocksf % 2 == 0 && ubf % 2 == 0 && waste2 < waste1 - 0.05) P->split = 2;
    }
    P->L = P->Lbins * P->split;
    P->Lpad = ((P->L + 63) / 64) * 64;
    // lane slots sorted by mT: a wave then holds momenta of similar energy, which is what makes the exact-zero
    // row culling of the main kernel wave-uniform more often (and keeps exp arguments of a wave close together)
    std::vector<double> mT(P->Lpad, 1.0), pT(P->Lpad, 0.0), sg(P->Lpad, 1.0), lb(P->Lpad, 0.0);
    std::vector<int32_t> lsub(P->Lpad, 0);
    std::vector<int> order(P->Lbins), slot_of(P->Lbins);
    std::vector<double> mT_nat(P->Lbins);
    for (int c = 0; c < P->ncls; c++)
        for (int i = 0; i < P->npT; i++) {
            double m = cmass[c], p = g->pT[i];
            mT_nat[c * P->npT + i] = std::sqrt(m * m + p * p);  // :259
            order[c * P->npT + i] = c * P->npT + i;
        }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return mT_nat[a] < mT_nat[b]; });
    {
        // dev A/B (IS3D_LANE_INTERLEAVE=1, developer build): the two waves of a workgroup take the even and the odd slots of their 128-slot span of
        // the mT order instead of its lower and upper half -- both then cover the same mT range, cull the same units and rows, and meet at the
        // per-batch barrier with (nearly) the same work done
        static const bool inter = is3d::dev_env("IS3D_LANE_INTERLEAVE") != nullptr;
        if (inter && o->dimension == 3 && !fq) {
            std::vector<int> o2(order);
            for (int b = 0; b + 128 <= P->Lbins; b += 128)
                for (int i = 0; i < 64; i++) { o2[b + i] = order[b + 2 * i]; o2[b + 64 + i] = order[b + 2 * i + 1]; }
            order.swap(o2);
        }
    }
    for (int s = 0; s < P->Lbins; s++) {
        const int nat = order[s], c = nat / P->npT, i = nat % P->npT;
        slot_of[nat] = s;
        for (int sl = 0; sl < P->split; sl++) {   // slot (bin s, sub sl) = sl * Lbins + s
            const int t = sl * P->Lbins + s;
            mT[t] = mT_nat[nat];
            pT[t] = g->pT[i];
            sg[t] = csign[c];
            lb[t] = cbar[c];
            lsub[t] = sl;
        }
    }
    for (int s = 0; s < P->L; s++) { P->mTmax = std::max(P->mTmax, mT[s]); P->pTmax = std::max(P->pTmax, std::fabs(pT[s])); }
    std::vector<int> lane_sp((size_t)P->npart * P->npT);
    for (int s = 0; s < P->npart; s++)
        for (int i = 0; i < P->npT; i++) lane_sp[(size_t)s * P->npT + i] = slot_of[cls[s] * P->npT + i];
    std::vector<double> deg(sp->degeneracy, sp->degeneracy + P->npart);
    std::vector<double> cosphi(P->J), sinphi(P->J);
    for (int j = 0; j < P->J; j++) { cosphi[j] = std::cos(g->phi[j]); sinphi[j] = std::sin(g->phi[j]); }  // :43-48
    std::vector<double> kgrid(P->K), kweight(P->K, 1.0);
    for (int k = 0; k < P->K; k++) {
        kgrid[k] = P->dim3 ? g->y[k] : g->eta[k];
        if (!P->dim3) kweight[k] = g->eta_w[k];
    }
    P->kmin = *std::min_element(kgrid.begin(), kgrid.end());
    P->kmax = *std::max_element(kgrid.begin(), kgrid.end());
    if (!P->dim3) {
        P->gw2d = 0.0;
        for (int k = 0; k < P->K; k++) P->gw2d = std::max(P->gw2d, std::fabs(kweight[k]) * std::cosh(kgrid[k]));
    }
    HIP_TRY(P->d_mT.upload(mT));
    HIP_TRY(P->d_pT.upload(pT));
    HIP_TRY(P->d_sign.upload(sg));
    HIP_TRY(P->d_lane_b.upload(lb));
    HIP_TRY(P->d_lane_sub.upload(lsub));
    if (P->e2tab) {
        std::vector<int32_t> ipT(P->Lpad, 0);
        for (int s = 0; s < P->Lbins; s++) ipT[s] = order[s] % P->npT;
        HIP_TRY(P->d_lane_ipT.upload(ipT));
        HIP_TRY(P->d_pTgrid.upload(std::vector<double>(g->pT, g->pT + P->npT)));
        for (int i = 0; i < P->npT; i++)
            if (!(g->pT[i] >= 0.0)) return fail(IS3D_EINVAL, "kernel_variant 5 needs pT >= 0 (pT Dmax must be the maximum of pT D_j)");
    }
    HIP_TRY(P->d_degeneracy.upload(deg));
    HIP_TRY(P->d_cls.upload(lane_sp));
    {
        // scaled p.dsigma of a lane: mT |A| + pT |B W| <= max(mT/mTmax, pT/pTmax) (mTmax |A| + pTmax |B W|) <= that factor (cf_pds_bound)
        std::vector<int32_t> pe(P->Lpad, 0);
        for (int s = 0; s < P->L; s++) {
            const double f = std::max(P->mTmax > 0.0 ? mT[s] / P->mTmax : 1.0, P->pTmax > 0.0 ? std::fabs(pT[s]) / P->pTmax : 0.0);
            int e = 0;
            (void)std::frexp(f * (1.0 + 1.0e-12), &e);     // f (1 + eps) = m 2^e, m in [0.5, 1): f < 2^e with room for the roundings
            pe[s] = std::min(e, 0);                        // the clamp keeps the scaled p.dsigma in [0, 1] whatever the lane
        }
        HIP_TRY(P->d_lane_pe.upload(pe));
    }
    HIP_TRY(P->d_cosphi.upload(cosphi));
    HIP_TRY(P->d_sinphi.upload(sinphi));
    HIP_TRY(P->d_kgrid.upload(kgrid));
    HIP_TRY(P->d_kweight.upload(kweight));
    {
        std::vector<double> kch(P->K), ksh(P->K);
        for (int k = 0; k < P->K; k++) { kch[k] = std::cosh(0.0 - kgrid[k]); ksh[k] = std::sinh(0.0 - kgrid[k]); }   // smooth_kernels.cpp:279-280 with y = 0
        HIP_TRY(P->d_kch.upload(kch));
        HIP_TRY(P->d_ksh.upload(ksh));
    }
    {
        std::vector<double> ck((size_t)7 * P->J), sk((size_t)7 * P->J);
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < P->J; j++) {
                ck[(size_t)k * P->J + j] = std::cos(((double)k + 1.0) * g->phi[j]);   // emissionfunction.cpp:1106
                sk[(size_t)k * P->J + j] = std::sin(((double)k + 1.0) * g->phi[j]);
            }
        HIP_TRY(P->d_coskphi.upload(ck));
        HIP_TRY(P->d_sinkphi.upload(sk));
        HIP_TRY(P->d_phiw.alloc(P->J));
        HIP_TRY(P->d_pTw.alloc(P->npT));
    }

    // ---- splines (deltafReader.cpp:300-322) ----
    const double *tabs[3] = {nullptr, nullptr, nullptr};
    int nspl;
    if (P->feqmod) {
        // df_mode 3: F, betabulk, betapi; df_mode 4 reads betapi only (the other two slots mirror it, unused)
        tabs[0] = (o->df_mode == 3) ? df->F : df->betapi;
        tabs[1] = (o->df_mode == 3) ? df->betabulk : df->betapi;
        tabs[2] = df->betapi;
        nspl = 3;
    } else if (!P->ce) { tabs[0] = df->c0; tabs[1] = df->c2; nspl = 2; }
    else { tabs[0] = df->F; tabs[1] = df->betabulk; tabs[2] = df->betapi; nspl = 3; }
    std::vector<double> xs(df->T, df->T + df->n_T);
    HIP_TRY(P->d_splx.upload(xs));
    P->spl.n = df->n_T;
    P->spl.x = P->d_splx.p;
    P->spl.nspl = nspl;
    for (int s = 0; s < 3; s++) { P->spl.y[s] = nullptr; P->spl.c[s] = nullptr; }
    for (int s = 0; s < nspl; s++) {
        std::vector<double> ys(tabs[s], tabs[s] + df->n_T), cs;
        if (!is3d::natural_cspline_init(xs, ys, cs)) return fail(IS3D_EINVAL, "spline construction failed");
        HIP_TRY(P->d_sply[s].upload(ys));
        HIP_TRY(P->d_splc[s].upload(cs));
        P->spl.y[s] = P->d_sply[s].p;
        P->spl.c[s] = P->d_splc[s].p;
    }
    if (P->baryon) {
        // full (mu_B, T) grids for the bilinear branch (deltafReader.cpp:412-484)
        const double *t5[5];
        if (!P->ce && !fq) { t5[0] = df->c0; t5[1] = df->c1; t5[2] = df->c2; t5[3] = df->c3; t5[4] = df->c4; }
        else { t5[0] = df->F; t5[1] = df->G; t5[2] = df->betabulk; t5[3] = df->betaV; t5[4] = df->betapi; }
        std::vector<double> bs(df->muB, df->muB + df->n_muB);
        HIP_TRY(P->d_bilT.upload(xs));
        HIP_TRY(P->d_bilB.upload(bs));
        P->bil.nT = df->n_T; P->bil.nB = df->n_muB;
        P->bil.swap = o->reference_bilinear_indexing != 0;
        P->bil.T = P->d_bilT.p; P->bil.muB = P->d_bilB.p;
        for (int k = 0; k < 5; k++) {
            std::vector<double> tv(t5[k], t5[k] + (size_t)df->n_T * df->n_muB);
            HIP_TRY(P->d_biltab[k].upload(tv));
            P->bil.tab[k] = P->d_biltab[k].p;
        }
    }
    if (P->feqmod) {
        P->ngl = fq->n_gla;
        P->detA_min = fq->deta_min;
        P->mass_pion0 = fq->mass_pion0;
        std::vector<double> gl((size_t)4 * P->ngl);
        for (int k = 0; k < P->ngl; k++) {
            gl[k] = fq->root1[k]; gl[P->ngl + k] = fq->weight1[k]; gl[2 * P->ngl + k] = fq->root2[k]; gl[3 * P->ngl + k] = fq->weight2[k];
        }
        HIP_TRY(P->d_gl.upload(gl));
        std::vector<double> jon;
        if (o->df_mode == 4) {
            std::vector<double> bp, l2, zz, cl, cz;
            is3d::jonah_tables(fq, bp, l2, zz, P->bp_max);
            // gsl_spline_init on (bulkPi_over_Peq, lambda_squared) and (bulkPi_over_Peq, z), deltafReader.cpp:311-320
            if (!is3d::natural_cspline_init(bp, l2, cl) || !is3d::natural_cspline_init(bp, zz, cz))
                return fail(IS3D_EINVAL, "df_mode 4: bulkPi/Peq(lambda) is not ascending at T_avg = %.6g GeV (GSL would abort here)", fq->T_avg);
            P->nj = (int)bp.size();
            for (const auto *v : {&bp, &l2, &zz, &cl, &cz}) jon.insert(jon.end(), v->begin(), v->end());
        } else {
            P->nj = 2;   // unused placeholder so that the prep kernel's LDS layout stays valid
            jon.assign(10, 0.0);
            jon[1] = 1.0;
        }
        HIP_TRY(P->d_jonah.upload(jon));
        std::vector<double> lane_mass(P->Lpad, 1.0);
        std::vector<int32_t> lane_cls(P->Lpad, 0);
        for (int sl = 0; sl < P->split; sl++)   // every slot of a bin carries the bin's class
            for (int s = 0; s < P->Lbins; s++) { lane_mass[sl * P->Lbins + s] = cmass[order[s] / P->npT]; lane_cls[sl * P->Lbins + s] = order[s] / P->npT; }
        HIP_TRY(P->d_lane_mass.upload(lane_mass));
        HIP_TRY(P->d_lane_cls.upload(lane_cls));
        HIP_TRY(P->d_cls_mass.upload(cmass));
        HIP_TRY(P->d_cls_sign.upload(csign));
        if (P->baryon) HIP_TRY(P->d_cls_baryon.upload(cbar));
    }

    // ---- tiling / workspace ----
    is3d::main_tile_shape(P->variant, P->dim3 ? 1 : 0, &P->JT, &P->KT);
    P->jtiles = (P->J + P->JT - 1) / P->JT;
    const bool tiled = P->variant != 1;
    if (!fq && is3d::prep_lds_bytes(df->n_T, nspl, P->J, P->K, P->baryon ? 1 : 0, tiled ? is3d::unit_rec_doubles(P->JT, P->KT, P->baryon ? 1 : 0) : 0, P->dim3 ? 1 : 0) > 160 * 1024)
        return fail(IS3D_EINVAL, "grids too large for the prep kernel's LDS staging");
    P->rblocks = tiled ? (P->K + P->KT - 1) / P->KT : 1;
    P->ktiles = P->dim3 ? (P->K + P->KT - 1) / P->KT : 1;   // k tiles are separate tasks only in 3+1D
    P->upc = (tiled && !P->dim3) ? P->rblocks : 1;            // 2+1D: eta blocks are consecutive units of one stream
    if (fq && is3d::prep_feqmod_lds_bytes(df->n_T, P->nj, P->ngl, P->J, P->K, P->jtiles, P->rblocks, is3d::unit_rec_doubles(P->JT, P->KT, 0)) > 160 * 1024)
        return fail(IS3D_EINVAL, "grids too large for the prep kernel's LDS staging");
    if (tiled)
        P->bytes_per_cell = sizeof(double) * (size_t)P->jtiles * P->rblocks * is3d::unit_rec_doubles(P->JT, P->KT, (P->baryon && !P->feqmod) ? 1 : 0);
    else
        P->bytes_per_cell = sizeof(double) * ((size_t)P->K * is3d::kS1Rec + (size_t)P->J * is3d::kS2Rec + (size_t)P->J * P->K);
    if (P->e2tab) P->bytes_per_cell += sizeof(double) * (size_t)P->jtiles * is3d::kE2Stride * P->JT;
    if (P->feqmod)   // fallback record, flag, list entry; df_mode 3: cell record + |renorm| per class
        P->bytes_per_cell += sizeof(double) * is3d::kFbRec + 2 * sizeof(int32_t) +
                             (o->df_mode == 3 ? sizeof(double) * ((size_t)is3d::kCrRec + P->ncls) : 0);
    // cap on the derived streams of one pass: the caller's, else 16 GiB or 45 % of the device's TOTAL memory, whichever is larger (288 GB of
    // HBM: a 1e6-cell surface with baryon slots -- 20.6 KB per cell -- stays a single pass; only what max_cells needs is allocated).  The
    // total, not what happens to be free: the pass count, the chunk count and with them the summation order of a surface that needs several
    // passes must not depend on what else occupies the GPU at the moment (the partial slab below adds up to 12 GiB on top; an allocation that
    // does not fit is reported as IS3D_ENOMEM with the sizes, and opts.workspace_bytes sets the cap explicitly)
    const int64_t ws = is3d::default_stream_cap_bytes(o->workspace_bytes);   // cf_launch.h: one rule for every plan
    int64_t pc = ws / (int64_t)P->bytes_per_cell;
    if (pc < 1) pc = 1;
    if (pc > max_cells) pc = max_cells;
    if (pc > 0x7fff0000LL) pc = 0x7fff0000LL;
    P->pass_cells = pc;
    P->max_cells = max_cells;

    // cell chunks: enough wave-tasks for >= ~24 rounds of the chip, chunks of >= 64 cells,
    // partial buffer <= 2 GiB
    {
        const int lane_waves = P->Lpad / 64;
        // waves per workgroup: all waves of a workgroup stream the same records; idle waves (lane-wave count not a
        // multiple) still occupy their SIMD slots, so take the size that wastes fewest, the larger one on ties
        P->wpb = 4;
        if (P->variant != 1) {
            int best_waste = 1 << 30;
            for (int w : {8, 4, 2}) {
                int waste = ((lane_waves + w - 1) / w) * w - lane_waves;
                if (waste * 64 < best_waste * 64 && waste < best_waste) { best_waste = waste; P->wpb = w; }
            }
            if (o->waves_per_group == 2 || o->waves_per_group == 4 || o->waves_per_group == 8) P->wpb = o->waves_per_group;
            if (o->waves_per_group == 1 && P->e2tab) P->wpb = 1;   // cf_main_tile3e: one-wave workgroups (no barrier partner)
            if (P->variant == 9) P->wpb = 1;                       // cf_main_tile3s: a workgroup IS one wave
            // cf_main_feqmod, 3+1D 8 x 7 without baryon slots: one-wave workgroups (cf_feqmod.hip, LDSD); the pipelined A/B form (variant 5) keeps two
            // -- the default there since round 4 (486 against 501 ms on the config-3 surface, profiles/r04_ab_feqmod.log); waves_per_group = 2 keeps the pair
            if (P->feqmod && P->dim3 && (P->variant == 3 || P->variant == 6) && (o->waves_per_group == 1 || o->waves_per_group == 0)) P->wpb = 1;
        }
        const int64_t tasks_per_chunk = (int64_t)lane_waves * P->jtiles * P->ktiles;
        const int64_t capacity = 256LL * 4 * 4;  // CUs x SIMDs x ~4 waves
        static const int64_t kChunkRounds = [] { const char *e = is3d::dev_env("IS3D_CHUNK_ROUNDS"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 12); }();   // dev A/B
        // Chunk count.  Two floors: enough tasks for ~12 rounds of the chip (load balance: lane-wave groups cull differently; 24 until the end of
        // round 3 -- on a 125 000-cell shard 144 instead of 288 chunks run the same 43.1 ms and reduce 0.3 ms less), and chunks of at
        // most ~1152 cells -- the streams of one (phi tile, chunk) pair are then ~5 MB, the readers' drift along them stays about the size of an
        // XCD's 4 MB L2, and the fabric / HBM-side traffic of the main kernel drops 3x (config 3: FETCH_SIZE 1.82e8 -> 5.99e7 KiB per launch)
        // at an unchanged step time (the main kernel gains what the 1.3 ms of extra partial-slab reduction cost; 1.0 point less culling because
        // every chunk warms its thresholds up anew).  profiles/r03_chunks.log
        int64_t nch = o->cell_chunks > 0 ? o->cell_chunks
                                         : std::max<int64_t>((kChunkRounds * capacity + tasks_per_chunk - 1) / tasks_per_chunk, (pc + 1151) / 1152);
        int64_t by_cells = std::max<int64_t>(1, pc / 64);
        nch = std::min(nch, by_cells);
        int64_t part_bytes_per_chunk = (int64_t)P->J * P->Kacc * P->Lpad * (int64_t)sizeof(double);
        // partial buffer: <= 12 GiB by default (one 9.8 MB slab per chunk on config 3: 869 chunks = 8.5 GB; it was 2 GiB = 219 chunks until round 3);
        // an explicit cell_chunks request may take up to 32 GiB
        int64_t by_mem = std::max<int64_t>(1, ((int64_t)(o->cell_chunks > 0 ? 32 : 12) << 30) / part_bytes_per_chunk);
        nch = std::max<int64_t>(1, std::min(nch, by_mem));
        P->nch_max = (int)nch;
    }
#define BIG_ALLOC(buf, count, what)                                                                                                  \
    do {                                                                                                                              \
        const size_t n_ = (count);                                                                                                    \
        hipError_t e_ = (buf).alloc(n_);                                                                                              \
        if (e_ == hipErrorOutOfMemory) {                                                                                              \
            (void)hipGetLastError();                                                                                                  \
            size_t fr_ = 0, to_ = 0;                                                                                                  \
            (void)hipMemGetInfo(&fr_, &to_);                                                                                          \
            return fail(IS3D_ENOMEM, "out of device memory allocating %s (%.2f GB; %.2f GB free of %.2f GB): %lld cells per pass x %zu B of streams, " \
                        "%d partial slabs -- lower opts.workspace_bytes (more passes) or opts.cell_chunks", what, n_ * sizeof(*(buf).p) / 1e9, \
                        fr_ / 1e9, to_ / 1e9, (long long)pc, P->bytes_per_cell, P->nch_max);                                          \
        }                                                                                                                             \
        if (e_ != hipSuccess) return fail(IS3D_ENODEVICE, "hipMalloc(%s) failed: %s", what, hipGetErrorString(e_));                  \
    } while (0)
    if (tiled) {
        // unpredicated direct-to-LDS staging over-reads a short last batch (cf_main_tile3e, cf_main_tile variant 8)
        const size_t slack = (size_t)is3d::tile3e_stream_slack_doubles(P->JT, P->KT) + 16 * (size_t)is3d::unit_rec_doubles(P->JT, P->KT, 1);
        BIG_ALLOC(P->d_TS, (size_t)pc * P->jtiles * P->rblocks * is3d::unit_rec_doubles(P->JT, P->KT, (P->baryon && !P->feqmod) ? 1 : 0) + slack, "the unit-record stream");
        if (P->e2tab) {
            BIG_ALLOC(P->d_TE, (size_t)pc * P->jtiles * is3d::kE2Stride * P->JT + slack, "the E2 table stream");
            P->ub3e = is3d::tile3e_units_per_batch(P->JT, P->KT, P->npT, P->wpb, P->baryon ? 1 : 0, P->variant == 10 ? 1 : P->variant == 12 ? 2 : 0);
            if (P->ub3e < 1) return fail(IS3D_EINVAL, "kernel_variant 5: a unit record plus its %d x %d E2 table does not fit the LDS budget", P->npT, P->JT);
        }
    } else {
        HIP_TRY(P->d_S1.alloc((size_t)pc * P->K * is3d::kS1Rec));
        HIP_TRY(P->d_S2.alloc((size_t)pc * P->J * is3d::kS2Rec));
        HIP_TRY(P->d_S3.alloc((size_t)pc * P->J * P->K));
    }
    BIG_ALLOC(P->d_partial, (size_t)(P->nch_max + is3d::kTaperExtra /* the tapered tail: chunk_plan */) * P->J * P->Kacc * P->Lpad, "the per-chunk partial spectra");
#undef BIG_ALLOC
    if (P->e2tab) HIP_TRY(P->d_cull_floor.alloc((size_t)P->jtiles * P->ktiles * P->Lpad));
    HIP_TRY(P->d_status.alloc(8));
    HIP_TRY(P->d_sticky.upload(std::vector<unsigned long long>{~0ULL, ~0ULL}));
    if (P->feqmod) {
        HIP_TRY(P->d_FB.alloc((size_t)pc * is3d::kFbRec));
        HIP_TRY(P->d_flag.alloc((size_t)pc));
        HIP_TRY(P->d_list.alloc((size_t)pc));
        HIP_TRY(P->d_count.alloc(1));
        if (o->df_mode == 3) {
            HIP_TRY(P->d_CR.alloc((size_t)pc * is3d::kCrRec));
            HIP_TRY(P->d_RN.alloc((size_t)pc * P->ncls));
        }
    }
    P->workspace = (int64_t)(P->d_S1.n + P->d_S2.n + P->d_S3.n + P->d_TS.n + P->d_TE.n + P->d_partial.n + P->d_FB.n + P->d_CR.n + P->d_RN.n) * (int64_t)sizeof(double) +
                   (int64_t)(P->d_flag.n + P->d_list.n) * (int64_t)sizeof(int32_t);
    *out = P.release();
    return IS3D_OK;
}

extern "C" int is3d_plan_create(is3d_plan **out, const is3d_species *sp, const is3d_grid *g, const is3d_df_tables *df,
                                const is3d_options *o, int64_t max_cells)
{
    return plan_create_impl(out, sp, g, df, nullptr, o, max_cells);
}

extern "C" int is3d_plan_create_feqmod(is3d_plan **out, const is3d_species *sp, const is3d_grid *g, const is3d_df_tables *df,
                                       const is3d_feqmod_tables *fq, const is3d_options *o, int64_t max_cells)
{
    if (!fq) return fail(IS3D_EINVAL, "null feqmod tables");
    return plan_create_impl(out, sp, g, df, fq, o, max_cells);
}

extern "C" int is3d_probe_shader_clock(int32_t device, double seconds, double *ghz)
{
    if (!ghz || !(seconds > 0.0) || seconds > 10.0) return fail(IS3D_EINVAL, "is3d_probe_shader_clock: ghz == NULL or seconds outside (0, 10]");
    *ghz = 0.0;
    HIP_TRY(hipSetDevice(device));
    int khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
    if (khz <= 0) return fail(IS3D_ENODEVICE, "is3d_probe_shader_clock: the device reports no wall clock rate");
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *d = nullptr, h[16] = {0};
    is3d::count_resource(1);
    hipError_t e = hipMalloc(&d, sizeof h);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, sizeof h, st);
    if (e == hipSuccess) e = is3d::launch_clock_probe((unsigned long long)(seconds * khz * 1e3), d, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d) (void)hipFree(d);
    (void)hipStreamDestroy(st);
    if (e != hipSuccess) return fail(IS3D_ENODEVICE, "is3d_probe_shader_clock: %s", hipGetErrorString(e));
    double sum = 0.0;
    int n = 0;
    for (int i = 0; i < 8; i++)
        if (h[2 * i + 1]) { sum += (double)h[2 * i] / (double)h[2 * i + 1]; n++; }
    const double ratio = n ? sum / n : 0.0;                 // shader ticks per reference tick
    *ghz = (ratio > 1.001 || ratio < 0.999) ? ratio * khz * 1e-6 : 0.0;
    return IS3D_OK;
}

extern "C" int is3d_math_probe(int32_t which, int64_t n, const double *x, double *y, int32_t device)
{
    if (which < 0 || which > 8 || n < 0 || (n > 0 && (!x || !y))) return fail(IS3D_EINVAL, "is3d_math_probe: which in 0..8, n >= 0, non-null arrays");
    if (is3d_device_count() < 1) return fail(IS3D_ENODEVICE, "no HIP device visible; this library has no CPU path");
    if (n == 0) return IS3D_OK;
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    DevBuf<double> dx, dy;
    HIP_TRY(dx.alloc((size_t)n));
    HIP_TRY(dy.alloc((size_t)n));
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(is3d::launch_math_probe(which, n, dx.p, dy.p, nullptr));
    HIP_TRY(hipMemcpy(y, dy.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return IS3D_OK;
}

extern "C" int64_t is3d_plan_output_size(const is3d_plan *P) { return P ? P->nout : 0; }
extern "C" int64_t is3d_plan_workspace_bytes(const is3d_plan *P) { return P ? P->workspace : 0; }
extern "C" const char *is3d_plan_main_kernel_name(const is3d_plan *P)
{
    if (P && P->feqmod) return "cf_main_feqmod";
    return is3d::main_kernel_name(P ? P->variant : 2);
}
extern "C" int is3d_plan_tile_shape(const is3d_plan *P, int32_t *JT, int32_t *R)
{
    if (!P || !JT || !R) return fail(IS3D_EINVAL, "null argument");
    *JT = P->JT;
    *R = P->KT;
    return IS3D_OK;
}
extern "C" int is3d_plan_set_timing(is3d_plan *P, int32_t enable)
{
    if (!P) return fail(IS3D_EINVAL, "null plan");
    P->timing = enable != 0;
    return IS3D_OK;
}
namespace is3d {
int plan_accumulate(const is3d_plan *P) { return P && P->opts.accumulate != 0; }
int plan_device(const is3d_plan *P) { return P ? P->device : -1; }
}
extern "C" void is3d_plan_destroy(is3d_plan *P)
{
    if (!P) return;
    (void)hipSetDevice(P->device);
    delete P;
}

static int chunks_for(const is3d_plan *P, int64_t n)
{
    int64_t by_cells = std::max<int64_t>(1, n / 64);
    return (int)std::max<int64_t>(1, std::min<int64_t>(P->nch_max, by_cells));
}

// ... with the tapered tail of the cell partition (cf_device.h: chunk_cells, chunk_taper): the grid of the main kernel drains in quarter-length
// tasks -- the fixed ~2.7 ms a launch cost beyond its per-cell time (profiles/r04_shard_sizes.json: 329.0 / 166.2 / 85.5 / 43.5 ms main at 1e6 /
// 5e5 / 2.5e5 / 1.25e5 cells; profiles/r04_ab_chunk_taper.log)
static void chunk_plan(const is3d_plan *P, int64_t n, int &nch, int &nsmall)
{
    nch = chunks_for(P, n);
    nsmall = is3d::chunk_taper(n, P->opts.cell_chunks, nch);
}

extern "C" int is3d_plan_execute(is3d_plan *P, const is3d_cells *cells, double *dN_out, void *hip_stream, is3d_status *status)
{
    if (!P || !cells || !dN_out) return fail(IS3D_EINVAL, "null argument");
    if (status) { memset(status, 0, sizeof *status); status->bad_cell = -1; }
    const int64_t n = cells->n_cells;
    if (n < 0 || n > P->max_cells) return fail(IS3D_EINVAL, "n_cells = %lld exceeds the plan's max_cells = %lld", (long long)n, (long long)P->max_cells);
    const is3d_options &o = P->opts;
    const bool need_eta = P->dim3;
    if (n > 0) {
        if (!cells->tau || !cells->dat || !cells->dax || !cells->day || !cells->dan || !cells->ux || !cells->uy || !cells->un ||
            !cells->T || !cells->P || !cells->E || (need_eta && !cells->eta))
            return fail(IS3D_EINVAL, "a required cell array is NULL");
        if (o.include_shear_deltaf && (!cells->pixx || !cells->pixy || !cells->pixn || !cells->piyy || !cells->piyn))
            return fail(IS3D_EINVAL, "include_shear_deltaf needs pixx, pixy, pixn, piyy, piyn");
        if (o.include_bulk_deltaf && !cells->bulkPi) return fail(IS3D_EINVAL, "include_bulk_deltaf needs bulkPi");
        if (P->baryondiff && (!cells->muB || !cells->nB || !cells->Vx || !cells->Vy || !cells->Vn))
            return fail(IS3D_EINVAL, "include_baryon && include_baryondiff_deltaf need muB, nB, Vx, Vy, Vn");
    }
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(P->device));

    const int npasses = n == 0 ? 0 : (int)((n + P->pass_cells - 1) / P->pass_cells);
    if (P->timing) {
        size_t need = (size_t)npasses * 3 + 1;
        while (P->ev_list.size() < need) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            P->ev_list.push_back(e);
        }
    }
    P->last_passes = npasses;
    unsigned long long init[8] = {~0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, ~0ULL};
    HIP_TRY(hipMemcpyAsync(P->d_status.p, init, sizeof init, hipMemcpyHostToDevice, st));

    int nch_used = 1, nch_small = 0;
    // tiled delta-f stream: p.dsigma travels times 2^-e (status[6] = bits of the bound, cf_device.h)
    const bool use_scale = !P->feqmod && P->variant != 1;
    if (n == 0) {
        // empty surface: spectrum is zero (reference: loops do not execute)
        if (!o.accumulate) HIP_TRY(hipMemsetAsync(dN_out, 0, (size_t)P->nout * sizeof(double), st));
    } else {
        // all passes use the chunk count of the first (largest) pass so that partial slots line up
        chunk_plan(P, std::min<int64_t>(n, P->pass_cells), nch_used, nch_small);
        if (use_scale) {
            is3d::CellPtrs cp{};
            cp.tau = cells->tau; cp.eta = cells->eta; cp.dat = cells->dat; cp.dax = cells->dax; cp.day = cells->day; cp.dan = cells->dan;
            HIP_TRY(is3d::launch_pds_bound(cp, n, P->dim3, P->kmin, P->kmax, P->gw2d, P->mTmax, P->pTmax, P->d_status.p + 6, st));
        }
        for (int pass = 0; pass < npasses; pass++) {
            const int64_t c0 = (int64_t)pass * P->pass_cells;
            const int32_t nc = (int32_t)std::min<int64_t>(P->pass_cells, n - c0);
            if (P->feqmod) {
                is3d::FqPrepParams fp{};
                fp.cells = {cells->tau, cells->eta, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un,
                            cells->T, cells->P, cells->E, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn, cells->bulkPi,
                            cells->muB, cells->nB, cells->Vx, cells->Vy, cells->Vn};
                fp.baryon = P->baryon; fp.baryondiff = P->baryondiff; fp.bil = P->bil;
                fp.cell0 = c0; fp.n_cells = nc; fp.J = P->J; fp.K = P->K;
                fp.dim3 = P->dim3; fp.mode = o.df_mode;
                fp.include_bulk = o.include_bulk_deltaf != 0; fp.include_shear = o.include_shear_deltaf != 0;
                fp.cosphi = P->d_cosphi.p; fp.sinphi = P->d_sinphi.p; fp.kgrid = P->d_kgrid.p; fp.kweight = P->d_kweight.p;
                fp.spl = P->spl;
                fp.nj = P->nj;
                fp.jx = P->d_jonah.p; fp.jl2 = fp.jx + P->nj; fp.jz = fp.jx + 2 * P->nj; fp.jcl = fp.jx + 3 * P->nj; fp.jcz = fp.jx + 4 * P->nj;
                fp.bp_max = P->bp_max;
                fp.mTmax = P->mTmax; fp.kmin = P->kmin; fp.kmax = P->kmax;
                fp.pTmax = P->pTmax; fp.scale_rows = (o.df_mode == 4 && o.outflow != 0) ? 1 : 0;   // cf_main_feqmod's CLAMP instantiations
                fp.ngl = P->ngl; fp.gl = P->d_gl.p;
                fp.detA_min = P->detA_min; fp.mass_pion0 = P->mass_pion0;
                fp.JT = P->JT; fp.R = P->KT; fp.jtiles = P->jtiles; fp.rblocks = P->rblocks;
                fp.TS = P->d_TS.p; fp.CR = P->d_CR.p; fp.FB = P->d_FB.p; fp.flag = P->d_flag.p;
                fp.status = P->d_status.p;
                if (P->timing) HIP_TRY(hipEventRecord(P->ev_list[pass * 3 + 0], st));
                HIP_TRY(is3d::launch_prep_feqmod(fp, st));
                if (o.df_mode == 3)
                    HIP_TRY(is3d::launch_feqmod_renorm(P->d_CR.p, P->d_gl.p, P->ngl, P->d_cls_mass.p, P->d_cls_sign.p,
                                                       P->baryon ? P->d_cls_baryon.p : nullptr, P->ncls, nc, fp.include_bulk, P->dim3,
                                                       P->d_RN.p, st));
                if (P->timing) HIP_TRY(hipEventRecord(P->ev_list[pass * 3 + 1], st));
                is3d::FqMainArgs a{};
                a.TS = P->d_TS.p; a.lane_mT = P->d_mT.p; a.lane_pT = P->d_pT.p; a.lane_sign = P->d_sign.p;
                a.RN = P->d_RN.p; a.lane_cls = P->d_lane_cls.p; a.ncls = P->ncls;
                a.lane_sub = P->d_lane_sub.p; a.g.split = P->split;
                a.lane_b = P->baryon ? P->d_lane_b.p : nullptr;
                a.partial = P->d_partial.p; a.stats = P->d_status.p;
                a.g.upc = P->upc; a.g.zskip = (o.zero_skip == 2) ? 0 : (o.zero_skip == 1 ? 1 : 2); a.g.baryon = 0;
                a.g.n_cells = nc; a.g.J = P->J; a.g.K = P->K; a.g.Lpad = P->Lpad; a.g.wpb = P->wpb;
                a.g.G = (P->Lpad / 64 + P->wpb - 1) / P->wpb;
                a.g.jtiles = P->jtiles; a.g.ktiles = P->ktiles; a.g.nch = nch_used; a.g.nch_small = nch_small;
                a.g.NT = P->jtiles * P->ktiles * nch_used; a.g.Kacc = P->Kacc; a.g.first_pass = (pass == 0);
                HIP_TRY(is3d::launch_main_feqmod(P->variant, P->dim3, o.outflow != 0, o.df_mode == 3, P->baryon ? 1 : 0, a, st));
                // flagged cells (breakdown, narrow rows): ordered list, then the linearised delta-f on top of chunk 0
                HIP_TRY(is3d::launch_feqmod_compact(P->d_flag.p, nc, P->d_list.p, P->d_count.p, P->d_status.p, st));
                is3d::FqLinearArgs la{};
                la.FB = P->d_FB.p; la.list = P->d_list.p; la.count = P->d_count.p;
                la.lane_mT = P->d_mT.p; la.lane_pT = P->d_pT.p; la.lane_sign = P->d_sign.p; la.lane_mass = P->d_lane_mass.p;
                la.lane_b = P->baryon ? P->d_lane_b.p : nullptr;
                la.cosphi = P->d_cosphi.p; la.sinphi = P->d_sinphi.p; la.kgrid = P->d_kgrid.p; la.kweight = P->d_kweight.p;
                la.partial = P->d_partial.p;
                la.J = P->J; la.K = P->K; la.Kacc = P->Kacc; la.Lpad = P->Lpad; la.dim3 = P->dim3; la.mode = o.df_mode; la.Lbins = P->Lbins;
                la.outflow = o.outflow != 0; la.regulate = o.regulate_deltaf != 0;
                HIP_TRY(is3d::launch_feqmod_linear(la, st));
                if (P->timing) HIP_TRY(hipEventRecord(P->ev_list[pass * 3 + 2], st));
                continue;
            }
            is3d::PrepParams pp{};
            pp.cells = {cells->tau, cells->eta, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un,
                        cells->T, cells->P, cells->E, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn, cells->bulkPi};
            pp.cell0 = c0;
            pp.n_cells = nc;
            pp.J = P->J; pp.K = P->K;
            pp.dim3 = P->dim3; pp.ce = P->ce;
            pp.include_bulk = o.include_bulk_deltaf != 0;
            pp.include_shear = o.include_shear_deltaf != 0;
            pp.baryon = P->baryon; pp.baryondiff = P->baryondiff;
            pp.bil = P->bil;
            pp.cells.muB = cells->muB; pp.cells.nB = cells->nB; pp.cells.Vx = cells->Vx; pp.cells.Vy = cells->Vy; pp.cells.Vn = cells->Vn;
            pp.cosphi = P->d_cosphi.p; pp.sinphi = P->d_sinphi.p;
            pp.kgrid = P->d_kgrid.p; pp.kweight = P->d_kweight.p; pp.kch = P->d_kch.p; pp.ksh = P->d_ksh.p;
            pp.spl = P->spl;
            pp.S1 = P->d_S1.p; pp.S2 = P->d_S2.p; pp.S3 = P->d_S3.p;
            pp.tiled = (P->variant != 1);
            pp.JT = P->JT; pp.R = P->KT; pp.jtiles = P->jtiles; pp.rblocks = P->rblocks;
            pp.TS = P->d_TS.p;
            pp.pds_bound = use_scale ? P->d_status.p + 6 : nullptr;
            pp.mTmax = P->mTmax; pp.kmin = P->kmin; pp.kmax = P->kmax;
            pp.status = P->d_status.p;
            pp.TE = P->e2tab ? P->d_TE.p : nullptr; pp.pTgrid = P->d_pTgrid.p; pp.npT = P->npT;
            if (P->timing) HIP_TRY(hipEventRecord(P->ev_list[pass * 3 + 0], st));
            HIP_TRY(is3d::launch_prep(pp, st));
            if (P->timing) HIP_TRY(hipEventRecord(P->ev_list[pass * 3 + 1], st));

            is3d::MainArgs a{};
            a.S1 = P->d_S1.p; a.S2 = P->d_S2.p; a.S3 = P->d_S3.p; a.TS = P->d_TS.p;
            a.g.upc = P->upc;
            a.g.zskip = (o.zero_skip == 2) ? 0 : (o.zero_skip == 1 ? 1 : 2);   // 0 default: exact-zero + accumulator-relative culling; 3: + surface-relative floors (below)
            a.lane_mT = P->d_mT.p; a.lane_pT = P->d_pT.p; a.lane_sign = P->d_sign.p; a.lane_b = P->d_lane_b.p;
            a.g.baryon = P->baryon;
            a.lane_pe = P->d_lane_pe.p;
            a.partial = P->d_partial.p;
            a.stats = P->d_status.p;
            a.g.n_cells = nc;
            a.g.J = P->J; a.g.K = P->K;
            a.g.Lpad = P->Lpad;
            a.g.wpb = P->wpb;
            a.g.G = (P->Lpad / 64 + P->wpb - 1) / P->wpb;
            a.g.jtiles = P->jtiles; a.g.ktiles = P->ktiles;
            a.g.nch = nch_used; a.g.nch_small = nch_small;
            a.g.NT = P->jtiles * P->ktiles * nch_used;
            a.g.Kacc = P->Kacc;
            a.g.first_pass = (pass == 0);
            a.TE = P->e2tab ? P->d_TE.p : nullptr; a.lane_ipT = P->d_lane_ipT.p; a.g.npT = P->npT; a.g.ub = P->ub3e; a.pTgrid = P->e2tab ? P->d_pTgrid.p : nullptr;
            a.g.split = P->split; a.lane_sub = P->d_lane_sub.p;
            // zero_skip 3, surface-relative cull (cf_main_tile3e with outflow && regulate_deltaf; include/is3d_amd.h): an eighth of the chunks
            // runs first with the accumulator-relative rule; their partial spectrum is a lower bound of the final one (all terms >= 0) and
            // floors the row-cull thresholds of the other chunks, which would otherwise each start from an empty accumulator
            static const int surf_den = [] { const char *e = is3d::dev_env("IS3D_SURFCULL_DEN"); const int v = e ? atoi(e) : 0; return v >= 2 ? v : 8; }();
            const bool surf = o.zero_skip == 3 && P->e2tab && o.outflow != 0 && o.regulate_deltaf != 0 && nch_used >= 2 * surf_den;
            if (surf) {
                const int nA = nch_used / surf_den;
                a.g.ch0 = 0; a.g.nch_run = nA;
                HIP_TRY(is3d::launch_main(P->variant, P->ce, P->dim3, true, true, a, st));
                HIP_TRY(is3d::launch_cull_floor(P->d_partial.p, nA, P->J, P->K, P->Kacc, P->Lpad, P->JT, P->KT, P->jtiles, P->ktiles, P->d_lane_pe.p,
                                                2.0, P->d_cull_floor.p, st));
                a.g.ch0 = nA; a.g.nch_run = nch_used - nA;
                a.cull_floor = P->d_cull_floor.p;
                HIP_TRY(is3d::launch_main(P->variant, P->ce, P->dim3, true, true, a, st));
            } else {
                HIP_TRY(is3d::launch_main(P->variant, P->ce, P->dim3, o.outflow != 0, o.regulate_deltaf != 0, a, st));
            }
            if (P->timing) HIP_TRY(hipEventRecord(P->ev_list[pass * 3 + 2], st));
        }
        HIP_TRY(is3d::launch_finalize(P->d_partial.p, P->d_cls.p, P->d_degeneracy.p, dN_out, P->nout, P->npart, P->npT, P->J,
                                      P->Kacc, P->Lpad, nch_used, P->prefactor, o.accumulate != 0,
                                      use_scale ? P->d_status.p + 6 : nullptr, st, P->split, P->Lbins));
        if (P->timing) HIP_TRY(hipEventRecord(P->ev_list[npasses * 3], st));
    }

    // executes that return their domain errors through `status` do not leave them for a later is3d_plan_check as well
    if (n > 0 && !status) HIP_TRY(is3d::launch_fold_status(P->d_status.p, P->d_sticky.p, st));
    if (status) {
        unsigned long long h[8];
        HIP_TRY(hipMemcpyAsync(h, P->d_status.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        status->n_classes = P->ncls;
        status->n_passes = npasses;
        status->kernel_variant = P->variant;
        status->n_cells_skipped = (int64_t)h[1];
        status->n_wave_rows = (int64_t)h[2];
        status->n_wave_rows_culled = (int64_t)h[3];
        status->n_cells_breakdown = (int64_t)h[4];
        status->n_cells_narrow = (int64_t)h[5];
        status->bad_cell = (h[0] == ~0ULL) ? -1 : (int64_t)h[0];
        if (h[7] != ~0ULL && (status->bad_cell < 0 || (int64_t)h[7] < status->bad_cell)) {
            status->bad_cell = (int64_t)h[7];
            status->code = IS3D_EDOMAIN;
            return fail(IS3D_EDOMAIN, "cell %lld: p.u/T can exceed 1e9 for the momentum grid (flow velocity / temperature outside the "
                        "kernel's exponent range; the reference's exp() overflows to inf there)", (long long)status->bad_cell);
        }
        if (status->bad_cell >= 0) {
            status->code = IS3D_EDOMAIN;
            return fail(IS3D_EDOMAIN, "cell %lld: T%s outside the coefficient table (the reference aborts in gsl_spline_eval here)",
                        (long long)status->bad_cell, (P->feqmod && o.df_mode == 4) ? " (or bulkPi/P)" : "");
        }
    }
    return IS3D_OK;
}

extern "C" int is3d_plan_check(is3d_plan *P, void *hip_stream, int64_t *bad_cell)
{
    if (!P) return fail(IS3D_EINVAL, "null plan");
    if (bad_cell) *bad_cell = -1;
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(P->device));
    unsigned long long h[2], reset[2] = {~0ULL, ~0ULL};
    HIP_TRY(hipMemcpyAsync(h, P->d_sticky.p, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(P->d_sticky.p, reset, sizeof reset, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    const unsigned long long first = std::min(h[0], h[1]);
    if (first == ~0ULL) return IS3D_OK;
    if (bad_cell) *bad_cell = (int64_t)first;
    if (h[1] <= h[0])
        return fail(IS3D_EDOMAIN, "cell %llu (of an execute since the last check): p.u/T can exceed 1e9 for the momentum grid; the cell was left "
                    "out of the spectrum (the reference's exp() overflows to inf there)", first);
    return fail(IS3D_EDOMAIN, "cell %llu (of an execute since the last check): T%s outside the coefficient table; the cell was left out of the "
                "spectrum (the reference aborts in gsl_spline_eval here)", first, (P->feqmod && P->opts.df_mode == 4) ? " (or bulkPi/P)" : "");
}

extern "C" int is3d_plan_observables(is3d_plan *P, const double *dN_dev, const double *pT_w, const double *phi_w, double *dNdy_dev,
                                     double *dN2pipTdpTdy_dev, double *vn_dev, void *hip_stream)
{
    if (!P || !dN_dev || !phi_w) return fail(IS3D_EINVAL, "null argument");
    if (dNdy_dev && !pT_w) return fail(IS3D_EINVAL, "dN/dy needs the pT weights");
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipMemcpyAsync(P->d_phiw.p, phi_w, (size_t)P->J * sizeof(double), hipMemcpyHostToDevice, st));
    if (pT_w) HIP_TRY(hipMemcpyAsync(P->d_pTw.p, pT_w, (size_t)P->npT * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(is3d::launch_observables(dN_dev, P->d_phiw.p, P->d_pTw.p, P->d_coskphi.p, P->d_sinkphi.p, dNdy_dev, dN2pipTdpTdy_dev,
                                     vn_dev, P->npart, P->npT, P->J, P->ny_eff, st));
    return IS3D_OK;
}

extern "C" int is3d_plan_timings(is3d_plan *P, is3d_status *status)
{
    if (!P || !status) return fail(IS3D_EINVAL, "null argument");
    status->ms_prep = status->ms_main = status->ms_finalize = 0.0;
    if (!P->timing || P->last_passes == 0) return IS3D_OK;
    HIP_TRY(hipSetDevice(P->device));
    HIP_TRY(hipEventSynchronize(P->ev_list[P->last_passes * 3]));
    for (int pass = 0; pass < P->last_passes; pass++) {
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, P->ev_list[pass * 3 + 0], P->ev_list[pass * 3 + 1]));
        HIP_TRY(hipEventElapsedTime(&b, P->ev_list[pass * 3 + 1], P->ev_list[pass * 3 + 2]));
        status->ms_prep += a;
        status->ms_main += b;
    }
    float c = 0;
    HIP_TRY(hipEventElapsedTime(&c, P->ev_list[(P->last_passes - 1) * 3 + 2], P->ev_list[P->last_passes * 3]));
    status->ms_finalize = c;
    status->n_passes = P->last_passes;
    status->kernel_variant = P->variant;
    status->n_classes = P->ncls;
    return IS3D_OK;
}

// ---------------------------------------------------------------------------------------------
// one-shot host entry
// ---------------------------------------------------------------------------------------------
static int smooth_spectra_impl(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                               const is3d_df_tables *df, const is3d_feqmod_tables *fq, const is3d_options *opts, double *dN_out,
                               is3d_status *status)
{
    if (!cells || !dN_out) return fail(IS3D_EINVAL, "null argument");
    is3d_plan *P = nullptr;
    int rc = plan_create_impl(&P, species, grid, df, fq, opts, std::max<int64_t>(cells->n_cells, 1));
    if (rc) { if (status) { memset(status, 0, sizeof *status); status->code = rc; status->bad_cell = -1; } return rc; }
    struct Guard { is3d_plan *p; ~Guard() { is3d_plan_destroy(p); } } guard{P};
    (void)is3d_plan_set_timing(P, 1);

    const int64_t n = cells->n_cells;
    const bool diff = opts->include_baryon && opts->include_baryondiff_deltaf;
    const double *src[23] = {cells->tau, cells->eta, cells->dat, cells->dax, cells->day, cells->dan, cells->ux, cells->uy, cells->un,
                             cells->T, cells->P, cells->E, cells->pixx, cells->pixy, cells->pixn, cells->piyy, cells->piyn, cells->bulkPi,
                             diff ? cells->muB : nullptr, diff ? cells->nB : nullptr, diff ? cells->Vx : nullptr,
                             diff ? cells->Vy : nullptr, diff ? cells->Vn : nullptr};
    DevBuf<double> dcell, dout;
    HIP_TRY(dcell.alloc((size_t)std::max<int64_t>(n, 1) * 23));
    HIP_TRY(dout.alloc((size_t)P->nout));
    hipEvent_t e0, e1, e2, e3;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2)); HIP_TRY(hipEventCreate(&e3));
    struct EvGuard { hipEvent_t e[4]; ~EvGuard() { for (auto x : e) (void)hipEventDestroy(x); } } evg{{e0, e1, e2, e3}};
    HIP_TRY(hipEventRecord(e0, nullptr));
    const double *dptr[23];
    for (int a = 0; a < 23; a++) {
        dptr[a] = nullptr;
        if (src[a] && n > 0) {
            HIP_TRY(hipMemcpyAsync(dcell.p + (size_t)a * n, src[a], (size_t)n * sizeof(double), hipMemcpyHostToDevice, nullptr));
            dptr[a] = dcell.p + (size_t)a * n;
        }
    }
    if (opts->accumulate) HIP_TRY(hipMemcpyAsync(dout.p, dN_out, (size_t)P->nout * sizeof(double), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipEventRecord(e1, nullptr));
    is3d_cells dc{};
    dc.n_cells = n;
    dc.tau = dptr[0]; dc.eta = dptr[1]; dc.dat = dptr[2]; dc.dax = dptr[3]; dc.day = dptr[4]; dc.dan = dptr[5];
    dc.ux = dptr[6]; dc.uy = dptr[7]; dc.un = dptr[8]; dc.T = dptr[9]; dc.P = dptr[10]; dc.E = dptr[11];
    dc.pixx = dptr[12]; dc.pixy = dptr[13]; dc.pixn = dptr[14]; dc.piyy = dptr[15]; dc.piyn = dptr[16]; dc.bulkPi = dptr[17];
    dc.muB = dptr[18]; dc.nB = dptr[19]; dc.Vx = dptr[20]; dc.Vy = dptr[21]; dc.Vn = dptr[22];
    is3d_status st{};
    rc = is3d_plan_execute(P, &dc, dout.p, nullptr, &st);
    if (rc) { if (status) *status = st; return rc; }
    HIP_TRY(hipEventRecord(e2, nullptr));
    HIP_TRY(hipMemcpyAsync(dN_out, dout.p, (size_t)P->nout * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipEventRecord(e3, nullptr));
    HIP_TRY(hipEventSynchronize(e3));
    (void)is3d_plan_timings(P, &st);
    float h2d = 0, d2h = 0;
    HIP_TRY(hipEventElapsedTime(&h2d, e0, e1));
    HIP_TRY(hipEventElapsedTime(&d2h, e2, e3));
    st.ms_h2d = h2d;
    st.ms_d2h = d2h;
    st.code = IS3D_OK;
    if (status) *status = st;
    return IS3D_OK;
}

extern "C" int is3d_smooth_spectra(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                                   const is3d_df_tables *df, const is3d_options *opts, double *dN_out, is3d_status *status)
{
    return smooth_spectra_impl(cells, species, grid, df, nullptr, opts, dN_out, status);
}

extern "C" int is3d_smooth_spectra_feqmod(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                                          const is3d_df_tables *df, const is3d_feqmod_tables *fq, const is3d_options *opts,
                                          double *dN_out, is3d_status *status)
{
    if (!fq) return fail(IS3D_EINVAL, "null feqmod tables");
    return smooth_spectra_impl(cells, species, grid, df, fq, opts, dN_out, status);
}
