// cf_feqmod.h -- modified-equilibrium smooth kernel (df_mode 3 "Mike", 4 "Jonah"): shared definitions and launch entry
// points of cf_feqmod.hip.  Device path of EmissionFunctionArray::calculate_dN_ptdptdphidy_feqmod
// (/root/reference/src/cpp/emissionfunction_smooth_kernels.cpp:396-996); include_baryon = 1 with df_mode 3 only (the reference
// exits for df_mode 4 at nonzero mu_B, deltafReader.cpp:470-474).
//
// The modified momentum is linear in the lab momentum,
//   p_LRF = mT a_k + pT b_j,  a_k = (-Xt ch + tau Xn sh, 0, -Zt ch + tau Zn sh),  b_j = (Xx cos + Xy sin, Yx cos + Yy sin, 0)
//   p_mod = A^-1 p_LRF        (ch, sh of y - eta; X, Y, Z the Milne basis; A = 1 + pi_LRF/(2 betapi) + bulk_mod)
// so that the argument of the distribution is the square root of a quadratic form in (mT, pT),
//   (E_mod / T_mod)^2 = mT^2 alphaf_k + mT pT betaf_jk + pT^2 gammaf_j
//   alphaf_k = (1 + |A^-1 a_k|^2)/T_mod^2,  betaf_jk = 2 (A^-1 a_k).(A^-1 b_j)/T_mod^2,  gammaf_j = (|A^-1 b_j|^2 - 1)/T_mod^2
// (m^2 = mT^2 - pT^2 folded in), and f = |renorm| / (exp(E_mod/T_mod) + sign).  cf_prep_feqmod writes these coefficients
// into the same tiled unit-record stream the delta-f kernel uses (cf_device.h) with the slots reused:
//   header jj : {B_j, gammaf_j, 0, 0} (free slots: the unit bounds of the cull, slot 15 the cell's scale exponent, cf_prep_feqmod)          row r, 2+1D : {A_k, alphaf_k, W_k, min_j betaf_jk, betaf_{j0..j0+JT-1,k}}
//                                              row r, 3+1D : {alphaf_k, min_j betaf_jk, A_k, W_k, betaf_{j0..j0+JT-1,k}} -- the two operands of a row's
//                                              liveness test in ONE 16-byte LDS read (fq_row_slots)
//   (include_baryon: header slot 2 of jj = 0 carries alpha_B,mod = alpha_B + Pi G / beta_Pi, :637; the lane's baryon number
//   times it is added to the exponent, :742, :927; A_ij ignores the baryon diffusion, ":660 leave for future work")
//   p.dsigma = rn (mT A_k + W_k pT B_j),  A_k = w_k ch dat + sh dan/tau  (the reference keeps dsigma_eta outside the eta
//   weight, :905), W_k = w_k;  rn = |renorm| (df_mode 4: folded into A_k and W_k by the prep kernel; df_mode 3: a
//   per-(cell, class) table RN read by the main kernel).
// Cells where feqmod breaks down (df_mode 3, emissionfunction.cpp:109-150) and the rows |y - eta| < detA of cells with
// detA < 0.01 (3+1D, :807-813) use the linearised delta-f instead: they are neutral in the stream and are evaluated by
// cf_feqmod_linear from the compacted list of such cells.
#pragma once
#include "cf_device.h"
#include <hip/hip_runtime_api.h>

namespace is3d {

// slots of a row's four scalars within the record (the phi entries follow from slot 4)
struct FqRowSlots { int A, AL, W, BM; };
constexpr FqRowSlots fq_row_slots(bool dim3) { return dim3 ? FqRowSlots{2, 0, 3, 1} : FqRowSlots{0, 1, 2, 3}; }

constexpr int kFbRec = 36;   // doubles per fallback record (cf_feqmod.hip::FbRec)
constexpr int kCrRec = 8;    // doubles per cell record of the df_mode 3 renormalisation kernel

struct FqPrepParams {
    CellPtrs cells;
    int64_t cell0;
    int32_t n_cells, J, K;
    int32_t dim3, mode;                 // mode: 3 | 4
    int32_t include_bulk, include_shear;
    int32_t baryon, baryondiff;         // include_baryon (df_mode 3), && include_baryondiff_deltaf: mu_B, n_B, V^mu are read (:572-584)
    BilinearDev bil;                    // baryon: F, G, betabulk, betaV, betapi on the (mu_B, T) grid
    const double *cosphi, *sinphi, *kgrid, *kweight;
    SplineDev spl;                      // y/c[0..2] = F, betabulk, betapi (F, betabulk unused in df_mode 4)
    int32_t nj;                         // Jonah tables (df_mode 4): abscissa bulkPi/Peq, lambda^2, z and their spline c's
    const double *jx, *jl2, *jz, *jcl, *jcz;
    double bp_max;                      // bulkPi_over_Peq_max
    int32_t ngl;                        // Gauss-Laguerre points; gl = [4][ngl]: root1, weight1, root2, weight2
    const double *gl;
    double detA_min, mass_pion0;
    int32_t JT, R, jtiles, rblocks;
    double *TS;
    double *CR;                         // [n_cells][kCrRec], df_mode 3
    double *FB;                         // [n_cells][kFbRec], written for flagged cells only
    int32_t *flag;                      // [n_cells] 0 feqmod | 1 breakdown (all rows linear) | 2 detA < 0.01 (narrow rows linear)
    unsigned long long *status;         // [0] min bad cell, [1] skipped, [7] min cell whose E_mod/T_mod can exceed 1e9
    double mTmax, kmin, kmax;           // largest lane mT; range of the k grid: bound of E_mod/T_mod for the exponential's domain (exp_p9: < 1.4e9)
    double pTmax;                       // largest lane pT (with mTmax: the bound of |p.dsigma| behind scale_rows)
    int32_t scale_rows;                 // df_mode 4 with outflow: A_k, W_k of a cell times 2^-e_c, header slot 15 = kExpShift + e_c (cf_prep_feqmod phase 2e)
};

struct FqMainArgs {
    const double *TS, *lane_mT, *lane_pT, *lane_sign;
    const double *lane_b;               // include_baryon: baryon number of each lane slot's class
    const double *RN;                   // df_mode 3: [n_cells][ncls] |renorm|
    const int32_t *lane_cls;            // df_mode 3: class of each lane slot
    const int32_t *lane_sub;            // 2+1D unit-strided lanes (g.split > 1): the slot's sub-index s, it takes the units u = s (mod split)
    int32_t ncls;
    double *partial;
    unsigned long long *stats;
    MainGeom g;
};

struct FqLinearArgs {
    const double *FB;
    const int32_t *list, *count;        // compacted fallback cells (ascending), *count entries
    const double *lane_mT, *lane_pT, *lane_sign, *lane_mass;
    const double *lane_b;               // include_baryon, else NULL
    const double *cosphi, *sinphi, *kgrid, *kweight;
    double *partial;                    // chunk 0 of the partial buffer: += after the main kernel
    int32_t J, K, Kacc, Lpad, dim3, mode, outflow, regulate;
    int32_t Lbins;                      // momentum bins: with unit-strided lanes only a bin's first slot (lanes < Lbins) takes the fallback cells
};

size_t prep_feqmod_lds_bytes(int nT, int nj, int ngl, int J, int K, int jtiles, int rblocks, int rec);
hipError_t launch_prep_feqmod(const FqPrepParams &p, hipStream_t st);
// RN[cell][cls] = |n_linear / n_mod| (/ detA in 3+1D), 0 where the reference skips the species (nan / inf) or the cell
hipError_t launch_feqmod_renorm(const double *CR, const double *gl, int ngl, const double *cls_mass, const double *cls_sign,
                                const double *cls_baryon /* NULL: include_baryon = 0 */, int ncls, int n_cells, int include_bulk,
                                int is_dim3, double *RN, hipStream_t st);
hipError_t launch_main_feqmod(int variant, int dim3, int outflow, int mode3, int baryon, const FqMainArgs &a, hipStream_t st);
// list = indices with flag != 0 in ascending order, *count = their number; status[4] += #flag 1, status[5] += #flag 2
hipError_t launch_feqmod_compact(const int32_t *flag, int n, int32_t *list, int32_t *count, unsigned long long *status,
                                 hipStream_t st);
hipError_t launch_feqmod_linear(const FqLinearArgs &a, hipStream_t st);

}  // namespace is3d
