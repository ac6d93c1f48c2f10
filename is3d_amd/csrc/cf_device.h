// cf_device.h -- shared definitions between the HIP kernels (cf_kernels.hip) and the plan/host code.
#pragma once
#include <stdint.h>

namespace is3d {

constexpr double kHbarC = 0.197327053;  // src/cpp/iS3D.h:9

// ---- derived per-cell coefficient streams written by cf_prep, read (wave-uniform) by cf_main ----
// S1[cell][k]   : {A_k, Cp_k, alpha_k, W_k}      k = y index (3+1D) or eta quadrature index (2+1D)
// S2[cell][j]   : {B_j, Dp_j, gamma_j, kappa}    j = phi index
// S3[cell][j][k]: beta_jk
// so that for a momentum (mT, pT, phi_j, y_k):
//   p.dsigma            = mT*A_k + W_k*pT*B_j                        (W_k = 1 in 3+1D)
//   p.u / T  (=: x)     = mT*Cp_k - pT*Dp_j
//   14-moment  df/feqbar = mT^2*alpha_k + mT*pT*beta_jk + pT^2*gamma_j
//   Chapman-E. df/feqbar = (mT^2*alpha_k + mT*pT*beta_jk + pT^2*gamma_j)/x     (the kappa*x term is folded into alpha, beta, gamma; the kappa slot is 0)
constexpr int kS1Rec = 4;
constexpr int kS2Rec = 4;

struct SplineDev {       // natural cubic spline, device copy: knots x, values y, second-derivative c
    int n;
    const double *x;
    const double *y[3];  // df_mode 1: {c0, c2, -}; df_mode 2: {F, betabulk, betapi}
    const double *c[3];
    int nspl;
};

struct CellPtrs {        // device pointers to the caller's SoA (is3d_cells), may be null when unused
    const double *tau, *eta, *dat, *dax, *day, *dan, *ux, *uy, *un, *T, *P, *E;
    const double *pixx, *pixy, *pixn, *piyy, *piyn, *bulkPi;
    const double *muB, *nB, *Vx, *Vy, *Vn;   // baryon diffusion arrays (include_baryon && include_baryondiff_deltaf)
};

struct BilinearDev {     // full (mu_B, T) coefficient grids for include_baryon = 1, device copies
    int nT, nB;
    const double *T, *muB;
    const double *tab[5];  // df_mode 1: c0 c1 c2 c3 c4 ; df_mode 2: F G betabulk betaV betapi ; each [nB][nT]
    int swap;              // opts.reference_bilinear_indexing: read tab[iT][imuB] like the reference's calculate_bilinear (deltafReader.cpp:404-407)
};

struct PrepParams {
    CellPtrs cells;
    int64_t cell0;       // first cell of this pass in the caller's arrays
    int32_t n_cells;     // cells in this pass
    int32_t J, K;
    int32_t dim3;        // 1: 3+1D (k = y), 0: 2+1D (k = eta quadrature)
    int32_t ce;          // 1: Chapman-Enskog, 0: 14-moment
    int32_t include_bulk, include_shear;
    int32_t baryon;      // include_baryon: bilinear coefficients, b*alpha_B in f_eq, b-linear delta-f terms
    int32_t baryondiff;  // ... && include_baryondiff_deltaf: muB, nB, V^mu read from the cell arrays
    BilinearDev bil;
    const double *cosphi, *sinphi;  // [J]
    const double *kgrid;            // [K] y values (3+1D) or eta nodes (2+1D)
    const double *kweight;          // [K] eta weights (2+1D), unused in 3+1D
    const double *kch, *ksh;        // [K] 2+1D: cosh(0 - eta_k), sinh(0 - eta_k)
    SplineDev spl;
    double *S1, *S2, *S3;           // flat streams (variant 1); unused when tiled
    // tiled stream (variants >= 2), see "unit record" below
    int32_t tiled;                  // 1: write TS, 0: write S1/S2/S3
    int32_t JT, R;                  // tile: JT phi's x R rows (rows = y's in 3+1D, eta nodes in 2+1D)
    int32_t jtiles, rblocks;        // ceil(J/JT), ceil(K/R)
    double *TS;
    const unsigned long long *pds_bound;   // tiled stream: bits of a bound on |p.dsigma| over all lanes and cells (cf_pds_bound)
    double mTmax, kmin, kmax;       // largest lane mT; range of the k grid (y in 3+1D, eta nodes in 2+1D): bound of p.u/T
    unsigned long long *status;     // [0] min bad cell (global index), [1] skipped count, [7] min cell whose p.u/T can exceed 1e9
    // E2 table stream (kernel variant 5, see "TE" below); TE == nullptr: not written
    int32_t w0_share;               // wave 0's share of a batch's units (% of another wave's) while it runs the next batch's phase 1
    int32_t pair_writer;            // tiled stream: the record writer handles two elements per lane and trip (set by launch_prep)
    int32_t dev_skip;               // dev (IS3D_PREP_SKIP, timing only, results invalid): bit 0 no unit records, bit 1 no E2 tables, bit 2 no phase 2; bit 4 (results valid): non-temporal record stores
    double *TE;
    const double *pTgrid;           // [npT] the pT grid (the lanes' pT values are exactly these)
    int32_t npT;
};

// ---- tiled stream TS (variants >= 2): what one workgroup of the main kernel streams through LDS ----
// unit record (REC = 4*JT + R*(4+JT) doubles) for tile (jt, rb) of one cell:
//   header  jj < JT : {B_j, Dp_j, gamma_j, x}                     j = jt*JT + jj (clamped to J-1); x: 3+1D records without baryon
//                     slots carry max_j Dp_j of the tile in x of jj = 0 and min_k Cp_k of the unit's rows in x of jj = 1 (unit-level cull)
//   row     r  < R  : {A_k, Cp_k, alpha_k, W_k, beta_{j0..j0+JT-1,k}}   k = rb*R + r; rows past K are
//                     neutral padding (A = W = 0, Cp copied from row K-1, alpha = beta = 0)
// 3+1D: stream s = jt*rblocks + rb, one unit per cell:        TS[(s*n_cells + cell)*REC]
// 2+1D: stream s = jt, rblocks units per cell (eta blocks):   TS[((s*n_cells + cell)*rblocks + rb)*REC]
// Scale: A_k and B_j are stored times 2^-e, one integer e per execute chosen (cf_pds_bound) so that |p.dsigma| 2^-e <= 1
// for every lane and cell; cf_finalize multiplies 2^e back.  With |p.dsigma| <= 1 the outflow test max(p.dsigma, 0) is
// the VOP3 clamp modifier of the add/fma that forms p.dsigma (no v_max_f64).  Exact: powers of two.
// include_baryon = 1 ("B" records): the header slot 3 holds L2_j, two doubles {alpha_B, Dmax} follow the header, and
// every row carries two more scalars {L_k, c} after W, where  b (mT L_k + pT L2_j)  is the part of df/feqbar
// that is linear in the momentum and proportional to the baryon number b of the lane (cf_kernels.hip::cf_prep); in 3+1D Dmax =
// max_j Dp_j of the tile and c of row 0 = min_k Cp_k of the unit's rows (the unit-level cull bounds; c = 0 in the other rows).
// E2 table stream TE (kernel variants 5, 6: 3+1D, with or without baryon slots): the phi-side factor of the factorised exponential,
//   E2[ipT][jj] = exp(pT_ipT Dp_j - pT_ipT Dmax),   Dmax = max_j Dp_j over the tile (header slot 3 of the unit record),
// depends on the lane only through its pT, and the grid has only npT (32) of them for thousands of lanes: cf_prep evaluates the
// npT x JT exponentials of a (cell, phi tile) ONCE and the main kernel's lanes read theirs from the LDS-staged table instead of
// each computing JT exponentials per live unit (the same for all row blocks and all lane-waves: 76x redundant on config 3).
//   TE[((jt*n_cells + cell)*JT + jj)*kE2Stride + ipT]   (pT fastest: the lanes' LDS reads of one jj hit consecutive words;
//                                                        grids of up to kE2Stride pT values, else the plan falls back to variant 3)
constexpr int kE2Stride = 32;
inline int unit_rec_doubles(int JT, int R, int baryon = 0) { return baryon ? 4 * JT + 2 + R * (6 + JT) : 4 * JT + R * (4 + JT); }

// ---- main kernel geometry ----
struct MainGeom {
    int32_t n_cells;   // cells in this pass
    int32_t J, K;
    int32_t Lpad;      // padded lane-slot count (multiple of 64)
    int32_t G;         // workgroups per stream = ceil(Lpad/64 / 4)
    int32_t jtiles, ktiles, nch;
    int32_t NT;        // streams = jtiles*ktiles*nch
    int32_t Kacc;      // accumulator slots along k: K (3+1D) or 1 (2+1D)
    int32_t first_pass;  // 1: store partials, 0: add to them
    int32_t upc;       // tiled stream: units per cell within a stream (1 in 3+1D, rblocks in 2+1D)
    int32_t zskip;     // 1: skip rows whose exponential is exactly zero for the whole wave; 2: also rows that cannot change a bit
                       // of any accumulator (tile delta-f kernel with outflow && regulate_deltaf; cf_kernels.hip)
    int32_t wpb;       // lane-waves (= waves) per workgroup of the tile kernel: 2, 4 or 8
    int32_t baryon;    // 1: "B" unit records, lanes carry a baryon number
    int32_t npT;       // variant 5: rows of the E2 table
    int32_t ub;        // variant 5: units per LDS batch (from the LDS budget of the workgroup)
    int32_t split;     // 2+1D: lane slots per momentum bin (unit-strided lanes, cf_main_tile); 1 = off
    int32_t ch0, nch_run;  // cf_main_tile3e: this launch runs chunks [ch0, ch0 + nch_run) of the nch (nch_run 0 = all of them)
    int32_t nch_small;     // the LAST nch_small of the nch chunks hold a quarter of the cells of the others (chunk_cells)
};

// Cells [c0, c1) of a chunk.  The chunks are equal but for a tapered tail: the last g.nch_small chunks are a quarter of the size of the others,
// so that the grid drains in quarter-length tasks (the fixed ~2.7 ms a launch of the main kernel cost beyond its per-cell time was the tail of
// full-length tasks: 6 % of a 125 000-cell shard's step).  nch_small = 0: c0 = chunk n_cells / nch, the partition of rounds 1-3.
// The host side of it (cf_plan.cpp, cf_vah.hip): given the base chunk count of `cells` cells, the last kTaperBig full-size chunks become
// 4 kTaperBig quarter-size ones -- four rounds of the chip in quarter-length tasks.  Not with an explicit opts.cell_chunks (the caller's count is
// kept as it is) and not for chunks of fewer than 256 cells.  Returns nch_small and raises nch by 3 kTaperBig; the partial buffer is allocated for
// kTaperExtra slabs beyond the base count.
constexpr int kTaperBig = 6, kTaperExtra = 3 * kTaperBig;
inline int chunk_taper(long long cells, int explicit_chunks, int &nch)
{
    if (explicit_chunks > 0 || nch < 4 * kTaperBig || cells / nch < 256) return 0;
    nch += kTaperExtra;
    return 4 * kTaperBig;
}

__host__ __device__ inline void chunk_cells(const MainGeom &g, int chunk, int &c0, int &c1)
{
    const int nbig = g.nch - g.nch_small;
    const long long Q = 4LL * nbig + g.nch_small;                                  // the surface in quarter-chunks
    const long long q0 = chunk < nbig ? 4LL * chunk : 4LL * nbig + (chunk - nbig);
    const long long q1 = chunk + 1 <= nbig ? 4LL * (chunk + 1) : 4LL * nbig + (chunk + 1 - nbig);
    c0 = (int)((q0 * g.n_cells) / Q);
    c1 = (int)((q1 * g.n_cells) / Q);
}

struct MainArgs {
    const double *S1, *S2, *S3, *TS, *lane_mT, *lane_pT, *lane_sign, *lane_b;
    double *partial;
    unsigned long long *stats;   // [2] += wave-rows visited, [3] += wave-rows culled as exactly zero
    MainGeom g;
    const int32_t *lane_pe = nullptr;   // per lane: exponent pe with max(mT/mTmax, pT/pTmax) < 2^pe (accumulator-relative cull)
    const double *TE = nullptr;         // variant 5: E2 table stream
    const int32_t *lane_ipT = nullptr;  // variant 5: per lane, the index of its pT in the grid
    const int32_t *lane_sub = nullptr;  // unit-strided lanes: per lane slot, which units (u = sub mod split) it takes
    const double *cull_floor = nullptr; // cf_main_tile3e, surface-relative cull: [jtiles * ktiles][Lpad] lower bounds of the row-cull threshold
                                        // (cf_cull_floor, from the partial spectrum of the chunks that ran first)
    const double *pTgrid = nullptr;     // cf_main_tile3e<E2L> (developer build): the pT grid, for tables built in LDS
};

}  // namespace is3d
