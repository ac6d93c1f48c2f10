/*
 * is3d_amd.h -- C ABI of the MI355X-native smooth Cooper-Frye spectra path.
 *
 * The reference (derekeverett/iS3D) has no FFI layer; the boundary this library is a drop-in for
 * is the C++ method
 *
 *   void EmissionFunctionArray::calculate_dN_pTdpTdphidy(double *Mass, double *Sign,
 *        double *Degeneracy, double *Baryon, double *T_fo, double *P_fo, double *E_fo,
 *        double *tau_fo, double *eta_fo, double *ux_fo, double *uy_fo, double *un_fo,
 *        double *dat_fo, double *dax_fo, double *day_fo, double *dan_fo, double *pixx_fo,
 *        double *pixy_fo, double *pixn_fo, double *piyy_fo, double *piyn_fo, double *bulkPi_fo,
 *        double *muB_fo, double *nB_fo, double *Vx_fo, double *Vy_fo, double *Vn_fo,
 *        Deltaf_Data *df_data)
 *   declared  src/cpp/emissionfunction.h:179, defined src/cpp/emissionfunction_smooth_kernels.cpp:28-393,
 *   called    src/cpp/emissionfunction.cpp:1519,
 *
 * whose implicit inputs are members of the object (FO_length, number_of_chosen_particles, the
 * pT/phi/y/eta Tables, DIMENSION, DF_MODE, OUTFLOW, REGULATE_DELTAF, INCLUDE_* --
 * src/cpp/emissionfunction.h:73-90, :139-142) and whose output is the member array
 * dN_pTdpTdphidy (emissionfunction.h:142), index
 *   iS3D = ipart + npart * (ipT + npT * (iphip + nphi * iy))      (smooth_kernels.cpp:363)
 * Every implicit input is an explicit argument here.  Plain pointers and sizes only.
 *
 * Conventions
 *   - all functions return IS3D_OK (0) or a negative IS3D_E* code; nothing calls exit()/abort()
 *     (reference: printf + exit(-1), GSL abort);  is3d_last_error() has the text;
 *   - the caller owns every buffer; the library never frees or keeps caller memory past a call
 *     (a plan keeps its own device copies of species/grid/tables);
 *   - pointers for disabled corrections (muB, nB, Vx, Vy, Vn; eta in 2+1D) may be NULL and are
 *     never dereferenced (reference passes uninitialised pointers there, emissionfunction.cpp:1357-1378);
 *   - dN_out has n_species * n_pT * n_phi * n_y_eff doubles, n_y_eff = (dimension == 2) ? 1 : n_y,
 *     species fastest; it is OVERWRITTEN unless opts->accumulate (reference semantics: +=, :375);
 *   - thread-safe for distinct plans; one plan serves one call at a time;
 *   - there is NO CPU fallback: without a HIP device every compute entry returns IS3D_ENODEVICE.
 */
#ifndef IS3D_AMD_H
#define IS3D_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IS3D_OK 0
#define IS3D_EINVAL (-1)     /* bad argument / unsupported option combination            */
#define IS3D_ENODEVICE (-2)  /* no HIP device, or a HIP runtime error (see last_error)    */
#define IS3D_EDOMAIN (-3)    /* a non-skipped cell's T is outside the coefficient table, */
                             /* or its flow is so fast that p.u/T can exceed 1e9         */
                             /* (reference: GSL domain error -> abort, deltafReader.cpp:339) */
#define IS3D_ENOMEM (-4)
#define IS3D_EIO (-5)        /* host reader/writer failure                               */
#define IS3D_EPEER (-6)      /* multi-GPU: another rank's execute failed before the all-reduce; the sum is incomplete */

/* Freezeout cells, structure of arrays, fp64, length n_cells each.  Replaces the *_fo pointer
 * arguments of calculate_dN_pTdpTdphidy (emissionfunction.h:179).  Units as after the reader's
 * hbar*c conversion (GeV, GeV/fm^3; readindata.cpp:367-410). */
typedef struct {
    int64_t n_cells;                    /* FO_length (emissionfunction.h:139) */
    const double *tau, *eta;            /* eta unused (may be NULL) when dimension == 2 */
    const double *dat, *dax, *day, *dan; /* covariant dsigma_mu */
    const double *ux, *uy, *un;         /* contravariant u^mu; u^tau recomputed (:133) */
    const double *T, *P, *E;
    const double *pixx, *pixy, *pixn, *piyy, *piyn; /* other components reconstructed (:166-170) */
    const double *bulkPi;
    const double *muB, *nB, *Vx, *Vy, *Vn; /* only read if include_baryon && include_baryondiff_deltaf (:186-197) */
} is3d_cells;

/* Replaces Mass, Sign, Degeneracy, Baryon (emissionfunction.cpp:1293-1307) */
typedef struct {
    int32_t n;                          /* number_of_chosen_particles */
    const double *mass, *sign, *degeneracy, *baryon;
} is3d_species;

/* Replaces pT_tab, phi_tab, y_tab, eta_tab column 1 (and eta weights, column 2)
 * (smooth_kernels.cpp:40-92).  cos/sin(phi) are formed inside, as the reference does. */
typedef struct {
    int32_t n_pT;  const double *pT;
    int32_t n_phi; const double *phi;
    int32_t n_y;   const double *y;     /* used when dimension == 3 */
    int32_t n_eta; const double *eta, *eta_w; /* used when dimension == 2 */
} is3d_grid;

/* Replaces Deltaf_Data: the coefficient tables as loaded by load_df_coefficient_data
 * (deltafReader.cpp:120-197), still carrying their temperature scaling (c0*T^4, c1*T^3, c2*T^4, c3*T^4,
 * c4*T^5, F/T, G, betabulk/T^4, betaV/T^3, betapi/T^4).  Every table is [n_muB][n_T], T fastest, row 0 =
 * the lowest mu_B (= 0 in the shipped files).
 *   include_baryon = 0: only row 0 of c0, c2 (df_mode 1) or F, betabulk, betapi (df_mode 2) is read
 *     (n_muB may be 1, the other pointers NULL); natural cubic splines in T are built inside
 *     (construct_cubic_splines, deltafReader.cpp:300-322; cubic_spline, :325-395).
 *   include_baryon = 1: bilinear interpolation in (T, mu_B) over the full grids of c0..c4 (df_mode 1) or
 *     F, G, betabulk, betaV, betapi (df_mode 2) (bilinear_interpolation, :412-484) -- with the intended
 *     [imuB][iT] indexing; the reference's calculate_bilinear swaps the two indices (:404-407). */
typedef struct {
    int32_t n_T;
    const double *T;                    /* GeV, ascending, uniform */
    int32_t n_muB;
    const double *muB;                  /* GeV, ascending, uniform; may be NULL when include_baryon = 0 */
    const double *c0, *c1, *c2, *c3, *c4;           /* df_mode 1 */
    const double *F, *G, *betabulk, *betaV, *betapi; /* df_mode 2 */
} is3d_df_tables;

/* Extra inputs of the modified-equilibrium path (df_mode 3, 4): replaces the Gauss_Laguerre *laguerre argument of
 * calculate_dN_ptdptdphidy_feqmod (emissionfunction.h:182), the PDG/Plasma inputs of
 * Deltaf_Data::compute_jonah_coefficients (deltafReader.cpp:222-297) and the members DETA_MIN, MASS_PION0
 * (emissionfunction.cpp:184, :188). */
typedef struct {
    int32_t n_gla;                      /* Gauss-Laguerre points per alpha (32 in tables/gla_roots_weights_32_points.txt) */
    const double *root1, *weight1;      /* alpha = 1 (Gauss_Laguerre::load_roots_and_weights, readindata.cpp:24-53) */
    const double *root2, *weight2;      /* alpha = 2 */
    int32_t n_pdg;                      /* ALL species of the PDG file (df_mode 4: the z(Pi/P), lambda(Pi/P) tables sum over them) */
    const double *pdg_mass, *pdg_degeneracy, *pdg_sign;
    double T_avg;                       /* Plasma::temperature: the surface average as read back from
                                           average_thermodynamic_quantities.dat (df_mode 4) */
    double deta_min;                    /* parameter deta_min */
    double mass_pion0;                  /* parameter mass_pion0 */
} is3d_feqmod_tables;

typedef struct {
    int32_t dimension;                  /* 2 | 3                      DIMENSION  */
    int32_t df_mode;                    /* 1 14-moment | 2 Chapman-Enskog | 3 modified equilibrium (Mike) | 4 (Jonah);
                                           3 and 4 only through the *_feqmod entries; 4 needs include_baryon = 0, as in the
                                           reference (deltafReader.cpp:470-474)                             DF_MODE */
    int32_t include_baryon;             /* 1: mu_B/T in f_eq, bilinear (T, mu_B) coefficients (muB_fo; with
                                           include_baryondiff_deltaf also nB_fo, Vx_fo, Vy_fo, Vn_fo) */
    int32_t include_bulk_deltaf;
    int32_t include_shear_deltaf;
    int32_t include_baryondiff_deltaf;  /* ignored unless include_baryon */
    int32_t regulate_deltaf;
    int32_t outflow;
    int32_t accumulate;                 /* 0: dN_out = result; 1: dN_out += result (reference) */
    int32_t device;                     /* HIP device ordinal; -1 = current device */
    /* tuning; 0 = library default */
    int32_t kernel_variant;             /* 0 default: 6 in 3+1D (pT grids of up to 32 values: the 8x7 tile with the phi-side exponentials read
                                           from a table stream and the rows of a unit tested for liveness before their exponentials; larger pT
                                           grids: 3, the 8x7 tile without the table -- 2, 6x7, with baryon slots), 7 in 2+1D (8x31 tile, unit-strided
                                           lanes); modified equilibrium (df_mode 3, 4): 3 in 3+1D, 7 in 2+1D.  The shipped library holds these
                                           kernels and honours one explicit choice -- 3 for a 3+1D delta-f surface without baryon slots; ANY other
                                           request runs the default, and status.kernel_variant says which kernel ran.  The A/B forms of rounds 1-5
                                           (1 direct kernel | 2, 4 other tile shapes | 5 hand-pipelined rows | 8 register-staged LDS copy | 9 unit
                                           records on the scalar path | 10 E2 column from global memory | 11 raw header values as FMA operands | 12 E2 tables built per workgroup in LDS | modified equilibrium: 5, 6 other row
                                           walks, 2-4 the 61-row tiles) exist only in the developer build of the library (make DEV=1) */
    int32_t cell_chunks;                /* number of cell chunks the main kernel grid is split into; 0 (default): the library's count, with
                                           a tapered tail (the last chunks a quarter of the size of the others); > 0: that many equal chunks */
    int64_t workspace_bytes;            /* cap on the derived-coefficient workspace per pass; 0: max(16 GiB, 45 % of the device's TOTAL memory) --
                                           the total, not what is free at the moment, so that the pass / chunk count (and with it the
                                           summation order) of a large surface does not depend on the GPU's other tenants.  An
                                           allocation that does not fit returns IS3D_ENOMEM naming the sizes */
    int32_t collapse_species;           /* 0 default(on) | 1 on | 2 off: evaluate one representative per
                                           distinct (mass, sign) and scale by degeneracy */
    int32_t zero_skip;                  /* wave-level culling of rows that cannot change the result (bitwise-identical spectra
                                           in all three settings): 1: rows whose exp(-p.u/T) is exactly +0; 0 (default): also
                                           rows whose every term is below half an ulp of every accumulator it would be added to
                                           (delta-f tile kernel with outflow && regulate_deltaf); 2: off; 3: as 0, and in the 3+1D
                                           delta-f kernel (cf_main_tile3e, outflow && regulate_deltaf, >= 16 cell chunks) an eighth of
                                           the chunks runs first and its partial spectrum -- a lower bound of the final one, every term
                                           being >= 0 -- floors the thresholds of the rest.  NOT bitwise: a skipped term is below 2^-57 of
                                           the final value of every bin it belongs to, the spectrum can lose at most n_cells 2^-57 of a
                                           bin (7e-12 at 1e6 cells), one-sided.  Measured on BASELINE config 3: 59.5 instead of 57.3 % of
                                           the wave-rows culled, main kernel -2.2 %, 6 of 4.9e6 bins differ by one ulp */
    int32_t waves_per_group;            /* 0 default | 2, 4, 8: lane-waves per workgroup of the tile kernel (they share
                                           one LDS-staged coefficient stream) | 1: one-wave workgroups, no barrier partner
                                           (variants 5, 6 only) */
    int32_t reference_bilinear_indexing; /* include_baryon = 1 only.  0 (default): the (T, mu_B) coefficient tables are read [imuB][iT], as they
                                           are stored.  1: bug-compatible with the reference's calculate_bilinear, which reads f_data[iT][imuB]
                                           (deltafReader.cpp:404-407) from arrays allocated [points_muB][points_T] (:36-61): the value at
                                           (mu_B row iT, T column imuB).  Reproduced wherever that read is inside the allocation
                                           (iT + 1 < n_muB, i.e. T < T[n_muB - 1] = 0.18 GeV on the shipped 101 x 81 grids); a live cell
                                           beyond it gives IS3D_EDOMAIN (the reference reads past its row pointers there: undefined) */
    int32_t reserved[4];
} is3d_options;

typedef struct {
    int32_t code;                       /* same as the return value */
    int32_t n_classes;                  /* distinct (mass, sign) classes actually evaluated */
    int64_t n_cells_skipped;            /* cells with u.dsigma <= 0 (contribute 0, :137) */
    int64_t bad_cell;                   /* first cell index with T outside the table, or -1 */
    int32_t n_passes;                   /* workspace passes over the cell axis */
    int32_t kernel_variant;             /* variant that ran */
    double ms_prep, ms_main, ms_finalize; /* device time of the three kernels (HIP events), summed over passes;
                                             filled by is3d_plan_timings / the host entry, else 0 */
    double ms_h2d, ms_d2h;              /* host entry only */
    int64_t n_wave_rows;                /* (cell, row, wave) triples the tile kernel visited ...          */
    int64_t n_wave_rows_culled;         /* ... and how many it skipped as exactly zero (zero_skip)        */
    int64_t n_cells_breakdown;          /* df_mode 3: cells where feqmod breaks down (linearised delta-f used, the count
                                           the reference prints, smooth_kernels.cpp:983-986)              */
    int64_t n_cells_narrow;             /* df_mode 3, 4 in 3+1D: cells with detA < 0.01, whose rows |y - eta| < detA use
                                           the linearised delta-f (:807-813)                               */
} is3d_status;

typedef struct is3d_plan is3d_plan;

const char *is3d_last_error(void);
const char *is3d_version(void);
/* number of visible HIP devices (>= 0), never fails */
int is3d_device_count(void);

/*
 * One-shot host entry: what a maintainer calls from calculate_dN_pTdpTdphidy.  All pointers are
 * HOST memory.  Uploads the cell arrays, runs prep -> main -> finalize on the device, downloads
 * the spectrum.
 */
int is3d_smooth_spectra(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                        const is3d_df_tables *df, const is3d_options *opts, double *dN_out,
                        is3d_status *status);

/* The same for df_mode 3 / 4: what a maintainer calls from calculate_dN_ptdptdphidy_feqmod
 * (emissionfunction.h:182, smooth_kernels.cpp:396-996, call site emissionfunction.cpp:1584).  df needs betapi
 * (df_mode 4) or F, betabulk, betapi (df_mode 3), row 0. */
int is3d_smooth_spectra_feqmod(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                               const is3d_df_tables *df, const is3d_feqmod_tables *fq, const is3d_options *opts,
                               double *dN_out, is3d_status *status);

/*
 * Device-resident API.  A plan holds the species classes, grids and spline tables on the device and
 * the workspaces sized for up to max_cells cells per execute.
 */
int is3d_plan_create(is3d_plan **plan, const is3d_species *species, const is3d_grid *grid,
                     const is3d_df_tables *df, const is3d_options *opts, int64_t max_cells);
/* plan for df_mode 3 / 4; execute, observables, timings as for any plan.  df_mode 4 builds the 301-point
 * lambda(Pi/P), z(Pi/P) tables on the host here (deltafReader.cpp:222-297), ~0.1 s. */
int is3d_plan_create_feqmod(is3d_plan **plan, const is3d_species *species, const is3d_grid *grid,
                            const is3d_df_tables *df, const is3d_feqmod_tables *fq, const is3d_options *opts,
                            int64_t max_cells);
/* length of dN_out in doubles */
int64_t is3d_plan_output_size(const is3d_plan *plan);
/* cells->* and dN_out are DEVICE pointers on the plan's device; hip_stream is a hipStream_t (NULL =
 * default stream).  Asynchronous with respect to the host except for the final status read-back
 * when status != NULL (then the stream is synchronised). */
int is3d_plan_execute(is3d_plan *plan, const is3d_cells *cells, double *dN_out, void *hip_stream,
                      is3d_status *status);
/* Domain errors of executes that ran with status == NULL (fully asynchronous: nothing is read back, a cell whose T leaves the
 * coefficient table is left out of the spectrum and the call returns IS3D_OK).  The plan keeps the lowest offending cell index of
 * all executes since the last check; this call synchronises hip_stream, returns IS3D_EDOMAIN (and *bad_cell, index within its
 * execute's cells; may be NULL) if there was one, and clears the record.  The reference aborts on such a cell (GSL domain error /
 * exit(-1)): a caller that skips the status must ask here before trusting the spectrum. */
int is3d_plan_check(is3d_plan *plan, void *hip_stream, int64_t *bad_cell);
/* Derived observables from a device-resident spectrum of this plan's shape (what the reference's writers reduce on
 * the host: emissionfunction.cpp:639-677, :729-772, :1053-1136).  pT_w / phi_w: HOST arrays of the quadrature weights
 * (column 2 of the pT / phi tables).  Outputs are DEVICE arrays, any may be NULL:
 *   dNdy          [n_species][n_y_eff]            sum_phi sum_pT w_phi w_pT dN
 *   dN2pipTdpTdy  [n_species][n_y_eff][n_pT]      sum_phi w_phi dN / (2 pi)
 *   vn            [n_species][n_y_eff][n_pT][7]   |sum_phi w_phi e^{i k phi} dN| / sum_phi w_phi dN, k = 1..7 (0 if the
 *                                                 denominator is below 1e-15)
 * Asynchronous on hip_stream. */
int is3d_plan_observables(is3d_plan *plan, const double *dN_dev, const double *pT_w, const double *phi_w, double *dNdy,
                          double *dN2pipTdpTdy, double *vn, void *hip_stream);
/* Enable (1) / disable (0) HIP-event timing of the three kernels on subsequent executes. */
int is3d_plan_set_timing(is3d_plan *plan, int32_t enable);
/* Synchronises the recorded events of the last execute and fills status->ms_*. */
int is3d_plan_timings(is3d_plan *plan, is3d_status *status);

/* Diagnostic (no reference counterpart): the shader clock a kernel actually runs at.  Launches one idle wave per XCD on a
 * private non-blocking stream of `device`; each reads the shader-clock counter (s_memtime) and the constant-rate counter
 * (s_memrealtime) `seconds` apart and the call returns the mean ratio in GHz.  Called from a second host thread while a
 * spectra kernel runs on another stream, it gives the clock that kernel's fp64 roofline should be priced at
 * (bench.py: roofline_valu.shader_clock_ghz).  *ghz = 0 if the two counters tick at the same rate on this device. */
int is3d_probe_shader_clock(int32_t device, double seconds, double *ghz);
/* Diagnostic (no reference counterpart): the elementary functions the kernels are built from (is3d_amd/csrc/cf_math.h), evaluated on the
 * device, y[i] = f(x[i]) for HOST arrays of n doubles -- so that their accuracy is a tested number, not a comment.  which: 0 exp_full
 * (Cody-Waite, degree 10) | 1 exp_p9 (one-fma reduction, degree 9; |x| < 1.4e9) | 2 exp_p9_sat | 3 exp_full_sat (|x| up to ~1e45) | 4 sqrt_g1 (v_rsq_f64 +
 * one Goldschmidt step) | 5 sqrt_nr | 6 rcp_nr1 (v_rcp_f64 + one Newton step) | 7 rcp_nr (two steps) | 8 exp_p9_scaled with e = 5 (32 e^x: the
 * power-of-two factor cf_main_feqmod takes out of p.dsigma comes back through the exponential's shift constant). */
int is3d_math_probe(int32_t which, int64_t n, const double *x, double *y, int32_t device);
/* Diagnostic (no reference counterpart): process-wide counts of the plans this library has created (is3d_plan_create*, the per-shard plans of
 * is3d_multi_plan_create and of the one-shot entries) and of the device allocations (hipMalloc) it has made, since the library was loaded.  What a
 * persistent plan promises -- a second execute creates nothing and allocates nothing -- is checked with these, not with a wall clock.  Either
 * pointer may be NULL. */
int is3d_resource_counters(int64_t *plans_created, int64_t *device_allocations);
/* Name of the dominant kernel as it appears in rocprofv3 traces, for the variant in use. */
const char *is3d_plan_main_kernel_name(const is3d_plan *plan);
/* tile of the main kernel: *JT phi's x *R rows (y's in 3+1D, eta nodes in 2+1D) */
int is3d_plan_tile_shape(const is3d_plan *plan, int32_t *JT, int32_t *R);
/* bytes of device workspace the plan holds */
int64_t is3d_plan_workspace_bytes(const is3d_plan *plan);
void is3d_plan_destroy(is3d_plan *plan);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU.  The reference is single-process; its spectrum is a plain sum over freezeout cells
 * (src/cpp/emissionfunction_smooth_kernels.cpp:363-375: dN_pTdpTdphidy[iS3D] += chunk sum), so the cell axis shards:
 * contiguous blocks of cells per GPU, no data-path collective, one sum of the n_species x n_bins spectrum at the end.
 * Two forms:
 *   (1) one process, several devices: is3d_smooth_spectra_multi -- one host thread + stream per shard;
 *   (2) one process per GPU (MPI / torchrun style hosts): an is3d_comm (RCCL communicator) + is3d_plan_execute_allreduce.
 * RCCL (librccl.so.1) is loaded on first use; a host that never asks for it does not need it installed.
 * --------------------------------------------------------------------------------------------- */
#define IS3D_REDUCE_ORDERED 0   /* shard spectra are added pairwise in a fixed binary tree by a device kernel (the partner's spectrum
                                   read over hipMemcpyPeer): round r adds shard i + 2^r into shard i for every i that is a multiple of
                                   2^(r+1) -- ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)) for eight shards -- the pairs of a round
                                   run concurrently on their own devices.  Bitwise reproducible for a given shard count, whatever
                                   the devices; a device may be listed more than once */
#define IS3D_REDUCE_RCCL 1      /* ncclAllReduce(ncclDouble, ncclSum) over a communicator of the listed devices (distinct) */

/* Host entry over several devices: what is3d_smooth_spectra / is3d_smooth_spectra_feqmod (fq != NULL: df_mode 3, 4) compute,
 * with the cells split into n_devices contiguous shards (sizes differ by at most one cell), shard s on devices[s].
 *   devices == NULL: ordinals 0 .. n_devices-1;  n_devices <= 0: every visible device.  opts->device is ignored.
 *   status (may be NULL): counters summed over the shards, bad_cell = lowest global cell index, ms_prep / ms_main / ms_finalize /
 *   ms_h2d = the slowest shard's, ms_d2h = reduction + download;  shard_status (may be NULL): n_devices entries, one per shard
 *   (bad_cell is shard-local there).
 * All shards run concurrently.  With one shard and IS3D_REDUCE_ORDERED this is is3d_smooth_spectra on devices[0]. */
int is3d_smooth_spectra_multi(const is3d_cells *cells, const is3d_species *species, const is3d_grid *grid,
                              const is3d_df_tables *df, const is3d_feqmod_tables *fq, const is3d_options *opts,
                              const int32_t *devices, int32_t n_devices, int32_t reduce, double *dN_out,
                              is3d_status *status, is3d_status *shard_status);
/* shard [*lo, *hi) of `rank` among n_ranks contiguous shards of n_cells cells (the split is3d_smooth_spectra_multi uses) */
int is3d_shard_bounds(int64_t n_cells, int32_t rank, int32_t n_ranks, int64_t *lo, int64_t *hi);

/* RCCL communicator for one-process-per-GPU hosts.  Rank 0 calls is3d_comm_unique_id and ships the 128 bytes to the other
 * ranks by whatever the host already has (MPI_Bcast, a torch.distributed store, a file); then every rank calls
 * is3d_comm_create (collective: ncclCommInitRank) with its HIP device. */
typedef struct is3d_comm is3d_comm;
#define IS3D_COMM_ID_BYTES 128
int is3d_comm_unique_id(uint8_t id[IS3D_COMM_ID_BYTES]);
int is3d_comm_create(is3d_comm **comm, const uint8_t id[IS3D_COMM_ID_BYTES], int32_t n_ranks, int32_t rank, int32_t device);
int is3d_comm_rank(const is3d_comm *comm, int32_t *rank, int32_t *n_ranks);
/* in-place sum over all ranks of n doubles at dN_dev (DEVICE pointer on the communicator's device), asynchronous on hip_stream:
 * the "+=" of smooth_kernels.cpp:375 across shards (ncclAllReduce, ncclDouble, ncclSum). */
int is3d_comm_allreduce(is3d_comm *comm, double *dN_dev, int64_t n, void *hip_stream);
void is3d_comm_destroy(is3d_comm *comm);
/* is3d_plan_execute on this rank's shard of the cells, then the sum of dN_out over the ranks on the same stream (ncclAllReduce):
 * every rank ends with the spectrum of the whole surface.  comm == NULL: plain is3d_plan_execute.  The plan must have been created
 * with opts.accumulate = 0 (the old contents would be summed n_ranks times): IS3D_EINVAL otherwise.
 * Failures.  A one-double error word is summed next to the spectrum (same RCCL group):
 *   - a rank whose execute fails in a way that leaves its stream usable (IS3D_EDOMAIN; IS3D_EINVAL for a bad shard or plan option)
 *     still joins -- after a domain error with the cells it could evaluate, after an argument error with zeros -- with its error word
 *     set, and returns its own error code;
 *   - the other ranks learn of it: with status != NULL (the call synchronises anyway) they return IS3D_EPEER; with status == NULL
 *     (fully asynchronous) at the next is3d_comm_check;
 *   - a rank that cannot join (dN_out or plan NULL, a HIP error) calls ncclCommAbort on its OWN communicator before it returns
 *     (unusable afterwards: every call returns IS3D_ENODEVICE) and its host should exit non-zero so that the launcher ends the job.
 *     Its abort is local: RCCL enqueues an all-reduce and returns at once, and nothing tells the peers' kernels that a rank will never
 *     arrive.  What protects the PEERS is their own deadline: every host-side wait the library makes behind a collective
 *     (the status read-back of this call, is3d_comm_check, is3d_comm_timings, is3d_comm_synchronize) polls the stream together with
 *     ncclCommGetAsyncError and, on an asynchronous RCCL error or after the communicator's timeout (is3d_comm_set_timeout; default 300 s,
 *     or IS3D_COMM_TIMEOUT_S at is3d_comm_create), aborts this rank's communicator and returns IS3D_ENODEVICE instead of blocking for ever.
 *     A host that synchronises the stream itself (hipStreamSynchronize) has no such deadline: use is3d_comm_synchronize. */
int is3d_plan_execute_allreduce(is3d_plan *plan, const is3d_cells *shard, double *dN_out, is3d_comm *comm, void *hip_stream,
                                is3d_status *status);
/* Error words of the collectives since the last check (waits for hip_stream, with the communicator's deadline): IS3D_OK, or IS3D_EPEER with
 * *n_failed (may be NULL) = how many rank-executes had failed before their all-reduce -- this rank's own failures included; IS3D_ENODEVICE
 * when the wait hit an asynchronous RCCL error or the deadline (the communicator is aborted then). */
int is3d_comm_check(is3d_comm *comm, void *hip_stream, int32_t *n_failed);
/* ncclCommAbort on this rank's communicator: for a host that has to leave a job.  Local -- the other ranks find out through their own
 * deadline (see is3d_plan_execute_allreduce), not through this call. */
int is3d_comm_abort(is3d_comm *comm);
/* Deadline, in seconds, of the host-side waits behind this communicator's collectives (default 300).  What it times (round 5): with ONE collective
 * enqueued since the last completed wait -- the usual step: execute, all-reduce, wait -- the clock starts when the stream REACHES the collective
 * (an event recorded just before it), so this rank's own kernels queued ahead of it do not count, however long they run.  With several collectives
 * queued before one wait, and as a backstop while the collective has not been reached, the wait gives up 20 x `seconds` after it was entered: a caller
 * that queues many steps before it waits must choose `seconds` above 1/20 of the compute it queues. */
int is3d_comm_set_timeout(is3d_comm *comm, double seconds);
/* hipStreamSynchronize(hip_stream) with that deadline and with ncclCommGetAsyncError polled beside it: IS3D_OK when everything enqueued on
 * the stream has finished, IS3D_ENODEVICE (communicator aborted) on an asynchronous RCCL error or when the deadline passes. */
int is3d_comm_synchronize(is3d_comm *comm, void *hip_stream);
/* Device time of the last collective on this rank (HIP events on its stream around the RCCL group; it includes the wait for the
 * slowest rank).  Waits for the closing event (with the communicator's deadline). */
int is3d_comm_timings(is3d_comm *comm, double *ms_allreduce);

/* Persistent form of is3d_smooth_spectra_multi: one plan, one workspace, one stream, the device blocks for cells and spectrum and
 * (IS3D_REDUCE_RCCL) the communicator set per shard are created ONCE; every execute then only uploads its cells, runs the shards concurrently and sums the
 * spectra.  max_cells bounds cells->n_cells of any execute.  The result of an execute is bitwise the one is3d_smooth_spectra_multi
 * returns for the same arguments. */
typedef struct is3d_multi_plan is3d_multi_plan;
int is3d_multi_plan_create(is3d_multi_plan **mplan, const is3d_species *species, const is3d_grid *grid, const is3d_df_tables *df,
                           const is3d_feqmod_tables *fq, const is3d_options *opts, const int32_t *devices, int32_t n_devices,
                           int32_t reduce, int64_t max_cells);
int is3d_multi_plan_execute(is3d_multi_plan *mplan, const is3d_cells *cells, double *dN_out, is3d_status *status,
                            is3d_status *shard_status);
int32_t is3d_multi_plan_shards(const is3d_multi_plan *mplan);
int64_t is3d_multi_plan_output_size(const is3d_multi_plan *mplan);
void is3d_multi_plan_destroy(is3d_multi_plan *mplan);

/* ---------------------------------------------------------------------------------------------
 * Anisotropic hydro (VAH, P_L matching): replaces EmissionFunctionArray::calculate_dN_pTdpTdphidy_VAH_PL
 * (src/cpp/emissionfunction.h, src/cpp/emissionfunction_smooth_kernels.cpp:2140-2393).  The reference never calls it (call
 * site commented out, emissionfunction.cpp:1650-1654) and src/cpp never loads the VAH coefficient tables: the per-cell
 * 14-moment coefficients c0..c4 are inputs, as in the method's own signature.  All ten pi_perp^{mu nu} components are inputs
 * too (this kernel does not reconstruct them); T is accepted and unused.  No outflow cut, no skipped cells.  opts: dimension,
 * include_bulk_deltaf, include_shear_deltaf, regulate_deltaf, accumulate, device, workspace_bytes, cell_chunks, collapse_species,
 * zero_skip (exact zeros only on this path: 0 and 1 are the same), kernel_variant (0 default = 3 in 3+1D: factored exponent on the 8 x 7
 * tile, cf_main_vah3, and the same kernel on 8 x 31 records with unit-strided lanes in 2+1D | 2: the round-1 kernel on the 6 x 7 / 8 x 61 tile,
 * kept for A/B in the developer build of the library -- the shipped one runs the default for it).  A cell whose E_a/Lambda could exceed 1e9 for the momentum grid returns IS3D_EDOMAIN (the reference's exp overflows there).
 * HOST pointers; dN_out as for is3d_smooth_spectra.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t n_cells;
    const double *tau, *eta;            /* eta unused (may be NULL) when dimension == 2 */
    const double *ux, *uy, *un;
    const double *dat, *dax, *day, *dan;
    const double *T;                    /* unused (only FORCE_F0 reads it in the reference, :2295-2298); may be NULL */
    const double *pitt, *pitx, *pity, *pitn, *pixx, *pixy, *pixn, *piyy, *piyn, *pinn;   /* pi_perp^{mu nu} */
    const double *bulkPi;               /* residual bulk pressure */
    const double *Wx, *Wy;              /* W_perp^mu; W^tau, W^eta reconstructed (:2244-2245) */
    const double *Lambda, *aL;          /* anisotropic variables */
    const double *c0, *c1, *c2, *c3, *c4;
} is3d_vah_cells;

int is3d_smooth_spectra_vah(const is3d_vah_cells *cells, const is3d_species *species, const is3d_grid *grid,
                            const is3d_options *opts, double *dN_out, is3d_status *status);

/* The anisotropic-hydro 14-moment coefficient tables deltaf_coefficients/vah/c{0..4}_vah1.dat.  src/cpp never loads them; the
 * loader the reference has is the CUDA tree's load_df_coefficient_data, "df_mode == 4 // va hydro PL matching 14 moment"
 * (src/cuda/deltafReader.cu:60-82 file names, :104-112 "n_Lambda\nn_alphaL\n" + one label line, :196-213 rows
 * "Lambda [fm^-1]  alpha_L  value", alpha_L outer, Lambda inner).  Every table is [n_aL][n_L], Lambda fastest, file units. */
typedef struct {
    int32_t n_L, n_aL;
    const double *L;                    /* Lambda nodes, fm^-1, ascending */
    const double *aL;                   /* alpha_L nodes, ascending */
    const double *c0, *c1, *c2, *c3, *c4;
} is3d_vah_df_tables;
/* Reads <dir>/c0_vah1.dat ... c4_vah1.dat (dir = "deltaf_coefficients/vah" in a run directory).  Two-call pattern: L == NULL ->
 * only *n_L, *n_aL; otherwise L[n_L], aL[n_aL], c[5 * n_aL * n_L] (table k at c + k n_aL n_L), capacity in doubles of `c`.
 * The node arrays are what the reference's scan leaves in L_array / aL_array (each row overwrites its entries; the c4 file is
 * scanned last).  IS3D_EIO: a file is missing or short; IS3D_EINVAL: the five headers disagree or the nodes do not ascend. */
int is3d_vah_df_read(const char *dir, int32_t *n_L, int32_t *n_aL, double *L, double *aL, double *c, int64_t capacity);
/* Per-cell coefficients as src/cuda/deltafReader.cu:224-278 sets them, evaluated on the device (HOST pointers in and out; Lambda in
 * GeV as the surface reader stores it, :228): the first alpha_L node i2 >= 1 with aL < aL[i2] and the first Lambda node i1 >= 1 with
 * Lambda/hbarc < L[i1] select the cell of the grid; bilinear interpolation in (Lambda, alpha_L) -- which below the first node
 * extrapolates, as the reference does -- and every coefficient divided by hbarc^3 (:262-266).  A cell with Lambda/hbarc >= L[n_L - 1]
 * or aL >= aL[n_aL - 1] (or a NaN) finds no node: the reference leaves its c0..c4 unset there; this call returns IS3D_EDOMAIN with
 * *bad_cell (may be NULL) = the lowest such index, and stores zeros for it.  c0..c4: n doubles each. */
int is3d_vah_coefficients(const is3d_vah_df_tables *tab, int64_t n, const double *Lambda, const double *aL, double *c0, double *c1,
                          double *c2, double *c3, double *c4, int32_t device, int64_t *bad_cell);
/* is3d_smooth_spectra_vah with the coefficients taken from the tables: cells->c0..c4 are ignored (may be NULL), a device kernel
 * interpolates them from (Lambda, aL) into the arrays the prep kernel reads.  IS3D_EDOMAIN + status->bad_cell for a cell outside
 * the grid (see is3d_vah_coefficients).  tab == NULL: is3d_smooth_spectra_vah. */
int is3d_smooth_spectra_vah_df(const is3d_vah_cells *cells, const is3d_species *species, const is3d_grid *grid,
                               const is3d_vah_df_tables *tab, const is3d_options *opts, double *dN_out, is3d_status *status);

/* Device-resident form (bench.py --workload config5, hosts whose surface is already in HBM): the plan holds the lane tables, the
 * coefficient tables (tab may be NULL: cells->c0..c4 are then inputs) and the workspaces for up to max_cells cells per execute;
 * cells->* and dN_out are DEVICE pointers on the plan's device, hip_stream a hipStream_t.  Asynchronous except for the status
 * read-back when status != NULL.  is3d_vah_plan_timings: device time of the last execute's kernels (coefficients + prep in ms_prep,
 * cf_main_vah in ms_main, reduction + species scatter in ms_finalize) when timing is enabled. */
typedef struct is3d_vah_plan is3d_vah_plan;
int is3d_vah_plan_create(is3d_vah_plan **plan, const is3d_species *species, const is3d_grid *grid, const is3d_vah_df_tables *tab,
                         const is3d_options *opts, int64_t max_cells);
int64_t is3d_vah_plan_output_size(const is3d_vah_plan *plan);
int64_t is3d_vah_plan_workspace_bytes(const is3d_vah_plan *plan);
int is3d_vah_plan_execute(is3d_vah_plan *plan, const is3d_vah_cells *cells, double *dN_out, void *hip_stream, is3d_status *status);
int is3d_vah_plan_set_timing(is3d_vah_plan *plan, int32_t enable);
int is3d_vah_plan_timings(is3d_vah_plan *plan, is3d_status *status);
int is3d_vah_plan_tile_shape(const is3d_vah_plan *plan, int32_t *JT, int32_t *R);
/* "cf_main_vah3" (3+1D, opts.kernel_variant 0 | 3: factored exponent, 8 x 7 tile) or "cf_main_vah" (2+1D; 3+1D with kernel_variant 2: the
 * round-1 kernel on the 6 x 7 tile, kept for A/B) as it appears in rocprofv3 traces */
const char *is3d_vah_plan_main_kernel_name(const is3d_vah_plan *plan);
void is3d_vah_plan_destroy(is3d_vah_plan *plan);

/* FO_data_reader::read_surf_VAH_PLMatch (mode 2; src/cpp/readindata.cpp:813-928): 31 numbers per cell -- tau x y eta | dat dax day
 * dan | ut ux uy un | E T P PL | pitt pitx pity pitn pixx pixy pixn piyy piyn pinn | Wt Wx Wy Wn | bulkPi -- E, T, P, PL, pi, W and
 * bulkPi multiplied by hbar*c, and the anisotropic variables inferred per cell: aL = aL_fit(PL/P), Lambda = T / (aL R200(aL) / 2)^(1/4)
 * (src/cpp/arsenal.cpp:999-1065), Lambda stored in GeV.  PL/P >= 3 is fatal in the reference ("pl is too large", exit(-1)):
 * IS3D_EINVAL naming the cell.  Two-call pattern as is3d_surface_read_vh (arrays == NULL -> only *n_cells).  arrays32: caller-allocated,
 * in the order of is3d_vah_cells' first 25 members -- tau eta ux uy un dat dax day dan T pitt pitx pity pitn pixx pixy pixn piyy piyn
 * pinn bulkPi Wx Wy Lambda aL -- then E P PL Wt Wn x y; the last seven may be NULL. */
int is3d_surface_read_vah(const char *path, int32_t dimension, int64_t *n_cells, double *const *arrays32);

/* ---------------------------------------------------------------------------------------------
 * Particle sampler (operation = 2): replaces EmissionFunctionArray::sample_dN_pTdpTdphidy
 * (src/cpp/emissionfunction.h:208-210, emissionfunction_sampling_kernels.cpp:833-1225, call sites emissionfunction.cpp:1543,
 * :1606) for viscous hydro, df_mode 1-4, fast = 0 | 1; include_baryon = 1 with df_mode 1-3.  The reference's serial
 * std::default_random_engine streams are replaced by counter-based Philox4x32-10 streams keyed by (seed, stream, global cell
 * index, event) -- same five stream roles and the same distributions; particle lists agree with the reference statistically,
 * not draw by draw (the construction is written out in cf_sampler.hip and DESIGN.md section 3c).
 * --------------------------------------------------------------------------------------------- */
typedef struct {                        /* Sampled_Particle (src/cpp/particle.h), 96 bytes */
    int64_t cell;                       /* global cell index: first_cell + index in the call */
    int32_t event;
    int32_t species;                    /* index into the species list (mc_id, mass: caller's arrays) */
    double tau, x, y, eta, t, z;        /* t = tau cosh(eta), z = tau sinh(eta) */
    double E, px, py, pz;
} is3d_particle;

typedef struct {
    int32_t n_events;                   /* Nevents (emissionfunction.cpp:202, :1531) */
    int32_t n_gla;                      /* Gauss-Laguerre points */
    uint64_t seed;                      /* sampler_seed */
    double y_cut;                       /* 2+1D: hadrons get a uniform rapidity in +-y_cut (Y_CUT); 3+1D: unused (y_max = 0.5) */
    int64_t first_cell;                 /* global index of cells[0]: a shard of a surface samples what the whole surface would */
    const double *x, *y;                /* cell positions x_fo, y_fo copied into the particles; may be NULL */
    const double *root1, *weight1;      /* Gauss-Laguerre alpha = 1 (equilibrium densities, max_particle_number) */
    const is3d_feqmod_tables *feqmod;   /* df_mode 3, 4 (and fast = 1 with df_mode 2): alpha = 2 nodes, PDG list, T_avg of the
                                           Jonah tables, deta_min, mass_pion0; NULL otherwise */
    int32_t fast;                       /* FAST: species densities at the surface-average temperature (:1044-1056) */
    int32_t batch_events;               /* tuning: events per count/scan/fill batch; 0 = as many as fit 2^25 (event, cell) threads.
                                           The particle list does not depend on it. */
    double T_avg;                       /* fast: Plasma::temperature as read back from average_thermodynamic_quantities.dat */
    double T_avg_switch;                /* fast, df_mode 3: the same after `if (SET_T_SWITCH) temperature = T_SWITCH` (:856);
                                           0 = T_avg */
    double muB_avg;                     /* fast with include_baryon: Plasma::baryon_chemical_potential (deltafReader.cpp:545, :858) */
} is3d_sampler_inputs;

typedef struct {
    int64_t n_cells_skipped;            /* u.dsigma <= 0 (:899) */
    int64_t n_hadrons_drawn;            /* sum of the Poisson numbers (before the flux / viscous keep test) */
    int64_t n_momentum_samples, n_acceptances;   /* "Momentum sampling efficiency" (:1224) */
    int32_t n_classes, reserved;
    int64_t n_cells_breakdown;          /* df_mode 3: cells sampled with the linear delta-f instead (:1038) */
    double ms_h2d, ms_prep, ms_count, ms_fill;   /* device time: upload, densities + cell records, count pass + scan, fill pass */
    double ms_density;                  /* part of ms_prep: cf_sampler_density (the Gauss-Laguerre density integrals per (cell, class)) */
    double ms_poisson;                  /* part of ms_count: cf_sampler_poisson + the compaction of the emitting (event, cell) pairs */
} is3d_sampler_stats;

/* All pointers HOST memory; opts: dimension, df_mode (1-4), include_bulk_deltaf, include_shear_deltaf, device.  Particles
 * come ordered by (event, cell, draw).  particles == NULL (or capacity 0): only *n_particles is computed.  If the buffer is
 * too small the first `capacity` particles are stored, *n_particles is the full count and IS3D_ENOMEM is returned. */
int is3d_sample_particles(const is3d_cells *cells, const is3d_species *species, const is3d_df_tables *df,
                          const is3d_sampler_inputs *in, const is3d_options *opts, is3d_particle *particles,
                          int64_t capacity, int64_t *n_particles, is3d_sampler_stats *stats);
/* Device-resident, persistent form (what bench.py --workload config5-sampler times; is3d_sample_particles is create + upload + execute +
 * download + destroy, so the two give the same list bit for bit): the species classes, splines / bilinear grids, Jonah tables and fast-mode
 * densities are set up once from (species, df, in, opts) -- `in` as for is3d_sample_particles; its n_events, seed, first_cell, x, y and
 * batch_events are execute-time arguments here -- and the workspaces (8 B x classes + 360 B per cell, + 8 B x species per cell for the running sums of the
 * species weights unless df_mode = 3; 25 B per (event, cell) of a batch) are
 * allocated at the first execute of a shape and kept.  execute: cells->* and x, y are DEVICE arrays on the plan's device (x, y may be NULL),
 * particles a DEVICE buffer of `capacity` entries (NULL: count only); runs on the null stream and returns when the list is complete.
 * stats->ms_h2d is 0 (nothing is uploaded).  Return codes as is3d_sample_particles; cells->n_cells > max_cells: IS3D_EINVAL. */
typedef struct is3d_sampler_plan is3d_sampler_plan;
int is3d_sampler_plan_create(is3d_sampler_plan **plan, const is3d_species *species, const is3d_df_tables *df, const is3d_sampler_inputs *in,
                             const is3d_options *opts, int64_t max_cells);
int is3d_sampler_plan_execute(is3d_sampler_plan *plan, const is3d_cells *cells_dev, const double *x_dev, const double *y_dev, int32_t n_events,
                              uint64_t seed, int64_t first_cell, int32_t batch_events, is3d_particle *particles_dev, int64_t capacity,
                              int64_t *n_particles, is3d_sampler_stats *stats);
void is3d_sampler_plan_destroy(is3d_sampler_plan *plan);
/* is3d_sample_particles over several devices: the same contiguous cell shards, shard s on devices[s] with first_cell advanced to
 * the shard's first cell -- the counter-based streams are keyed by the GLOBAL cell index, so the hadrons are exactly those one device
 * samples; the shard lists are merged into the single-device order (event, cell, draw).  No collective at all.  Same calling
 * pattern and return codes as is3d_sample_particles; stats are summed (times: the slowest shard's).  opts->device is ignored. */
int is3d_sample_particles_multi(const is3d_cells *cells, const is3d_species *species, const is3d_df_tables *df,
                                const is3d_sampler_inputs *in, const is3d_options *opts, const int32_t *devices, int32_t n_devices,
                                is3d_particle *particles, int64_t capacity, int64_t *n_particles, is3d_sampler_stats *stats);

/* EmissionFunctionArray::calculate_total_yield (src/cpp/emissionfunction_sampling_kernels.cpp:653-830; call sites
 * emissionfunction.cpp:1527, :1591): the mean particle yield of the surface, from which an oversampled run takes its number of
 * events, Nevents = min(ceil(min_num_hadrons / |yield|), max_num_samples) (emissionfunction.cpp:1524-1533).  Species densities
 * as Deltaf_Data::compute_particle_densities forms them at the surface averages (deltafReader.cpp:536-650), per-cell terms as
 * estimate_mean_particle_number (:200-236); 2+1D: times 2 y_cut (:822-826).  Deterministic (no random numbers).
 * in: n_gla, root1, weight1, y_cut and feqmod (the alpha = 2 nodes for every df_mode; df_mode 4 also the PDG list and T_avg of
 * the Jonah tables).  opts: dimension, df_mode, include_bulk_deltaf, include_baryon, include_baryondiff_deltaf, device.
 * densities (may be NULL): 3 * species->n doubles {Equilibrium_Density, Bulk_Density, Diffusion_Density} (emissionfunction.cpp:1289-1306). */
typedef struct {
    double T, E, P, muB, nB;            /* Plasma::load_thermodynamic_averages: the five lines of average_thermodynamic_quantities.dat */
    const double *root3, *weight3;      /* Gauss-Laguerre alpha = 3 (df_mode 1: J30, J31, deltafReader.cpp:596-600); NULL otherwise */
} is3d_yield_inputs;
int is3d_total_yield(const is3d_cells *cells, const is3d_species *species, const is3d_df_tables *df, const is3d_sampler_inputs *in,
                     const is3d_yield_inputs *avg, const is3d_options *opts, double *mean_yield, double *densities);

/* write_particle_list_OSC (src/cpp/emissionfunction.cpp:863-901): results/particle_list_osc.dat, "# N" per non-empty event
 * then "mcid t x y z E px py pz" rows; particles ordered by event. */
int is3d_write_particle_list_osc(const char *path, int32_t n_events, int64_t n_particles, const is3d_particle *particles,
                                 const int64_t *mc_id);

/* test_sampler = 1: the binned self-consistency outputs (sample_dN_dy ... sample_dN_dX, sampling_kernels.cpp:31-152; writers
 * emissionfunction.cpp:903-1257) from a particle list: <dir>/dN_dy/, dN_deta/, momentum_distribution/, vn/,
 * spacetime_distribution/ (must exist), mean_yield.dat, yield_list.dat.  Bins = the parameters y_cut, y_bins, eta_cut, eta_bins,
 * pT_lower_cut, pT_upper_cut, pT_bins, tau_min, tau_max, tau_bins, r_min, r_max, r_bins (emissionfunction.cpp:205-222). */
typedef struct {
    double y_cut, eta_cut, pT_lower_cut, pT_upper_cut, tau_min, tau_max, r_min, r_max;
    int32_t y_bins, eta_bins, pT_bins, tau_bins, r_bins, reserved;
} is3d_sampler_test_bins;
int is3d_write_sampler_tests(const char *results_dir, const is3d_sampler_test_bins *bins, int32_t n_events, int32_t n_species,
                             const int64_t *mc_id, int64_t n_particles, const is3d_particle *particles, double mean_yield);

/* ---------------------------------------------------------------------------------------------
 * Driver: IS3D::run_particlization (src/cpp/iS3D.cpp:74-192; class IS3D, src/cpp/iS3D.h:19-96).  Reads iS3D_parameters.dat,
 * PDG/, tables/, deltaf_coefficients/ from the current directory and writes results/ exactly as the command line tool does
 * (which is this call with surface = NULL).  surface != NULL is the embedding path (read_fo_surf_from_memory +
 * run_particlization(0), iS3D.cpp:27-72, :100-134): HOST arrays already in GeV / fm units, x and y the cell positions (may be
 * NULL).  result may be NULL; otherwise it receives library-allocated copies -- operation 2: the sampled particles ordered
 * by event (what the reference returns in final_particles_, :178-184); operation 1: the spectrum -- to be released with
 * is3d_run_result_free.  include/iS3D_amd.hpp wraps this in a class with the reference's member names.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t operation, n_events, n_species, reserved;
    int64_t n_particles;
    is3d_particle *particles;           /* operation 2 */
    int64_t *mc_id;                     /* [n_species] chosen particles, order of PDG/chosen_particles.dat */
    double *mass;                       /* [n_species] */
    int64_t n_spectrum;
    double *spectrum;                   /* operation 1: dN_pTdpTdphidy, species fastest */
} is3d_run_result;

int is3d_run_particlization(const is3d_cells *surface, const double *x, const double *y, int32_t kernel_variant,
                            is3d_run_result *result);
/* The same on an explicit device list (operation 1: the cells are sharded over the devices as is3d_smooth_spectra_multi does;
 * operation 2 samples the shards on their devices and merges the lists, is3d_sample_particles_multi).  devices == NULL && n_devices > 0: ordinals 0 .. n_devices-1.  n_devices <= 0 is what
 * is3d_run_particlization and the command line tool do: the environment decides -- IS3D_DEVICES = "0,2,3" | "all" (default:
 * every visible device), IS3D_REDUCE = "ordered" (default) | "rccl". */
int is3d_run_particlization_on(const is3d_cells *surface, const double *x, const double *y, int32_t kernel_variant,
                               const int32_t *devices, int32_t n_devices, int32_t reduce, is3d_run_result *result);
void is3d_run_result_free(is3d_run_result *result);

/* ---------------------------------------------------------------------------------------------
 * Host I/O in the reference's file formats (C++ implementation, C ABI so that tests and other
 * hosts can reach it).  All paths are explicit; the CLI driver passes the reference's hard-coded
 * CWD-relative names.
 * --------------------------------------------------------------------------------------------- */

/* ParameterReader::readFromFile + getVal (src/cpp/ParameterReader.cpp:38-155): `name = value # comment`.
 * Returns IS3D_EIO if the file cannot be read, IS3D_EINVAL if `name` is absent (reference: exit). */
int is3d_param_get(const char *path, const char *name, double *value);

/* Table(filename) (src/cpp/Table.cpp:179-195, arsenal.cpp:406-453): n-column numeric text.  Two-call
 * pattern: data == NULL returns the shape; otherwise fills data[row * n_cols + col] (capacity in
 * doubles).  A last line without a trailing newline is dropped, as in the reference. */
int is3d_table_read(const char *path, int64_t *n_rows, int32_t *n_cols, double *data, int64_t capacity);

/* FO_data_reader::read_surf_VH (mode 1; src/cpp/readindata.cpp:320-468).  Two-call pattern as above
 * (cells == NULL -> only *n_cells).  Fills caller-allocated arrays named in `cells` (non-NULL
 * members of at least n_cells doubles; x,y positions are not kept).  Also returns the
 * surface-volume-weighted averages {T, E, P, muB, nB} (readindata.cpp:422-466) in avg5 if non-NULL. */
int is3d_surface_read_vh(const char *path, int32_t include_baryon, int32_t include_baryondiff_deltaf,
                         int32_t dimension, int64_t *n_cells, double *const *cell_arrays23, double *avg5);

/* FO_data_reader::read_surf_switch (src/cpp/readindata.cpp:133-144) for the viscous-hydro surface formats the
 * smooth path accepts: mode 0 read_surf_VH_old (:148-318), 1 read_surf_VH (:320-468), 4 read_surf_VH_MUSIC (:552-668),
 * 5 read_surf_VH_Vorticity (:470-551: the mode-1 columns, V^tau inside the diffusion block, six thermal-vorticity columns that the
 * viscous-hydro kernels calculate_spectra runs on such a surface do not read; unlike the reference's reader this one also returns the
 * surface averages), 6 read_surf_VH_MUSIC_New (:671-810), 7 read_surf_VH_hiceventgen (:1059-1196).  Same calling pattern and array order
 * as is3d_surface_read_vh; every format is converted to the kernel's conventions the way the reference does (tau
 * Jacobians, hbar*c, p = T s - e, u = gamma v); muB is stored whenever the array is given (modes 4, 6, 7 always
 * carry it), nB and V^mu are zero in modes 4, 6, 7.  Other modes: IS3D_EINVAL. */
int is3d_surface_read(const char *path, int32_t mode, int32_t include_baryon, int32_t include_baryondiff_deltaf,
                      int32_t dimension, int64_t *n_cells, double *const *cell_arrays23, double *avg5);

/* The same readers (modes 0, 1, 4, 5, 6, 7 and the anisotropic-hydro mode 2) behind ONE read and ONE parse of the text, the arrays owned by
 * the library, and a binary sidecar next to the text so that a second run on the same surface skips the parse (no reference counterpart: the
 * reference re-parses `input/surface.dat` with operator>> on every run, readindata.cpp:320-468; the drop-in keeps its result, not its cost).
 *   cache = 0: parse the text, touch nothing else | 1: use `<path>.is3dcache` when it matches the text file as it is now -- size, mtime (ns), a
 *   hash of sampled blocks of its contents, the parse parameters (mode, include_baryon, include_baryondiff_deltaf, dimension) and the sidecar's
 *   own length -- else parse the text and (re)write it (temp file + rename, on a thread of the library's while the caller computes; a directory
 *   that cannot be written to is not an error) | 2: as 1 with a hash of the WHOLE text (reads the text: slower, for the wary).
 *   IS3D_NO_CACHE=1 in the environment forces 0.  The cached arrays and averages are bit for bit what the text parse produced.
 * is3d_surface_arrays: modes 0-7 except 2: n_arrays = 25 -- cell_arrays23 order (T P E tau eta ux uy un dat dax day dan pixx pixy pixn piyy
 * piyn bulkPi muB nB Vx Vy Vn; NULL for the ones the flags leave out) then the positions x, y (columns 2, 3: the sampler's) -- and avg5 as
 * is3d_surface_read; mode 2: n_arrays = 32, the arrays32 of is3d_surface_read_vah, avg5 untouched.  Pointers stay valid until is3d_surface_close.
 *   The sidecar's key (round 5): the text's size, mtime, CHANGE time, inode and device, the sampled (or whole-file) content hash and the parse
 *   parameters -- an edit that restores size and mtime (cp -p, rsync -t, utime) still changes the inode's ctime and is re-parsed.
 * is3d_surface_source: 0 the text was parsed, no sidecar written | 1 parsed, sidecar written (waits for the writer) | 2 loaded from the sidecar.
 * is3d_surface_from_sidecar: 1 when the arrays came from the sidecar, else 0; known when is3d_surface_open returns, never waits. */
typedef struct is3d_surface is3d_surface;
int is3d_surface_open(const char *path, int32_t mode, int32_t include_baryon, int32_t include_baryondiff_deltaf, int32_t dimension,
                      int32_t cache, is3d_surface **surface);
int64_t is3d_surface_cells(const is3d_surface *surface);
int32_t is3d_surface_source(is3d_surface *surface);
int32_t is3d_surface_from_sidecar(const is3d_surface *surface);
int is3d_surface_arrays(const is3d_surface *surface, const double **arrays, int32_t n_arrays, double avg5[5]);
void is3d_surface_close(is3d_surface *surface);

/* PDG_Data::read_resonances_conventional (src/cpp/readindata.cpp:1440-1568), reduced to what the
 * smooth path uses.  Two-call pattern (mc_id == NULL -> only *n).  Arrays of capacity entries. */
int is3d_pdg_read(const char *path, int32_t *n, int64_t *mc_id, double *mass, double *gspin,
                  double *baryon, double *sign, int32_t capacity);

/* PDG_Data::read_resonances_smash_box with read_mcid (src/cpp/readindata.cpp:1571-1685, :1201-1418): the line-oriented list of hrg_eos = 3
 * (PDG/pdg_box.dat: "name mass width parity id [id...]", '#' comments) -- degeneracy, baryon number, statistics and the antiparticle
 * entries follow from the digits of the Monte-Carlo ids.  Same outputs and two-call pattern as is3d_pdg_read. */
int is3d_pdg_read_box(const char *path, int32_t *n, int64_t *mc_id, double *mass, double *gspin,
                      double *baryon, double *sign, int32_t capacity);

/* Deltaf_Data::load_df_coefficient_data (src/cpp/deltafReader.cpp:65-219) for one file, mu_B = 0 row.
 * Two-call pattern (T == NULL -> only *n_T). */
int is3d_df_table_read(const char *path, int32_t *n_T, double *T, double *value, int32_t capacity);
/* The same file with every mu_B row (include_baryon = 1): value[iB * n_T + iT].  Two-call pattern
 * (T == NULL -> only *n_T, *n_muB); capacity in doubles of `value`. */
int is3d_df_table_read_full(const char *path, int32_t *n_T, int32_t *n_muB, double *T, double *muB, double *value,
                            int64_t capacity);

/* Gauss_Laguerre::load_roots_and_weights (src/cpp/readindata.cpp:24-53; tables/gla_roots_weights_32_points.txt).
 * Two-call pattern (root == NULL -> only the shape); root/weight[alpha * n_points + k], capacity in doubles each. */
int is3d_gla_read(const char *path, int32_t *n_alpha, int32_t *n_points, double *root, double *weight, int64_t capacity);

/* Writers (src/cpp/emissionfunction.cpp:381-450, :729-772, :1053-1136): append to
 * <dir>/dN_pTdpTdphidy.dat, <dir>/dN_pTdpTdphidy_<mcid>.dat, <dir>/dN_dy_<mcid>.dat,
 * <dir>/vn_continuous/vn_<mcid>.dat in the reference's formatting.  pT/phi/y carry nodes and
 * weights (w may be NULL for write_spectra only). */
int is3d_write_results(const char *results_dir, int32_t dimension, int32_t n_species, const int64_t *mc_id,
                       int32_t n_pT, const double *pT, const double *pT_w, int32_t n_phi, const double *phi,
                       const double *phi_w, int32_t n_y, const double *y, const double *dN);

#ifdef __cplusplus
}
#endif
#endif /* IS3D_AMD_H */
