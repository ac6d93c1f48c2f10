/*
 * cf_oracle.c -- CPU restatement (plain C, optional OpenMP) of iS3D's smooth Cooper-Frye
 * momentum-spectrum path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker the HIP path is compared against.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library (is3d_amd/csrc) never links,
 * imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or expected outputs for this path
 * (SURVEY.md section 0.1), its only third-party arithmetic (GSL cspline, version unpinned, not
 * vendored) is absent from this image, and executing the reference binary was denied by the
 * environment.  The restatement is therefore anchored on
 *   (a) the reference source text cited line by line below,
 *   (b) closed-form known answers for the 1-cell toy surface (tests/test_oracle.py),
 *   (c) an independent extended-precision numpy restatement (tests/golden/make_golden.py),
 *   (d) scipy's independent natural cubic spline for the GSL replacement.
 *
 * What is restated (all paths relative to /root/reference):
 *   src/cpp/emissionfunction_smooth_kernels.cpp:28-393   EmissionFunctionArray::calculate_dN_pTdpTdphidy
 *   src/cpp/deltafReader.cpp:300-395, :486-504           construct_cubic_splines / cubic_spline / evaluate_df_coefficients
 *   GSL gsl_interp_cspline (natural) + gsl_linalg_solve_symm_tridiag (published algorithm)
 *
 * Intended semantics where the reference has defects:
 *   - the cell's eta is private to the cell (reference: racy shared etaValues[0], :120-123);
 *   - a cell skipped by u.dsigma <= 0 contributes exactly 0 (reference: stale scratch, :137);
 *   - the sum over cells is a plain sum (reference: per-chunk simd reduction then += , :367-375).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_HBARC 0.197327053 /* src/cpp/iS3D.h:9 */

/* ------------------------------------------------------------------------------------------
 * Natural cubic spline = gsl_interp_cspline.  deltafReader.cpp:303-320 builds one spline per
 * coefficient over the mu_B = 0 row; deltafReader.cpp:339-358 evaluates it at the cell's T.
 *
 * GSL cspline_init: c[0] = c[n-1] = 0; interior c from the symmetric tridiagonal system
 *   diag[i] = 2 (h_i + h_{i+1}), offdiag[i] = h_{i+1}, rhs[i] = 3 (dy_{i+1}/h_{i+1} - dy_i/h_i)
 * solved by gsl_linalg_solve_symm_tridiag (LDL^T forward/back substitution).
 * ------------------------------------------------------------------------------------------ */
int oracle_cspline_init(int n, const double *x, const double *y, double *c)
{
    if (n < 3) return -1;
    int i;
    int max_index = n - 1;
    int sys_size = max_index - 1; /* linear system is sys_size x sys_size */
    double *g = (double *)malloc(sizeof(double) * n);
    double *diag = (double *)malloc(sizeof(double) * n);
    double *offdiag = (double *)malloc(sizeof(double) * n);
    double *alpha = (double *)malloc(sizeof(double) * n);
    double *gamma = (double *)malloc(sizeof(double) * n);
    double *z = (double *)malloc(sizeof(double) * n);
    double *cc = (double *)malloc(sizeof(double) * n);

    c[0] = 0.0;
    c[max_index] = 0.0;

    for (i = 0; i < sys_size; i++) {
        const double h_i = x[i + 1] - x[i];
        const double h_ip1 = x[i + 2] - x[i + 1];
        const double ydiff_i = y[i + 1] - y[i];
        const double ydiff_ip1 = y[i + 2] - y[i + 1];
        const double g_i = (h_i != 0.0) ? 1.0 / h_i : 0.0;
        const double g_ip1 = (h_ip1 != 0.0) ? 1.0 / h_ip1 : 0.0;
        offdiag[i] = h_ip1;
        diag[i] = 2.0 * (h_ip1 + h_i);
        g[i] = 3.0 * (ydiff_ip1 * g_ip1 - ydiff_i * g_i);
    }

    if (sys_size == 1) {
        c[1] = g[0] / diag[0];
    } else {
        /* solve_tridiag for symmetric positive definite systems, N = sys_size */
        const int N = sys_size;
        alpha[0] = diag[0];
        gamma[0] = offdiag[0] / alpha[0];
        for (i = 1; i < N - 1; i++) {
            alpha[i] = diag[i] - offdiag[i - 1] * gamma[i - 1];
            gamma[i] = offdiag[i] / alpha[i];
        }
        alpha[N - 1] = diag[N - 1] - offdiag[N - 2] * gamma[N - 2];
        /* update RHS */
        z[0] = g[0];
        for (i = 1; i < N; i++) z[i] = g[i] - gamma[i - 1] * z[i - 1];
        for (i = 0; i < N; i++) cc[i] = z[i] / alpha[i];
        /* back substitution */
        c[1 + (N - 1)] = cc[N - 1];
        for (i = N - 2; i >= 0; i--) c[1 + i] = cc[i] - gamma[i] * c[1 + i + 1];
    }
    free(g); free(diag); free(offdiag); free(alpha); free(gamma); free(z); free(cc);
    return 0;
}

/* gsl_interp_bsearch: index i with x[i] <= xq < x[i+1], right end closed. */
static int bsearch_interval(const double *x, double xq, int lo, int hi)
{
    while (hi > lo + 1) {
        int i = (hi + lo) / 2;
        if (x[i] > xq) hi = i; else lo = i;
    }
    return lo;
}

/* gsl_spline_eval: domain error (GSL_EDOM -> default handler aborts) outside [x0, x_{n-1}]. */
int oracle_cspline_eval(int n, const double *x, const double *y, const double *c, double xq, double *out)
{
    if (!(xq >= x[0] && xq <= x[n - 1])) return -1;
    int i = bsearch_interval(x, xq, 0, n - 1);
    const double x_lo = x[i], x_hi = x[i + 1];
    const double dx = x_hi - x_lo;
    if (!(dx > 0.0)) return -2;
    const double y_lo = y[i], y_hi = y[i + 1];
    const double dy = y_hi - y_lo;
    const double delx = xq - x_lo;
    const double c_i = c[i], c_ip1 = c[i + 1];
    const double b_i = (dy / dx) - dx * (c_ip1 + 2.0 * c_i) / 3.0;
    const double d_i = (c_ip1 - c_i) / (3.0 * dx);
    *out = y_lo + delx * (b_i + delx * (c_i + delx * d_i));
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int n_T;
    const double *T;        /* knots, GeV (deltafReader.cpp:184-196, first points_T rows) */
    const double *c0, *c2;  /* 14 moment, scaled by T^4  (deltafReader.cpp:337-344) */
    const double *F, *betabulk, *betapi; /* Chapman-Enskog: F/T, betabulk/T^4, betapi/T^4 (:352-358) */
    /* include_baryon = 1 only: the full (mu_B, T) tables as load_df_coefficient_data stores them,
     * data[iB][iT] (deltafReader.cpp:168-196); order c0 c1 c2 c3 c4 F G betabulk betaV betapi */
    int n_muB;
    const double *muB;
    const double *t2d[10];
} oracle_df_tables;

typedef struct {
    int dimension, df_mode;
    int include_baryon, include_bulk_deltaf, include_shear_deltaf, include_baryondiff_deltaf;
    int regulate_deltaf, outflow;
} oracle_opts;

typedef struct {
    double c0, c1, c2, c3, c4, F, G, betabulk, betaV, betapi;
} df_coeff;

/* deltafReader.cpp:325-395 (include_baryon = 0 branch of :486-504) */
static int eval_df(const oracle_df_tables *t, const double *sc0, const double *sc2, const double *sF,
                   const double *sbb, const double *sbp, int df_mode, double T, df_coeff *df)
{
    memset(df, 0, sizeof(*df));
    double T4 = T * T * T * T;
    double v;
    if (df_mode == 1) {
        if (oracle_cspline_eval(t->n_T, t->T, t->c0, sc0, T, &v)) return -1;
        df->c0 = v / T4;
        df->c1 = 0.0;
        if (oracle_cspline_eval(t->n_T, t->T, t->c2, sc2, T, &v)) return -1;
        df->c2 = v / T4;
        df->c3 = 0.0;
        df->c4 = 0.0;
    } else if (df_mode == 2) {
        if (oracle_cspline_eval(t->n_T, t->T, t->F, sF, T, &v)) return -1;
        df->F = v * T;
        df->G = 0.0;
        if (oracle_cspline_eval(t->n_T, t->T, t->betabulk, sbb, T, &v)) return -1;
        df->betabulk = v * T4;
        df->betaV = 1.0;
        if (oracle_cspline_eval(t->n_T, t->T, t->betapi, sbp, T, &v)) return -1;
        df->betapi = v * T4;
    } else {
        return -2;
    }
    return 0;
}

/* Deltaf_Data::bilinear_interpolation (deltafReader.cpp:412-484), INTENDED indexing: the reference's
 * calculate_bilinear reads f_data[iT][imuB] (:404-407) although the tables are stored [imuB][iT]
 * (:168-196) -- a transposed-index defect (out of bounds for T > 0.18 GeV); the restatement uses
 * f_data[imuB][iT].  Returns -1 outside the table (reference: printf + exit(-1), :423-427). */
static int eval_df_bilinear(const oracle_df_tables *t, int df_mode, double T, double muB, df_coeff *df)
{
    memset(df, 0, sizeof(*df));
    const int nT = t->n_T, nB = t->n_muB;
    if (nB < 2 || !t->muB) return -4;
    const double T_min = t->T[0], muB_min = t->muB[0];
    const double dT = fabs(t->T[1] - t->T[0]), dmuB = fabs(t->muB[1] - t->muB[0]);
    int iTL = (int)floor((T - T_min) / dT), iTR = iTL + 1;
    int iBL = (int)floor((muB - muB_min) / dmuB), iBR = iBL + 1;
    if (!(iTL >= 0 && iTR < nT) || !(iBL >= 0 && iBR < nB)) return -1;
    const double TL = t->T[iTL], TR = t->T[iTR], BL = t->muB[iBL], BR = t->muB[iBR];
    double v[10];
    for (int k = 0; k < 10; k++) {
        const double *f = t->t2d[k];
        if (!f) return -4;
        double f_LL = f[(size_t)iBL * nT + iTL], f_LR = f[(size_t)iBR * nT + iTL];
        double f_RL = f[(size_t)iBL * nT + iTR], f_RR = f[(size_t)iBR * nT + iTR];
        v[k] = ((f_LL * (TR - T) + f_RL * (T - TL)) * (BR - muB) + (f_LR * (TR - T) + f_RR * (T - TL)) * (muB - BL)) / (dT * dmuB);
    }
    double T3 = T * T * T, T4 = T3 * T, T5 = T4 * T;
    if (df_mode == 1) {                                        /* :436-452 */
        df->c0 = v[0] / T4; df->c1 = v[1] / T3; df->c2 = v[2] / T4; df->c3 = v[3] / T4; df->c4 = v[4] / T5;
    } else if (df_mode == 2) {                                 /* :454-468 */
        df->F = v[5] * T; df->G = v[6]; df->betabulk = v[7] * T4; df->betaV = v[8] * T3; df->betapi = v[9] * T4;
    } else {
        return -2;
    }
    return 0;
}

/* out10 = {c0,c1,c2,c3,c4,F,G,betabulk,betaV,betapi} by the bilinear branch */
int oracle_df_coefficients_bilinear(const oracle_df_tables *t, int df_mode, double T, double muB, double *out10)
{
    df_coeff df;
    int rc = eval_df_bilinear(t, df_mode, T, muB, &df);
    out10[0] = df.c0; out10[1] = df.c1; out10[2] = df.c2; out10[3] = df.c3; out10[4] = df.c4;
    out10[5] = df.F; out10[6] = df.G; out10[7] = df.betabulk; out10[8] = df.betaV; out10[9] = df.betapi;
    return rc;
}

/* construct_cubic_splines (deltafReader.cpp:300-322) on the mu_B = 0 rows; s holds 5 x n_T second-derivative arrays */
static void init_splines(const oracle_df_tables *t, double *s)
{
    const int n = t->n_T;
    const double *rows[5] = {t->c0, t->c2, t->F, t->betabulk, t->betapi};
    for (int k = 0; k < 5; k++)
        if (rows[k]) oracle_cspline_init(n, t->T, rows[k], s + (size_t)k * n);
}

/* Exposed so tests can pin the coefficient evaluation on its own. out = {c0,c2,F,betabulk,betapi} */
int oracle_df_coefficients(const oracle_df_tables *t, int df_mode, double T, double *out5)
{
    int n = t->n_T, rc;
    double *s = (double *)calloc((size_t)5 * n, sizeof(double));
    init_splines(t, s);
    df_coeff df;
    rc = eval_df(t, s, s + n, s + 2 * n, s + 3 * n, s + 4 * n, df_mode, T, &df);
    out5[0] = df.c0; out5[1] = df.c2; out5[2] = df.F; out5[3] = df.betabulk; out5[4] = df.betapi;
    free(s);
    return rc;
}

/* Everything about one cell that the momentum loops need: smooth_kernels.cpp:118-242 */
typedef struct {
    int skip;
    double tau, tau2, eta, dat, dax, day, dan, ux, uy, un, ut, T;
    double pitt, pitx, pity, pitn, pixx, pixy, pixn, piyy, piyn, pinn, bulkPi;
    double alphaB, baryon_enthalpy_ratio, Vt, Vx, Vy, Vn;
    double c3, c4, betaV;
    double shear_coeff, bulk0_coeff, bulk1_coeff, bulk2_coeff;
} cell_ctx;

typedef struct {
    const double *T, *P, *E, *tau, *eta, *ux, *uy, *un, *dat, *dax, *day, *dan;
    const double *pixx, *pixy, *pixn, *piyy, *piyn, *bulkPi, *muB, *nB, *Vx, *Vy, *Vn;
} cell_arrays;

static int load_cell(const cell_arrays *a, long ic, const oracle_opts *o, const oracle_df_tables *t,
                     const double *sc0, const double *sc2, const double *sF, const double *sbb,
                     const double *sbp, cell_ctx *c)
{
    memset(c, 0, sizeof(*c));
    double tau = a->tau[ic];                                   /* :118 */
    double tau2 = tau * tau;                                   /* :119 */
    c->tau = tau; c->tau2 = tau2;
    c->eta = (o->dimension == 3) ? a->eta[ic] : 0.0;           /* :120-123, private per cell */
    double dat = a->dat[ic], dax = a->dax[ic], day = a->day[ic], dan = a->dan[ic]; /* :125-128 */
    double ux = a->ux[ic], uy = a->uy[ic], un = a->un[ic];     /* :130-132 */
    double ut = sqrt(1.0 + ux * ux + uy * uy + tau2 * un * un); /* :133 */
    double udsigma = ut * dat + ux * dax + uy * day + un * dan; /* :135 */
    c->dat = dat; c->dax = dax; c->day = day; c->dan = dan;
    c->ux = ux; c->uy = uy; c->un = un; c->ut = ut;
    if (udsigma <= 0.0) { c->skip = 1; return 0; }             /* :137 */

    double ux2 = ux * ux, uy2 = uy * uy, ut2 = ut * ut;        /* :139-141 */
    double utperp = sqrt(1.0 + ux * ux + uy * uy);             /* :142 */
    double T = a->T[ic], P = a->P[ic], E = a->E[ic];           /* :144-146 */
    c->T = T;

    if (o->include_shear_deltaf) {                             /* :159-171 */
        double pixx = a->pixx[ic], pixy = a->pixy[ic], pixn = a->pixn[ic];
        double piyy = a->piyy[ic], piyn = a->piyn[ic];
        double pinn = (pixx * (ux2 - ut2) + piyy * (uy2 - ut2) + 2.0 * (pixy * ux * uy + tau2 * un * (pixn * ux + piyn * uy))) / (tau2 * utperp * utperp);
        double pitn = (pixn * ux + piyn * uy + tau2 * pinn * un) / ut;
        double pity = (pixy * ux + piyy * uy + tau2 * piyn * un) / ut;
        double pitx = (pixx * ux + pixy * uy + tau2 * pixn * un) / ut;
        double pitt = (pitx * ux + pity * uy + tau2 * pitn * un) / ut;
        c->pixx = pixx; c->pixy = pixy; c->pixn = pixn; c->piyy = piyy; c->piyn = piyn;
        c->pinn = pinn; c->pitn = pitn; c->pity = pity; c->pitx = pitx; c->pitt = pitt;
    }
    if (o->include_bulk_deltaf) c->bulkPi = a->bulkPi[ic];     /* :173-175 */

    double muB = 0.0;
    if (o->include_baryon && o->include_baryondiff_deltaf) {   /* :186-197 */
        muB = a->muB[ic];
        double nB = a->nB[ic];
        c->Vx = a->Vx[ic]; c->Vy = a->Vy[ic]; c->Vn = a->Vn[ic];
        c->Vt = (c->Vx * ux + c->Vy * uy + tau2 * c->Vn * un) / ut;
        c->alphaB = muB / T;
        c->baryon_enthalpy_ratio = nB / (E + P);
    }

    df_coeff df;                                               /* :200, deltafReader.cpp:486-504 */
    if (o->include_baryon) {
        int brc = eval_df_bilinear(t, o->df_mode, T, muB, &df);
        if (brc) return brc;
    } else if (eval_df(t, sc0, sc2, sF, sbb, sbp, o->df_mode, T, &df)) return -1;
    c->c3 = df.c3; c->c4 = df.c4; c->betaV = df.betaV;

    switch (o->df_mode) {                                      /* :220-242 */
    case 1:
        c->shear_coeff = 0.5 / (T * T * (E + P));
        c->bulk0_coeff = df.c0 - df.c2;
        c->bulk1_coeff = df.c1;
        c->bulk2_coeff = 4.0 * df.c2 - df.c0;
        break;
    case 2:
        c->shear_coeff = 0.5 / (df.betapi * T);
        c->bulk0_coeff = df.F / (T * T * df.betabulk);
        c->bulk1_coeff = df.G / df.betabulk;
        c->bulk2_coeff = 1.0 / (3.0 * T * df.betabulk);
        break;
    default:
        return -2;
    }
    return 0;
}

/* One (cell, species, pT, phi, y) value: the eta loop of smooth_kernels.cpp:271-335, operation
 * order kept as in the reference.  Returns pdotdsigma_f_eta_sum. */
static inline double eta_sum(const cell_ctx *c, const oracle_opts *o, double mass2, double sign,
                             double baryon, double mT, double mT_over_tau, double px, double py,
                             double y, int eta_pts, const double *etaValues, const double *etaWeights)
{
    double chem = baryon * c->alphaB;                          /* :254 */
    double sum = 0.0;
    for (int ieta = 0; ieta < eta_pts; ieta++) {
        double eta = etaValues[ieta];
        double eta_weight = etaWeights[ieta];
        double pt = mT * cosh(y - eta);                        /* :279 */
        double pn = mT_over_tau * sinh(y - eta);               /* :280 */
        double tau2_pn = c->tau2 * pn;                         /* :281 */
        double pdotdsigma = eta_weight * (pt * c->dat + px * c->dax + py * c->day + pn * c->dan); /* :283 */
        if (o->outflow && pdotdsigma <= 0.0) continue;         /* :285 */
        double pdotu = pt * c->ut - px * c->ux - py * c->uy - tau2_pn * c->un; /* :287 */
        double feq = 1.0 / (exp(pdotu / c->T - chem) + sign);  /* :289 */
        double feqbar = 1.0 - sign * feq;                      /* :290 */
        double pimunu_pmu_pnu = c->pitt * pt * pt + c->pixx * px * px + c->piyy * py * py + c->pinn * tau2_pn * tau2_pn
            + 2.0 * (-(c->pitx * px + c->pity * py) * pt + c->pixy * px * py + tau2_pn * (c->pixn * px + c->piyn * py - c->pitn * pt)); /* :293-294 */
        double Vmu_pmu = c->Vt * pt - c->Vx * px - c->Vy * py - c->Vn * tau2_pn; /* :297 */
        double df;
        if (o->df_mode == 1) {                                 /* :303-312 */
            double df_shear = c->shear_coeff * pimunu_pmu_pnu;
            double df_bulk = (c->bulk0_coeff * mass2 + (c->bulk1_coeff * baryon + c->bulk2_coeff * pdotu) * pdotu) * c->bulkPi;
            double df_diff = (c->c3 * baryon + c->c4 * pdotu) * Vmu_pmu;
            df = feqbar * (df_shear + df_bulk + df_diff);
        } else {                                               /* :313-321 */
            double df_shear = c->shear_coeff * pimunu_pmu_pnu / pdotu;
            double df_bulk = (c->bulk0_coeff * pdotu + c->bulk1_coeff * baryon + c->bulk2_coeff * (pdotu - mass2 / pdotu)) * c->bulkPi;
            double df_diff = (c->baryon_enthalpy_ratio - baryon / pdotu) * Vmu_pmu / c->betaV;
            df = feqbar * (df_shear + df_bulk + df_diff);
        }
        if (o->regulate_deltaf) df = fmax(-1.0, fmin(df, 1.0)); /* :328 */
        double f = feq * (1.0 + df);                           /* :330 */
        sum += (pdotdsigma * f);                               /* :332 */
    }
    return sum;
}

typedef struct {
    int pT_tab_length; const double *pT;
    int phi_tab_length; const double *phi;
    int y_tab_length; const double *y;
    int eta_tab_length; const double *eta, *eta_w;
} oracle_grid;

static int check_inputs(const oracle_opts *o, const oracle_grid *g)
{
    if (o->dimension != 2 && o->dimension != 3) return -3;
    if (o->df_mode != 1 && o->df_mode != 2) return -2;
    /* include_baryon = 1 needs the full (mu_B, T) tables: checked in eval_df_bilinear (-4 when absent) */
    if (g->pT_tab_length < 1 || g->phi_tab_length < 1) return -3;
    return 0;
}

/*
 * Variant B ("port"): intended semantics, no scratch array; per-thread partial spectra combined in
 * thread order.  dN_pTdpTdphidy has npart * npT * nphi * y_tab_length entries (sized with
 * y_tab_length even in 2+1D, emissionfunction.cpp:276) and is ACCUMULATED INTO (+=, :375).
 * Index: iS3D = ipart + npart * (ipT + npT * (iphip + nphi * iy))   (:363)
 * Returns 0, or <0: -1 T outside the coefficient table (GSL would abort), -2 df_mode, -3 bad
 * dimension/grid, -4 include_baryon = 1 without the full (mu_B, T) tables.
 */
int oracle_dN_pTdpTdphidy(long FO_length, int npart, const double *Mass, const double *Sign,
                          const double *Degeneracy, const double *Baryon, const cell_arrays *a,
                          const oracle_df_tables *t, const oracle_grid *g, const oracle_opts *o,
                          double *dN_pTdpTdphidy)
{
    int rc = check_inputs(o, g);
    if (rc) return rc;
    const double prefactor = pow(2.0 * M_PI * ORACLE_HBARC, -3); /* :36 */
    const int npT = g->pT_tab_length, nphi = g->phi_tab_length;
    double *cosphi = (double *)malloc(sizeof(double) * nphi), *sinphi = (double *)malloc(sizeof(double) * nphi);
    for (int i = 0; i < nphi; i++) { cosphi[i] = cos(g->phi[i]); sinphi[i] = sin(g->phi[i]); } /* :43-48 */

    int y_pts = g->y_tab_length, eta_pts = 1;                  /* :59-66 */
    if (o->dimension == 2) { y_pts = 1; eta_pts = g->eta_tab_length; }
    double *yValues = (double *)malloc(sizeof(double) * (y_pts > 0 ? y_pts : 1));
    if (o->dimension == 2) yValues[0] = 0.0;                   /* :75 */
    else for (int iy = 0; iy < y_pts; iy++) yValues[iy] = g->y[iy]; /* :88-91 */

    int n = t->n_T;
    double *s = (double *)calloc((size_t)5 * n, sizeof(double));
    init_splines(t, s);

    const long long nspec = (long long)npart * npT * nphi * y_pts;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    double *part = (double *)calloc((size_t)nspec * nthreads, sizeof(double));
    int err = 0;

#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *acc = part + (size_t)tid * nspec;
#pragma omp for schedule(static)
        for (long ic = 0; ic < FO_length; ic++) {
            cell_ctx c;
            int lrc = load_cell(a, ic, o, t, s, s + n, s + 2 * n, s + 3 * n, s + 4 * n, &c);
            if (lrc) {
#pragma omp atomic write
                err = lrc;
                continue;
            }
            if (c.skip) continue;
            double eta1 = c.eta, w1 = 1.0;                     /* :86-87, :122 */
            const double *etaValues = (o->dimension == 2) ? g->eta : &eta1;
            const double *etaWeights = (o->dimension == 2) ? g->eta_w : &w1;
            for (int ipart = 0; ipart < npart; ipart++) {      /* :246-254 */
                double mass = Mass[ipart], mass2 = mass * mass, sign = Sign[ipart];
                double degeneracy = Degeneracy[ipart], baryon = Baryon[ipart];
                for (int ipT = 0; ipT < npT; ipT++) {          /* :256-260 */
                    double pT = g->pT[ipT];
                    double mT = sqrt(mass2 + pT * pT);
                    double mT_over_tau = mT / c.tau;
                    for (int iphip = 0; iphip < nphi; iphip++) { /* :262-265 */
                        double px = pT * cosphi[iphip], py = pT * sinphi[iphip];
                        for (int iy = 0; iy < y_pts; iy++) {   /* :267-339 */
                            double v = eta_sum(&c, o, mass2, sign, baryon, mT, mT_over_tau, px, py,
                                               yValues[iy], eta_pts, etaValues, etaWeights);
                            long long iS3D = (long long)ipart + (long long)npart * ((long long)ipT + (long long)npT * ((long long)iphip + (long long)nphi * (long long)iy));
                            acc[iS3D] += (prefactor * degeneracy * v); /* :339, :371 */
                        }
                    }
                }
            }
        }
    }
    if (!err) {
        for (long long i = 0; i < nspec; i++) {
            double tot = 0.0;
            for (int th = 0; th < nthreads; th++) tot += part[(size_t)th * nspec + i];
            dN_pTdpTdphidy[i] += tot;                          /* :375 */
        }
    }
    free(part); free(s); free(yValues); free(cosphi); free(sinphi);
    return err;
}

/*
 * Variant A ("reference-shaped"): the reference's own structure -- FO_chunk = 10000 cells
 * (:37), one scratch array npart*FO_chunk*npT*nphi*y_tab_length (:98), an OpenMP loop over the
 * cells of a chunk that STORES every value (:106-349), then a collapse(4) loop over bins with a
 * simd reduction over the chunk's cells (:354-383).  Differences from the reference, on purpose:
 * private eta, and the scratch slots of a skipped cell are zeroed.  Used only to time what the
 * reference's memory-bound structure costs on the host (bench.py cpu_baseline, "sample").
 */
int oracle_dN_pTdpTdphidy_chunked(long FO_length, int npart, const double *Mass, const double *Sign,
                                  const double *Degeneracy, const double *Baryon, const cell_arrays *a,
                                  const oracle_df_tables *t, const oracle_grid *g, const oracle_opts *o,
                                  long FO_chunk, double *dN_pTdpTdphidy)
{
    int rc = check_inputs(o, g);
    if (rc) return rc;
    if (FO_chunk <= 0) FO_chunk = 10000;
    const double prefactor = pow(2.0 * M_PI * ORACLE_HBARC, -3);
    const int npT = g->pT_tab_length, nphi = g->phi_tab_length;
    double *cosphi = (double *)malloc(sizeof(double) * nphi), *sinphi = (double *)malloc(sizeof(double) * nphi);
    for (int i = 0; i < nphi; i++) { cosphi[i] = cos(g->phi[i]); sinphi[i] = sin(g->phi[i]); }
    int y_pts = g->y_tab_length, eta_pts = 1;
    if (o->dimension == 2) { y_pts = 1; eta_pts = g->eta_tab_length; }
    double *yValues = (double *)malloc(sizeof(double) * (y_pts > 0 ? y_pts : 1));
    if (o->dimension == 2) yValues[0] = 0.0;
    else for (int iy = 0; iy < y_pts; iy++) yValues[iy] = g->y[iy];
    int n = t->n_T;
    double *s = (double *)calloc((size_t)5 * n, sizeof(double));
    init_splines(t, s);

    /* the reference sizes the scratch with y_tab_length even in 2+1D (:98); only iy < y_pts is touched */
    size_t scratch_n = (size_t)npart * (size_t)FO_chunk * npT * nphi * (size_t)y_pts;
    double *all = (double *)calloc(scratch_n, sizeof(double));
    if (!all) { free(s); free(yValues); free(cosphi); free(sinphi); return -5; }
    int err = 0;

    for (long nchunk = 0; nchunk < (FO_length / FO_chunk) + 1; nchunk++) {   /* :102 */
        long endFO = FO_chunk;
        if (nchunk == (FO_length / FO_chunk)) endFO = FO_length - (nchunk * FO_chunk); /* :105 */
#pragma omp parallel for
        for (long icell = 0; icell < endFO; icell++) {
            long ic = nchunk * FO_chunk + icell;
            cell_ctx c;
            int lrc = load_cell(a, ic, o, t, s, s + n, s + 2 * n, s + 3 * n, s + 4 * n, &c);
            if (lrc) {
#pragma omp atomic write
                err = lrc;
                c.skip = 1;
            }
            double eta1 = c.eta, w1 = 1.0;
            const double *etaValues = (o->dimension == 2) ? g->eta : &eta1;
            const double *etaWeights = (o->dimension == 2) ? g->eta_w : &w1;
            for (int ipart = 0; ipart < npart; ipart++) {
                double mass = Mass[ipart], mass2 = mass * mass, sign = Sign[ipart];
                double degeneracy = Degeneracy[ipart], baryon = Baryon[ipart];
                for (int ipT = 0; ipT < npT; ipT++) {
                    double pT = g->pT[ipT];
                    double mT = sqrt(mass2 + pT * pT);
                    double mT_over_tau = mT / c.tau;
                    for (int iphip = 0; iphip < nphi; iphip++) {
                        double px = pT * cosphi[iphip], py = pT * sinphi[iphip];
                        for (int iy = 0; iy < y_pts; iy++) {
                            double v = c.skip ? 0.0 : eta_sum(&c, o, mass2, sign, baryon, mT, mT_over_tau, px, py,
                                                              yValues[iy], eta_pts, etaValues, etaWeights);
                            long long iSpectra = (long long)icell + (long long)endFO * ((long long)ipart + (long long)npart * ((long long)ipT + (long long)npT * ((long long)iphip + (long long)nphi * (long long)iy))); /* :337 */
                            all[iSpectra] = (prefactor * degeneracy * v);   /* :339 */
                        }
                    }
                }
            }
        }
        if (endFO != 0) {                                      /* :351-385 */
#pragma omp parallel for collapse(4)
            for (int ipart = 0; ipart < npart; ipart++)
                for (int ipT = 0; ipT < npT; ipT++)
                    for (int iphip = 0; iphip < nphi; iphip++)
                        for (int iy = 0; iy < y_pts; iy++) {
                            long long iS3D = (long long)ipart + (long long)npart * ((long long)ipT + (long long)npT * ((long long)iphip + (long long)nphi * (long long)iy));
                            double tmp = 0.0;
#pragma omp simd reduction(+ : tmp)
                            for (long icell = 0; icell < endFO; icell++) {
                                long long iSpectra = (long long)icell + (long long)endFO * iS3D;
                                tmp += all[iSpectra];
                            }
                            dN_pTdpTdphidy[iS3D] += tmp;
                        }
        }
    }
    free(all); free(s); free(yValues); free(cosphi); free(sinphi);
    return err;
}

void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
