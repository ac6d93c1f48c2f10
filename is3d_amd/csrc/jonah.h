// jonah.h -- host table setup of the df_mode 4 ("Jonah") modified equilibrium, shared by the smooth-spectra plan (cf_plan.cpp)
// and the particle sampler (cf_sampler.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/is3d_amd.h"

namespace is3d {

// Deltaf_Data::compute_jonah_coefficients (deltafReader.cpp:222-297): lambda in [-1, 2] on 301 points; for each the
// hadron-gas energy density and pressure with momenta rescaled by (1 + lambda), by 32-point Gauss-Laguerre (alpha = 2)
// quadrature over ALL species of the PDG file at the surface-averaged temperature; z = E/E_mod, Pi/P = (P_mod/P) z - 1.
// Host table setup (the reference does the same once per run); out = {Pi/P, lambda^2, z}.
inline void jonah_tables(const is3d_feqmod_tables *fq, std::vector<double> &bp, std::vector<double> &l2, std::vector<double> &zz,
                         double &bp_max)
{
    const int n = 301;
    const double lambda_min = -1.0, lambda_max = 2.0;
    const double delta_lambda = (lambda_max - lambda_min) / ((double)n - 1.0);
    const double T = fq->T_avg;
    bp.assign(n, 0.0); l2.assign(n, 0.0); zz.assign(n, 0.0);
    auto sums = [&](double lambda, double &E, double &P) {
        E = 0.0; P = 0.0;
        const double scale2 = (1.0 + lambda) * (1.0 + lambda);
        for (int s = 0; s < fq->n_pdg; s++) {
            const double mass = fq->pdg_mass[s], mbar = mass / T, sign = fq->pdg_sign[s];
            if (mass == 0.0) continue;   // photons skipped, :257
            double e = 0.0, pr = 0.0;
            for (int k = 0; k < fq->n_gla; k++) {
                const double pbar = fq->root2[k], w = fq->weight2[k];
                const double Ebar = std::sqrt(pbar * pbar + mbar * mbar);
                const double Es = std::sqrt(pbar * pbar * scale2 + mbar * mbar);
                const double thermal = std::exp(pbar) / (std::exp(Ebar) + sign);
                e += w * (Es * thermal);                               // E_mod_int, gaussThermal.cpp
                pr += w * (pbar * pbar * scale2 / Es * thermal);       // P_mod_int
            }
            E += fq->pdg_degeneracy[s] * e;
            P += (1.0 / 3.0) * fq->pdg_degeneracy[s] * pr;
        }
    };
    double E0, P0;
    sums(0.0, E0, P0);
    bp_max = -1.0;
    for (int i = 0; i < n; i++) {
        const double lambda = lambda_min + (double)i * delta_lambda;
        double Em, Pm;
        sums(lambda, Em, Pm);
        const double z = E0 / Em;
        bp[i] = (Pm / P0) * z - 1.0;
        l2[i] = lambda * lambda;
        zz[i] = z;
        bp_max = std::max(bp_max, bp[i]);
    }
}

}  // namespace is3d
