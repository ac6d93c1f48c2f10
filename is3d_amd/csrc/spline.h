// spline.h -- natural cubic spline coefficients, the replacement for the GSL objects the reference
// builds in Deltaf_Data::construct_cubic_splines (/root/reference/src/cpp/deltafReader.cpp:300-322:
// gsl_spline_alloc(gsl_interp_cspline, n) + gsl_spline_init).  GSL is neither vendored by the
// reference nor present in this image; this follows the published algorithm of gsl_interp_cspline:
// natural end conditions c_0 = c_{n-1} = 0 and, for the interior second-derivative coefficients,
// the symmetric tridiagonal system
//     h_{i-1} c_{i-1} + 2 (h_{i-1} + h_i) c_i + h_i c_{i+1} = 3 (dy_i / h_i - dy_{i-1} / h_{i-1})
// solved by an LDL^T sweep.  Evaluation (device side, cf_kernels.hip::spline_eval_lds):
//     y = y_i + d (b_i + d (c_i + d d_i)),  b_i = dy_i/h_i - h_i (c_{i+1} + 2 c_i)/3,  d_i = (c_{i+1} - c_i)/(3 h_i).
#pragma once
#include <vector>

namespace is3d {

inline bool natural_cspline_init(const std::vector<double> &x, const std::vector<double> &y, std::vector<double> &c)
{
    const int n = (int)x.size();
    if (n < 3 || (int)y.size() != n) return false;
    c.assign(n, 0.0);
    const int m = n - 2;  // interior unknowns c_1 .. c_{n-2}
    std::vector<double> diag(m), off(m), rhs(m);
    for (int i = 0; i < m; i++) {
        const double h0 = x[i + 1] - x[i], h1 = x[i + 2] - x[i + 1];
        if (!(h0 > 0.0) || !(h1 > 0.0)) return false;
        diag[i] = 2.0 * (h1 + h0);
        off[i] = h1;
        rhs[i] = 3.0 * ((y[i + 2] - y[i + 1]) * (1.0 / h1) - (y[i + 1] - y[i]) * (1.0 / h0));
    }
    if (m == 1) {
        c[1] = rhs[0] / diag[0];
        return true;
    }
    std::vector<double> alpha(m), gamma(m), z(m);
    alpha[0] = diag[0];
    gamma[0] = off[0] / alpha[0];
    for (int i = 1; i < m - 1; i++) {
        alpha[i] = diag[i] - off[i - 1] * gamma[i - 1];
        gamma[i] = off[i] / alpha[i];
    }
    alpha[m - 1] = diag[m - 1] - off[m - 2] * gamma[m - 2];
    z[0] = rhs[0];
    for (int i = 1; i < m; i++) z[i] = rhs[i] - gamma[i - 1] * z[i - 1];
    for (int i = 0; i < m; i++) z[i] /= alpha[i];
    c[m] = z[m - 1];
    for (int i = m - 2; i >= 0; i--) c[i + 1] = z[i] - gamma[i] * c[i + 2];
    return true;
}

}  // namespace is3d
