"""GPU (-m gpu): the elementary functions of is3d_amd/csrc/cf_math.h against numpy long double, through is3d_math_probe.  The accuracy
figures quoted in DESIGN.md and in the kernels' comments are the assertions here."""
import numpy as np
import pytest

from is3d_amd import api

pytestmark = pytest.mark.gpu
LD = np.longdouble


def _rel(got, ref):
    ref = np.asarray(ref, dtype=LD)
    return float(np.max(np.abs(got.astype(LD) - ref) / np.abs(ref)))


def test_exponentials():
    rng = np.random.default_rng(1)
    x = np.concatenate([-rng.random(200000) * 700.0, -rng.random(50000) * 2.0, rng.random(20000) * 20.0, [0.0, -1e-300, -0.5 * np.log(2), -708.0]])
    ref = np.exp(x.astype(LD))
    assert _rel(api.math_probe("exp_full", x), ref) < 2.5e-15          # Cody-Waite + degree 10: 1.4e-15 on the reduced interval, + the reduction
    assert _rel(api.math_probe("exp_full_sat", x), ref) < 2.5e-15
    e9 = _rel(api.math_probe("exp_p9", x), ref)
    assert e9 < 7e-14, e9                                              # one-fma reduction (|n| 2.3e-17) + degree 9 (4.5e-14)
    assert _rel(api.math_probe("exp_p9_sat", x), ref) < 7e-14
    assert np.array_equal(api.math_probe("exp_p9", x), api.math_probe("exp_p9_sat", x))   # the two conversions agree wherever both are defined
    # a power-of-two factor through the shift constant (cf_main_feqmod: the cell's p.dsigma scale): 2^5 e^x, the bits of 32 * exp_p9(x)
    norm = x[x > -700.0]
    assert np.array_equal(api.math_probe("exp_p9_x32", norm), 32.0 * api.math_probe("exp_p9", norm))
    # the exact-zero rule of the row culls: e^v is exactly +0 for v < -745.2 in every variant, and the saturating forms take any argument
    far = np.array([-745.25, -746.0, -1000.0, -1.0e6, -1.3e9])
    for f in ("exp_full", "exp_p9", "exp_full_sat", "exp_p9_sat"):
        assert not api.math_probe(f, far).any(), f
    # the saturating forms: exactly +0 for arguments far beyond the shift trick's domain -- up to ~1e45, where the reduced argument's
    # polynomial overflows; what this test found: they are NOT defined for every double (e^-1e300 came out inf, e^-inf nan), so the prep
    # kernels bound the argument of every per-evaluation exponential instead (status[7]: IS3D_EDOMAIN) and the main kernels use exp_p9
    huge = np.array([-1.0e12, -1.0e20, -1.0e30, -1.0e40])
    for f in ("exp_full_sat", "exp_p9_sat"):
        assert not api.math_probe(f, huge).any(), f
    # denormal results stay within a few ulps of the denormal grid
    den = np.array([-709.0, -720.0, -740.0, -744.0])
    got = api.math_probe("exp_p9", den)
    assert np.all(np.abs(got - np.exp(den)) <= 4 * 4.9406564584124654e-324 + 1e-13 * np.exp(den))


def test_square_roots_and_reciprocals():
    rng = np.random.default_rng(2)
    x = np.exp(rng.uniform(np.log(1e-200), np.log(1e200), 200000))
    assert _rel(api.math_probe("sqrt_g1", x), np.sqrt(x.astype(LD))) < 4e-15     # v_rsq_f64 (4.5e-8) + one Goldschmidt step: 1.5 e0^2
    assert _rel(api.math_probe("sqrt_nr", x), np.sqrt(x.astype(LD))) < 3e-16
    d = np.concatenate([x[:100000], -x[100000:]])
    assert _rel(api.math_probe("rcp_nr1", d), 1 / d.astype(LD)) < 4e-15           # v_rcp_f64 + one Newton step
    assert _rel(api.math_probe("rcp_nr", d), 1 / d.astype(LD)) < 3e-16
    api.math_probe("rcp_nr", d[:1])                                                 # sets the argument types
    assert api.load().is3d_math_probe(10, 0, None, None, -1) == api.IS3D_EINVAL      # an unknown function is refused
