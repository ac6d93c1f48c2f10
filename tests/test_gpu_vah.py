"""GPU (-m gpu): the anisotropic-hydro (VAH, P_L matching) smooth kernel (is3d_smooth_spectra_vah; BASELINE config 5) through
the C ABI against the oracle's restatement of calculate_dN_pTdpTdphidy_VAH_PL (smooth_kernels.cpp:2140-2393)."""
import numpy as np
import pytest

from conftest import relerr
from is3d_amd import api, inputs, synth
from oracle import oracle  # the checker

pytestmark = pytest.mark.gpu
TOL = 2e-9


@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("flags", [dict(), dict(regulate_deltaf=0), dict(include_bulk_deltaf=0), dict(include_shear_deltaf=0)])
def test_vah_parity(fx, dim, flags):
    cells = synth.synth_vah_surface(70 if dim == 3 else 9, dim, seed=900 + dim)
    cells["dat"][3] *= -1.0                      # u.dsigma < 0 is NOT skipped on this path (negative contributions stay)
    sp = inputs.species([211, 321, 2212, -2212, 3122, 333]) if dim == 3 else fx["pikp"]
    o = dict(dimension=dim, **flags)
    ref = oracle.dN_pTdpTdphidy_vah(cells, sp, fx["grid"], o)
    got, st = api.smooth_spectra_vah(cells, sp, fx["grid"], o)
    assert relerr(got, ref, floor=1e-270) < TOL, relerr(got, ref, floor=1e-270)
    assert (ref < 0).any() or dim == 2


def test_vah_species_collapse_passes_and_accumulate(fx):
    cells = synth.synth_vah_surface(300, 3, seed=910)
    sp = fx["urqmd"]
    g = dict(fx["grid"], pT=fx["grid"]["pT"][::4], phi=fx["grid"]["phi"][::3])
    o = dict(dimension=3)
    ref = oracle.dN_pTdpTdphidy_vah(cells, sp, g, o)
    one, st = api.smooth_spectra_vah(cells, sp, g, o)
    assert st["n_classes"] == 75 and relerr(one, ref, floor=1e-270) < TOL
    for extra in (dict(workspace_bytes=1 << 20), dict(cell_chunks=3), dict(zero_skip=2), dict(collapse_species=2)):
        got, st2 = api.smooth_spectra_vah(cells, sp, g, dict(o, **extra))
        if "workspace_bytes" in extra:
            assert st2["n_passes"] > 1
        assert relerr(got, one, floor=1e-270) < 1e-12, extra
    acc = 2.0 * one
    api.smooth_spectra_vah(cells, sp, g, dict(o, accumulate=1), out=acc)
    assert relerr(acc, 3.0 * one, floor=1e-270) < 1e-12
    empty, _ = api.smooth_spectra_vah({k: v[:0] for k, v in cells.items()}, sp, g, o)
    assert (empty == 0).all()
    with pytest.raises(api.Is3dError):
        api.smooth_spectra_vah({k: v for k, v in cells.items() if k != "aL"}, sp, g, o)


def test_vah_config5_size_properties(fx):
    """BASELINE config 5's stated size for the VAH kernel: 1e6 cells x 305 species x 16 128 bins.  No reference behaviour exists
    (the reference never calls this kernel), so properties: finite, shard additivity over three uneven shards, dsigma-linearity on
    a slice, reproducibility."""
    n = 1000000
    cells = synth.synth_vah_surface(n, 3)
    sp = fx["urqmd"]
    o = dict(dimension=3)
    whole, st = api.smooth_spectra_vah(cells, sp, fx["grid"], o)
    assert np.isfinite(whole).all() and st["n_classes"] == 75
    parts = np.zeros_like(whole)
    for lo, hi in ((0, 333333), (333333, 700001), (700001, n)):
        p, _ = api.smooth_spectra_vah({k: v[lo:hi] for k, v in cells.items()}, sp, fx["grid"], o)
        parts += p
    assert relerr(parts, whole, floor=1e-250) < 1e-10
    again, _ = api.smooth_spectra_vah(cells, sp, fx["grid"], o)
    assert np.array_equal(again, whole)
    sl = {k: v[:50000] for k, v in cells.items()}
    a, _ = api.smooth_spectra_vah(sl, sp, fx["grid"], o)
    b, _ = api.smooth_spectra_vah({k: (3.0 * v if k in ("dat", "dax", "day", "dan") else v) for k, v in sl.items()}, sp, fx["grid"], o)
    assert relerr(b, 3.0 * a, floor=1e-250) < 1e-14
    print("vah 1e6 cells: main kernel ms", st["ms_main"], "prep", st["ms_prep"])


def test_vah_full_size_properties(fx):
    """2e5 cells x 305 species: shard additivity and the equilibrium limit against the delta-f kernel with all corrections off."""
    n = 200000
    cells = synth.synth_vah_surface(n, 3)
    sp = fx["urqmd"]
    o = dict(dimension=3)
    whole, st = api.smooth_spectra_vah(cells, sp, fx["grid"], o)
    assert np.isfinite(whole).all()
    a, _ = api.smooth_spectra_vah({k: v[:70000] for k, v in cells.items()}, sp, fx["grid"], o)
    b, _ = api.smooth_spectra_vah({k: v[70000:] for k, v in cells.items()}, sp, fx["grid"], o)
    assert relerr(a + b, whole, floor=1e-250) < 1e-10
    eq = {k: v.copy() for k, v in cells.items()}
    eq["aL"][:] = 1.0
    eq["Lambda"] = eq["T"].copy()
    for k in ("c0", "c1", "c2", "c3", "c4"):
        eq[k][:] = 0.0
    got, _ = api.smooth_spectra_vah(eq, sp, fx["grid"], o)
    vh = synth.synth_surface(n, 3)
    ref, _ = api.smooth_spectra(vh, sp, fx["grid"], fx["df"], dict(dimension=3, df_mode=1, include_bulk_deltaf=0, include_shear_deltaf=0, outflow=0))
    assert relerr(got, ref, floor=1e-250) < 1e-9
    print("vah main kernel ms", st["ms_main"], "prep", st["ms_prep"])
