"""GPU (-m gpu): the anisotropic-hydro (VAH, P_L matching) smooth kernel (is3d_smooth_spectra_vah; BASELINE config 5) through
the C ABI against the oracle's restatement of calculate_dN_pTdpTdphidy_VAH_PL (smooth_kernels.cpp:2140-2393)."""
import numpy as np
import pytest

from conftest import honoured, relerr
from is3d_amd import api, inputs, synth
from oracle import oracle  # the checker

pytestmark = pytest.mark.gpu
TOL = 2e-9


@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("flags", [dict(), dict(regulate_deltaf=0), dict(include_bulk_deltaf=0), dict(include_shear_deltaf=0)])
@pytest.mark.devlib
def test_vah_parity(fx, dim, flags):
    cells = synth.synth_vah_surface(70 if dim == 3 else 9, dim, seed=900 + dim)
    cells["dat"][3] *= -1.0                      # u.dsigma < 0 is NOT skipped on this path (negative contributions stay)
    sp = inputs.species([211, 321, 2212, -2212, 3122, 333]) if dim == 3 else fx["pikp"]
    o = dict(dimension=dim, **flags)
    ref = oracle.dN_pTdpTdphidy_vah(cells, sp, fx["grid"], o)
    for variant in honoured("vah", dim, (0, 2, 3)):   # 3 (= the default): cf_main_vah3, factored exponent -- 8 x 7 tile in 3+1D, 8 x 31 with unit-strided lanes in 2+1D; 2: the round-1 kernel
        got, st = api.smooth_spectra_vah(cells, sp, fx["grid"], dict(o, kernel_variant=variant))
        assert relerr(got, ref, floor=1e-270) < TOL, (variant, relerr(got, ref, floor=1e-270))
        assert st["kernel_variant"] == (2 if variant == 2 else 3)
        off, _ = api.smooth_spectra_vah(cells, sp, fx["grid"], dict(o, kernel_variant=variant, zero_skip=2))
        assert np.array_equal(off, got), variant         # the row / unit culls skip exact zeros only
    assert (ref < 0).any() or dim == 2
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra_vah(cells, sp, fx["grid"], dict(o, kernel_variant=5))
    assert e.value.code == api.IS3D_EINVAL


@pytest.mark.devlib
def test_vah_exponent_domain_is_refused(fx):
    """The VAH kernels' exponential takes its integer part by the shift trick (|E_a/Lambda| < 1.4e9): cf_prep_vah reports a cell that could
    exceed 1e9 (a Lambda of 1e-12 GeV here) instead of wrapping an exponent; the reference's exp() overflows to inf there."""
    cells = synth.synth_vah_surface(40, 3, seed=3)
    cells["Lambda"][11] = 1e-12
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra_vah(cells, fx["pikp"], fx["grid"], dict(dimension=3))
    assert e.value.code == api.IS3D_EDOMAIN and e.value.bad_cell == 11 and "1e9" in str(e.value)
    for o in (dict(dimension=3, kernel_variant=2), dict(dimension=2)):                                    # every VAH kernel shares the check
        with pytest.raises(api.Is3dError) as e:
            api.smooth_spectra_vah(cells, fx["pikp"], fx["grid"], o)
        assert e.value.code == api.IS3D_EDOMAIN and e.value.bad_cell == 11


@pytest.mark.devlib
def test_vah_species_collapse_passes_and_accumulate(fx):
    cells = synth.synth_vah_surface(300, 3, seed=910)
    sp = fx["urqmd"]
    g = dict(fx["grid"], pT=fx["grid"]["pT"][::4], phi=fx["grid"]["phi"][::3])
    o = dict(dimension=3)
    ref = oracle.dN_pTdpTdphidy_vah(cells, sp, g, o)
    one, st = api.smooth_spectra_vah(cells, sp, g, o)
    assert st["n_classes"] == 75 and relerr(one, ref, floor=1e-270) < TOL
    for extra in (dict(workspace_bytes=1 << 19), dict(cell_chunks=3), dict(zero_skip=2), dict(collapse_species=2), dict(kernel_variant=2)):
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


def test_vah_device_coefficients_are_the_oracles_bit_for_bit():
    """is3d_vah_coefficients (device kernel cf_vah_coeffs) == oracle_vah_coefficients (src/cuda/deltafReader.cu:216-278): the same
    node search and the same expression with every rounding written out -- bitwise equal, extrapolation below the first node
    included; beyond the last node (and NaN) the reference leaves the coefficients unset: IS3D_EDOMAIN with the lowest such cell."""
    tab = inputs.vah_df_tables()
    h = 0.197327053
    rng = np.random.default_rng(5)
    n = 100003
    lam = (0.55 + 0.69 * rng.random(n)) * h
    al = 0.15 + 1.84 * rng.random(n)
    ref, found = oracle.vah_coefficients(tab, lam, al)
    assert found.all() and ((lam / h) < 0.6).any() and (al < 0.2).any()      # all inside or below: every cell gets a value
    got = api.vah_coefficients(tab, lam, al)
    for k in range(5):
        assert np.array_equal(got["c%d" % k], ref["c%d" % k]), (k, relerr(got["c%d" % k], ref["c%d" % k]))
    lam[[70001, 313]] = 1.2501 * h            # beyond the last Lambda node
    al[90000] = 2.0                           # exactly the last alpha_L node: aL < aL[i2] fails for every i2
    lam[95000] = np.nan
    ref, found = oracle.vah_coefficients(tab, lam, al)
    assert list(np.nonzero(~found)[0]) == [313, 70001, 90000, 95000]
    with pytest.raises(api.Is3dError) as e:
        api.vah_coefficients(tab, lam, al)
    assert e.value.code == api.IS3D_EDOMAIN and e.value.bad_cell == 313 and "cell 313" in str(e.value)
    for k in range(5):
        v = e.value.values["c%d" % k]
        assert np.array_equal(v[found], ref["c%d" % k][found]) and not v[~found].any()
    with pytest.raises(api.Is3dError) as e:   # tables must ascend
        api.vah_coefficients(dict(tab, aL=tab["aL"][::-1].copy()), lam[:4], al[:4])
    assert e.value.code == api.IS3D_EINVAL


@pytest.mark.parametrize("dim", [3, 2])
def test_vah_spectrum_from_the_coefficient_tables(fx, dim):
    """BASELINE config 5 from files: the cells carry (Lambda, alpha_L) only, c0..c4 come from the tables on the device == the oracle
    kernel fed with the oracle's coefficients; the device-resident plan gives the same bits as the host entry; a cell outside the
    tables is IS3D_EDOMAIN with its index."""
    import torch
    tab = inputs.vah_df_tables()
    cells = synth.synth_vah_surface(120 if dim == 3 else 9, dim, seed=940 + dim)
    n = len(cells["tau"])
    sp = inputs.species([211, 321, 2212, 3122]) if dim == 3 else fx["pikp"]
    coef, found = oracle.vah_coefficients(tab, cells["Lambda"], cells["aL"])
    assert found.all()
    o = dict(dimension=dim)
    ref = oracle.dN_pTdpTdphidy_vah(dict(cells, **coef), sp, fx["grid"], o)
    got, st = api.smooth_spectra_vah(cells, sp, fx["grid"], o, tab=inputs.vah_df_tables())
    assert relerr(got, ref, floor=1e-270) < TOL, relerr(got, ref, floor=1e-270)
    same, _ = api.smooth_spectra_vah(dict(cells, **coef), sp, fx["grid"], o)       # coefficients as inputs: the same kernel, the same bits
    assert np.array_equal(same, got)
    assert st["n_wave_rows"] > 0
    # device-resident plan, torch-owned memory, a non-default stream
    dev = torch.device("cuda:0")
    fields = [f for f in api.VAH_FIELDS[:25] if f != "T"]
    tens = {k: torch.from_numpy(np.ascontiguousarray(cells[k])).to(dev) for k in fields}
    plan = api.VahPlan(sp, fx["grid"], o, tab=tab, max_cells=n + 5)
    plan.set_timing(True)
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st2 = plan.execute(n, {k: v.data_ptr() for k, v in tens.items()}, out.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert np.array_equal(out.cpu().numpy(), got) and st2["code"] == 0 and st2["n_wave_rows"] == st["n_wave_rows"]
    t = plan.timings()
    assert t["ms_main"] > 0 and t["ms_prep"] > 0
    tens["aL"][7] = 2.5
    with pytest.raises(api.Is3dError) as e:
        plan.execute(n, {k: v.data_ptr() for k, v in tens.items()}, out.data_ptr(), 0)
    assert e.value.code == api.IS3D_EDOMAIN and e.value.bad_cell == 7
    with pytest.raises(api.Is3dError) as e:
        plan.execute(n + 6, {k: v.data_ptr() for k, v in tens.items()}, out.data_ptr(), 0)
    assert e.value.code == api.IS3D_EINVAL
    plan.close()
    # passes over the cell axis keep the global index of an offending cell
    bad = {k: v.copy() for k, v in cells.items()}
    bad["Lambda"][n - 2] = 0.3
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra_vah(bad, sp, fx["grid"], dict(o, workspace_bytes=1 << 18), tab=tab)
    assert e.value.code == api.IS3D_EDOMAIN and e.value.bad_cell == n - 2


def test_vah_surface_file_to_spectrum(fx, tmp_path):
    """Mode-2 surface file -> is3d_surface_read_vah -> spectrum with table coefficients == oracle reader -> oracle coefficients ->
    oracle kernel."""
    tab = inputs.vah_df_tables()
    path = str(tmp_path / "surface.dat")
    src = synth.synth_vah_surface(64, 3, seed=77)
    synth.write_surface_vah_dat(path, src)
    cells = api.surface_read_vah(path, 3)
    o = dict(dimension=3)
    got, _ = api.smooth_spectra_vah(cells, fx["pikp"], fx["grid"], o, tab=tab)
    rc = oracle.read_surf_VAH_PLMatch(path)
    coef, found = oracle.vah_coefficients(tab, rc["Lambda"], rc["aL"])
    assert found.all()
    ref = oracle.dN_pTdpTdphidy_vah(dict(rc, **coef), fx["pikp"], fx["grid"], o)
    assert relerr(got, ref, floor=1e-270) < TOL


@pytest.mark.devlib
def test_vah_odd_grids(fx):
    """Grid lengths that are not multiples of the tiles (phi 5, pT 3, y 5 / 29 / 41 -- more than 32 rows; eta 41 and 7 in 2+1D), both 3+1D kernels,
    with the coefficients from the tables."""
    rng = np.random.default_rng(3)
    tab = inputs.vah_df_tables()
    cells = synth.synth_vah_surface(21, 3, seed=8)
    g = dict(pT=np.array([0.1, 0.7, 2.5]), phi=np.sort(rng.random(5) * 2 * np.pi), y=np.linspace(-2, 2, 5), eta=fx["grid"]["eta"], eta_w=fx["grid"]["eta_w"])
    coef, found = oracle.vah_coefficients(tab, cells["Lambda"], cells["aL"])
    assert found.all()
    for ygrid in (np.linspace(-2, 2, 5), np.linspace(-3.5, 3.5, 29), np.linspace(-4, 4, 41)):
        gg = dict(g, y=ygrid)
        ref = oracle.dN_pTdpTdphidy_vah(dict(cells, **coef), fx["pikp"], gg, dict(dimension=3))
        for variant in honoured("vah", 3, (2, 3)):
            got, _ = api.smooth_spectra_vah(cells, fx["pikp"], gg, dict(dimension=3, kernel_variant=variant), tab=tab)
            assert relerr(got, ref, floor=1e-270) < TOL, (len(ygrid), variant)
    c2 = synth.synth_vah_surface(5, 2, seed=9)
    coef2, _ = oracle.vah_coefficients(tab, c2["Lambda"], c2["aL"])
    for neta in (41, 7):
        eta = np.linspace(-2.0, 2.0, neta)
        w = np.full(neta, 1.0)
        w[[0, -1]] *= 0.5
        gg = dict(g, eta=eta, eta_w=w)
        ref = oracle.dN_pTdpTdphidy_vah(dict(c2, **coef2), fx["pikp"], gg, dict(dimension=2))
        got, _ = api.smooth_spectra_vah(c2, fx["pikp"], gg, dict(dimension=2), tab=tab)
        assert relerr(got, ref, floor=1e-270) < TOL, neta


@pytest.mark.devlib
def test_vah_2d_factored_kernel_against_the_round1_kernel(fx):
    """2+1D: cf_main_vah3<DIM3 = false> (factored exponent, 8 x 31 tile, unit-strided lanes: pi / K / p are 96 bins, four lane slots each) against
    the round-1 cf_main_vah (8 x 61, expanded quadratic form) on 700 cells: the same sums in another order and with another polynomial for the
    exponential -- <= 2e-13 on every bin that is not cancellation noise; cell counts that do not fill the LDS batches; a species list whose bin
    count gives no split; several passes."""
    cells = synth.synth_vah_surface(700, 2, seed=77)
    tab = inputs.vah_df_tables()
    o = dict(dimension=2)
    old, st_old = api.smooth_spectra_vah(cells, fx["pikp"], fx["grid"], dict(o, kernel_variant=2), tab=tab)
    new, st_new = api.smooth_spectra_vah(cells, fx["pikp"], fx["grid"], o, tab=tab)
    assert st_old["kernel_variant"] == (2 if api.DEV_LIB else 3) and st_new["kernel_variant"] == 3   # (the round-1 kernel exists in the developer build; the shipped library runs the default for it)
    scale = np.abs(old).max()
    assert float(np.max(np.abs(new - old) / np.maximum(np.abs(old), 1e-12 * scale))) < 2e-13
    for n in (1, 3, 65):
        sub = {k: v[:n] for k, v in cells.items()}
        a, _ = api.smooth_spectra_vah(sub, fx["pikp"], fx["grid"], dict(o, kernel_variant=2), tab=tab)
        b, _ = api.smooth_spectra_vah(sub, fx["pikp"], fx["grid"], o, tab=tab)
        assert float(np.max(np.abs(b - a) / np.maximum(np.abs(a), 1e-12 * np.abs(a).max()))) < 2e-13, n
    many = inputs.species([211, 321, 2212, -2212, 3122, 333, 111, 221])      # 8 classes x 32 pT = 256 bins: whole waves, no split
    a, _ = api.smooth_spectra_vah(cells, many, fx["grid"], dict(o, kernel_variant=2), tab=tab)
    b, _ = api.smooth_spectra_vah(cells, many, fx["grid"], o, tab=tab)
    assert float(np.max(np.abs(b - a) / np.maximum(np.abs(a), 1e-12 * np.abs(a).max()))) < 2e-13
    c, st = api.smooth_spectra_vah(cells, fx["pikp"], fx["grid"], dict(o, workspace_bytes=1 << 22), tab=tab)   # passes
    assert st["n_passes"] > 1 and float(np.max(np.abs(c - new) / np.maximum(np.abs(new), 1e-12 * scale))) < 1e-13


@pytest.mark.devlib
def test_vah_golden_vectors_on_device():
    """The committed independent vectors (tests/golden/golden_vah.npz: scipy coefficients, numpy long-double spectra) against the device:
    coefficients to 2e-12, spectra -- with the coefficients interpolated on the device from the tables -- to the parity tolerance."""
    import os
    from conftest import ROOT
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_vah.npz"))
    tab = inputs.vah_df_tables()
    c = api.vah_coefficients(tab, z["coef_Lambda"], z["coef_aL"])
    for k in range(5):
        assert relerr(c["c%d" % k], z["coef_c%d" % k], floor=1e-300) < 2e-12, k
    sp = inputs.species([int(i) for i in z["species"]])
    for dim in (3, 2):
        cells = {k[len("cells%d_" % dim):]: z[k] for k in z.files if k.startswith("cells%d_" % dim)}
        grid = {k[len("grid%d_" % dim):]: z[k] for k in z.files if k.startswith("grid%d_" % dim)}
        for reg in (1, 0):
            for variant in honoured("vah", dim, (0, 2)):
                got, _ = api.smooth_spectra_vah(cells, sp, grid, dict(dimension=dim, regulate_deltaf=reg, kernel_variant=variant), tab=tab)
                assert relerr(got, z["dN%d_reg%d" % (dim, reg)], floor=1e-270) < TOL, (dim, reg, variant)


def test_vah_full_size_config5_from_tables_stratified_oracle_sample(fx):
    """BASELINE config 5, smooth leg, as bench.py --workload config5 runs it: 1e6-cell anisotropic-hydro surface, coefficients from the
    tables.  (i) the whole surface: finite, additive over two shards; (ii) a stratified oracle sample on a 1e5-cell slice: one species of
    EVERY one of the 75 (mass, sign) classes x 4 pT x 3 phi (one per phi tile) x all 21 rapidities = 18 900 bins against oracle
    coefficients -> oracle kernel; species of a class follow from their representative exactly."""
    tab = inputs.vah_df_tables()
    n = 1000000
    cells = synth.synth_vah_surface(n, 3)
    sp = fx["urqmd"]
    o = dict(dimension=3)
    whole, st = api.smooth_spectra_vah(cells, sp, fx["grid"], o, tab=tab)
    assert np.isfinite(whole).all() and st["n_classes"] == 75 and st["kernel_variant"] == 3
    lo, _ = api.smooth_spectra_vah({k: v[:400000] for k, v in cells.items()}, sp, fx["grid"], o, tab=tab)
    hi, _ = api.smooth_spectra_vah({k: v[400000:] for k, v in cells.items()}, sp, fx["grid"], o, tab=tab)
    assert relerr(lo + hi, whole, floor=1e-250) < 1e-10
    sl = {k: v[600000:700000] for k, v in cells.items()}
    got, _ = api.smooth_spectra_vah(sl, sp, fx["grid"], o, tab=tab)
    seen, reps = set(), []
    for s, (m, sg) in enumerate(zip(sp["mass"], sp["sign"])):
        if (m, sg) not in seen:
            seen.add((m, sg))
            reps.append(s)
    assert len(reps) == 75
    ipT, iphi = [0, 9, 20, 31], [2, 11, 21]
    coef, found = oracle.vah_coefficients(tab, sl["Lambda"], sl["aL"])
    assert found.all()
    sub_grid = dict(fx["grid"], pT=fx["grid"]["pT"][ipT], phi=fx["grid"]["phi"][iphi])
    ref = oracle.dN_pTdpTdphidy_vah(dict(sl, **coef), inputs.species([int(sp["mc_id"][s]) for s in reps]), sub_grid, o).reshape(21, 3, 4, 75)
    g5 = got.reshape(21, 24, 32, 305)
    sub = g5[:, iphi][:, :, ipT][:, :, :, reps]
    assert sub.size == 18900
    assert relerr(sub, ref, floor=1e-250) < TOL, relerr(sub, ref, floor=1e-250)
    cls_of = {}
    for s, (m, sg) in enumerate(zip(sp["mass"], sp["sign"])):
        cls_of.setdefault((m, sg), s)
    for s in (7, 150, 299):
        r = cls_of[(sp["mass"][s], sp["sign"][s])]
        assert relerr(g5[..., s] * sp["degeneracy"][r], g5[..., r] * sp["degeneracy"][s], floor=1e-250) < 1e-15
