"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs,
against the committed golden vectors, and -- at BASELINE.json's full sizes -- through size-independent
properties plus oracle spot checks on sampled bins.

Tolerance: north_star asks <= 1e-6 relative (fp64).  The kernels reach ~1e-10; tests assert 2e-9 so that a
regression of the arithmetic (not of the last bits) fails.  `relerr` uses an absolute floor of 1e-280 for
the exact-zero / denormal tails (pT up to 40 GeV), see SURVEY.md section 4.
"""
import os

import numpy as np
import pytest

from conftest import honoured, ROOT, relerr
from is3d_amd import api, inputs, synth
from oracle import oracle  # the checker

pytestmark = pytest.mark.gpu
TOL = 2e-9


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    assert api.load().is3d_device_count() >= 1
    return torch


def run_plan(torch, cells, sp, grid, df, opts, n=None, out=None, first=0):
    """Device-resident entry: torch tensors provide memory and the stream; returns (numpy spectrum, status)."""
    dev = torch.device("cuda:0")
    n_all = len(cells["tau"])
    n = n_all - first if n is None else n
    tens = {k: torch.from_numpy(np.ascontiguousarray(cells[k])).to(dev) for k in synth.CELL_FIELDS}
    plan = api.Plan(sp, grid, df, opts, max_cells=max(n, 1))
    o = out if out is not None else torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    ptrs = {k: v.data_ptr() + 8 * first for k, v in tens.items()}
    st = plan.execute(n, ptrs, o.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    plan.close()
    return o.cpu().numpy(), st


SP7 = [211, 321, 2212, -2212, 3122, 333, 22]
DEFAULT3 = 6   # the library's default kernel variant for 3+1D without baryon terms


@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("df_mode", [1, 2])
@pytest.mark.parametrize("flags", [dict(), dict(outflow=0, regulate_deltaf=0), dict(outflow=0), dict(regulate_deltaf=0),
                                   dict(include_bulk_deltaf=0), dict(include_shear_deltaf=0)])
@pytest.mark.devlib
def test_parity_matrix(fx, dim, df_mode, flags):
    cells = synth.synth_surface(70 if dim == 3 else 9, dim, seed=100 + dim)
    sp = inputs.species(SP7) if dim == 3 else fx["pikp"]
    o = dict(dimension=dim, df_mode=df_mode, **flags)
    ref = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], o)
    for variant in honoured("df", dim, (1, 2, 3, 4, 5, 6, 7)):
        got, st = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=variant))
        # variants 5, 6 (E2 table stream) exist for the 3+1D kernel, 7 (unit-strided lanes) for 2+1D; elsewhere the request falls
        # back to that mode's default
        want_variant = variant if variant < 5 else ((variant if variant < 7 else DEFAULT3) if dim == 3 else (7 if variant == 7 else 2))
        assert st["kernel_variant"] == want_variant and st["code"] == 0
        assert relerr(got, ref) < TOL, (variant, relerr(got, ref))


@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("df_mode", [1, 2])
@pytest.mark.parametrize("diff", [1, 0])
@pytest.mark.devlib
def test_parity_include_baryon(fx, dim, df_mode, diff):
    """include_baryon = 1 -- b mu_B/T in f_eq, bilinear (T, mu_B) coefficients (intended
    indexing), bulk1 and baryon-diffusion terms (smooth_kernels.cpp:186-197, :254, :297, :306-307, :316-317)."""
    dff = inputs.df_tables_full()
    cells = synth.synth_surface(40 if dim == 3 else 6, dim, seed=200 + dim, baryon=True)
    sp = inputs.species([211, 2212, -2212, 3122, -3122, 321])
    for flags in (dict(), dict(outflow=0, regulate_deltaf=0), dict(include_bulk_deltaf=0), dict(include_shear_deltaf=0)):
        o = dict(dimension=dim, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=diff, **flags)
        ref = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], dff, o)
        # 5, 6: the E2-table kernel with baryon slots (3+1D; 2+1D falls back to the default); the shipped library: the defaults (6 | 7)
        for variant in (honoured("df", dim, (2, 3, 4, 5, 6), baryon=True) or [0]):
            got, st = api.smooth_spectra(cells, sp, fx["grid"], dff, dict(o, kernel_variant=variant))
            assert relerr(got, ref) < TOL, (variant, flags, relerr(got, ref))
            if dim == 3 and variant:
                assert st["kernel_variant"] == variant
        # culling (exact zeros; accumulator-relative with the default flags) changes no bit with baryon slots either
        if dim == 3:
            a0, st0 = api.smooth_spectra(cells, sp, fx["grid"], dff, dict(o, zero_skip=0))
            a2, _ = api.smooth_spectra(cells, sp, fx["grid"], dff, dict(o, zero_skip=2))
            assert st0["kernel_variant"] == DEFAULT3 and np.array_equal(a0, a2)
    # classes now carry the baryon number: p and pbar are different classes, Lambda/Lambdabar too
    _, st = api.smooth_spectra(cells, sp, fx["grid"], dff, dict(dimension=dim, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=diff))
    assert st["n_classes"] == 6
    _, st0 = api.smooth_spectra(cells, sp, fx["grid"], dff, dict(dimension=dim, df_mode=df_mode))
    assert st0["n_classes"] == 4


def test_include_baryon_domain_and_validation(fx):
    dff = inputs.df_tables_full()
    cells = synth.synth_surface(12, 3, seed=9, baryon=True)
    sp = inputs.species([2212, -2212])
    o = dict(dimension=3, df_mode=1, include_baryon=1, include_baryondiff_deltaf=1)
    cells["muB"][5] = 0.81                      # outside the (T, mu_B) table: reference prints and exits (:423-427)
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra(cells, sp, fx["grid"], dff, o)
    assert e.value.code == api.IS3D_EDOMAIN and "cell 5" in str(e.value)
    with pytest.raises(api.Is3dError) as e:     # the mu_B = 0 rows alone are not enough
        api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    assert e.value.code == api.IS3D_EINVAL
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra(cells, sp, fx["grid"], dff, dict(o, kernel_variant=1))
    assert e.value.code == api.IS3D_EINVAL


def test_config1_known_answers_on_device(fx, pins):
    """The reference's own toy surface (config 1), closed form of SURVEY.md section 4."""
    h = 0.197327053
    v = dict(tau=0.5, eta=0.0, dat=1000.0, dax=0.0, day=0.0, dan=0.0, ux=0.0, uy=0.0, un=0.0, E=1.839 * h, T=0.786 * h, P=0.270 * h,
             pixx=0.0, pixy=0.0, pixn=0.0, piyy=0.0, piyn=0.0, bulkPi=0.0)
    cell = {k: np.array([x]) for k, x in v.items()}
    for dfm in (1, 2):
        o3, _ = api.smooth_spectra(cell, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=dfm))
        o2, _ = api.smooth_spectra(cell, fx["pikp"], fx["grid"], fx["df"], dict(dimension=2, df_mode=dfm))
        o3, o2 = o3.reshape(21, 24, 32, 3), o2.reshape(1, 24, 32, 3)
        for key, (y0, y2, bi) in pins["kat_config1"].items():
            s, i = (int(x) for x in key.split(","))
            assert abs(o3[10, 0, i, s] / y0 - 1) < 1e-10 and abs(o3[14, 0, i, s] / y2 - 1) < 1e-10 and abs(o2[0, 0, i, s] / bi - 1) < 1e-10
        assert o3[0, 0, 31, 0] == 0.0   # exp overflow tail is exactly zero, never NaN
        assert np.isfinite(o3).all() and np.isfinite(o2).all()


def test_golden_vectors(fx, pins):
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_64cell.npz"))
    meta = pins["golden_64cell"]
    s3 = synth.synth_surface(64, 3, seed=meta["seed3"])
    s2 = synth.synth_surface(16, 2, seed=meta["seed2"])
    for dfm in (1, 2):
        got, _ = api.smooth_spectra(s3, inputs.species(meta["species3"]), fx["grid"], fx["df"], dict(dimension=3, df_mode=dfm))
        assert relerr(got, z["s3_df%d" % dfm]) < TOL
        got, _ = api.smooth_spectra(s2, fx["pikp"], fx["grid"], fx["df"], dict(dimension=2, df_mode=dfm))
        assert relerr(got, z["s2_df%d" % dfm]) < TOL
    hp = np.load(os.path.join(ROOT, "tests", "golden", "golden_highprec.npz"))
    cellsets = {nm: {k: hp["cells_%s_%s" % (nm, k)] for k in synth.CELL_FIELDS} for nm in ("hand3", "seed3", "seed2")}
    for nm in ("seedb3", "seedb2"):
        cellsets[nm] = {k: hp["cells_%s_%s" % (nm, k)] for k in synth.CELL_FIELDS + synth.BARYON_FIELDS}
    cellsets["hand2"] = {k: v[:2] for k, v in cellsets["hand3"].items()}
    dff = inputs.df_tables_full()
    for case in pins["highprec_cases"]:
        if case.get("feqmod"):
            continue                      # modified-equilibrium cases: test_feqmod_* below
        sp = inputs.species(case["species"]) if "species" in case else fx["pikp"]
        df = dff if case["opts"].get("include_baryon") else fx["df"]
        got, _ = api.smooth_spectra(cellsets[case["cells"]], sp, fx["grid"], df, case["opts"])
        assert relerr(got, hp[case["key"]]) < TOL, case["key"]


def test_species_collapse_is_exact_and_full_urqmd_list(fx):
    """305 urqmd species -> 75 (mass, sign) classes; the collapsed and the uncollapsed run agree to rounding and both
    match the oracle; output order = order of the chosen list, species fastest."""
    cells = synth.synth_surface(6, 3, seed=77)
    sp = fx["urqmd"]
    o = dict(dimension=3, df_mode=2)
    a, sta = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    b, stb = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, collapse_species=2))
    assert sta["n_classes"] == 75 and stb["n_classes"] == 305
    assert relerr(a, b) < 1e-13
    ref = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], o)
    assert relerr(a, ref) < TOL


@pytest.mark.parametrize("n", [0, 1, 2, 63, 65, 257])
def test_ragged_and_empty_surfaces(fx, torch_mod, n):
    cells = synth.synth_surface(n, 3, seed=5)
    g = dict(fx["grid"], pT=fx["grid"]["pT"][::2])
    o = dict(dimension=3, df_mode=1)
    ref = oracle.dN_pTdpTdphidy(cells, fx["pikp"], g, fx["df"], o)
    got, st = api.smooth_spectra(cells, fx["pikp"], g, fx["df"], o)
    assert relerr(got, ref) < TOL
    got2, _ = run_plan(torch_mod, cells, fx["pikp"], g, fx["df"], o)
    assert np.array_equal(got, got2)          # host entry and device-resident entry run the same kernels
    if n == 0:
        assert not got.any()


@pytest.mark.devlib
def test_odd_grids(fx):
    """Grid lengths that are not multiples of the kernel tiles (phi 5, y 5 and 29, pT 3, eta 41 and 7)."""
    rng = np.random.default_rng(3)
    cells = synth.synth_surface(21, 3, seed=8)
    g = dict(pT=np.array([0.1, 0.7, 2.5]), phi=np.sort(rng.random(5) * 2 * np.pi), y=np.linspace(-2, 2, 5), eta=fx["grid"]["eta"], eta_w=fx["grid"]["eta_w"])
    # 41 and 67 rapidities: more than 32 rows in 3+1D -- cf_prep then runs 4-cell batches and must still reserve LDS for the unit-cull
    # bounds behind the row arrays (prep_lds_bytes once sized them with K <= 32 while the kernel laid them out for p.dim3)
    for ygrid in (np.linspace(-2, 2, 5), np.linspace(-3.5, 3.5, 29), np.linspace(-4, 4, 41), np.linspace(-5, 5, 67)):
        gg = dict(g, y=ygrid)
        for dfm in (1, 2):
            ref = oracle.dN_pTdpTdphidy(cells, fx["pikp"], gg, fx["df"], dict(dimension=3, df_mode=dfm))
            for variant in honoured("df", 3, (1, 2, 3, 4, 5, 6)):
                got, _ = api.smooth_spectra(cells, fx["pikp"], gg, fx["df"], dict(dimension=3, df_mode=dfm, kernel_variant=variant))
                assert relerr(got, ref) < TOL
    c2 = synth.synth_surface(5, 2, seed=9)
    for neta in (41, 7):
        eta = np.linspace(-2.0, 2.0, neta)
        w = np.full(neta, eta[1] - eta[0])
        w[[0, -1]] *= 0.5
        gg = dict(g, eta=eta, eta_w=w)
        ref = oracle.dN_pTdpTdphidy(c2, fx["pikp"], gg, fx["df"], dict(dimension=2, df_mode=2))
        for variant in honoured("df", 2, (1, 2, 3, 4, 7)):
            got, _ = api.smooth_spectra(c2, fx["pikp"], gg, fx["df"], dict(dimension=2, df_mode=2, kernel_variant=variant))
            assert relerr(got, ref) < TOL


def test_prep_record_writers_write_the_same_streams():
    """cf_prep's record writers: one element per lane (0), two per lane (1) and the default (3: the duo writer for 3+1D records, the rows
    writer for the long 2+1D ones) -- and plain against non-temporal record stores -- give bitwise the same spectra and cull counts.  The
    switches that select them (IS3D_PREP_PAIR, IS3D_PREP_SKIP) exist only in the developer build of the library (`make DEV=1`,
    is3d_amd/lib_dev; the shipped library reads no such variable), so the comparison runs in a child process on that build:
    tests/dev_writer_check.py."""
    import subprocess
    import sys
    from conftest import ROOT
    dev = os.path.join(ROOT, "is3d_amd", "lib_dev", "libis3d_amd.so")
    if not os.path.exists(dev):
        pytest.skip("developer build absent (make -C is3d_amd/csrc DEV=1)")
    env = dict(os.environ, IS3D_USE_DEV_LIB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev_writer_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "writers agree" in r.stdout


@pytest.mark.devlib
def test_skipped_cells_and_domain_error(fx):
    """u.dsigma <= 0 cells contribute exactly 0 and are counted; their T is never looked up (smooth_kernels.cpp:137
    precedes :200); a live cell outside the coefficient table is IS3D_EDOMAIN (reference: GSL abort)."""
    cells = synth.synth_surface(40, 3, seed=12)
    for k in ("dat", "dax", "day", "dan"):
        cells[k][[3, 17, 39]] *= -1.0
    cells["T"][17] = 0.05
    cells["eta"][39] = np.nan            # garbage in a skipped cell must not leak
    o = dict(dimension=3, df_mode=2)
    ref = oracle.dN_pTdpTdphidy(cells, fx["pikp"], fx["grid"], fx["df"], o)
    for variant in (honoured("df", 3, (1, 2)) or [0]):
        got, st = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], dict(o, kernel_variant=variant))
        assert st["n_cells_skipped"] == 3 and st["bad_cell"] == -1
        assert relerr(got, ref) < TOL and np.isfinite(got).all()
    cells["T"][20] = 0.2004
    cells["T"][30] = 0.09
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o)
    assert e.value.code == api.IS3D_EDOMAIN and "cell 20" in str(e.value)
    # a Lorentz factor beyond the exponent range of the device exp (p.u/T > 1e9 possible): refused, not garbage
    fast = synth.synth_surface(40, 3, seed=12)
    fast["ux"][11] = 3.0e5
    for variant in (honoured("df", 3, (1, 2)) or [0]):
        with pytest.raises(api.Is3dError) as e:
            api.smooth_spectra(fast, fx["pikp"], fx["grid"], fx["df"], dict(o, kernel_variant=variant))
        assert e.value.code == api.IS3D_EDOMAIN and "cell 11" in str(e.value) and "p.u/T" in str(e.value)
    # gamma ~ 200 is still inside: p.u/T up to ~4e8 for the 40 GeV bins, every exponential exactly 0 or finite
    fast["ux"][11] = 200.0
    ref = oracle.dN_pTdpTdphidy(fast, fx["pikp"], fx["grid"], fx["df"], o)
    got, st = api.smooth_spectra(fast, fx["pikp"], fx["grid"], fx["df"], o)
    assert relerr(got, ref) < TOL and np.isfinite(got).all()


def test_passes_chunks_and_accumulate(fx, torch_mod):
    """A workspace too small for the surface forces several passes over the cell axis; the chunk count is a tuning
    knob; accumulate = 1 adds to the caller's array (reference semantics, :375)."""
    cells = synth.synth_surface(300, 3, seed=21)
    sp = inputs.species(SP7)
    o = dict(dimension=3, df_mode=1)
    base, st0 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    assert st0["n_passes"] == 1
    small, st1 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, workspace_bytes=1 << 20))
    assert st1["n_passes"] > 2 and relerr(small, base) < 1e-13
    for ch in (1, 3, 50):
        got, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, cell_chunks=ch))
        assert relerr(got, base) < 1e-13
    again, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    assert np.array_equal(again, base)                       # fixed-order reduction: bitwise reproducible
    acc, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, accumulate=1), out=base.copy())
    assert relerr(acc, 2.0 * base) < 1e-15
    # 2+1D with unit-strided lanes (96 bins, four lane slots per bin): passes and chunk counts keep whole groups of four units
    c2 = synth.synth_surface(37, 2, seed=22)
    o2 = dict(dimension=2, df_mode=2)
    b2, s2 = api.smooth_spectra(c2, fx["pikp"], fx["grid"], fx["df"], o2)
    assert s2["kernel_variant"] == 7 and s2["n_passes"] == 1
    small2, s2p = api.smooth_spectra(c2, fx["pikp"], fx["grid"], fx["df"], dict(o2, workspace_bytes=1 << 20))
    assert s2p["n_passes"] > 2 and relerr(small2, b2) < 1e-13
    for ch in (1, 2, 7):
        got, _ = api.smooth_spectra(c2, fx["pikp"], fx["grid"], fx["df"], dict(o2, cell_chunks=ch))
        assert relerr(got, b2) < 1e-13


def test_tapered_partition_passes_and_explicit_chunks(fx):
    """The default cell partition has a tapered tail (is3d::chunk_cells: the last chunks a quarter of the size of the others) once its chunks
    hold 256 cells or more: 80 000 cells x 305 species here.  Against the uniform partition of an explicit cell_chunks, and split into passes
    that reuse the first pass's chunk count for fewer cells: the same spectrum to rounding (another summation order)."""
    cells = synth.synth_surface(80000, 3, seed=23)
    sp = inputs.species("urqmd")
    o = dict(dimension=3, df_mode=2)
    base, st0 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    assert st0["n_passes"] == 1 and st0["code"] == 0
    uni, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, cell_chunks=144))
    assert relerr(uni, base) < 1e-12 and not np.array_equal(uni, base)      # (not the same partition: the taper is in use)
    multi, st1 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, workspace_bytes=600 << 20))
    assert st1["n_passes"] >= 2 and relerr(multi, base) < 1e-12
    again, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    assert np.array_equal(again, base)


def test_workgroup_sizes_and_batch_tails_change_no_bit(fx):
    """waves_per_group (1, 2, 4, 8 lane-waves sharing one LDS-staged stream; 1 = no barrier partner, cf_main_tile3e only) and the
    cell count modulo the units per LDS batch are scheduling only: a lane sees the same units in the same order, so the spectrum
    is bitwise the same.  Cell counts around the batch sizes exercise the unpredicated staging pieces that over-read a short
    last batch (into the slack behind the streams / unused units of the LDS buffer)."""
    sp = inputs.species("urqmd")
    o = dict(dimension=3, df_mode=2, cell_chunks=2)
    for n in (1, 5, 6, 7, 13, 97):
        cells = synth.synth_surface(n, 3, seed=31)
        base, st = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
        assert st["kernel_variant"] == DEFAULT3 and np.isfinite(base).all()
        for w in (1, 2, 4, 8):
            got, st = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, waves_per_group=w))
            assert st["kernel_variant"] == DEFAULT3 and np.array_equal(got, base), (n, w)
    cells = synth.synth_surface(97, 3, seed=31)
    ref = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], o)
    got, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, waves_per_group=1))
    assert relerr(got, ref) < TOL


def test_surface_relative_cull_is_bounded_not_bitwise(fx):
    """zero_skip 3 (cf_main_tile3e with outflow && regulate_deltaf): an eighth of the chunks runs first, its partial spectrum floors the
    row-cull thresholds of the rest.  Not bitwise: a skipped term is below 2^-57 of the FINAL accumulator per cell, so the spectrum may
    lose at most N_cells 2^-57 of a bin, one-sided; it culls more rows than the accumulator-relative rule.  Where the rule does not apply
    (too few chunks, no outflow clamp) zero_skip 3 is zero_skip 0."""
    n = 24000
    cells = synth.synth_surface(n, 3, seed=91)
    sp = inputs.species("urqmd")
    o = dict(dimension=3, df_mode=2, cell_chunks=64)
    full, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=2))
    rel, s0 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=0))
    surf, s3 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=3))
    assert np.array_equal(rel, full)
    assert s3["n_wave_rows"] == s0["n_wave_rows"] and s3["n_wave_rows_culled"] > s0["n_wave_rows_culled"]
    assert (surf <= full).all()
    err = np.abs(surf - full) / np.maximum(np.abs(full), 1e-300)
    assert err.max() <= n * 2.0 ** -57 and err.max() < 1e-12
    few, sf = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=3, cell_chunks=4))
    few0, sf0 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=0, cell_chunks=4))
    assert np.array_equal(few, few0) and sf["n_wave_rows_culled"] == sf0["n_wave_rows_culled"]
    # several workspace passes: the floors of a later pass come from partial sums that already hold the earlier passes (still a lower bound)
    mp, smp = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=3, workspace_bytes=1 << 27))
    assert smp["n_passes"] > 1      # (another summation order than `full`: compared to rounding, not one-sidedly)
    assert (np.abs(mp - full) / np.maximum(np.abs(full), 1e-300)).max() < 1e-12
    a, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=3, outflow=0))
    b, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=2, outflow=0))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dim,df_mode,species", [(3, 2, "urqmd"), (3, 1, "pikp"), (2, 1, "pikp"), (2, 2, "urqmd")])
@pytest.mark.devlib
def test_row_culling_changes_no_bit(fx, dim, df_mode, species):
    """zero_skip: 2 evaluates every row; 1 skips wave-rows whose exp(-p.u/T) is exactly +0; 0 (default) also skips rows whose
    every term is below half an ulp of every accumulator it would be added to (outflow && regulate_deltaf).  All three give the
    same bits; the default culls most (the y / eta range of the surface is several units wide here, as in config 3)."""
    n = (2000 if species == "urqmd" else 3000) if dim == 3 else 150
    cells = synth.synth_surface(n, dim, seed=77 + dim)
    sp = inputs.species(species)
    o = dict(dimension=dim, df_mode=df_mode, cell_chunks=3)
    full, st2 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=2))
    exact, st1 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=1))
    rel, st0 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, zero_skip=0))
    assert np.array_equal(exact, full) and np.array_equal(rel, full)
    assert st2["n_wave_rows_culled"] == 0 and st0["n_wave_rows_culled"] >= st1["n_wave_rows_culled"]
    if dim == 3 and species == "urqmd":
        assert st0["n_wave_rows_culled"] > 1.5 * st1["n_wave_rows_culled"] > 0
    # without the outflow clamp the accumulators are not monotone: only the exact-zero rule applies, and it is still bit-exact
    o2 = dict(o, outflow=0)
    a, sa = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o2, zero_skip=0))
    b, sb = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o2, zero_skip=2))
    assert np.array_equal(a, b) and sa["n_wave_rows_culled"] == st1["n_wave_rows_culled"]
    if species == "pikp":
        assert relerr(rel, oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], dict(dimension=dim, df_mode=df_mode))) < TOL
    if dim == 2:
        # the default in 2+1D is variant 7 (8 x 31; 96 momentum bins: unit-strided lanes, four lane slots per bin; 2 400 bins: one); the
        # plain 8 x 61 tile (developer build) agrees to rounding
        assert st0["kernel_variant"] == 7
        if api.DEV_LIB:
            v2, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=2))
            assert relerr(rel, v2) < 5e-11
    if dim == 3:
        # variant 6 (the default here) reads the phi-side exponentials from the table stream cf_prep writes and tests the rows' liveness before
        # their exponentials; it agrees with variant 3 (the 8 x 7 tile without the table) to the rounding of the exponent's argument (bmax - mT C'_k
        # cancels two numbers of order 1e4, one ulp of which is 2e-12) and culls the same rows to a per cent (the thresholds are refreshed per LDS
        # batch, and the batches differ)
        assert st0["kernel_variant"] == 6
        v3, s3 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=3))
        assert s3["kernel_variant"] == 3 and relerr(rel, v3) < 5e-11
        assert abs(st0["n_wave_rows_culled"] - s3["n_wave_rows_culled"]) <= 3e-2 * s3["n_wave_rows_culled"]
        if api.DEV_LIB:
            # variant 5 (developer build): the table stream with hand-pipelined rows -- culling changes no bit of it either
            v5, s5 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=5))
            v5e, s5e = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=5, zero_skip=1))
            v5f, s5f = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=5, zero_skip=2))
            assert s5["kernel_variant"] == 5 and np.array_equal(v5, v5f) and np.array_equal(v5e, v5f)
            assert relerr(v5, v3) < 5e-11
            assert relerr(rel, v5) < 5e-11 and abs(st0["n_wave_rows_culled"] - s5["n_wave_rows_culled"]) <= 2e-2 * s5["n_wave_rows_culled"]
            assert abs(s5["n_wave_rows_culled"] - s3["n_wave_rows_culled"]) <= 2e-2 * s3["n_wave_rows_culled"] and s5f["n_wave_rows_culled"] == 0
            # variant 9 (developer build; measured and dropped in round 5): the unit records on the scalar path -- same rows culled, spectrum to rounding
            v9, s9 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=9))
            v9f, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=9, zero_skip=2))
            assert s9["kernel_variant"] == 9 and np.array_equal(v9, v9f) and relerr(v9, rel) < 5e-11
            # variant 10 (developer build; measured and dropped in round 5): the E2 column from global memory, records-only LDS batches -- the default's
            # arithmetic on the default's operands: bitwise the default's spectrum, with one- and two-wave workgroups, culling on or off
            for extra in (dict(), dict(waves_per_group=1), dict(zero_skip=2)):
                v10, s10 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=10, **extra))
                assert s10["kernel_variant"] == 10 and np.array_equal(v10, rel), extra
            # variant 12 (developer build; measured and dropped in round 5): the E2 tables built per workgroup in LDS with cf_prep's own expression --
            # bitwise the default's spectrum at two and four waves per workgroup, culling on or off; variant 11 (raw header values as FMA operands): to rounding
            for extra in (dict(), dict(waves_per_group=4), dict(zero_skip=2)):
                v12, s12 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=12, **extra))
                assert s12["kernel_variant"] == 12 and np.array_equal(v12, rel), extra
            v11, s11 = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=11))
            v11f, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(o, kernel_variant=11, zero_skip=2))
            assert s11["kernel_variant"] == 11 and np.array_equal(v11, v11f) and relerr(v11, rel) < 5e-11


@pytest.mark.parametrize("dim", [3, 2])
def test_device_observables(fx, torch_mod, dim):
    """dN/dy, dN/(2 pi pT dpT dy) and v_n on the device == the reductions of the reference's
    writers (emissionfunction.cpp:639-677, :729-772, :1053-1136) applied to the oracle's spectrum."""
    torch = torch_mod
    g = fx["grid_w"]
    cells = synth.synth_surface(50 if dim == 3 else 8, dim, seed=300 + dim)
    sp = inputs.species(SP7)
    o = dict(dimension=dim, df_mode=2)
    ref = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], o)
    ny = 21 if dim == 3 else 1
    r4 = ref.reshape(ny, 24, 32, 7)
    dev = torch.device("cuda:0")
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    plan = api.Plan(sp, fx["grid"], fx["df"], o, max_cells=len(cells["tau"]))
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.execute(len(cells["tau"]), {k: v.data_ptr() for k, v in tens.items()}, out.data_ptr(), st)
    dndy = torch.zeros(7 * ny, dtype=torch.float64, device=dev)
    s2pi = torch.zeros(7 * ny * 32, dtype=torch.float64, device=dev)
    vn = torch.zeros(7 * ny * 32 * 7, dtype=torch.float64, device=dev)
    plan.observables(out.data_ptr(), g["pT_w"], g["phi_w"], dndy.data_ptr(), s2pi.data_ptr(), vn.data_ptr(), st)
    torch.cuda.synchronize()
    want_dndy = np.einsum("j,i,kjis->sk", g["phi_w"], g["pT_w"], r4)
    assert relerr(dndy.cpu().numpy().reshape(7, ny), want_dndy) < TOL
    want_s2pi = np.einsum("j,kjis->ski", g["phi_w"], r4) / (2.0 * np.pi)
    assert relerr(s2pi.cpu().numpy().reshape(7, ny, 32), want_s2pi) < TOL
    den = np.einsum("j,kjis->ski", g["phi_w"], r4)
    got_vn = vn.cpu().numpy().reshape(7, ny, 32, 7)
    for k in range(7):
        num = np.abs(np.einsum("j,kjis->ski", np.exp(1j * (k + 1) * g["phi"]) * g["phi_w"], r4))
        want = np.where(den < 1e-15, 0.0, num / np.where(den == 0, 1.0, den))
        ok = den > 1e-200            # where the spectrum is ~1e-300 the ratio is numerical noise in both
        assert np.max(np.abs(got_vn[:, :, :, k] - want)[ok]) < 1e-7
        assert (got_vn[:, :, :, k][den < 1e-15] == 0.0).all()
    # only some outputs requested
    plan.observables(out.data_ptr(), g["pT_w"], g["phi_w"], dndy_ptr=dndy.data_ptr(), stream=st)
    torch.cuda.synchronize()
    plan.close()


@pytest.mark.parametrize("dim,df_mode", [(3, 2), (3, 1), (2, 2)])
def test_reference_bilinear_indexing_option(fx, dim, df_mode):
    """opts.reference_bilinear_indexing = 1: the (T, mu_B) tables read as the reference's calculate_bilinear reads them
    (f_data[iT][imuB], deltafReader.cpp:404-407), against the oracle with the same option; results differ from the intended
    indexing; a live cell with T >= T[n_muB - 1] (0.18 GeV: the reference reads past its row pointers) is IS3D_EDOMAIN."""
    dff = inputs.df_tables_full()
    cells = synth.synth_surface(60 if dim == 3 else 6, dim, seed=500 + dim, baryon=True)
    sp = inputs.species([211, 2212, -2212, 3122])
    o = dict(dimension=dim, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=1, reference_bilinear_indexing=1)
    ref = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], dff, o)
    got, st = api.smooth_spectra(cells, sp, fx["grid"], dff, o)
    assert relerr(got, ref) < TOL
    intended, _ = api.smooth_spectra(cells, sp, fx["grid"], dff, dict(o, reference_bilinear_indexing=0))
    assert relerr(intended, ref) > 1e-4
    cells["T"][5] = 0.1815                      # inside the table, outside what the reference's read can reach
    api.smooth_spectra(cells, sp, fx["grid"], dff, dict(o, reference_bilinear_indexing=0))
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra(cells, sp, fx["grid"], dff, o)
    assert e.value.code == api.IS3D_EDOMAIN and "cell 5" in str(e.value)


def test_status_less_executes_keep_their_domain_errors(fx, torch_mod):
    """is3d_plan_execute with status == NULL is fully asynchronous and neutralises a cell outside the coefficient table; the
    plan keeps the lowest such cell until is3d_plan_check (the reference aborts there: deltafReader.cpp:339)."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    cells = synth.synth_surface(300, 3, seed=8)
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    plan = api.Plan(fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=2), max_cells=300)
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    plan.execute(300, ptrs, out.data_ptr(), stream, want_status=False)
    plan.check(stream)                                   # clean surface: nothing to report
    tens["T"][[41, 207]] = 0.25
    plan.execute(300, ptrs, out.data_ptr(), stream, want_status=False)
    tens["T"][[41, 207]] = 0.15
    plan.execute(300, ptrs, out.data_ptr(), stream, want_status=False)   # a later clean execute does not erase the record
    with pytest.raises(api.Is3dError) as e:
        plan.check(stream)
    assert e.value.code == api.IS3D_EDOMAIN and "cell 41" in str(e.value)
    plan.check(stream)                                   # cleared
    plan.close()


@pytest.mark.parametrize("df_mode", [1, 2, 3, 4])
def test_device_spectra_integrate_to_the_ideal_gas_density(fx, df_mode):
    """A check that does NOT go through the oracle: static cells (u = (1, 0, 0, 0)), pi^{mu nu} = Pi = 0, so every df_mode reduces
    to f_eq and the spectrum, integrated with the reference's own pT / phi / y quadratures (the weights of
    write_dN_dy_toFile, emissionfunction.cpp:729-772; trapezoid in y), must give volume x n_eq with the closed form
    n_eq = g m^2 T / (2 pi^2 hbarc^3) sum_k (-sign)^(k+1) K_2(k m / T) / k.  Agreement is limited by those quadratures (3e-6)."""
    from scipy import special
    g = fx["grid_w"]
    sp = inputs.species([211, 321, 2212, 3122])
    T, n = 0.15, 3
    z = np.zeros(n)
    cells = dict(tau=np.ones(n), eta=np.array([0.0, 0.3, -0.2]), dat=np.array([1.0, 2.0, 0.5]), dax=z.copy(), day=z.copy(), dan=z.copy(), ux=z.copy(),
                 uy=z.copy(), un=z.copy(), T=T * np.ones(n), P=0.08 * np.ones(n), E=0.3 * np.ones(n), pixx=z.copy(), pixy=z.copy(), pixn=z.copy(),
                 piyy=z.copy(), piyn=z.copy(), bulkPi=z.copy())
    fq = inputs.feqmod_tables(T) if df_mode >= 3 else None
    got, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], dict(dimension=3, df_mode=df_mode), fq=fq)
    N = np.einsum("k,j,i,kjis->s", np.full(21, 0.5), g["phi_w"], g["pT_w"], got.reshape(21, 24, 32, 4))   # pT_w carries the pT Jacobian
    k = np.arange(1, 60)
    want = np.array([gd * m * m * T / (2 * np.pi ** 2 * 0.197327053 ** 3) * np.sum((-s) ** (k + 1) * special.kn(2, k * m / T) / k)
                     for m, gd, s in zip(sp["mass"], sp["degeneracy"], sp["sign"])]) * cells["dat"].sum()
    assert np.max(np.abs(N / want - 1)) < 1e-5, N / want - 1


def _subset_oracle(fx, cells, sp_ids, ipT, iphi, opts):
    g = fx["grid"]
    sub = dict(g, pT=g["pT"][ipT], phi=g["phi"][iphi])
    return oracle.dN_pTdpTdphidy(cells, inputs.species(sp_ids), sub, fx["df"], opts)


def test_full_size_config3_properties_and_spot_checks(fx, torch_mod):
    """BASELINE config 3: 1e6-cell 3+1D surface, Chapman-Enskog, 305 species.  (i) additivity over cell shards --
    the property the multi-GPU path relies on; (ii) linearity in dsigma; (iii) bitwise reproducibility;
    (iv) oracle parity on sampled (species, pT, phi) columns over ALL cells and all 21 rapidities."""
    torch = torch_mod
    n = 1000000
    cells = synth.synth_surface(n, 3)
    sp = fx["urqmd"]
    o = dict(dimension=3, df_mode=2)
    whole, st = run_plan(torch, cells, sp, fx["grid"], fx["df"], o)
    assert st["n_classes"] == 75 and np.isfinite(whole).all() and (whole >= 0).all()
    # (i) two shards, second accumulated onto the first
    dev = torch.device("cuda:0")
    lo, _ = run_plan(torch, cells, sp, fx["grid"], fx["df"], o, n=n // 2)
    acc = torch.from_numpy(lo.copy()).to(dev)
    both, _ = run_plan(torch, cells, sp, fx["grid"], fx["df"], dict(o, accumulate=1), n=n - n // 2, first=n // 2, out=acc)
    assert relerr(both, whole) < 1e-12
    # (ii) dsigma -> 2 dsigma on a 1e5-cell slice
    sl = {k: v[:100000] for k, v in cells.items()}
    a, _ = run_plan(torch, sl, sp, fx["grid"], fx["df"], o)
    sl2 = {k: (2.0 * v if k in ("dat", "dax", "day", "dan") else v) for k, v in sl.items()}
    b, _ = run_plan(torch, sl2, sp, fx["grid"], fx["df"], o)
    assert relerr(b, 2.0 * a) < 1e-15
    # (iii)
    again, _ = run_plan(torch, cells, sp, fx["grid"], fx["df"], o)
    assert np.array_equal(again, whole)
    # (iv) sampled columns against the oracle over all 1e6 cells
    w5 = whole.reshape(21, 24, 32, 305)
    ids = [int(sp["mc_id"][s]) for s in (0, 120, 304)]
    ipT, iphi = [2, 17], [1, 13]
    ref = _subset_oracle(fx, cells, ids, ipT, iphi, o).reshape(21, 2, 2, 3)
    got = w5[:, iphi][:, :, ipT][:, :, :, [0, 120, 304]]
    assert relerr(got, ref) < TOL


def test_full_size_config3_stratified_oracle_sample(fx, torch_mod):
    """BASELINE config 3, a stratified sample of > 1 % of the 4 919 040 bins against the oracle: one species of EVERY one of the
    75 (mass, sign) classes x 6 pT (low / mid / high) x 6 phi (two per phi tile of the kernel) x all 21 rapidities = 56 700 bins,
    over a 1e5-cell slice of the 1e6-cell surface; the shard-additivity test above ties slices to the whole surface bitwise-
    reproducibly.  The slice runs through the default kernel (variant 6) and through variant 3."""
    torch = torch_mod
    n = 100000
    cells = synth.synth_surface(1000000, 3)
    sl = {k: v[300000:300000 + n] for k, v in cells.items()}
    sp = fx["urqmd"]
    o = dict(dimension=3, df_mode=2)
    got, st = run_plan(torch, sl, sp, fx["grid"], fx["df"], o)
    assert st["n_classes"] == 75 and st["kernel_variant"] == DEFAULT3
    seen, reps = set(), []
    for s, (m, sg) in enumerate(zip(sp["mass"], sp["sign"])):
        if (m, sg) not in seen:
            seen.add((m, sg))
            reps.append(s)
    assert len(reps) == 75
    ipT, iphi = [0, 5, 12, 18, 25, 31], [1, 6, 9, 14, 17, 22]
    ref = _subset_oracle(fx, sl, [int(sp["mc_id"][s]) for s in reps], ipT, iphi, o).reshape(21, 6, 6, 75)
    g5 = got.reshape(21, 24, 32, 305)
    sub = g5[:, iphi][:, :, ipT][:, :, :, reps]
    assert sub.size == 56700 and sub.size > 0.01 * got.size
    assert relerr(sub, ref) < TOL
    # species of one class differ by the degeneracy only: every other species follows from its representative exactly
    cls_of = {}
    for s, (m, sg) in enumerate(zip(sp["mass"], sp["sign"])):
        cls_of.setdefault((m, sg), s)
    for s in (7, 150, 299):
        r = cls_of[(sp["mass"][s], sp["sign"][s])]
        assert relerr(g5[..., s] * sp["degeneracy"][r], g5[..., r] * sp["degeneracy"][s]) < 1e-15
    old, _ = run_plan(torch, sl, sp, fx["grid"], fx["df"], dict(o, kernel_variant=3))
    assert relerr(got, old) < 5e-11


def test_full_size_config2_properties_and_spot_checks(fx, torch_mod):
    """BASELINE config 2: 1e5-cell 2+1D boost-invariant surface, 14-moment, pi/K/p, 241-point eta quadrature."""
    torch = torch_mod
    n = 100000
    cells = synth.synth_surface(n, 2)
    o = dict(dimension=2, df_mode=1)
    whole, _ = run_plan(torch, cells, fx["pikp"], fx["grid"], fx["df"], o)
    assert np.isfinite(whole).all() and (whole >= 0).all()
    parts = np.zeros_like(whole)
    for r in range(4):
        lo, hi = r * n // 4, (r + 1) * n // 4
        p, _ = run_plan(torch, cells, fx["pikp"], fx["grid"], fx["df"], o, n=hi - lo, first=lo)
        parts += p
    assert relerr(parts, whole) < 1e-12
    w4 = whole.reshape(1, 24, 32, 3)
    ipT, iphi = [0, 9, 20], [0, 7, 23]
    ref = _subset_oracle(fx, cells, [211, 321, 2212], ipT, iphi, o).reshape(1, 3, 3, 3)
    assert relerr(w4[:, iphi][:, :, ipT], ref) < TOL


def test_shader_clock_probe():
    """is3d_probe_shader_clock (diagnostic used by bench.py): idle waves read s_memtime against s_memrealtime; an MI355X runs
    between 1 and 2.5 GHz, or the call reports 0 when the two counters are the same clock."""
    ghz = api.probe_shader_clock(0.05)
    assert ghz == 0.0 or 1.0 < ghz < 2.6, ghz
    with pytest.raises(api.Is3dError):
        api.probe_shader_clock(0.0)


@pytest.mark.parametrize("df_mode", [2, 1])
def test_config3_size_include_baryon_stratified_oracle_sample(fx, df_mode):
    """SURVEY.md 8f rank 1 at BASELINE config-3 size, as `bench.py --include-baryon` runs it: include_baryon = 1 with baryon diffusion on a 60 000-cell
    slice of the 1e6-cell surface x 305 species (124 classes: the baryon number joins the class key) through cf_main_tile3e<.., BARYON> against the
    oracle on a stratified sample -- one species of every fourth (mass, sign, baryon) class x 4 pT x 4 phi x all 21 rapidities -- with the bilinear
    (T, mu_B) coefficients and the b mu_B / T exponent (smooth_kernels.cpp:186-197, :254, :297, :303-321); culling on / off: the same bits."""
    n = 60000
    cells = synth.synth_surface(1000000, 3, baryon=True)
    sl = {k: v[500000:500000 + n] for k, v in cells.items()}
    sp = fx["urqmd"]
    dff = inputs.df_tables_full()
    o = dict(dimension=3, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=1)
    got, st = api.smooth_spectra(sl, sp, fx["grid"], dff, o)
    assert st["n_classes"] == 124 and st["kernel_variant"] == DEFAULT3 and st["n_wave_rows_culled"] > 0.3 * st["n_wave_rows"]
    seen, reps = set(), []
    for s, key in enumerate(zip(sp["mass"], sp["sign"], sp["baryon"])):
        if key not in seen:
            seen.add(key)
            reps.append(s)
    assert len(reps) == 124
    reps = reps[::4]
    assert any(sp["baryon"][s] > 0 for s in reps) and any(sp["baryon"][s] < 0 for s in reps)
    ipT, iphi = [0, 9, 20, 31], [2, 7, 13, 22]
    g = fx["grid"]
    sub_grid = dict(g, pT=g["pT"][ipT], phi=g["phi"][iphi])
    ref = oracle.dN_pTdpTdphidy(sl, inputs.species([int(sp["mc_id"][s]) for s in reps]), sub_grid, dff, o)
    g5 = got.reshape(21, 24, 32, 305)
    sub = g5[:, iphi][:, :, ipT][:, :, :, reps]
    assert sub.size == 21 * 16 * len(reps) and relerr(sub, ref.reshape(21, 4, 4, len(reps))) < TOL
    off, st_off = api.smooth_spectra(sl, sp, fx["grid"], dff, dict(o, zero_skip=2))
    assert np.array_equal(off, got) and st_off["n_wave_rows_culled"] == 0
    # a proton and an antiproton bin differ (mu_B > 0 on this surface), a pi+ and a pi- bin do not
    ids = list(sp["mc_id"])
    p, pbar, pip, pim = (ids.index(i) for i in (2212, -2212, 211, -211))
    assert g5[..., p].sum() > 5.0 * g5[..., pbar].sum() and np.array_equal(g5[..., pip], g5[..., pim])   # e^(2 mu_B / T) with mu_B 0.05 .. 0.4 GeV


def test_config3_size_14_moment_stratified_oracle_sample(fx):
    """BASELINE config 1's physics (14-moment delta-f) at config-3 size, as `bench.py --df-mode 1` runs it: cf_main_tile3e<CE = false> on a 60 000-cell
    slice x 305 species against the oracle on the stratified sample of the Chapman-Enskog test above (every fifth class)."""
    n = 60000
    cells = synth.synth_surface(1000000, 3)
    sl = {k: v[700000:700000 + n] for k, v in cells.items()}
    sp = fx["urqmd"]
    o = dict(dimension=3, df_mode=1)
    got, st = api.smooth_spectra(sl, sp, fx["grid"], fx["df"], o)
    assert st["n_classes"] == 75 and st["kernel_variant"] == DEFAULT3
    seen, reps = set(), []
    for s, key in enumerate(zip(sp["mass"], sp["sign"])):
        if key not in seen:
            seen.add(key)
            reps.append(s)
    reps = reps[::5]
    ipT, iphi = [0, 9, 20, 31], [2, 7, 13, 22]
    ref = _subset_oracle(fx, sl, [int(sp["mc_id"][s]) for s in reps], ipT, iphi, o).reshape(21, 4, 4, len(reps))
    g5 = got.reshape(21, 24, 32, 305)
    assert relerr(g5[:, iphi][:, :, ipT][:, :, :, reps], ref) < TOL
    off, _ = api.smooth_spectra(sl, sp, fx["grid"], fx["df"], dict(o, zero_skip=2))
    assert np.array_equal(off, got)
