"""GPU (-m gpu): randomised differential test of the HIP path against the oracle -- seeded, so it is the same set of cases every run.
Grid lengths (1 .. 40 pT, 1 .. 30 phi, 1 .. 44 y, 2 .. 70 eta nodes), species subsets, cell counts, flags, kernel variants and
workspace / chunk settings are drawn at random: the point is the combinations nobody wrote a test for (grids that are not multiples of
the tiles, more than 32 rows or pT values, one-element grids, passes over the cell axis with odd chunk counts)."""
import numpy as np
import pytest

from conftest import honoured, relerr
from is3d_amd import api, inputs, synth
from oracle import oracle  # the checker

pytestmark = pytest.mark.gpu
TOL = 2e-9


def _grid(rng, g, dim):
    npT, nphi = int(rng.integers(1, 41)), int(rng.integers(1, 31))
    pT = np.sort(rng.random(npT) * rng.choice([1.0, 3.0, 12.0])) + 0.01
    phi = np.sort(rng.random(nphi) * 2 * np.pi)
    ny = int(rng.integers(1, 45))
    y = np.sort(rng.random(ny) * 8.0 - 4.0)
    neta = int(rng.integers(2, 71))
    eta = np.linspace(-rng.uniform(1.0, 5.0), rng.uniform(1.0, 5.0), neta)
    w = np.full(neta, eta[1] - eta[0])
    w[[0, -1]] *= 0.5
    return dict(pT=pT, phi=phi, y=y, eta=eta, eta_w=w)


def _species(rng, fx):
    ids = fx["urqmd"]["mc_id"]
    k = int(rng.integers(1, 9))
    return inputs.species([int(i) for i in rng.choice(ids, size=k, replace=False)])


@pytest.mark.parametrize("case", range(60))
@pytest.mark.devlib
def test_random_configuration_matches_the_oracle(fx, case):
    rng = np.random.default_rng(7000 + case)
    dim = int(rng.choice([3, 3, 2]))
    g = _grid(rng, fx["grid"], dim)
    if dim == 2:
        g = dict(g, pT=g["pT"][:12], phi=g["phi"][:10])          # the eta sum multiplies the oracle's work
    sp = _species(rng, fx)
    n = int(rng.integers(1, 70 if dim == 3 else 8))
    kind = case % 4                                               # 0, 1: delta-f kernels; 2: modified equilibrium; 3: anisotropic hydro
    o = dict(dimension=dim)
    for flag in ("include_bulk_deltaf", "include_shear_deltaf", "regulate_deltaf"):
        o[flag] = int(rng.random() < 0.75)
    extra = {}
    if rng.random() < 0.4:
        extra["workspace_bytes"] = int(rng.integers(1 << 16, 1 << 21))
    if rng.random() < 0.4:
        extra["cell_chunks"] = int(rng.integers(1, 6))
    extra["zero_skip"] = int(rng.choice([0, 1, 2]))
    if kind in (0, 1):
        baryon = rng.random() < 0.3
        cells = synth.synth_surface(n, dim, seed=9000 + case, baryon=baryon)
        o.update(df_mode=int(rng.choice([1, 2])), outflow=int(rng.random() < 0.75))
        df = fx["df"]
        if baryon:
            df = inputs.df_tables_full()
            o.update(include_baryon=1, include_baryondiff_deltaf=int(rng.random() < 0.5))
        fx = dict(fx, df=df)
        ref = oracle.dN_pTdpTdphidy(cells, sp, g, fx["df"], o)
        scale = np.abs(ref)
        if not (o["outflow"] and o["regulate_deltaf"]):
            # terms of both signs can cancel in a bin: judge such a bin against the size of what was added up, S = sum of |p.dsigma| f (1 + clamp df)
            # (two oracle runs with the outflow cut, dsigma as it is and negated), with a cancellation factor of 1e4 allowed
            pos = oracle.dN_pTdpTdphidy(cells, sp, g, fx["df"], dict(o, outflow=1, regulate_deltaf=1))
            neg = oracle.dN_pTdpTdphidy({k: (-v if k in ("dat", "dax", "day", "dan") else v) for k, v in cells.items()}, sp, g, fx["df"],
                                        dict(o, outflow=1, regulate_deltaf=1))
            scale = np.maximum(scale, 1e-4 * (pos + neg))
        variants = [0] + list(rng.choice([2, 3, 4, 5, 6, 7] if baryon else [1, 2, 3, 4, 5, 6, 7], size=2, replace=False))   # variant 1 has no baryon slots
        variants = [0] + honoured("df", dim, [int(v) for v in variants[1:]], baryon=bool(baryon))   # the shipped library: the ones it holds; all of them in the developer build
        # and the cell-axis split: a random number of shards on the one device sums to the same spectrum
        shards = int(rng.integers(2, 6))
        multi, _, _ = api.smooth_spectra_multi(cells, sp, g, fx["df"], o, devices=[0] * shards)
        assert float(np.max(np.abs(multi - ref) / np.maximum(scale, 1e-280))) < TOL, (case, "shards", shards)
        for v in variants:
            got, st = api.smooth_spectra(cells, sp, g, fx["df"], dict(o, kernel_variant=int(v), **extra))
            err = float(np.max(np.abs(got - ref) / np.maximum(scale, 1e-280)))
            assert err < TOL, (case, v, st["kernel_variant"], err, relerr(got, ref), {k: len(x) for k, x in g.items()}, n)
    elif kind == 2:
        cells = synth.synth_surface(n, dim, seed=9000 + case)
        o.update(df_mode=int(rng.choice([3, 4])), outflow=int(rng.random() < 0.75))
        fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
        ref, _ = oracle.dN_pTdpTdphidy_feqmod(cells, sp, g, fx["df"], fq, o)
        for v in [0] + honoured("fq", dim, [int(rng.choice([2, 3, 4]))]):
            got, st = api.smooth_spectra(cells, sp, g, fx["df"], dict(o, kernel_variant=v, **extra), fq=fq)
            assert relerr(got, ref) < TOL, (case, v, relerr(got, ref), {k: len(x) for k, x in g.items()}, n)
    else:
        tab = inputs.vah_df_tables()
        cells = synth.synth_vah_surface(n, dim, seed=9000 + case)
        coef, found = oracle.vah_coefficients(tab, cells["Lambda"], cells["aL"])
        assert found.all()
        ref = oracle.dN_pTdpTdphidy_vah(dict(cells, **coef), sp, g, o)
        for v in honoured("vah", dim, (0, 2)):
            got, st = api.smooth_spectra_vah(cells, sp, g, dict(o, kernel_variant=v, **extra), tab=tab)
            assert relerr(got, ref, floor=1e-270) < TOL, (case, v, relerr(got, ref, floor=1e-270), {k: len(x) for k, x in g.items()}, n)


@pytest.mark.parametrize("case", range(16))
def test_random_sampler_configuration_gives_the_oracles_list(fx, case):
    """The particle sampler on random small configurations (dimension, df_mode 1-4, fast mode, species subsets, event counts, seeds, y_cut,
    shard offsets, event batching): the device list is the oracle's list -- same hadrons, same order, momenta to rounding."""
    rng = np.random.default_rng(8000 + case)
    dim = int(rng.choice([3, 2]))
    dfm = int(rng.choice([1, 2, 3, 4]))
    n = int(rng.integers(1, 500))
    cells = synth.synth_surface(n, dim, seed=9500 + case)
    sp = _species(rng, fx)
    T_avg = inputs.surface_average_T(cells)
    fq = inputs.feqmod_tables(T_avg)
    fast = int(rng.random() < 0.4)
    o = dict(dimension=dim, df_mode=dfm, include_bulk_deltaf=int(rng.random() < 0.8), include_shear_deltaf=int(rng.random() < 0.8))
    kw = dict(n_events=int(rng.integers(1, 40)), seed=int(rng.integers(1, 1 << 40)), y_cut=float(rng.uniform(0.3, 2.0)),
              first_cell=int(rng.integers(0, 10 ** 9)), fq=fq if (dfm >= 3 or fast) else None, fast=fast, T_avg=fq["T_avg"])
    ref, rst = oracle.sample_particles(cells, sp, fx["df"], fq, o, **kw)
    got, st = api.sample_particles(cells, sp, fx["df"], fq, o, batch_events=int(rng.choice([0, 1, 3])), **kw)
    assert len(got) == len(ref["E"]), (case, len(got), len(ref["E"]))
    assert np.array_equal(got["cell"], ref["cell"]) and np.array_equal(got["event"], ref["event"]) and np.array_equal(got["species"], ref["species"])
    for f in ("tau", "x", "y", "eta", "t", "z", "E", "px", "py", "pz"):
        assert np.allclose(got[f], ref[f], rtol=1e-11, atol=1e-13), (case, f)
    assert st["n_hadrons_drawn"] == rst["drawn"] and st["n_momentum_samples"] == rst["samples"]
