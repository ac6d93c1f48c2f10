"""GPU (-m gpu): the device particle sampler (is3d_sample_particles) through the C ABI against the
oracle's restatement on the same counter-based streams: the particle LISTS must agree -- same hadrons, same order, momenta
to rounding (device libm vs glibc differ in the last bits of log/exp/sin/cos) -- and, at larger statistics, against the
smooth Cooper-Frye spectrum computed by the device path itself (the reference's own test_sampler methodology)."""
import numpy as np
import pytest

from conftest import relerr
from is3d_amd import api, inputs, synth
from oracle import oracle  # the checker

pytestmark = pytest.mark.gpu
FLOAT_FIELDS = ["tau", "x", "y", "eta", "t", "z", "E", "px", "py", "pz"]


def compare_lists(got, ref):
    assert len(got) == len(ref["E"]), (len(got), len(ref["E"]))
    assert np.array_equal(got["cell"], ref["cell"]) and np.array_equal(got["event"], ref["event"]) and np.array_equal(got["species"], ref["species"])
    for f in FLOAT_FIELDS:
        assert np.allclose(got[f], ref[f], rtol=1e-11, atol=1e-13), f


@pytest.mark.parametrize("dim,df_mode,species", [(3, 2, "pikp"), (3, 1, "pikp"), (2, 1, "pikp"), (3, 2, "urqmd")])
def test_sampler_lists_match_the_oracle(fx, dim, df_mode, species):
    n = 400 if species == "pikp" else 120
    cells = synth.synth_surface(n, dim, seed=800 + dim)
    cells = {k: v.copy() for k, v in cells.items()}
    for k in ("dat", "dax", "day", "dan"):
        cells[k][[5, 77]] *= -1.0                       # skipped cells
    sp = inputs.species(species)
    gla = inputs.feqmod_tables(0.15)
    o = dict(dimension=dim, df_mode=df_mode)
    n_events = 150 if species == "pikp" else 30
    ref, rst = oracle.sample_particles(cells, sp, fx["df"], gla, o, n_events=n_events, seed=4242, y_cut=0.7)
    got, st = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=n_events, seed=4242, y_cut=0.7)
    assert len(ref["E"]) > 100
    compare_lists(got, ref)
    assert st["n_particles"] == rst["n_kept"] and st["n_hadrons_drawn"] == rst["drawn"]
    assert st["n_momentum_samples"] == rst["samples"] and st["n_acceptances"] == rst["acceptances"]
    assert st["n_cells_skipped"] == 2
    if species == "urqmd":
        assert st["n_classes"] == 75
    # flags: bulk / shear switched off change the viscous weight, not the draws
    o2 = dict(o, include_bulk_deltaf=0, include_shear_deltaf=0)
    ref2, _ = oracle.sample_particles(cells, sp, fx["df"], gla, o2, n_events=n_events, seed=4242, y_cut=0.7)
    got2, _ = api.sample_particles(cells, sp, fx["df"], gla, o2, n_events=n_events, seed=4242, y_cut=0.7)
    compare_lists(got2, ref2)
    assert len(ref2["E"]) != len(ref["E"]) or not np.array_equal(ref2["E"], ref["E"])


def test_sampler_sharding_capacity_and_errors(fx):
    cells = synth.synth_surface(600, 3, seed=811)
    sp = fx["pikp"]
    gla = inputs.feqmod_tables(0.15)
    o = dict(dimension=3, df_mode=2)
    whole, st = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=25, seed=7)
    # shards with their global offsets reproduce the whole surface's hadrons (multi-GPU: one shard per rank, lists concatenated)
    parts = []
    for lo, hi in ((0, 250), (250, 600)):
        sub = {k: v[lo:hi] for k, v in cells.items()}
        pr, _ = api.sample_particles(sub, sp, fx["df"], gla, o, n_events=25, seed=7, first_cell=lo)
        parts.append(pr)
    merged = np.concatenate(parts)
    merged = merged[np.lexsort((merged["cell"], merged["event"]))]
    # within one (event, cell) the draw order is preserved by the stable sort keys above only if lexsort is stable: it is
    assert len(merged) == len(whole) and all(np.array_equal(merged[f], whole[f]) for f in whole.dtype.names)
    # event batching (count / scan / fill per batch of events) does not change the list
    for be in (1, 7):
        b, _ = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=25, seed=7, batch_events=be)
        assert len(b) == len(whole) and all(np.array_equal(b[f], whole[f]) for f in whole.dtype.names)
    empty, ste = api.sample_particles({k: v[:0] for k, v in cells.items()}, sp, fx["df"], gla, o, n_events=3, seed=7)
    assert len(empty) == 0 and ste["n_particles"] == 0
    # a buffer that is too small: IS3D_ENOMEM, the count is still reported
    with pytest.raises(api.Is3dError) as e:
        api.sample_particles(cells, sp, fx["df"], gla, o, n_events=25, seed=7, capacity=10)
    assert e.value.code == -4 and str(len(whole)) in str(e.value)
    # other seeds give other events; same seed the same list
    again, _ = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=25, seed=7)
    assert all(np.array_equal(again[f], whole[f]) for f in whole.dtype.names)
    other, _ = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=25, seed=8)
    assert len(other) != len(whole) or not np.array_equal(other["E"], whole["E"])
    for bad in (dict(dimension=3, df_mode=5), dict(dimension=3, df_mode=4), dict(dimension=3, df_mode=2, include_baryon=1), dict(dimension=4, df_mode=1)):
        with pytest.raises(api.Is3dError) as e:
            api.sample_particles(cells, sp, fx["df"], gla, bad, n_events=1, seed=1)
        assert e.value.code == -1
    with pytest.raises(api.Is3dError) as e:                  # photons cannot be sampled (reference: exit)
        api.sample_particles(cells, inputs.species([211, 22]), fx["df"], gla, o, n_events=1, seed=1)
    assert e.value.code == -1 and "photon" in str(e.value)
    hot = {k: v.copy() for k, v in cells.items()}
    hot["T"][9] = 0.3
    with pytest.raises(api.Is3dError) as e:
        api.sample_particles(hot, sp, fx["df"], gla, o, n_events=1, seed=1)
    assert e.value.code == -3 and "cell 9" in str(e.value)


@pytest.mark.parametrize("dim,df_mode,fast", [(3, 4, 0), (3, 3, 0), (2, 4, 0), (2, 3, 1), (3, 2, 1), (3, 4, 1)])
def test_modified_equilibrium_and_fast_lists_match_the_oracle(fx, dim, df_mode, fast):
    """df_mode 3 / 4: momenta drawn at T_mod and rescaled with A (rescale_momentum), n_linear / z n_eq mean numbers, breakdown
    cells on the linear delta-f; fast = 1: densities at the average temperature, breakdown test at T_switch."""
    cells = synth.synth_surface(400, dim, seed=830 + dim + df_mode)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["bulkPi"][::9] = -5.0 * cells["P"][::9]          # df_mode 3: breakdown; df_mode 4: clamped to -(1 - 1e-5) P
    sp = fx["pikp"]
    T_avg = inputs.surface_average_T(cells)
    fq = inputs.feqmod_tables(T_avg)
    o = dict(dimension=dim, df_mode=df_mode)
    kw = dict(n_events=150, seed=31337, y_cut=0.9, fq=fq, fast=fast, T_avg=fq["T_avg"], T_avg_switch=0.151)
    ref, rst = oracle.sample_particles(cells, sp, fx["df"], fq, o, **kw)
    got, st = api.sample_particles(cells, sp, fx["df"], fq, o, **kw)
    assert len(ref["E"]) > 100
    compare_lists(got, ref)
    assert st["n_cells_breakdown"] == rst["breakdown"] and (rst["breakdown"] > 0) == (df_mode == 3)
    assert st["n_hadrons_drawn"] == rst["drawn"] and st["n_momentum_samples"] == rst["samples"]


def test_sampled_yields_follow_the_device_smooth_spectrum(fx):
    """2e4 cells, ~1e6 hadrons: species yields and <pT> from the sampled list against the integrals of the smooth spectrum
    the tile kernel computes for the same surface (both on the device; 4.5 sigma)."""
    n = 20000
    cells = synth.synth_surface(n, 3, seed=812)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["eta"] *= 0.25
    sp = fx["pikp"]
    gla = inputs.feqmod_tables(0.15)
    o = dict(dimension=3, df_mode=2)
    g = fx["grid_w"]
    smooth, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    s4 = smooth.reshape(len(g["y"]), len(g["phi"]), len(g["pT"]), 3)
    dndy = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"], s4)
    pt1 = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"] * g["pT"], s4)
    h = g["y"][1] - g["y"][0]
    N_smooth, pT_smooth = dndy.sum(axis=0) * h, pt1.sum(axis=0) / dndy.sum(axis=0)
    n_events = int(np.ceil(1.0e6 / N_smooth.sum()))
    p, st = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=n_events, seed=2026)
    assert st["n_particles"] == len(p) > 5e5
    for s in range(3):
        sel = p["species"] == s
        want = N_smooth[s] * n_events
        assert abs(sel.sum() - want) < 4.5 * np.sqrt(want), (s, sel.sum(), want)
        pT = np.hypot(p["px"][sel], p["py"][sel])
        assert abs(pT.mean() - pT_smooth[s]) < 4.5 * pT.std() / np.sqrt(sel.sum())
    m = sp["mass"][p["species"]]
    assert relerr(p["E"] ** 2, p["px"] ** 2 + p["py"] ** 2 + p["pz"] ** 2 + m ** 2) < 1e-12


@pytest.mark.parametrize("dim,df_mode,diff", [(3, 2, 1), (3, 1, 1), (3, 3, 1), (2, 2, 1), (2, 3, 1), (3, 2, 0)])
def test_sampler_with_baryon_lists_match_the_oracle(fx, dim, df_mode, diff):
    """include_baryon = 1: the baryon number is part of the class key, chem = b mu_B/T in the density integrals and the momentum
    weight, c1 c3 c4 | G betaV terms in the viscous weight, diffusion in the df_mode-3 momentum rescaling; breakdown cells
    (df_mode 3) fall back to the linear delta-f with the baryon terms."""
    cells = synth.synth_surface(300, dim, seed=850 + dim + df_mode, baryon=True)
    cells = {k: v.copy() for k, v in cells.items()}
    if df_mode == 3:
        cells["bulkPi"][::9] = -5.0 * cells["P"][::9]
    sp = inputs.species([211, 321, 2212, -2212, 3122, -3122])
    dff = inputs.df_tables_full()
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
    o = dict(dimension=dim, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=diff)
    kw = dict(n_events=150, seed=90210, y_cut=0.8, fq=fq if df_mode == 3 else None)
    ref, rst = oracle.sample_particles(cells, sp, dff, fq, o, **kw)
    got, st = api.sample_particles(cells, sp, dff, fq, o, **kw)
    assert len(ref["E"]) > 100
    compare_lists(got, ref)
    assert st["n_hadrons_drawn"] == rst["drawn"] and st["n_momentum_samples"] == rst["samples"]
    assert st["n_cells_breakdown"] == rst["breakdown"] and (rst["breakdown"] > 0) == (df_mode == 3)
    assert st["n_classes"] == 6                                    # p / pbar and Lambda / Lambdabar are separate classes now
    if diff:
        n_p, n_pbar = (got["species"] == 2).sum(), (got["species"] == 3).sum()
        assert n_p > 2 * n_pbar
    for bad in (dict(o, df_mode=4),):
        with pytest.raises(api.Is3dError) as e:
            api.sample_particles(cells, sp, dff, fq, bad, n_events=1, seed=1, fq=fq)
        assert e.value.code == -1
    if df_mode >= 2:
        # fast = 1: species densities at the surface-average (T, mu_B) (deltafReader.cpp:536-650), breakdown test at T_switch
        fkw = dict(kw, fq=fq, fast=1, T_avg=fq["T_avg"], T_avg_switch=0.1505, muB_avg=float(np.mean(cells["muB"])))
        fref, frst = oracle.sample_particles(cells, sp, dff, fq, o, **fkw)
        fgot, fst = api.sample_particles(cells, sp, dff, fq, o, **fkw)
        assert len(fref["E"]) > 100 and len(fref["E"]) != len(ref["E"])
        compare_lists(fgot, fref)
        assert fst["n_hadrons_drawn"] == frst["drawn"] and fst["n_cells_breakdown"] == frst["breakdown"]
        with pytest.raises(api.Is3dError) as e:                       # average mu_B outside the table
            api.sample_particles(cells, sp, dff, fq, o, **dict(fkw, muB_avg=0.9))
        assert e.value.code == -3
    if diff and dim == 3:
        out = {k: v.copy() for k, v in cells.items()}
        out["muB"][7] = 0.95
        with pytest.raises(api.Is3dError) as e:
            api.sample_particles(out, sp, dff, fq, o, n_events=1, seed=1, fq=fq if df_mode == 3 else None)
        assert e.value.code == -3 and "cell 7" in str(e.value)


@pytest.mark.parametrize("dim,df_mode,baryon", [(3, 1, 0), (3, 2, 0), (2, 3, 0), (3, 4, 0), (2, 4, 0), (3, 1, 1), (3, 2, 1), (2, 3, 1)])
def test_total_yield_matches_the_oracle(fx, dim, df_mode, baryon):
    """is3d_total_yield (calculate_total_yield, sampling_kernels.cpp:653-830): species densities at the surface averages on the
    host, per-cell weights and the reduction on the device; against the oracle's serial restatement."""
    cells = synth.synth_surface(20000, dim, seed=400 + dim, baryon=bool(baryon))
    cells["dat"][::7] *= -1.0                          # some cells with u.dsigma <= 0 (skipped, :689)
    sp = inputs.species([211, 321, 2212, -2212, 3122, -3122, 333])
    df = inputs.df_tables_full() if baryon else fx["df"]
    avg = inputs.surface_averages(cells)
    gla = inputs.feqmod_tables(avg[0])
    o = dict(dimension=dim, df_mode=df_mode, include_baryon=baryon, include_baryondiff_deltaf=baryon)
    want, wdens = oracle.total_yield(cells, sp, df, gla, avg, o, y_cut=0.8)
    got, gdens = api.total_yield(cells, sp, df, gla, avg, o, y_cut=0.8)
    assert np.allclose(gdens, wdens, rtol=1e-13, atol=0)
    assert abs(got / want - 1) < 1e-11 and want > 0
    # a temperature outside the table aborts the reference in evaluate_df_coefficients (:761)
    cells["T"][123] = 0.31
    with pytest.raises(api.Is3dError) as e:
        api.total_yield(cells, sp, df, gla, avg, o, y_cut=0.8)
    assert e.value.code == api.IS3D_EDOMAIN and "cell 123" in str(e.value)


def test_sampler_config5_size(fx):
    """BASELINE config 5's stated size for the sampler: the 1e6-cell config-3 surface, 305 species, 20 events (6e5 hadrons).
    (i) the list restricted to the first 5e4 cells IS the oracle's list for that slice (same hadrons, same order) -- what
    tests/bench_sampler.py prints; (ii) sharding by first_cell: four shards concatenated per event give the same list;
    (iii) event batching does not change it; (iv) the mean multiplicity per event is the analytic yield (is3d_total_yield)
    within the average-temperature approximation of that estimate."""
    n, nev, nc = 1000000, 20, 50000
    cells = synth.synth_surface(n, 3)
    sp = fx["urqmd"]
    gla = inputs.feqmod_tables(0.15)
    o = dict(dimension=3, df_mode=2)
    p, st = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=nev, seed=20260002)
    assert st["n_particles"] == len(p) > 3e5 and np.all(np.diff(p["event"]) >= 0)
    # (i)
    ref, rst = oracle.sample_particles({k: v[:nc] for k, v in cells.items()}, sp, fx["df"], gla, o, n_events=nev, seed=20260002)
    sel = p["cell"] < nc
    assert int(sel.sum()) == len(ref["E"]) and np.array_equal(p["species"][sel], ref["species"]) and np.array_equal(p["cell"][sel], ref["cell"])
    assert np.allclose(p["E"][sel], ref["E"], rtol=1e-11, atol=0) and np.allclose(p["pz"][sel], ref["pz"], rtol=1e-9, atol=1e-13)
    # (ii)
    shards = []
    for r in range(4):
        lo, hi = api.shard_bounds(n, r, 4)
        q, _ = api.sample_particles({k: v[lo:hi] for k, v in cells.items()}, sp, fx["df"], gla, o, n_events=nev, seed=20260002, first_cell=lo)
        shards.append(q)
    merged = np.concatenate(shards)
    merged = merged[np.lexsort((merged["cell"], merged["event"]))]
    assert len(merged) == len(p) and all(np.array_equal(merged[f], p[f]) for f in ("event", "cell", "species", "E", "px", "py", "pz", "tau", "eta"))
    # (iii)
    q, _ = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=nev, seed=20260002, batch_events=3)
    assert len(q) == len(p) and np.array_equal(q["E"], p["E"]) and np.array_equal(q["cell"], p["cell"])
    # (iv)
    avg = inputs.surface_averages(cells)
    N, _ = api.total_yield(cells, sp, fx["df"], inputs.feqmod_tables(avg[0]), avg, o)
    assert abs(len(p) / nev / N - 1) < 0.08


def test_sampler_plan_is_the_one_shot_entry_on_resident_data(fx):
    """is3d_sampler_plan_* (device-resident cell arrays and particle buffer, persistent workspaces): the same list as is3d_sample_particles for the
    same (seed, first_cell, events) -- df_mode 2 (species by bisection of the stored running sums) and df_mode 3 (linear inversion: its weights may be
    negative) --, a second execute of the same shape allocates nothing, another seed gives another list, a shard with its global offset gives the
    whole surface's hadrons for its cells, and the refusals: more cells than the plan holds, a buffer too small, zero events."""
    import torch
    dev = torch.device("cuda:0")
    n = 3000
    cells = synth.synth_surface(n, 3, seed=821)
    sp = inputs.species("urqmd")
    fields = list(synth.CELL_FIELDS) + ["x", "y"]
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in fields}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    for df_mode in (2, 3):
        gla = inputs.feqmod_tables(inputs.surface_average_T(cells))
        o = dict(dimension=3, df_mode=df_mode)
        fq = gla if df_mode == 3 else None
        ref, rst = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=12, seed=99, fq=fq)
        plan = api.SamplerPlan(sp, fx["df"], gla, o, max_cells=n, fq=fq)
        count, st = plan.execute(n, ptrs, 12, 99, x_ptr=ptrs["x"], y_ptr=ptrs["y"])                     # count only
        assert count == len(ref) == rst["n_particles"] and st["ms_h2d"] == 0.0
        buf = torch.zeros(count * api.PARTICLE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        p0, a0 = api.resource_counters()
        got_n, st = plan.execute(n, ptrs, 12, 99, particles_ptr=buf.data_ptr(), capacity=count, x_ptr=ptrs["x"], y_ptr=ptrs["y"])
        assert api.resource_counters() == (p0, a0)                                                      # workspaces of this shape exist already
        got = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=api.PARTICLE_DTYPE)[:got_n]
        assert got_n == count and all(np.array_equal(got[f], ref[f]) for f in got.dtype.names), df_mode
        assert st["n_hadrons_drawn"] == rst["n_hadrons_drawn"] and st["ms_density"] > 0 and st["ms_density"] < st["ms_prep"] and st["ms_poisson"] <= st["ms_count"]
        # another seed: another list (the buffer may be too small now: IS3D_ENOMEM carries the full count)
        c2, _ = plan.execute(n, ptrs, 12, 100)
        assert c2 != count or c2 > 0
        if c2 > count:
            with pytest.raises(api.Is3dError) as e:
                plan.execute(n, ptrs, 12, 100, particles_ptr=buf.data_ptr(), capacity=count)
            assert e.value.code == api.IS3D_ENOMEM
        # a shard of the resident arrays with its global offset: the whole surface's hadrons of those cells
        lo, hi = 1000, 2200
        sub = {k: v + 8 * lo for k, v in ptrs.items()}
        cs, _ = plan.execute(hi - lo, sub, 12, 99, first_cell=lo)
        sel = (ref["cell"] >= lo) & (ref["cell"] < hi)
        assert cs == int(sel.sum())
        b2 = torch.zeros(max(cs, 1) * api.PARTICLE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        plan.execute(hi - lo, sub, 12, 99, particles_ptr=b2.data_ptr(), capacity=cs, first_cell=lo, x_ptr=sub["x"], y_ptr=sub["y"])
        g2 = np.frombuffer(b2.cpu().numpy().tobytes(), dtype=api.PARTICLE_DTYPE)[:cs]
        assert all(np.array_equal(g2[f], ref[f][sel]) for f in g2.dtype.names)
        with pytest.raises(api.Is3dError) as e:
            plan.execute(n + 1, ptrs, 12, 99)
        assert e.value.code == api.IS3D_EINVAL and "created for" in str(e.value)
        with pytest.raises(api.Is3dError) as e:
            plan.execute(n, ptrs, 0, 99)
        assert e.value.code == api.IS3D_EINVAL
        plan.close()
