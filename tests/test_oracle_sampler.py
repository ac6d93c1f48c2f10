"""CPU: the particle sampler restatement (oracle/cf_oracle.c) -- the counter-based RNG against the
published Philox known answers, and the sampler against the smooth Cooper-Frye spectrum of the same surface, which is the
reference's own validation method for it (test_sampler: sampled dN/dy, pT spectra vs the smooth integrals,
emissionfunction.cpp:905-1260).  The device sampler is held to these lists bit for bit in tests/test_gpu_sampler.py."""
import numpy as np
import pytest

from is3d_amd import inputs, synth
from oracle import oracle


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11)."""
    assert oracle.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_streams():
    """u = ((a >> 5) 2^26 + (b >> 6)) 2^-53 from consecutive outputs of block (blk, stream, cell, event), key = seed."""
    seed, stream, cell, event = 0x0123456789abcdef, 3, 17, 5
    u = oracle.rng_uniforms(seed, stream, cell, event, 6)
    key = [seed & 0xffffffff, seed >> 32]
    want = []
    for blk in range(3):
        o = oracle.philox4x32_10([blk, stream, cell, event], key)
        want += [((o[0] >> 5) * 67108864.0 + (o[1] >> 6)) / 9007199254740992.0, ((o[2] >> 5) * 67108864.0 + (o[3] >> 6)) / 9007199254740992.0]
    assert np.array_equal(u, np.array(want))
    big = oracle.rng_uniforms(11, 0, 0, 0, 200000)
    assert 0.0 <= big.min() and big.max() < 1.0 and abs(big.mean() - 0.5) < 4 / np.sqrt(12 * big.size)
    assert not np.array_equal(oracle.rng_uniforms(11, 0, 1, 0, 8), big[:8]) and not np.array_equal(oracle.rng_uniforms(11, 1, 0, 0, 8), big[:8])


def smooth_yields(cells, sp, fx, o):
    """per species: N = int dy pT dpT dphi dN/(pT dpT dphi dy), <pT>, from the smooth oracle spectrum (pT weights carry the pT Jacobian)"""
    g = fx["grid_w"]
    ny = 1 if o["dimension"] == 2 else len(g["y"])
    s = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], o).reshape(ny, len(g["phi"]), len(g["pT"]), len(sp["mass"]))
    dndy = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"], s)
    pt1 = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"] * g["pT"], s)
    if o["dimension"] == 2:
        return dndy[0], pt1[0] / dndy[0]
    h = g["y"][1] - g["y"][0]
    return dndy.sum(axis=0) * h, pt1.sum(axis=0) / dndy.sum(axis=0)     # the spectrum vanishes at the ends of the y grid here


@pytest.mark.parametrize("dim,df_mode", [(3, 2), (3, 1), (2, 1)])
def test_sampler_reproduces_the_smooth_spectrum(fx, dim, df_mode):
    ncell = 160
    cells = synth.synth_surface(ncell, dim, seed=700 + dim)
    cells = {k: v.copy() for k, v in cells.items()}
    if dim == 3:
        cells["eta"] *= 0.25                       # |eta| <= 1: every hadron lands well inside the y grid (+-5)
    cells["dat"][3] *= -1.0
    cells["dax"][3] *= -1.0
    cells["day"][3] *= -1.0
    cells["dan"][3] *= -1.0                        # one cell with u.dsigma <= 0: never emits
    sp = fx["pikp"]
    o = dict(dimension=dim, df_mode=df_mode)
    y_cut = 0.8
    N_smooth, pT_smooth = smooth_yields(cells, sp, fx, o)
    if dim == 2:
        N_smooth = N_smooth * 2.0 * y_cut           # boost invariant: dN/dy at y = 0 times the sampled rapidity window
    n_events = int(np.ceil(60000.0 / N_smooth.sum()))
    gla = inputs.feqmod_tables(0.15)
    p, st = oracle.sample_particles(cells, sp, fx["df"], gla, o, n_events=n_events, seed=20260004, y_cut=y_cut)
    assert st["n_kept"] == len(p["E"]) and st["acceptances"] == st["drawn"] and st["samples"] >= st["drawn"]
    assert not (p["cell"] == 3).any()
    # ordering: event, then cell
    key = p["event"] * ncell + p["cell"]
    assert (np.diff(key) >= 0).all()
    mass = sp["mass"][p["species"]]
    assert np.allclose(p["E"] ** 2 - p["px"] ** 2 - p["py"] ** 2 - p["pz"] ** 2, mass ** 2, rtol=0, atol=2e-9 * p["E"] ** 2)   # on shell
    assert np.allclose(0.5 * np.log((p["E"] + p["pz"]) / (p["E"] - p["pz"])), p["rapidity"], atol=1e-9)
    assert np.allclose(p["t"] ** 2 - p["z"] ** 2, p["tau"] ** 2, rtol=1e-12)
    if dim == 2:
        assert np.abs(p["rapidity"]).max() < y_cut
    for s in range(3):
        sel = p["species"] == s
        n = sel.sum()
        want = N_smooth[s] * n_events
        assert abs(n - want) < 4.5 * np.sqrt(want), (s, n, want)
        pT = np.hypot(p["px"][sel], p["py"][sel])
        assert abs(pT.mean() - pT_smooth[s]) < 4.5 * pT.std() / np.sqrt(n), (s, pT.mean(), pT_smooth[s])


def test_sampler_is_a_pure_function_of_seed_cell_event(fx):
    """Counter-based streams: the hadrons of (cell, event) do not depend on which other cells or events are sampled."""
    cells = synth.synth_surface(120, 3, seed=31)
    sp = fx["pikp"]
    gla = inputs.feqmod_tables(0.15)
    o = dict(dimension=3, df_mode=2)
    a, _ = oracle.sample_particles(cells, sp, fx["df"], gla, o, n_events=6, seed=99)
    b, _ = oracle.sample_particles(cells, sp, fx["df"], gla, o, n_events=6, seed=99)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c, _ = oracle.sample_particles(cells, sp, fx["df"], gla, o, n_events=6, seed=100)
    assert len(c["E"]) != len(a["E"]) or not np.array_equal(c["E"], a["E"])
    # a shard with its global offset samples exactly the hadrons the whole surface gives those cells (multi-GPU sharding)
    sub = {k: v[40:90] for k, v in cells.items()}
    d, _ = oracle.sample_particles(sub, sp, fx["df"], gla, o, n_events=4, seed=99, first_cell=40)
    sel = (a["cell"] >= 40) & (a["cell"] < 90) & (a["event"] < 4)
    assert sel.sum() > 0 and all(np.array_equal(d[k], a[k][sel]) for k in a)
    with pytest.raises(RuntimeError):
        oracle.sample_particles(cells, sp, fx["df"], gla, dict(dimension=3, df_mode=5), n_events=1, seed=1)


@pytest.mark.parametrize("df_mode", [4, 3])
def test_modified_equilibrium_sampler_reproduces_the_feqmod_spectrum(fx, df_mode):
    """df_mode 3 / 4: momenta drawn at T_mod and rescaled p = A p_mod (rescale_momentum, sampling_kernels.cpp:619-650), mean
    numbers n_linear / z n_eq (max_particle_number :306-348), no viscous weight; cells where feqmod breaks down (df_mode 3)
    fall back to the linear delta-f with its weight.  Yields and <pT> against the smooth feqmod restatement."""
    ncell = 160
    cells = synth.synth_surface(ncell, 3, seed=720 + df_mode)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["eta"] *= 0.25
    if df_mode == 3:
        cells["bulkPi"][::8] = -5.0 * cells["P"][::8]          # breakdown cells
    sp = fx["pikp"]
    o = dict(dimension=3, df_mode=df_mode)
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
    g = fx["grid_w"]
    smooth, nb = oracle.dN_pTdpTdphidy_feqmod(cells, sp, fx["grid"], fx["df"], fq, o)
    s4 = smooth.reshape(len(g["y"]), len(g["phi"]), len(g["pT"]), 3)
    dndy = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"], s4)
    pt1 = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"] * g["pT"], s4)
    h = g["y"][1] - g["y"][0]
    N_smooth, pT_smooth = dndy.sum(axis=0) * h, pt1.sum(axis=0) / dndy.sum(axis=0)
    n_events = int(np.ceil(60000.0 / N_smooth.sum()))
    p, st = oracle.sample_particles(cells, sp, fx["df"], fq, o, n_events=n_events, seed=20260005, fq=fq)
    assert st["breakdown"] == nb and (nb > 0) == (df_mode == 3)
    for s in range(3):
        sel = p["species"] == s
        want = N_smooth[s] * n_events
        assert abs(sel.sum() - want) < 4.5 * np.sqrt(want), (s, sel.sum(), want)
        pT = np.hypot(p["px"][sel], p["py"][sel])
        assert abs(pT.mean() - pT_smooth[s]) < 4.5 * pT.std() / np.sqrt(sel.sum()), (s, pT.mean(), pT_smooth[s])
    mass = sp["mass"][p["species"]]
    assert np.allclose(p["E"] ** 2 - p["px"] ** 2 - p["py"] ** 2 - p["pz"] ** 2, mass ** 2, rtol=0, atol=2e-9 * p["E"] ** 2)


@pytest.mark.parametrize("df_mode", [2, 4])
def test_fast_mode_on_an_isothermal_surface(fx, df_mode):
    """fast = 1 takes the species densities at the surface-average temperature (sampling_kernels.cpp:1044-1056,
    deltafReader.cpp:536-650); on an isothermal surface that is the same number up to rounding, so the list is the one of the
    regular mode."""
    cells = synth.synth_surface(200, 3, seed=733)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["T"][:] = 0.151
    sp = fx["pikp"]
    fq = inputs.feqmod_tables(0.151)
    o = dict(dimension=3, df_mode=df_mode)
    a, sa = oracle.sample_particles(cells, sp, fx["df"], fq, o, n_events=200, seed=5, fq=fq)
    b, sb = oracle.sample_particles(cells, sp, fx["df"], fq, o, n_events=200, seed=5, fq=fq, fast=1, T_avg=0.151)
    assert sa["n_kept"] == sb["n_kept"] > 50 and all(np.array_equal(a[k], b[k]) for k in a)
    # a different average temperature changes the mean numbers (and with them the list)
    c, sc = oracle.sample_particles(cells, sp, fx["df"], fq, o, n_events=200, seed=5, fq=fq, fast=1, T_avg=0.140)
    assert sc["drawn"] < sb["drawn"]


@pytest.mark.parametrize("df_mode", [2, 1, 3])
def test_sampler_with_baryon_reproduces_the_smooth_spectrum(fx, df_mode):
    """include_baryon = 1 (sampling_kernels.cpp:942-964, :282-359, :361-453, :456-617, :619-650): chem = b mu_B/T in the mean
    numbers and the momentum weights, bulk1 / diffusion terms in the viscous weight, diffusion in the momentum rescaling of
    df_mode 3.  Proton and antiproton yields and <pT> against the smooth baryon spectra of the same surface."""
    ncell = 160
    cells = synth.synth_surface(ncell, 3, seed=760 + df_mode, baryon=True)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["eta"] *= 0.25
    sp = inputs.species([211, 2212, -2212])
    dff = inputs.df_tables_full()
    o = dict(dimension=3, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=1)
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
    g = fx["grid_w"]
    if df_mode == 3:
        smooth, nb = oracle.dN_pTdpTdphidy_feqmod(cells, sp, fx["grid"], dff, fq, o)
        assert nb == 0
    else:
        smooth = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], dff, o)
    s4 = smooth.reshape(len(g["y"]), len(g["phi"]), len(g["pT"]), 3)
    dndy = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"], s4)
    pt1 = np.einsum("j,i,kjis->ks", g["phi_w"], g["pT_w"] * g["pT"], s4)
    h = g["y"][1] - g["y"][0]
    N_smooth, pT_smooth = dndy.sum(axis=0) * h, pt1.sum(axis=0) / dndy.sum(axis=0)
    assert N_smooth[1] > 3.0 * N_smooth[2]                             # mu_B > 0: more protons than antiprotons
    n_events = int(np.ceil(90000.0 / N_smooth.sum()))
    p, st = oracle.sample_particles(cells, sp, dff, fq, o, n_events=n_events, seed=20260006, fq=fq if df_mode == 3 else None)
    for s in range(3):
        sel = p["species"] == s
        want = N_smooth[s] * n_events
        assert abs(sel.sum() - want) < 4.5 * np.sqrt(want), (s, sel.sum(), want)
        pT = np.hypot(p["px"][sel], p["py"][sel])
        assert abs(pT.mean() - pT_smooth[s]) < 4.5 * pT.std() / np.sqrt(sel.sum()), (s, pT.mean(), pT_smooth[s])
    with pytest.raises(RuntimeError):
        oracle.sample_particles(cells, sp, dff, fq, dict(o, df_mode=4), n_events=1, seed=1, fq=fq)
