"""GPU (-m gpu): the iS3D-compatible command line driver end to end -- reference file formats in, reference
file formats out -- against the oracle run on the same parsed inputs."""
import os
import subprocess

import numpy as np
import pytest

import refformat
from conftest import ROOT, relerr
from is3d_amd import api, inputs, synth
from oracle import oracle  # the checker

pytestmark = pytest.mark.gpu


def read_spectra_file(path, header=False):
    rows = []
    with open(path) as f:
        if header:
            assert f.readline() == "y\tphip\tpT\tdN_pTdpTdphidy\n"
        for ln in f:
            if ln.strip():
                rows.append([float(x) for x in ln.split("\t")])
    return np.array(rows)


@pytest.mark.parametrize("dim,df_mode", [(3, 1), (3, 2), (2, 1)])
def test_cli_run_directory(tmp_path, fx, dim, df_mode):
    ids = [211, -2212, 3122, 321]          # antibaryon ids exist only because the reader synthesises them
    cells = synth.synth_surface(23 if dim == 3 else 7, dim, seed=40 + dim)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(dimension=dim, df_mode=df_mode))
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Number of chosen particles: 4" in r.stdout and "Total number of freezeout cells: %d" % len(cells["tau"]) in r.stdout

    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    sp = inputs.species(ids)
    g = fx["grid_w"]
    ref = oracle.dN_pTdpTdphidy(parsed, sp, fx["grid"], fx["df"], dict(dimension=dim, df_mode=df_mode))
    ny = 21 if dim == 3 else 1
    ref4 = ref.reshape(ny, 24, 32, 4)
    allsp = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy.dat"))
    assert allsp.shape == (4 * ny * 24 * 32, 4)
    want = np.transpose(ref4, (3, 0, 1, 2)).reshape(-1)             # file order: species, y, phi, pT
    assert relerr(allsp[:, 3], want, floor=1e-250) < 2e-8           # files carry 9 significant digits
    yv = g["y"] if dim == 3 else np.zeros(1)
    assert np.allclose(allsp[: 32, 2], g["pT"], rtol=1e-8) and np.allclose(allsp[:ny * 24 * 32:24 * 32, 0], yv, atol=1e-12)
    one = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy_-2212.dat"), header=True)
    assert relerr(one[:, 3], ref4[:, :, :, 1].reshape(-1), floor=1e-250) < 2e-8
    dndy = np.loadtxt(os.path.join(root, "results", "dN_dy_3122.dat"), ndmin=2)
    want_dy = np.einsum("j,i,kji->k", g["phi_w"], g["pT_w"], ref4[:, :, :, 2])
    assert np.allclose(dndy[:, 1], want_dy, rtol=3e-8)
    vn = np.loadtxt(os.path.join(root, "results", "vn_continuous", "vn_211.dat"), ndmin=2)
    assert vn.shape == (ny * 32, 9)
    num = np.einsum("j,kji->ki", np.exp(2j * g["phi"]) * g["phi_w"], ref4[:, :, :, 0])
    den = np.einsum("j,kji->ki", g["phi_w"], ref4[:, :, :, 0])
    v2 = np.where(den < 1e-15, 0.0, np.abs(num) / np.where(den == 0, 1, den)).reshape(-1)
    assert np.allclose(vn[:, 3], v2, rtol=1e-6, atol=1e-7)
    avg = [float(x) for x in open(os.path.join(root, "average_thermodynamic_quantities.dat")).read().split()]
    assert len(avg) == 5 and 0.14 < avg[0] < 0.16 and avg[3] == 0.0


@pytest.mark.parametrize("hrg_eos", [2, 3])
def test_cli_other_particle_lists(tmp_path, fx, hrg_eos):
    """hrg_eos = 2 (the reference's shipped default: PDG/pdg_smash.dat, deltaf_coefficients/vh/smash/, the conventional reader) and hrg_eos = 3
    (PDG/pdg_box.dat, the line-oriented list read by read_resonances_smash_box, deltaf_coefficients/vh/smash_box/; readindata.cpp:1687-1713,
    deltafReader.h:27-29): the driver finds its files and the spectra are the oracle's for the species the list defines."""
    ids = [211, -321, 2212, -2212, 3122]
    cells = synth.synth_surface(19, 3, seed=47)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(dimension=3, df_mode=2, hrg_eos=hrg_eos))
    assert not os.path.exists(os.path.join(root, "PDG", "pdg-urqmd_v3.3+.dat"))
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Number of chosen particles: 5" in r.stdout
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    if hrg_eos == 3:
        ent = {e[0]: e for e in refformat.box_entries(refformat.BOX_ROWS)}
        sp = dict(mass=np.array([ent[i][1] for i in ids]), degeneracy=np.array([ent[i][2] for i in ids]),
                  baryon=np.array([ent[i][3] for i in ids]), sign=np.array([ent[i][4] for i in ids]))
    else:
        sp = inputs.species(ids)
    ref = oracle.dN_pTdpTdphidy(parsed, sp, fx["grid"], fx["df"], dict(dimension=3, df_mode=2))
    ref4 = ref.reshape(21, 24, 32, 5)
    for k, i in enumerate(ids):
        one = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy_%d.dat" % i), header=True)
        assert relerr(one[:, 3], ref4[:, :, :, k].reshape(-1), floor=1e-250) < 2e-8, i


@pytest.mark.parametrize("df_mode", [1, 2])
def test_cli_include_baryon(tmp_path, fx, df_mode):
    """25-column surface (muB, nB, Vx, Vy, Vn), full (T, mu_B) coefficient files, bilinear branch."""
    ids = [211, 2212, -2212]
    cells = synth.synth_surface(9, 3, seed=55, baryon=True)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(dimension=3, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=1))
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    ref = oracle.dN_pTdpTdphidy(parsed, inputs.species(ids), fx["grid"], inputs.df_tables_full(),
                                dict(dimension=3, df_mode=df_mode, include_baryon=1, include_baryondiff_deltaf=1))
    allsp = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy.dat"))
    want = np.transpose(ref.reshape(21, 24, 32, 3), (3, 0, 1, 2)).reshape(-1)
    assert relerr(allsp[:, 3], want, floor=1e-250) < 2e-8
    avg = [float(x) for x in open(os.path.join(root, "average_thermodynamic_quantities.dat")).read().split()]
    assert 0.05 < avg[3] < 0.4 and 0.02 < avg[4] < 0.1     # mu_B and n_B averages are filled now


def test_cli_feqmod_with_baryon(tmp_path, fx):
    """df_mode = 3 with include_baryon = 1 end to end: 25-column surface, full (T, mu_B) coefficient files, Gauss-Laguerre file."""
    ids = [211, 2212, -2212]
    cells = synth.synth_surface(11, 3, seed=57, baryon=True)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["bulkPi"][4] = -5.0 * cells["P"][4]            # one breakdown cell: linearised delta-f with the baryon terms
    o = dict(dimension=3, df_mode=3, include_baryon=1, include_baryondiff_deltaf=1)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, o)
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    T_avg = float(open(os.path.join(root, "average_thermodynamic_quantities.dat")).read().split()[0])
    ref, nb = oracle.dN_pTdpTdphidy_feqmod(parsed, inputs.species(ids), fx["grid"], inputs.df_tables_full(), inputs.feqmod_tables(T_avg), o)
    assert nb == 1 and "feqmod breaks down for 1 cells" in r.stdout
    allsp = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy.dat"))
    want = np.transpose(ref.reshape(21, 24, 32, 3), (3, 0, 1, 2)).reshape(-1)
    assert relerr(allsp[:, 3], want, floor=1e-250) < 2e-8


@pytest.mark.parametrize("mode", [0, 4, 5, 6, 7])
def test_cli_other_surface_formats(tmp_path, fx, mode):
    """The other viscous-hydro surface formats end to end: 2+1D boost-invariant MUSIC /
    hic-eventgen files, the old 26-column gpu-vh file and the gpu-vh file with thermal-vorticity columns (mode 5: the reference runs the
    viscous-hydro kernels on it, emissionfunction.cpp:1503-1643)."""
    dim = 3 if mode in (0, 5) else 2
    ids = [211, 321, 2212]
    cells = synth.synth_surface(11, dim, seed=70 + mode)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(dimension=dim, df_mode=2, mode=mode))
    refformat.write_surface_mode(os.path.join(root, "input", "surface.dat"), cells, mode)
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    parsed, _ = api.surface_read(os.path.join(root, "input", "surface.dat"), mode, 0, 0, dim)   # reader parity: tests/test_host_io.py
    ref = oracle.dN_pTdpTdphidy(parsed, inputs.species(ids), fx["grid"], fx["df"], dict(dimension=dim, df_mode=2))
    ny = 21 if dim == 3 else 1
    allsp = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy.dat"))
    want = np.transpose(ref.reshape(ny, 24, 32, 3), (3, 0, 1, 2)).reshape(-1)
    assert relerr(allsp[:, 3], want, floor=1e-250) < 2e-8
    # and the physics is the one of the mode-1 file of the same cells
    ref1 = oracle.dN_pTdpTdphidy(cells, inputs.species(ids), fx["grid"], fx["df"], dict(dimension=dim, df_mode=2))
    assert relerr(ref, ref1, floor=1e-250) < 1e-10


def test_cli_refuses_what_it_does_not_implement(tmp_path):
    cells = synth.synth_surface(3, 3, seed=1)
    for bad in (dict(operation=3), dict(operation=2, include_baryon=1, df_mode=4), dict(mode=2), dict(mode=3), dict(df_mode=5), dict(df_mode=4, include_baryon=1)):
        root = refformat.make_run_dir(str(tmp_path / ("r%d" % len(os.listdir(tmp_path)))), cells, [211], bad)
        r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "iS3D-amd:" in r.stderr
        assert not os.listdir(os.path.join(root, "results", "vn_continuous"))


@pytest.mark.parametrize("dim,df_mode", [(3, 4), (3, 3), (2, 4)])
def test_cli_feqmod(tmp_path, fx, dim, df_mode):
    """df_mode 3 / 4 end to end: Gauss-Laguerre file, deta_min / mass_pion0, the surface averages read back as text
    (Plasma::load_thermodynamic_averages), the full PDG list for the Jonah tables."""
    ids = [211, -2212, 3122, 321]
    cells = synth.synth_surface(23 if dim == 3 else 7, dim, seed=60 + dim)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["bulkPi"][3] = -5.0 * cells["P"][3]            # one breakdown cell in df_mode 3
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(dimension=dim, df_mode=df_mode))
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    T_avg = float(open(os.path.join(root, "average_thermodynamic_quantities.dat")).read().split()[0])
    assert abs(T_avg / inputs.surface_average_T(parsed) - 1) < 1e-12
    fq = inputs.feqmod_tables(T_avg)
    ref, nb = oracle.dN_pTdpTdphidy_feqmod(parsed, inputs.species(ids), fx["grid"], fx["df"], fq, dict(dimension=dim, df_mode=df_mode))
    assert "feqmod breaks down for %d cells" % nb in r.stdout and nb == (1 if df_mode == 3 else 0)
    ny = 21 if dim == 3 else 1
    allsp = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy.dat"))
    want = np.transpose(ref.reshape(ny, 24, 32, 4), (3, 0, 1, 2)).reshape(-1)
    assert relerr(allsp[:, 3], want, floor=1e-250) < 2e-8


@pytest.mark.parametrize("dim,oversample", [(3, 0), (3, 1), (2, 0)])
def test_cli_sampler(tmp_path, fx, dim, oversample):
    """operation = 2 end to end: surface (with positions) + parameters in, results/particle_list_osc.dat out, the list equal to
    the oracle's for the same seed; oversample = 1 sizes the number of events from the analytic mean yield
    (calculate_total_yield, emissionfunction.cpp:1524-1533)."""
    ids = [211, 321, 2212, -2212]
    cells = synth.synth_surface(5000 if dim == 3 else 3000, dim, seed=90 + dim)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(operation=2, dimension=dim, df_mode=2, oversample=oversample,
                                                                  min_num_hadrons=300, sampler_seed=17))
    if dim == 3 and oversample == 0:
        # the reference opens tables/eta/eta_trapezoid_table_41pt.dat when it samples (iS3D.cpp:164-167; the sampler never reads it): a run directory
        # that holds only that one is accepted
        os.rename(os.path.join(root, "tables", "eta", "eta_trapezoid_table_241pt.dat"), os.path.join(root, "tables", "eta", "eta_trapezoid_table_41pt.dat"))
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "iS3D Sampling Seed : 17" in r.stdout and "Sampling particles with Chapman Enskog df..." in r.stdout
    n_events = int(r.stdout.split(" event(s)")[0].split("Sampling ")[-1])
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    gla = inputs.feqmod_tables(0.15)
    sp = inputs.species(ids)
    if oversample:
        avg = [float(v) for v in open(os.path.join(root, "average_thermodynamic_quantities.dat")).read().split()]
        Ntot, _ = oracle.total_yield(parsed, sp, fx["df"], gla, avg, dict(dimension=dim, df_mode=2), y_cut=0.7)
        max_samples = int(api.param_get(os.path.join(root, "iS3D_parameters.dat"), "max_num_samples"))
        assert n_events == max(1, min(int(np.ceil(300.0 / float(np.float32(abs(Ntot))))), max_samples)) and n_events > 1
        assert abs(float(r.stdout.split("Total particle yield: ")[1].split()[0]) / Ntot - 1) < 1e-6
    else:
        assert n_events == 1
    ref, _ = oracle.sample_particles(parsed, sp, fx["df"], gla, dict(dimension=dim, df_mode=2), n_events=n_events, seed=17, y_cut=0.7)
    lines = [ln for ln in open(os.path.join(root, "results", "particle_list_osc.dat")).read().split("\n") if ln]
    headers = [int(ln[2:]) for ln in lines if ln.startswith("#")]
    rows = np.array([[float(v) for v in ln.split(" ")] for ln in lines if not ln.startswith("#")]).reshape(-1, 9)
    assert len(rows) > 10
    assert sum(headers) == len(rows) == len(ref["E"]) and len(headers) == len(np.unique(ref["event"]))
    assert np.array_equal(rows[:, 0].astype(np.int64), np.array(ids)[ref["species"]])
    for col, f in enumerate(["t", "x", "y", "z", "E", "px", "py", "pz"], start=1):
        assert np.allclose(rows[:, col], ref[f], rtol=1e-11, atol=1e-13), f


def test_cli_sampler_with_baryon(tmp_path, fx):
    """operation = 2 with include_baryon = 1: 25-column surface, full (T, mu_B) coefficient files; the list equals the oracle's."""
    ids = [211, 2212, -2212]
    cells = synth.synth_surface(4000, 3, seed=95, baryon=True)
    o = dict(dimension=3, df_mode=2, include_baryon=1, include_baryondiff_deltaf=1)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(o, operation=2, oversample=0, sampler_seed=19))
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    ref, _ = oracle.sample_particles(parsed, inputs.species(ids), inputs.df_tables_full(), inputs.feqmod_tables(0.15), o, n_events=1, seed=19, y_cut=0.7)
    lines = [ln for ln in open(os.path.join(root, "results", "particle_list_osc.dat")).read().split("\n") if ln]
    rows = np.array([[float(v) for v in ln.split(" ")] for ln in lines if not ln.startswith("#")]).reshape(-1, 9)
    assert len(rows) == len(ref["E"]) > 10
    assert np.array_equal(rows[:, 0].astype(np.int64), np.array(ids)[ref["species"]])
    for col, f in enumerate(["t", "x", "y", "z", "E", "px", "py", "pz"], start=1):
        assert np.allclose(rows[:, col], ref[f], rtol=1e-11, atol=1e-13), f


@pytest.mark.parametrize("operation", [2, 1])
def test_embedding_class(tmp_path, fx, operation):
    """include/iS3D_amd.hpp: the reference's embedding API (class IS3D: read_fo_surf_from_memory + run_particlization(0) +
    final_particles_, iS3D.h:19-96) from a C++ host compiled with g++ against the C ABI; surface handed over in memory, in
    GeV units; the sampled list equals the oracle's, the spectrum the oracle's spectrum."""
    ids = [211, 321, 2212, -2212]
    cells = synth.synth_surface(3000, 3, seed=95)
    root = refformat.make_run_dir(str(tmp_path / "run"), synth.synth_surface(2, 3, seed=1), ids,
                                  dict(operation=operation, dimension=3, df_mode=2, sampler_seed=23))
    os.remove(os.path.join(root, "input", "surface.dat"))          # the in-memory path must not need it
    exe = str(tmp_path / "embed_main")
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "embed_main.cpp"), "-o", exe,
                           "-L", os.path.dirname(api.LIB_PATH), "-lis3d_amd", "-Wl,-rpath," + os.path.dirname(api.LIB_PATH)])
    cols = ["tau", "x", "y", "eta", "dat", "dax", "day", "dan", "E", "T", "P", "ux", "uy", "un", "pixx", "pixy", "pixn", "piyy", "piyn", None, "bulkPi"]
    tab = np.stack([cells[c] if c else np.full(3000, 7.0) for c in cols], axis=1)      # pinn: junk, must be ignored
    np.savetxt(str(tmp_path / "surf21.txt"), tab, fmt="%.17g")
    r = subprocess.run([exe, str(tmp_path / "surf21.txt")], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    assert "Reading in freezeout surface from memory" in r.stdout
    sp = inputs.species(ids)
    if operation == 2:
        ref, _ = oracle.sample_particles(cells, sp, fx["df"], inputs.feqmod_tables(0.15), dict(dimension=3, df_mode=2), n_events=1, seed=23, y_cut=0.7)
        rows = np.array([[float(v) for v in ln.split()[1:]] for ln in r.stdout.split("\n") if ln.startswith("P ")]).reshape(-1, 14)
        assert len(rows) == len(ref["E"]) > 5 and "EVENTS 1 SPECTRUM 0" in r.stdout
        assert np.array_equal(rows[:, 1].astype(int), ref["species"]) and np.array_equal(rows[:, 2].astype(int), np.array(ids)[ref["species"]])
        assert np.array_equal(rows[:, 3], sp["mass"][ref["species"]])
        for col, f in enumerate(["tau", "x", "y", "eta", "t", "z", "E", "px", "py", "pz"], start=4):
            assert np.allclose(rows[:, col], ref[f], rtol=1e-11, atol=1e-13), f
        assert os.path.getsize(os.path.join(root, "results", "particle_list_osc.dat")) > 0
    else:
        want = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], dict(dimension=3, df_mode=2))
        line = [ln for ln in r.stdout.split("\n") if ln.startswith("SUM ")][0].split()
        assert "SPECTRUM %d" % want.size in r.stdout
        assert abs(float(line[1]) / want.sum() - 1) < 1e-9 and abs(float(line[3]) / want[0] - 1) < 1e-9


def test_cli_shipped_default_shape(tmp_path, fx):
    """The shape of the reference's shipped iS3D_parameters.dat: operation = 2, hrg_eos = 2 (PDG/pdg_smash.dat, deltaf_coefficients/vh/smash/),
    dimension = 2, df_mode = 4 (Jonah), fast = 1, set_FO_temperature = 1, test_sampler = 1 -> binned test distributions instead of the particle list."""
    ids = [211, 321, 2212]
    cells = synth.synth_surface(4000, 2, seed=97)
    root = refformat.make_run_dir(str(tmp_path), cells, ids, dict(operation=2, dimension=2, df_mode=4, fast=1, test_sampler=1, oversample=1,
                                                                  min_num_hadrons=20000, sampler_seed=5, hrg_eos=2))
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Sampling particles with Jonah's modified distribution..." in r.stdout and "Using fast mode: (Tavg, muBavg) = (0.151000" in r.stdout
    assert not os.path.exists(os.path.join(root, "results", "particle_list_osc.dat"))
    n_events = int(r.stdout.split(" event(s)")[0].split("Sampling ")[-1])
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat"))
    T_avg = float(open(os.path.join(root, "average_thermodynamic_quantities.dat")).read().split()[0])
    fq = inputs.feqmod_tables(T_avg)
    ref, _ = oracle.sample_particles(parsed, inputs.species(ids), fx["df"], fq, dict(dimension=2, df_mode=4), n_events=n_events, seed=5,
                                     y_cut=0.7, fq=fq, fast=1, T_avg=fq["T_avg"], T_avg_switch=0.151)
    assert len(ref["E"]) > 5000
    got = np.loadtxt(os.path.join(root, "results", "dN_dy", "dN_dy_211_test.dat"))
    h, _ = np.histogram(ref["rapidity"][ref["species"] == 0], bins=14, range=(-0.7, 0.7))
    assert np.allclose(got[:, 1], h / (0.1 * n_events), rtol=6e-6)
    got = np.loadtxt(os.path.join(root, "results", "momentum_distribution", "dN_2pipTdpTdy_2212_test.dat"))
    sel = ref["species"] == 2
    h, e = np.histogram(np.hypot(ref["px"], ref["py"])[sel], bins=30, range=(0, 3))
    assert np.allclose(got[:, 1], h / (2 * np.pi * 1.4 * 0.1 * 0.5 * (e[1:] + e[:-1]) * n_events), rtol=6e-6)
    ylist = [int(v) for v in open(os.path.join(root, "results", "yield_list.dat")).read().split()[3:]]
    assert len(ylist) == n_events and sum(ylist) == len(ref["E"])


@pytest.mark.parametrize("dim", [3, 2])
def test_cli_mode2_anisotropic_hydro(tmp_path, fx, dim):
    """mode = 2, df_mode = 4, operation = 1: a read_surf_VAH_PLMatch surface and the deltaf_coefficients/vah tables in, the three
    result files out == oracle reader -> oracle coefficients -> oracle VAH kernel (BASELINE config 5 from files).  The reference's own
    binary writes zeros here (its call site is commented out, emissionfunction.cpp:1650-1654)."""
    ids = [211, 321, 2212]
    cells = synth.synth_vah_surface(19 if dim == 3 else 5, dim, seed=60 + dim)
    vh = synth.synth_surface(3, dim)            # make_run_dir wants a mode-1 surface to write first; it is replaced below
    root = refformat.make_run_dir(str(tmp_path), vh, ids, dict(dimension=dim, df_mode=4, mode=2))
    synth.write_surface_vah_dat(os.path.join(root, "input", "surface.dat"), cells)
    tab = inputs.vah_df_tables()
    refformat.write_vah_df_tables(os.path.join(root, "deltaf_coefficients", "vah"), tab)
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "vahydro" in r.stdout and "Total number of freezeout cells: %d" % len(cells["tau"]) in r.stdout
    rc = oracle.read_surf_VAH_PLMatch(os.path.join(root, "input", "surface.dat"))
    coef, found = oracle.vah_coefficients(tab, rc["Lambda"], rc["aL"])
    assert found.all()
    sp = inputs.species(ids)
    ref = oracle.dN_pTdpTdphidy_vah(dict(rc, **coef), sp, fx["grid"], dict(dimension=dim))
    ny = 21 if dim == 3 else 1
    ref4 = ref.reshape(ny, 24, 32, 3)
    allsp = read_spectra_file(os.path.join(root, "results", "dN_pTdpTdphidy.dat"))
    want = np.transpose(ref4, (3, 0, 1, 2)).reshape(-1)
    assert relerr(allsp[:, 3], want, floor=1e-250) < 2e-8
    assert os.path.exists(os.path.join(root, "results", "dN_dy_2212.dat")) and os.path.exists(os.path.join(root, "results", "vn_continuous", "vn_321.dat"))
    # refusals: the sampler (an empty stub in the reference), another df_mode, a cell outside the tables
    for bad, msg in ((dict(operation=2), "stub"), (dict(df_mode=1), "df_mode = 4")):
        root2 = refformat.make_run_dir(str(tmp_path / ("bad%d" % len(msg))), vh, ids, dict(dict(dimension=dim, df_mode=4, mode=2), **bad))
        r = subprocess.run([api.CLI_PATH], cwd=root2, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and msg in r.stderr
    hot = dict(cells, T=cells["T"] * 2.5)       # Lambda ~ T beyond the last node of the tables
    synth.write_surface_vah_dat(os.path.join(root, "input", "surface.dat"), hot)
    r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "beyond the last node" in r.stderr
