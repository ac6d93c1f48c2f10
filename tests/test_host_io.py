"""CPU: the C++ readers/writers behind the C ABI (is3d_amd/csrc/host_io.cpp) against independent Python
parsing of the same files -- formats and quirks of the reference."""
import os

import numpy as np
import pytest

import refformat
from conftest import REFERENCE
from is3d_amd import api, inputs, synth


def test_parameter_reader_semantics(tmp_path):
    """ParameterReader.cpp:38-155: '#' comments, blanks removed everywhere, case-insensitive names, last one wins."""
    p = tmp_path / "iS3D_parameters.dat"
    p.write_text("operation = 2  # first\n\n   \t\nDF_Mode\t=\t1.0e0 # case\nT_switch = 0.151\n d f _ m o d e = 2\n# only a comment\nx = -3.5e-2#c\noperation = 1\n")
    assert api.param_get(str(p), "operation") == 1.0
    assert api.param_get(str(p), "df_mode") == 2.0          # blanks inside the name are stripped too (arsenal.cpp:552-565)
    assert api.param_get(str(p), "T_SWITCH") == 0.151
    assert api.param_get(str(p), "x") == -3.5e-2
    with pytest.raises(api.Is3dError) as e:
        api.param_get(str(p), "dimension")                  # reference: exit(-1)
    assert e.value.code == api.IS3D_EINVAL
    with pytest.raises(api.Is3dError) as e:
        api.param_get(str(tmp_path / "nope.dat"), "x")
    assert e.value.code == api.IS3D_EIO
    (tmp_path / "bad.dat").write_text("operation 1\n")
    with pytest.raises(api.Is3dError):
        api.param_get(str(tmp_path / "bad.dat"), "operation")


def test_table_reader_drops_unterminated_last_line(tmp_path):
    """readBlockData (arsenal.cpp:406-453): rows = newline-terminated lines; columns fixed by the first line."""
    p = tmp_path / "t.dat"
    p.write_text("\t0.5\t0.25\n 1.5 0.75 9\n2.5\t1.25\n\t")
    t = api.table_read(str(p))
    assert t.shape == (3, 2) and np.array_equal(t, [[0.5, 0.25], [1.5, 0.75], [2.5, 1.25]])
    p.write_text("1 2\n3 4\n5 6")          # no final newline: the last line is lost, as in the reference
    assert api.table_read(str(p)).shape == (2, 2)
    p.write_text("1 2\n3\n")
    with pytest.raises(api.Is3dError):
        api.table_read(str(p))
    p.write_text("\n1 2\n")
    with pytest.raises(api.Is3dError):
        api.table_read(str(p))


def test_number_parsing_is_correctly_rounded(tmp_path):
    """The table parser (threads + Clinger fast path in front of strtod) returns what strtod / operator>> return: the correctly
    rounded double, for every decimal format a hydro code prints and for the edge cases that must fall through to strtod."""
    import random
    random.seed(3)
    rows, vals = [], []
    for i in range(30000):
        kind = random.randrange(6)
        if kind == 0:
            s = "%.6g" % random.uniform(-1e3, 1e3)
        elif kind == 1:
            s = "%.9e" % (random.uniform(-1, 1) * 10 ** random.randint(-30, 30))
        elif kind == 2:
            s = "%d" % random.randint(-10 ** 9, 10 ** 9)
        elif kind == 3:
            s = "%.15g" % random.uniform(-1, 1)
        elif kind == 4:
            s = "%.17g" % random.uniform(-1, 1)
        else:
            s = random.choice(["0", "-0.0", "1e22", "1e-22", "1e23", "123456789012345678", "0.000001", "9007199254740993", "9007199254740992",
                               "1.e5", "+.5", "5.", "1e400", "4.9e-324", "2.2250738585072011e-308", "0.1e-21", "12345678901234567890123"])
        rows.append(s + "\t" + s)
        vals.append(float(s))
    p = str(tmp_path / "numbers.dat")
    open(p, "w").write("\n".join(rows) + "\n")
    t = api.table_read(p)
    assert t.shape == (30000, 2) and np.array_equal(t[:, 0], np.array(vals)) and np.array_equal(t[:, 1], t[:, 0])
    assert np.array_equal(np.signbit(t[:, 0]), np.signbit(np.array(vals)))


def test_grid_tables_round_trip(tmp_path):
    g = inputs.grid()
    refformat.write_table(str(tmp_path / "phi.dat"), g["phi"], g["phi_w"], leading_tab=True, dangling_fragment=True)
    t = api.table_read(str(tmp_path / "phi.dat"))
    assert t.shape == (24, 2) and np.array_equal(t[:, 0], g["phi"]) and np.array_equal(t[:, 1], g["phi_w"])


def test_surface_reader_mode1(tmp_path):
    cells = synth.synth_surface(37, 3, seed=11)
    path = str(tmp_path / "surface.dat")
    synth.write_surface_dat(path, cells)
    got, avg = api.surface_read_vh(path)
    ref = refformat.read_surface_like_reference(path)
    for k in synth.CELL_FIELDS:
        assert np.array_equal(got[k], ref[k]), k
        assert np.allclose(got[k], cells[k], rtol=4e-16, atol=0), k   # text round trip: (v / hbarc) * hbarc
    # averages of readindata.cpp:422-466
    ut = np.sqrt(1 + ref["ux"] ** 2 + ref["uy"] ** 2 + ref["tau"] ** 2 * ref["un"] ** 2)
    uds = ut * ref["dat"] + ref["ux"] * ref["dax"] + ref["uy"] * ref["day"] + ref["un"] * ref["dan"]
    dsds = ref["dat"] ** 2 - ref["dax"] ** 2 - ref["day"] ** 2 - ref["dan"] ** 2 / ref["tau"] ** 2
    mag = np.abs(uds) + np.sqrt(np.abs(uds * uds - dsds))
    assert np.allclose(avg[:3], [np.sum(ref[k] * mag) / np.sum(mag) for k in ("T", "E", "P")], rtol=1e-13)
    assert avg[3] == 0.0 and avg[4] == 0.0


def test_surface_reader_toy_file_and_errors(tmp_path):
    p = tmp_path / "surface.dat"
    p.write_text("0.5 0 0 0 1000.0 0 0 0 0 0 0 1.839  0.786  0.270 0 0 0 0 0 0\n")   # the reference's input/surface.dat
    got, _ = api.surface_read_vh(str(p))
    assert len(got["tau"]) == 1 and got["dat"][0] == 1000.0 and got["T"][0] == 0.786 * 0.197327053
    p.write_text("0.5 0 0 0 1000.0 0 0 0 0 0 0 1.839  0.786  0.270 0 0 0 0 0 0")     # unterminated -> zero cells
    got, _ = api.surface_read_vh(str(p))
    assert len(got["tau"]) == 0
    p.write_text("0.5 0 0 0 1000.0 0 0 0\n1 2 3 4 5 6 7 8\n")                          # token stream runs dry
    with pytest.raises(api.Is3dError):
        api.surface_read_vh(str(p))


def test_surface_reader_other_formats(tmp_path):
    """Modes 0 (old gpu-vh), 4 / 6 (MUSIC old / new), 7 (hic-eventgen): same cells written in each format parse to the
    same kernel-convention arrays as the mode-1 file (tau Jacobians, hbar*c, p = T s - e, u = gamma v), and the
    format-specific overrides hold (eta -> 0 in 4/6/7, dsigma_eta -> 0, mu_B always carried by 4/6/7)."""
    path = str(tmp_path / "surface.dat")
    c3 = synth.synth_surface(29, 3, seed=21, baryon=True)
    c2 = synth.synth_surface(17, 2, seed=22, baryon=True)
    keys = [k for k in synth.CELL_FIELDS if k not in ("eta", "dan")]

    refformat.write_surface_mode(path, c3, 0, include_baryon=1, include_baryondiff=1)
    got, avg = api.surface_read(path, 0, 1, 1, 3)
    for k in synth.CELL_FIELDS + synth.BARYON_FIELDS:
        assert np.allclose(got[k], c3[k], rtol=5e-16, atol=0), k
    refformat.write_surface_mode(path, c3, 0)
    got0, _ = api.surface_read(path, 0, 0, 0, 3)
    assert np.array_equal(got0["T"], got["T"]) and not got0["muB"].any()
    with pytest.raises(api.Is3dError):      # old format: dsigma_eta != 0 in a 2+1D run is fatal (readindata.cpp:180-184)
        api.surface_read(path, 0, 0, 0, 2)

    for mode in (4, 6):
        for cells, dim in ((c3, 3), (c2, 2)):
            refformat.write_surface_mode(path, cells, mode)
            got, avg = api.surface_read(path, mode, 1, 0, dim)
            for k in keys:
                assert np.allclose(got[k], cells[k], rtol=4e-15, atol=0), (mode, k)
            assert not got["eta"].any()                                    # both MUSIC readers put the cell at eta = 0
            if mode == 6 or dim == 2:
                assert not got["dan"].any()
            else:
                assert np.allclose(got["dan"], cells["dan"], rtol=4e-15)
            assert np.allclose(got["muB"], cells["muB"], rtol=4e-16) and not got["nB"].any() and not got["Vx"].any()
            assert 0.05 < avg[3] < 0.4 and avg[4] == 0.0

    refformat.write_surface_mode(path, c2, 7)
    got, avg = api.surface_read(path, 7, 0, 0, 2)
    for k in keys:
        assert np.allclose(got[k], c2[k], rtol=4e-15, atol=1e-18), k
    assert not got["eta"].any() and not got["dan"].any() and not got["un"].any()
    # mode 5 (gpu-vh + thermal vorticity, readindata.cpp:470-551): the viscous-hydro kernels run on it, the six vorticity columns are not theirs
    for ib, idf in ((1, 1), (1, 0), (0, 0)):
        refformat.write_surface_mode(path, c3, 5, include_baryon=ib, include_baryondiff=idf)
        got5, avg5 = api.surface_read(path, 5, ib, idf, 3)
        for k in synth.CELL_FIELDS + (synth.BARYON_FIELDS if idf else []):
            assert np.allclose(got5[k], c3[k], rtol=5e-16, atol=0), (k, ib, idf)
        assert (np.allclose(got5["muB"], c3["muB"], rtol=5e-16) if ib else not got5["muB"].any())
    synth.write_surface_dat(path, {k: v for k, v in c3.items() if k not in ("muB", "nB", "Vx", "Vy", "Vn")})   # the 20-column mode-1 file
    m1, avg1 = api.surface_read(path, 1, 0, 0, 3)
    assert all(np.array_equal(m1[k], got5[k]) for k in synth.CELL_FIELDS) and np.allclose(avg1, avg5, rtol=1e-14)
    with pytest.raises(api.Is3dError) as e:
        api.surface_read(path, 3, 0, 0, 3)
    assert e.value.code == api.IS3D_EINVAL
    # mode 1 through the switch == the dedicated entry
    synth.write_surface_dat(path, c3)
    a, _ = api.surface_read(path, 1, 1, 1, 3)
    b, _ = api.surface_read_vh(path, 1, 1, 3)
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_pdg_reader(tmp_path):
    fx = inputs.load_fixture()
    particles = [r for r in fx["pdg_urqmd"] if r[3] >= 0]
    path = str(tmp_path / "pdg.dat")
    refformat.write_pdg(path, particles)
    got = api.pdg_read(path)
    ref = np.array(fx["pdg_urqmd"], dtype=np.float64)
    assert len(got["mc_id"]) == 327
    assert np.array_equal(got["mc_id"], ref[:, 0].astype(np.int64))
    for i, k in enumerate(["mass", "gspin", "baryon", "sign"]):
        assert np.array_equal(got[k], ref[:, i + 1]), k
    # antibaryon follows its baryon; sign = -1 for even baryon number, also for baryon = 2 (readindata.cpp:1544)
    i = list(got["mc_id"]).index(2212)
    assert got["mc_id"][i + 1] == -2212 and got["baryon"][i + 1] == -1 and got["sign"][i + 1] == 1
    # a file without trailing whitespace loses its last entry (the reference's `count - 1`, :1540)
    refformat.write_pdg(path, particles[:3], trailing_blank=False)
    with open(path, "rb") as f:
        raw = f.read().rstrip()
    with open(path, "wb") as f:
        f.write(raw)
    assert len(api.pdg_read(path)["mc_id"]) == 2


def test_df_table_reader(tmp_path):
    df = inputs.df_tables()
    path = str(tmp_path / "c0.dat")
    refformat.write_df_table(path, df["T"], df["c0"], "c0_T4 [fm^3/GeV^3 * GeV^4]")
    T, v = api.df_table_read(path)
    assert np.array_equal(T, df["T"]) and np.array_equal(v, df["c0"])   # only the mu_B = 0 block is kept


def test_gla_reader(tmp_path):
    """Gauss_Laguerre::load_roots_and_weights (readindata.cpp:24-53): header 'n_alpha n_points', rows 'alpha root weight'."""
    import refformat
    root = refformat.make_run_dir(str(tmp_path), synth.synth_surface(2, 3, seed=1), [211], dict())
    r, w = api.gla_read(os.path.join(root, "tables", "gla_roots_weights_32_points.txt"))
    g = inputs.load_fixture()["gla_32"]
    assert r.shape == (4, 32) and np.array_equal(r[1], g["root1"]) and np.array_equal(w[1], g["weight1"]) and np.array_equal(r[3], g["root3"])
    assert np.array_equal(r[2], g["root2"]) and np.array_equal(w[2], g["weight2"])
    # generalized Gauss-Laguerre: sum_k w_k = Gamma(alpha + 1), sum_k w_k x_k = Gamma(alpha + 2)
    assert abs(w[1].sum() - 1.0) < 1e-13 and abs(w[2].sum() - 2.0) < 1e-13 and abs((w[1] * r[1]).sum() - 2.0) < 1e-12
    with pytest.raises(api.Is3dError):
        api.gla_read(os.path.join(root, "tables", "missing.txt"))


def test_oscar_writer(tmp_path):
    """write_particle_list_OSC (emissionfunction.cpp:863-901): "# N" per non-empty event, rows "mcid t x y z E px py pz" with
    setprecision(16) scientific; empty events are not written."""
    p = np.zeros(5, dtype=api.PARTICLE_DTYPE)
    p["event"] = [0, 0, 2, 2, 2]
    p["species"] = [0, 1, 1, 0, 0]
    for i, f in enumerate(["t", "x", "y", "z", "E", "px", "py", "pz"]):
        p[f] = np.arange(5) * 0.37 + i + 1.0 / 3.0
    path = str(tmp_path / "particle_list_osc.dat")
    api.write_particle_list_osc(path, 4, p, [211, -2212])
    lines = open(path).read().split("\n")
    assert lines[0] == "# 2" and lines[3] == "# 3" and lines[-1] == "" and len(lines) == 8
    row = lines[1].split(" ")
    assert row[0] == "211" and len(row) == 9 and row[1] == "%.16e" % p["t"][0] and row[8] == "%.16e" % p["pz"][0]
    assert lines[4].split(" ")[0] == "-2212"
    back = np.array([[float(v) for v in ln.split(" ")[1:]] for ln in lines if ln and not ln.startswith("#")])
    assert np.array_equal(back[:, 4], p["E"]) and np.array_equal(back[:, 0], p["t"])
    q = p.copy()
    q["event"] = [0, 2, 0, 2, 2]           # not ordered by event
    with pytest.raises(api.Is3dError):
        api.write_particle_list_osc(path, 4, q, [211, -2212])


def test_sampler_test_writers(tmp_path):
    """test_sampler = 1 outputs (sampling_kernels.cpp:31-152, emissionfunction.cpp:903-1257) against numpy histograms."""
    rng = np.random.default_rng(5)
    n = 4000
    p = np.zeros(n, dtype=api.PARTICLE_DTYPE)
    p["event"] = np.sort(rng.integers(0, 3, n))
    p["species"] = rng.integers(0, 2, n)
    p["px"], p["py"], p["pz"] = rng.normal(0, 0.6, n), rng.normal(0, 0.6, n), rng.normal(0, 2.0, n)
    m = np.array([0.138, 0.938])[p["species"]]
    p["E"] = np.sqrt(m ** 2 + p["px"] ** 2 + p["py"] ** 2 + p["pz"] ** 2)
    p["eta"], p["tau"] = rng.normal(0, 2.0, n), rng.uniform(0.5, 11.0, n)
    p["x"], p["y"] = rng.normal(0, 3.0, n), rng.normal(0, 3.0, n)
    root = str(tmp_path)
    for d in ("dN_dy", "dN_deta", "momentum_distribution", "vn", "spacetime_distribution"):
        os.makedirs(os.path.join(root, d))
    bins = dict(y_cut=1.5, eta_cut=6.0, pT_lower_cut=0.0, pT_upper_cut=3.0, tau_min=0.0, tau_max=12.0, r_min=0.0, r_max=10.0,
                y_bins=10, eta_bins=24, pT_bins=15, tau_bins=12, r_bins=10)
    api.write_sampler_tests(root, bins, 3, [211, 2212], p, mean_yield=12.5)
    yp = 0.5 * np.log((p["E"] + p["pz"]) / (p["E"] - p["pz"]))
    sel = p["species"] == 1
    got = np.loadtxt(os.path.join(root, "dN_dy", "dN_dy_2212_test.dat"))
    h, edges = np.histogram(yp[sel], bins=10, range=(-1.5, 1.5))
    assert np.allclose(got[:, 0], 0.5 * (edges[1:] + edges[:-1]), atol=1e-6) and np.allclose(got[:, 1], h / (0.3 * 3), rtol=6e-6)
    avg = float(open(os.path.join(root, "dN_dy", "dN_dy_2212_average_test.dat")).read())
    assert abs(avg - h.sum() / (3.0 * 3)) < 1e-5 * avg
    mid = np.abs(yp) <= 1.5
    pT = np.hypot(p["px"], p["py"])
    got = np.loadtxt(os.path.join(root, "momentum_distribution", "dN_2pipTdpTdy_211_test.dat"))
    h, edges = np.histogram(pT[mid & ~sel], bins=15, range=(0.0, 3.0))
    pm = 0.5 * (edges[1:] + edges[:-1])
    assert np.allclose(got[:, 1], h / (2 * np.pi * 3.0 * 0.2 * pm * 3), rtol=6e-6)
    got = np.loadtxt(os.path.join(root, "vn", "vn_211_test.dat"))
    phi = np.arctan2(p["py"], p["px"])
    ib = np.floor(pT / 0.2).astype(int)
    k = (mid & ~sel) & (ib == 3)
    assert got.shape == (15, 8) and abs(got[3, 2] - abs(np.exp(2j * phi[k]).sum()) / k.sum()) < 2e-6
    got = np.loadtxt(os.path.join(root, "dN_deta", "dN_deta_211_test.dat"))
    h, _ = np.histogram(p["eta"][~sel], bins=24, range=(-6, 6))
    assert np.allclose(got[:, 1], h / (0.5 * 3), rtol=6e-6)
    got = np.loadtxt(os.path.join(root, "spacetime_distribution", "dN_taudtaudy_sampled_2212_test.dat"))
    h, edges = np.histogram(p["tau"][mid & sel], bins=12, range=(0, 12))
    assert np.allclose(got[:, 1], h / (0.5 * (edges[1:] + edges[:-1]) * 1.0 * 3 * 3.0), rtol=6e-6)
    got = np.loadtxt(os.path.join(root, "spacetime_distribution", "dN_twopirdrdy_sampled_2212_test.dat"))
    h, edges = np.histogram(np.hypot(p["x"], p["y"])[mid & sel], bins=10, range=(0, 10))
    assert np.allclose(got[:, 1], h / (2 * np.pi * 0.5 * (edges[1:] + edges[:-1]) * 1.0 * 3 * 3.0), rtol=6e-6)
    lines = open(os.path.join(root, "yield_list.dat")).read().split()
    assert [int(v) for v in lines[3:]] == [int((p["event"] == e).sum()) for e in range(3)]
    assert float(open(os.path.join(root, "mean_yield.dat")).read()) == 12.5


def test_writers_format(tmp_path):
    """emissionfunction.cpp:381-450, :729-772, :1053-1136."""
    g = inputs.grid()
    npart, npT, nphi, ny = 2, 32, 24, 21
    rng = np.random.default_rng(0)
    dN = rng.random(npart * npT * nphi * ny) * 10.0 ** rng.integers(-30, 3, npart * npT * nphi * ny)
    os.makedirs(tmp_path / "results" / "vn_continuous")
    res = str(tmp_path / "results")
    api.write_results(res, 3, [211, -2212], g["pT"], g["pT_w"], g["phi"], g["phi_w"], g["y"], dN)
    d4 = dN.reshape(ny, nphi, npT, npart)
    lines = open(os.path.join(res, "dN_pTdpTdphidy.dat")).read().split("\n")
    assert len(lines) == npart * ny * nphi * (npT + 1) + 1
    assert lines[0] == "%.8e\t%.8e\t%.8e\t%.8e" % (g["y"][0], g["phi"][0], g["pT"][0], d4[0, 0, 0, 0])
    assert lines[npT] == ""                                                      # blank line after each phi block
    assert lines[npT + 1].split("\t")[1] == "%.8e" % g["phi"][1]
    per = open(os.path.join(res, "dN_pTdpTdphidy_-2212.dat")).read().split("\n")
    assert per[0] == "y\tphip\tpT\tdN_pTdpTdphidy"
    assert per[1] == "%.8e\t%.8e\t%.8e\t%.8e" % (g["y"][0], g["phi"][0], g["pT"][0], d4[0, 0, 0, 1])
    # dN/dy: default float format with 8 significant digits (no `scientific`)
    dy = [ln.split("\t") for ln in open(os.path.join(res, "dN_dy_211.dat")).read().strip().split("\n")]
    assert len(dy) == ny
    want = np.einsum("j,i,kji->k", g["phi_w"], g["pT_w"], d4[:, :, :, 0])
    assert dy[3][0] == "%5s" % ("%.8g" % g["y"][3])                              # setw(5) pads the y column
    assert np.allclose([float(r[1]) for r in dy], want, rtol=2e-8)
    # v_n: |sum w e^{ik phi} dN| / sum w dN, k = 1..7; rows per pT, blank line per y
    vn = open(os.path.join(res, "vn_continuous", "vn_211.dat")).read().split("\n")
    assert len(vn) == ny * (npT + 1) + 1
    row = vn[5].split("\t")
    assert len(row) == 9
    num = np.sum(np.exp(2j * g["phi"]) * g["phi_w"] * d4[0, :, 5, 0])
    assert abs(float(row[3]) - abs(num) / np.sum(g["phi_w"] * d4[0, :, 5, 0])) < 1e-7
    # append mode: a second call doubles the file (emissionfunction.cpp:395)
    api.write_results(res, 3, [211, -2212], g["pT"], g["pT_w"], g["phi"], g["phi_w"], g["y"], dN)
    assert len(open(os.path.join(res, "dN_pTdpTdphidy.dat")).read().split("\n")) == 2 * (len(lines) - 1) + 1
    # 2+1D: y printed as 0, one rapidity slice
    os.makedirs(tmp_path / "r2" / "vn_continuous")
    api.write_results(str(tmp_path / "r2"), 2, [211], g["pT"], g["pT_w"], g["phi"], g["phi_w"], g["y"], dN[:npT * nphi])
    l2 = open(tmp_path / "r2" / "dN_pTdpTdphidy.dat").read().split("\n")
    assert len(l2) == nphi * (npT + 1) + 1 and l2[0].startswith("0.00000000e+00\t")
    with pytest.raises(api.Is3dError):
        api.write_results(str(tmp_path / "missing_dir"), 2, [211], g["pT"], g["pT_w"], g["phi"], g["phi_w"], g["y"], dN[:npT * nphi])


def test_writer_is_byte_for_byte_the_iostream_text(tmp_path):
    """The spectra writer formats y / phi / pT once per row position and the value with std::to_chars(scientific, 8): every byte must be what
    the reference's `f << scientific << setprecision(8) << y << "\\t" << ...` prints (= printf "%.8e"), for ordinary values, zeros of both
    signs, subnormals, 9-digit rounding carries and non-finite values, over more species than one 64-species group of the writer's threads."""
    g = inputs.grid()
    npart, npT, nphi, ny = 70, 5, 3, 4
    rng = np.random.default_rng(7)
    n = npart * npT * nphi * ny
    dN = rng.random(n) * 10.0 ** rng.integers(-320, 300, n) * rng.choice([1.0, -1.0], n)
    dN[:12] = [0.0, -0.0, 5e-324, 2.2250738585072014e-308, 9.999999995e5, 9.9999999949e5, 1.0, 1.7976931348623157e308, np.inf, -np.inf, np.nan, 1.234567885e-7]
    os.makedirs(tmp_path / "results" / "vn_continuous")
    res = str(tmp_path / "results")
    ids = [100 + i for i in range(npart)]
    pT, phi, y = g["pT"][:npT], g["phi"][:nphi], g["y"][:ny]
    api.write_results(res, 3, ids, pT, g["pT_w"][:npT], phi, g["phi_w"][:nphi], y, dN)
    d4 = dN.reshape(ny, nphi, npT, npart)

    def fmt(v):
        return "%.8e" % v

    want_all = []
    for ip in range(npart):
        blk = []
        for iy in range(ny):
            for j in range(nphi):
                for i in range(npT):
                    blk.append("%s\t%s\t%s\t%s\n" % (fmt(y[iy]), fmt(phi[j]), fmt(pT[i]), fmt(d4[iy, j, i, ip])))
                blk.append("\n")
        blk = "".join(blk)
        want_all.append(blk)
        assert open(os.path.join(res, "dN_pTdpTdphidy_%d.dat" % ids[ip])).read() == "y\tphip\tpT\tdN_pTdpTdphidy\n" + blk, ip
    assert open(os.path.join(res, "dN_pTdpTdphidy.dat")).read() == "".join(want_all)


@pytest.mark.reference
def test_readers_on_the_reference_files():
    """The shipped data files parse to the fixture the GPU box uses (is3d_amd/data/inputs_urqmd.json)."""
    g = inputs.grid()
    for key, rel in [("pT", "tables/pT_gauss_legendre_table.dat"), ("phi", "tables/phi_gauss_legendre_table.dat"),
                     ("y", "tables/y_trapezoid_table_21pt.dat"), ("eta", "tables/eta/eta_trapezoid_table_241pt.dat")]:
        t = api.table_read(os.path.join(REFERENCE, rel))
        assert np.array_equal(t[:, 0], g[key]) and np.array_equal(t[:, 1], g[key + "_w"])
    df = inputs.df_tables()
    for name in ("c0", "c2", "F", "betabulk", "betapi"):
        T, v = api.df_table_read(os.path.join(REFERENCE, "deltaf_coefficients/vh/urqmd", name + ".dat"))
        assert np.array_equal(T, df["T"]) and np.array_equal(v, df[name])
    pdg = api.pdg_read(os.path.join(REFERENCE, "PDG/pdg-urqmd_v3.3+.dat"))
    ref = np.array(inputs.load_fixture()["pdg_urqmd"], dtype=np.float64)
    assert np.array_equal(pdg["mc_id"], ref[:, 0].astype(np.int64)) and np.array_equal(pdg["mass"], ref[:, 1])
    chosen = api.table_read(os.path.join(REFERENCE, "PDG/chosen_particles_urqmd_v3.3+.dat"))
    assert chosen.shape == (305, 1) and np.array_equal(chosen[:, 0], inputs.load_fixture()["chosen_urqmd"])
    surf, _ = api.surface_read_vh(os.path.join(REFERENCE, "input/surface.dat"))
    assert len(surf["tau"]) == 1 and surf["P"][0] == 0.270 * 0.197327053
    assert api.param_get(os.path.join(REFERENCE, "iS3D_parameters.dat"), "df_mode") == 4.0


def test_pdg_box_reader(tmp_path):
    """is3d_pdg_read_box (hrg_eos = 3): read_resonances_smash_box with read_mcid (readindata.cpp:1571-1685, :1201-1418) -- comment and blank lines,
    a comment behind the ids, up to four ids per line, UTF-8 names; degeneracy / baryon number / statistics / antiparticles from the id's digits."""
    path = str(tmp_path / "pdg_box.dat")
    refformat.write_pdg_box(path, refformat.BOX_ROWS)
    got = api.pdg_read(path, box=True)
    want = refformat.box_entries(refformat.BOX_ROWS)
    assert len(got["mc_id"]) == len(want) == 27
    for k, (i, m, g, b, sg) in enumerate(want):
        assert (got["mc_id"][k], got["mass"][k], got["gspin"][k], got["baryon"][k], got["sign"][k]) == (i, m, g, b, sg), k
    assert 221 in got["mc_id"] and -221 not in got["mc_id"] and -311 in got["mc_id"] and got["gspin"][list(got["mc_id"]).index(225)] == 5.0
    with open(path, "a", encoding="utf-8") as f:
        f.write("d   1.876  0  +  1000010020\n")            # the deuteron: the reference prints an error and carries on with wrong numbers
    with pytest.raises(api.Is3dError) as e:
        api.pdg_read(path, box=True)
    assert e.value.code == api.IS3D_EIO and "1000010020" in str(e.value)


@pytest.mark.reference
def test_smash_particle_lists_of_the_reference():
    """hrg_eos = 2 (the reference's shipped default) reads PDG/pdg_smash.dat with the conventional reader: against an independent token parse here;
    hrg_eos = 3 reads PDG/pdg_box.dat with the line reader: against refformat.box_entries on an independent line parse.  Both chosen lists are
    subsets of what is read (the reference requires it, iS3D_parameters.dat:18)."""
    tok = open(os.path.join(REFERENCE, "PDG/pdg_smash.dat"), encoding="utf-8").read().split()
    want, i = [], 0
    while i < len(tok):
        mc, mass, g, b, nd = int(tok[i]), float(tok[i + 2]), int(tok[i + 4]), int(tok[i + 5]), int(tok[i + 11])
        want.append((mc, mass, g, b))
        if b > 0:
            want.append((-mc, mass, g, -b))
        i += 12 + 8 * nd
    got = api.pdg_read(os.path.join(REFERENCE, "PDG/pdg_smash.dat"))
    assert len(got["mc_id"]) == len(want) == 493
    assert [(int(a), float(m), int(g), int(b)) for a, m, g, b in zip(got["mc_id"], got["mass"], got["gspin"], got["baryon"])] == want
    chosen = api.table_read(os.path.join(REFERENCE, "PDG/chosen_particles_smash.dat"))[:, 0]
    assert len(chosen) == 444 and set(chosen.astype(np.int64)) <= set(got["mc_id"])
    rows = []
    for line in open(os.path.join(REFERENCE, "PDG/pdg_box.dat"), encoding="utf-8"):
        if not line.strip() or line.startswith("#"):
            continue
        f = line.split("#")[0].split()
        rows.append((f[0], float(f[1]), [int(x) for x in f[4:8]]))
    box = api.pdg_read(os.path.join(REFERENCE, "PDG/pdg_box.dat"), box=True)
    wantb = refformat.box_entries(rows)
    assert len(box["mc_id"]) == len(wantb) == 400
    assert [(int(a), float(m), float(g), float(b), float(s)) for a, m, g, b, s in
            zip(box["mc_id"], box["mass"], box["gspin"], box["baryon"], box["sign"])] == wantb
    chosen = api.table_read(os.path.join(REFERENCE, "PDG/chosen_particles_box.dat"))[:, 0]
    assert len(chosen) == 399 and set(chosen.astype(np.int64)) <= set(box["mc_id"])
    # ... and their coefficient tables (deltafReader.h:27-29): the mu_B = 0 rows against numpy's own parse
    for d in ("smash", "smash_box"):
        for name in ("c0", "c2", "F", "betabulk", "betapi"):
            path = os.path.join(REFERENCE, "deltaf_coefficients/vh", d, name + ".dat")
            T, v = api.df_table_read(path)
            raw = np.loadtxt(path, skiprows=3)
            assert len(T) == 101 and np.array_equal(T, raw[:101, 0]) and np.array_equal(v, raw[:101, 2]) and np.all(raw[:101, 1] == 0.0)


def test_vah_table_reader(tmp_path):
    """is3d_vah_df_read on files in the shipped layout (src/cuda/deltafReader.cu:104-127, :196-213): dimensions, one label line read by
    fgets(header, 100), rows with alpha_L outer and Lambda inner; the node arrays are what the LAST rows leave behind."""
    tab = inputs.vah_df_tables()
    d = str(tmp_path / "vah")
    refformat.write_vah_df_tables(d, tab)
    got = api.vah_df_read(d)
    for k in ("L", "aL", "c0", "c1", "c2", "c3", "c4"):
        assert np.array_equal(got[k], tab[k]), k
    assert got["c3"].shape == (180, 80)
    # a label line of more than 99 characters: fgets(header, 100, file) stops after 99 and the scan then starts inside the label --
    # the reference's fscanf fails there silently; here it is an error
    refformat.write_vah_df_tables(d, tab, header_pad=120)
    with pytest.raises(api.Is3dError) as e:
        api.vah_df_read(d)
    assert e.value.code == api.IS3D_EIO
    refformat.write_vah_df_tables(d, tab)
    os.remove(os.path.join(d, "c3_vah1.dat"))
    with pytest.raises(api.Is3dError) as e:
        api.vah_df_read(d)
    assert e.value.code == api.IS3D_EIO and "c3" in str(e.value)
    small = {k: (v[:7] if k == "L" else v[:5] if k == "aL" else v[:5, :7]) for k, v in tab.items()}
    refformat.write_vah_df_tables(d, small)
    with open(os.path.join(d, "c1_vah1.dat"), "r+") as f:   # a header that disagrees with c0's
        f.write("6")
    with pytest.raises(api.Is3dError) as e:
        api.vah_df_read(d)
    assert e.value.code == api.IS3D_EINVAL
    small["L"] = small["L"][::-1].copy()
    refformat.write_vah_df_tables(d, small)
    with pytest.raises(api.Is3dError) as e:
        api.vah_df_read(d)
    assert e.value.code == api.IS3D_EINVAL and "ascend" in str(e.value)


def test_surface_reader_mode2(tmp_path):
    """is3d_surface_read_vah == the numpy restatement of read_surf_VAH_PLMatch (readindata.cpp:813-928, arsenal.cpp:999-1065): hbar*c
    conversions, u^tau dropped, (alpha_L, Lambda) from the conformal-factorisation fit; PL/P >= 3 is fatal."""
    from oracle import oracle
    path = str(tmp_path / "surface.dat")
    for dim, n in ((3, 57), (2, 11)):
        cells = synth.synth_vah_surface(n, dim, seed=300 + dim)
        synth.write_surface_vah_dat(path, cells)
        got = api.surface_read_vah(path, dim)
        ref = oracle.read_surf_VAH_PLMatch(path)
        for k in api.VAH_SURFACE_ORDER:
            assert np.allclose(got[k], ref[k], rtol=4e-15, atol=0), k
        for k in ("tau", "eta", "ux", "uy", "un", "dat", "dax", "day", "dan", "x", "y"):
            assert np.allclose(got[k], cells[k], rtol=1e-15, atol=0), k
        for k in ("T", "E", "P", "PL", "pitt", "pitn", "pinn", "piyn", "Wx", "Wy", "bulkPi"):
            assert np.allclose(got[k], cells[k], rtol=5e-16, atol=0), k
        assert (got["aL"] > 0.6).all() and (got["aL"] < 1.6).all()
        # isotropic pressure: alpha_L = 1 and Lambda = T (the fit is exact to 2e-8 there)
        iso = dict(cells, PL=cells["P"].copy())
        synth.write_surface_vah_dat(path, iso)
        g2 = api.surface_read_vah(path, dim)
        assert np.allclose(g2["aL"], 1.0, rtol=1e-7) and np.allclose(g2["Lambda"], g2["T"], rtol=1e-7)
    bad = dict(cells, PL=cells["P"] * np.where(np.arange(n) == 4, 3.5, 1.0))
    synth.write_surface_vah_dat(path, bad)
    with pytest.raises(api.Is3dError) as e:
        api.surface_read_vah(path, 2)
    assert e.value.code == api.IS3D_EINVAL and "cell 4" in str(e.value)
    with open(path, "w") as f:
        f.write("1 2 3\n")
    with pytest.raises(api.Is3dError) as e:
        api.surface_read_vah(path, 3)
    assert e.value.code == api.IS3D_EIO


@pytest.mark.reference
def test_vah_tables_of_the_reference_parse_to_the_fixture():
    got = api.vah_df_read(os.path.join(REFERENCE, "deltaf_coefficients/vah"))
    tab = inputs.vah_df_tables()
    for k in ("L", "aL", "c0", "c1", "c2", "c3", "c4"):
        assert np.array_equal(got[k], tab[k]), k


def test_surface_open_is_the_two_call_readers_in_one_pass(tmp_path):
    """is3d_surface_open (one read, one parse, arrays owned by the library) gives bit for bit what is3d_surface_read / _read_vah give, for
    every format of the smooth path, plus the position columns x, y the sampler wants; cache = 0 touches nothing next to the file."""
    path = str(tmp_path / "surface.dat")
    c3 = synth.synth_surface(41, 3, seed=31, baryon=True)
    for mode, ib, idf in [(0, 1, 1), (1, 0, 0), (1, 1, 1), (4, 0, 0), (5, 1, 1), (6, 0, 0), (7, 0, 0)]:
        if mode == 1:
            synth.write_surface_dat(path, c3 if ib else {k: v for k, v in c3.items() if k not in synth.BARYON_FIELDS})
        else:
            refformat.write_surface_mode(path, c3, mode, include_baryon=ib, include_baryondiff=idf)
        ref, avg_ref = api.surface_read(path, mode, ib, idf, 3)
        got, avg, source = api.surface_open(path, mode, ib, idf, 3, cache=0)
        assert source == 0 and not os.path.exists(path + ".is3dcache")
        for k in api.SURFACE_READ_ORDER:
            present = k in synth.CELL_FIELDS or (k == "muB" and ib) or (k in ("nB", "Vx", "Vy", "Vn") and idf)
            if present:
                assert np.array_equal(got[k], ref[k]), (mode, k)
            else:
                assert got[k] is None, (mode, k)
        assert np.array_equal(avg, avg_ref)
        cols = np.loadtxt(path, ndmin=2)
        assert np.array_equal(got["x"], cols[:, 1]) and np.array_equal(got["y"], cols[:, 2])
    vc = synth.synth_vah_surface(23, 3, seed=32)
    synth.write_surface_vah_dat(path, vc)
    ref = api.surface_read_vah(path, 3)
    got, avg, source = api.surface_open(path, 2, dimension=3, cache=0)
    assert avg is None and source == 0
    for k in api.VAH_SURFACE_ORDER:
        assert np.array_equal(got[k], ref[k]), k
    # an unterminated single line is zero cells, a missing file IS3D_EIO, a format the path does not read IS3D_EINVAL
    (tmp_path / "one.dat").write_text("0.5 0 0 0 1000.0 0 0 0 0 0 0 1.839  0.786  0.270 0 0 0 0 0 0")
    got, _, _ = api.surface_open(str(tmp_path / "one.dat"), 1, cache=0)
    assert len(got["tau"]) == 0
    with pytest.raises(api.Is3dError) as e:
        api.surface_open(str(tmp_path / "nope.dat"), 1)
    assert e.value.code == api.IS3D_EIO
    with pytest.raises(api.Is3dError) as e:
        api.surface_open(path, 3)
    assert e.value.code == api.IS3D_EINVAL


def test_surface_sidecar_cache_and_its_invalidation_rules(tmp_path, monkeypatch):
    """The binary sidecar `<path>.is3dcache` ("offer a binary cache"): written after the first parse, used -- bit for bit the
    parsed arrays and averages -- when size, mtime, sampled-content hash and parse parameters match the text file as it is now, ignored and
    rewritten otherwise; IS3D_NO_CACHE=1 neither reads nor writes one; cache = 2 hashes the whole text."""
    monkeypatch.delenv("IS3D_NO_CACHE", raising=False)
    path = str(tmp_path / "surface.dat")
    side = path + ".is3dcache"
    cells = synth.synth_surface(6000, 3, seed=33, baryon=True)       # ~ 3 MB of text: larger than the sampled head and tail
    synth.write_surface_dat(path, cells)
    first, avg1, s1 = api.surface_open(path, 1, 1, 1, 3)
    assert s1 == 1 and os.path.exists(side)
    n_arrays = 25
    assert os.path.getsize(side) == 128 + n_arrays * 6000 * 8
    again, avg2, s2 = api.surface_open(path, 1, 1, 1, 3)
    assert s2 == 2 and np.array_equal(avg1, avg2)
    for k in first:
        assert np.array_equal(first[k], again[k]), k
    ref, avg_ref = api.surface_read(path, 1, 1, 1, 3)
    for k in api.SURFACE_READ_ORDER:
        assert np.array_equal(again[k], ref[k]), k
    assert np.array_equal(avg2, avg_ref)
    # (a) other parse parameters: the sidecar of (include_baryon, diffusion) = (1, 1) is not the one of (0, 0) -- reparse, rewrite
    _, _, s = api.surface_open(path, 1, 1, 0, 3)
    assert s == 1 and os.path.getsize(side) == 128 + 21 * 6000 * 8
    _, _, s = api.surface_open(path, 1, 1, 0, 3)
    assert s == 2
    _, _, s = api.surface_open(path, 1, 1, 0, 2)            # dimension is a parse parameter too (mode 0 / 4 treat dsigma_eta by it)
    assert s == 1
    # (b) mtime moved (same bytes): reparse
    st = os.stat(path)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns + 1000))
    _, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 1
    _, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 2
    # (c) content changed, size and mtime kept: a change inside a sampled block is seen by cache = 1 ...
    st = os.stat(path)
    text = bytearray(open(path, "rb").read())
    size = len(text)
    i = text.index(b"e", 100)                               # a digit of the first rows' mantissa -> 9 (head block)
    text[i - 1:i] = b"9" if text[i - 1:i] != b"9" else b"8"
    open(path, "wb").write(text)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    got, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 1
    ref, _ = api.surface_read(path, 1, 1, 0, 2)
    assert all(np.array_equal(got[k], ref[k]) for k in synth.CELL_FIELDS)
    # ... one outside every sampled block (first / last 64 KiB, 256 blocks of 4 KiB at multiples of size // 257) only by cache = 2
    st = os.stat(path)
    kblk = 65536 // (size // 257) + 2                       # first sampled 4-KiB block that starts beyond the head
    off = (size // 257) * kblk + 4096 + 700
    assert 65536 < off < (size // 257) * (kblk + 1) - 64 and off < size - 65536
    j = text.index(b"e", off)
    text[j - 1:j] = b"7" if text[j - 1:j] != b"7" else b"6"
    open(path, "wb").write(text)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    # round 5: the key holds the inode's CHANGE time too -- utime() restored the mtime above, nothing in user space restores a ctime, so the
    # default (cache = 1) re-parses this edit as well (round 4 served the stale sidecar here: the sampled hash alone misses it)
    got, _, s = api.surface_open(path, 1, 1, 0, 2, cache=1)
    assert s == 1
    ref, _ = api.surface_read(path, 1, 1, 0, 2)
    assert all(np.array_equal(got[k], ref[k]) for k in synth.CELL_FIELDS)
    _, _, s = api.surface_open(path, 1, 1, 0, 2, cache=1)
    assert s == 2
    _, _, s = api.surface_open(path, 1, 1, 0, 2, cache=2)   # a sidecar written without the whole-file hash does not satisfy cache = 2
    assert s == 1
    _, _, s = api.surface_open(path, 1, 1, 0, 2, cache=2)
    assert s == 2
    # a copy of the text WITH its sidecar (another inode): re-parsed once, the safe direction
    import shutil
    path2 = str(tmp_path / "copy.dat")
    shutil.copy2(path, path2)
    shutil.copy2(side, path2 + ".is3dcache")
    _, _, s = api.surface_open(path2, 1, 1, 0, 2)
    assert s == 1
    _, _, s = api.surface_open(path2, 1, 1, 0, 2)
    assert s == 2
    # (d) a truncated or foreign sidecar is ignored and replaced
    blob = open(side, "rb").read()
    open(side, "wb").write(blob[:-8])
    _, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 1 and os.path.getsize(side) == len(blob)
    open(side, "wb").write(b"not a cache")
    _, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 1
    # a header whose n_cells makes popcount(mask) * n_cells * 8 wrap around 2^64 to the file's own length: bounded before it is multiplied,
    # ignored like any other unusable sidecar (round 4: the size check passed and vector::resize threw through the C ABI)
    import struct
    blob = bytearray(open(side, "rb").read())
    n_here, = struct.unpack_from("<q", blob, 64)
    assert n_here == 6000
    struct.pack_into("<q", blob, 64, n_here + (1 << 61))
    open(side, "wb").write(blob)
    got, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 1 and len(got["tau"]) == 6000
    # (e) size changed (a row appended)
    with open(path, "ab") as f:
        f.write(open(path, "rb").readline())
    got, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 1 and len(got["tau"]) == 6001
    # (f) IS3D_NO_CACHE=1: no read, no write
    os.remove(side)
    monkeypatch.setenv("IS3D_NO_CACHE", "1")
    _, _, s = api.surface_open(path, 1, 1, 0, 2)
    assert s == 0 and not os.path.exists(side)
    monkeypatch.delenv("IS3D_NO_CACHE")
    # mode 2 (anisotropic hydro) has its own 32-array sidecar
    vc = synth.synth_vah_surface(50, 3, seed=34)
    vpath = str(tmp_path / "vah.dat")
    synth.write_surface_vah_dat(vpath, vc)
    a, _, s = api.surface_open(vpath, 2)
    b, _, s2 = api.surface_open(vpath, 2)
    assert (s, s2) == (1, 2) and all(np.array_equal(a[k], b[k]) for k in api.VAH_SURFACE_ORDER)
