"""ctypes front end of oracle/libcf_oracle.so -- TEST INFRASTRUCTURE ONLY (see cf_oracle.c header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package is3d_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CELL_FIELDS = ["T", "P", "E", "tau", "eta", "ux", "uy", "un", "dat", "dax", "day", "dan",
               "pixx", "pixy", "pixn", "piyy", "piyn", "bulkPi", "muB", "nB", "Vx", "Vy", "Vn"]
_dp = C.POINTER(C.c_double)


class _CellArrays(C.Structure):
    _fields_ = [(n, _dp) for n in CELL_FIELDS]


class _DfTables(C.Structure):
    _fields_ = [("n_T", C.c_int), ("T", _dp), ("c0", _dp), ("c2", _dp), ("F", _dp), ("betabulk", _dp), ("betapi", _dp),
                ("n_muB", C.c_int), ("muB", _dp), ("t2d", _dp * 10)]


DF_NAMES_2D = ["c0", "c1", "c2", "c3", "c4", "F", "G", "betabulk", "betaV", "betapi"]


class _Opts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ["dimension", "df_mode", "include_baryon", "include_bulk_deltaf",
                                       "include_shear_deltaf", "include_baryondiff_deltaf", "regulate_deltaf", "outflow",
                                       "reference_bilinear_indexing"]]


class _Grid(C.Structure):
    _fields_ = [("pT_tab_length", C.c_int), ("pT", _dp), ("phi_tab_length", C.c_int), ("phi", _dp),
                ("y_tab_length", C.c_int), ("y", _dp), ("eta_tab_length", C.c_int), ("eta", _dp), ("eta_w", _dp)]


def build(force=False):
    so = os.path.join(_HERE, "libcf_oracle.so")
    src = os.path.join(_HERE, "cf_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcf_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_cspline_init.argtypes = [C.c_int, _dp, _dp, _dp]
        _LIB.oracle_cspline_eval.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, _dp]
        _LIB.oracle_df_coefficients.argtypes = [C.POINTER(_DfTables), C.c_int, C.c_double, _dp]
        sig = [C.c_long, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(_CellArrays), C.POINTER(_DfTables),
               C.POINTER(_Grid), C.POINTER(_Opts)]
        _LIB.oracle_dN_pTdpTdphidy.argtypes = sig + [_dp]
        _LIB.oracle_dN_pTdpTdphidy_chunked.argtypes = sig + [C.c_long, _dp]
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def cspline_init(x, y):
    x, y = _f64(x), _f64(y)
    c = np.zeros_like(x)
    rc = lib().oracle_cspline_init(len(x), _p(x), _p(y), _p(c))
    assert rc == 0
    return c


def cspline_eval(x, y, c, xq):
    x, y, c = _f64(x), _f64(y), _f64(c)
    out = C.c_double()
    rc = lib().oracle_cspline_eval(len(x), _p(x), _p(y), _p(c), float(xq), C.byref(out))
    if rc:
        raise ValueError("oracle_cspline_eval: x outside the table (GSL would abort)")
    return out.value


def _df_struct(df):
    keep = {k: _f64(df[k]) for k in ["T", "c0", "c2", "F", "betabulk", "betapi"]}
    st = _DfTables(len(keep["T"]), *[_p(keep[k]) for k in ["T", "c0", "c2", "F", "betabulk", "betapi"]])
    if "2d" in df:   # full (mu_B, T) tables for include_baryon = 1
        keep["muB"] = _f64(df["muB"])
        st.n_muB = len(keep["muB"])
        st.muB = _p(keep["muB"])
        for i, name in enumerate(DF_NAMES_2D):
            keep["2d_" + name] = _f64(df["2d"][name])
            assert keep["2d_" + name].shape == (st.n_muB, st.n_T)
            st.t2d[i] = _p(keep["2d_" + name])
    return st, keep


def df_coefficients(df, df_mode, T):
    """-> dict(c0, c2, F, betabulk, betapi) with the temperature scaling undone (deltafReader.cpp:337-358)."""
    st, keep = _df_struct(df)
    out = np.zeros(5)
    rc = lib().oracle_df_coefficients(C.byref(st), int(df_mode), float(T), _p(out))
    if rc:
        raise ValueError("T outside the coefficient table")
    return dict(zip(["c0", "c2", "F", "betabulk", "betapi"], out))


def df_coefficients_bilinear(df, df_mode, T, muB, reference_indexing=False):
    """-> dict of the ten coefficients by the bilinear branch, temperature scaling undone.  reference_indexing: read the tables
    as the reference's calculate_bilinear does (f_data[iT][imuB], deltafReader.cpp:404-407) instead of the intended [imuB][iT]."""
    st, keep = _df_struct(df)
    out = np.zeros(10)
    lib().oracle_df_coefficients_bilinear.argtypes = [C.POINTER(_DfTables), C.c_int, C.c_double, C.c_double, _dp]
    rc = lib().oracle_df_coefficients_bilinear(C.byref(st), int(df_mode) + (100 if reference_indexing else 0), float(T), float(muB), _p(out))
    if rc:
        raise ValueError("(T, muB) outside the coefficient table (rc=%d)" % rc)
    return dict(zip(DF_NAMES_2D, out))


DEFAULT_OPTS = dict(dimension=3, df_mode=1, include_baryon=0, include_bulk_deltaf=1, include_shear_deltaf=1,
                    include_baryondiff_deltaf=0, regulate_deltaf=1, outflow=1, reference_bilinear_indexing=0)


def dN_pTdpTdphidy(cells, species, grid, df, opts, chunked=False, FO_chunk=10000, out=None):
    """cells: dict of 1-D arrays (CELL_FIELDS; missing ones = unused);
    species: dict(mass, sign, degeneracy, baryon); grid: dict(pT, phi, y, eta, eta_w);
    df: dict(T, c0, c2, F, betabulk, betapi); opts: dict like DEFAULT_OPTS.
    Returns the spectrum, flat, length npart*npT*nphi*ny_eff with ny_eff = 1 in 2+1D; species fastest."""
    o = dict(DEFAULT_OPTS)
    o.update(opts)
    n = len(cells["tau"])
    keep = {}
    ca = _CellArrays()
    for f in CELL_FIELDS:
        if f in cells and cells[f] is not None:
            keep[f] = _f64(cells[f])
            assert keep[f].shape == (n,), f
            setattr(ca, f, _p(keep[f]))
    sp = {k: _f64(species[k]) for k in ["mass", "sign", "degeneracy", "baryon"]}
    npart = len(sp["mass"])
    g = {k: _f64(grid[k]) for k in ["pT", "phi", "y", "eta", "eta_w"]}
    gs = _Grid(len(g["pT"]), _p(g["pT"]), len(g["phi"]), _p(g["phi"]), len(g["y"]), _p(g["y"]),
               len(g["eta"]), _p(g["eta"]), _p(g["eta_w"]))
    st, keep_df = _df_struct(df)
    os_ = _Opts(*[int(o[k]) for k, _ in _Opts._fields_])
    ny = 1 if o["dimension"] == 2 else len(g["y"])
    size = npart * len(g["pT"]) * len(g["phi"]) * ny
    if out is None:
        out = np.zeros(size)
    assert out.dtype == np.float64 and out.size == size
    args = [n, npart, _p(sp["mass"]), _p(sp["sign"]), _p(sp["degeneracy"]), _p(sp["baryon"]),
            C.byref(ca), C.byref(st), C.byref(gs), C.byref(os_)]
    if chunked:
        rc = lib().oracle_dN_pTdpTdphidy_chunked(*args, int(FO_chunk), _p(out))
    else:
        rc = lib().oracle_dN_pTdpTdphidy(*args, _p(out))
    if rc:
        raise RuntimeError("oracle_dN_pTdpTdphidy failed rc=%d" % rc)
    return out


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))
    return num_threads()


class _FeqmodTables(C.Structure):
    _fields_ = [("n_pts", C.c_int), ("root1", _dp), ("weight1", _dp), ("root2", _dp), ("weight2", _dp), ("n_pdg", C.c_int),
                ("pdg_mass", _dp), ("pdg_degeneracy", _dp), ("pdg_sign", _dp), ("T_avg", C.c_double), ("deta_min", C.c_double),
                ("mass_pion0", C.c_double)]


def _feqmod_struct(fq):
    keep = {k: _f64(fq[k]) for k in ["root1", "weight1", "root2", "weight2", "pdg_mass", "pdg_degeneracy", "pdg_sign"]}
    st = _FeqmodTables(len(keep["root1"]), _p(keep["root1"]), _p(keep["weight1"]), _p(keep["root2"]), _p(keep["weight2"]),
                       len(keep["pdg_mass"]), _p(keep["pdg_mass"]), _p(keep["pdg_degeneracy"]), _p(keep["pdg_sign"]),
                       float(fq["T_avg"]), float(fq["deta_min"]), float(fq["mass_pion0"]))
    return st, keep


def jonah_tables(fq):
    """-> (lambda^2[301], z[301], bulkPi/Peq[301], bulkPi_over_Peq_max): Deltaf_Data::compute_jonah_coefficients."""
    st, keep = _feqmod_struct(fq)
    out = np.zeros(3 * 301 + 1)
    lib().oracle_jonah_tables.argtypes = [C.POINTER(_FeqmodTables), _dp]
    lib().oracle_jonah_tables(C.byref(st), _p(out))
    t = out[:903].reshape(301, 3)
    return t[:, 0].copy(), t[:, 1].copy(), t[:, 2].copy(), float(out[903])


def dN_pTdpTdphidy_feqmod(cells, species, grid, df, fq, opts, out=None):
    """Modified-equilibrium smooth spectra (df_mode 3 | 4), calculate_dN_ptdptdphidy_feqmod.  Returns (spectrum, n_breakdown)."""
    o = dict(DEFAULT_OPTS)
    o.update(opts)
    n = len(cells["tau"])
    keep = {}
    ca = _CellArrays()
    for f in CELL_FIELDS:
        if f in cells and cells[f] is not None:
            keep[f] = _f64(cells[f])
            setattr(ca, f, _p(keep[f]))
    sp = {k: _f64(species[k]) for k in ["mass", "sign", "degeneracy", "baryon"]}
    npart = len(sp["mass"])
    g = {k: _f64(grid[k]) for k in ["pT", "phi", "y", "eta", "eta_w"]}
    gs = _Grid(len(g["pT"]), _p(g["pT"]), len(g["phi"]), _p(g["phi"]), len(g["y"]), _p(g["y"]), len(g["eta"]), _p(g["eta"]), _p(g["eta_w"]))
    st, keep_df = _df_struct(df)
    fs, keep_fq = _feqmod_struct(fq)
    os_ = _Opts(*[int(o[k]) for k, _ in _Opts._fields_])
    ny = 1 if o["dimension"] == 2 else len(g["y"])
    size = npart * len(g["pT"]) * len(g["phi"]) * ny
    if out is None:
        out = np.zeros(size)
    nb = C.c_long(0)
    L = lib()
    L.oracle_dN_pTdpTdphidy_feqmod.argtypes = [C.c_long, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(_CellArrays), C.POINTER(_DfTables),
                                               C.POINTER(_FeqmodTables), C.POINTER(_Grid), C.POINTER(_Opts), _dp, C.POINTER(C.c_long)]
    rc = L.oracle_dN_pTdpTdphidy_feqmod(n, npart, _p(sp["mass"]), _p(sp["sign"]), _p(sp["degeneracy"]), _p(sp["baryon"]), C.byref(ca),
                                        C.byref(st), C.byref(fs), C.byref(gs), C.byref(os_), _p(out), C.byref(nb))
    if rc:
        raise RuntimeError("oracle_dN_pTdpTdphidy_feqmod failed rc=%d" % rc)
    return out, int(nb.value)


# ---- particle sampler (operation = 2) ----------------------------------------------------------------------------
PARTICLE_FIELDS = ["event", "cell", "species", "tau", "x", "y", "eta", "t", "z", "E", "px", "py", "pz", "rapidity"]


def philox4x32_10(ctr, key):
    L = lib()
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    L.oracle_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def rng_uniforms(seed, stream, cell, event, n):
    L = lib()
    L.oracle_rng_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _dp]
    out = np.zeros(n)
    L.oracle_rng_uniforms(seed, stream, cell, event, n, _p(out))
    return out


class _SamplerOpts(C.Structure):
    _fields_ = [("n_events", C.c_int), ("fast", C.c_int), ("seed", C.c_uint64), ("y_cut", C.c_double), ("first_cell", C.c_long),
                ("T_avg", C.c_double), ("T_avg_switch", C.c_double), ("muB_avg", C.c_double)]


def sample_particles(cells, species, df, gla, opts, n_events=1, seed=1, y_cut=0.5, capacity=None, first_cell=0, fq=None, fast=0,
                     T_avg=0.0, T_avg_switch=None, muB_avg=0.0):
    """sample_dN_pTdpTdphidy (df_mode 1..4) with the counter-based RNG defined in cf_oracle.c.  gla: dict with root1, weight1
    (is3d_amd.inputs.feqmod_tables() has them); fq: the feqmod tables (df_mode 3, 4, and the alpha = 2 nodes of fast mode);
    fast = 1: species densities at T_avg (breakdown test at T_avg_switch, default T_avg).  cells may carry x, y.
    Returns (dict of arrays per PARTICLE_FIELDS, stats dict)."""
    o = dict(DEFAULT_OPTS)
    o.update(opts)
    n = len(cells["tau"])
    keep = {}
    ca = _CellArrays()
    for f in CELL_FIELDS:
        if f in cells and cells[f] is not None:
            keep[f] = _f64(cells[f])
            setattr(ca, f, _p(keep[f]))
    xs = _f64(cells["x"]) if cells.get("x") is not None else None
    ys = _f64(cells["y"]) if cells.get("y") is not None else None
    sp = {k: _f64(species[k]) for k in ["mass", "sign", "degeneracy"]}
    sp["baryon"] = _f64(species["baryon"]) if "baryon" in species else np.zeros(len(sp["mass"]))
    npart = len(sp["mass"])
    st, keep_df = _df_struct(df)
    r1, w1 = _f64(gla["root1"]), _f64(gla["weight1"])
    if fq is None and (int(o["df_mode"]) >= 3 or fast):
        fq = gla
    fs, keep_fq = _feqmod_struct(fq) if fq is not None else (None, None)
    os_ = _Opts(*[int(o[k]) for k, _ in _Opts._fields_])
    so = _SamplerOpts(int(n_events), int(fast), int(seed), float(y_cut), int(first_cell), float(T_avg),
                      float(T_avg if T_avg_switch is None else T_avg_switch), float(muB_avg))
    L = lib()
    L.oracle_sample_particles.restype = C.c_long
    L.oracle_sample_particles.argtypes = [C.c_long, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(_CellArrays), _dp, _dp, C.POINTER(_DfTables), C.c_int,
                                          _dp, _dp, C.POINTER(_FeqmodTables), C.POINTER(_Opts), C.POINTER(_SamplerOpts), _dp, C.c_long,
                                          C.POINTER(C.c_long)]
    stats = (C.c_long * 4)()
    cap = int(capacity) if capacity is not None else 0
    while True:
        out = np.zeros((max(cap, 1), len(PARTICLE_FIELDS)))
        rc = L.oracle_sample_particles(n, npart, _p(sp["mass"]), _p(sp["sign"]), _p(sp["degeneracy"]), _p(sp["baryon"]), C.byref(ca),
                                       _p(xs) if xs is not None else None, _p(ys) if ys is not None else None, C.byref(st), len(r1), _p(r1),
                                       _p(w1), C.byref(fs) if fs is not None else None, C.byref(os_), C.byref(so), _p(out), cap, stats)
        if rc < 0:
            raise RuntimeError("oracle_sample_particles failed rc=%d" % rc)
        if rc <= cap or capacity is not None:
            break
        cap = int(rc)
    k = min(int(rc), cap)
    res = {f: out[:k, i].copy() for i, f in enumerate(PARTICLE_FIELDS)}
    for f in ("event", "cell", "species"):
        res[f] = res[f].astype(np.int64)
    return res, dict(n_kept=int(rc), samples=int(stats[0]), acceptances=int(stats[1]), drawn=int(stats[2]), breakdown=int(stats[3]))


def total_yield(cells, species, df, gla, avg5, opts, y_cut=0.5, fq=None):
    """calculate_total_yield restated (oracle_total_yield).  gla: dict with root1, weight1, root2, weight2 (+ root3, weight3 for
    df_mode 1); avg5 = (T, E, P, muB, nB) surface averages; fq: feqmod tables (df_mode 4; defaults to gla).
    Returns (yield, densities[3][n_species])."""
    o = dict(DEFAULT_OPTS)
    o.update(opts)
    n = len(cells["tau"])
    keep = {}
    ca = _CellArrays()
    for f in CELL_FIELDS:
        if f in cells and cells[f] is not None:
            keep[f] = _f64(cells[f])
            setattr(ca, f, _p(keep[f]))
    sp = {k: _f64(species[k]) for k in ["mass", "sign", "degeneracy"]}
    sp["baryon"] = _f64(species["baryon"]) if "baryon" in species else np.zeros(len(sp["mass"]))
    npart = len(sp["mass"])
    st, keep_df = _df_struct(df)
    r = {k: _f64(gla[k]) for k in ["root1", "weight1", "root2", "weight2"]}
    r3 = _f64(gla["root3"]) if "root3" in gla else None
    w3 = _f64(gla["weight3"]) if "weight3" in gla else None
    fs, keep_fq = _feqmod_struct(fq if fq is not None else gla) if int(o["df_mode"]) == 4 else (None, None)
    os_ = _Opts(*[int(o[k]) for k, _ in _Opts._fields_])
    a5 = _f64(avg5)
    L = lib()
    L.oracle_total_yield.argtypes = [C.c_long, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(_CellArrays), C.POINTER(_DfTables), C.c_int,
                                     _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(_FeqmodTables), C.POINTER(_Opts), _dp, C.c_double, _dp, _dp]
    out = np.zeros(1)
    dens = np.zeros((3, npart))
    rc = L.oracle_total_yield(n, npart, _p(sp["mass"]), _p(sp["sign"]), _p(sp["degeneracy"]), _p(sp["baryon"]), C.byref(ca), C.byref(st),
                              len(r["root1"]), _p(r["root1"]), _p(r["weight1"]), _p(r["root2"]), _p(r["weight2"]),
                              _p(r3) if r3 is not None else None, _p(w3) if w3 is not None else None,
                              C.byref(fs) if fs is not None else None, C.byref(os_), _p(a5), float(y_cut), _p(out), _p(dens))
    if rc:
        raise RuntimeError("oracle_total_yield failed rc=%d" % rc)
    return float(out[0]), dens


# ---- anisotropic hydro (VAH, P_L matching) smooth kernel ---------------------------------------------------------
VAH_FIELDS = ["tau", "eta", "ux", "uy", "un", "dat", "dax", "day", "dan", "T", "pitt", "pitx", "pity", "pitn", "pixx", "pixy", "pixn",
              "piyy", "piyn", "pinn", "bulkPi", "Wx", "Wy", "Lambda", "aL", "c0", "c1", "c2", "c3", "c4"]


class _VahCells(C.Structure):
    _fields_ = [(f, _dp) for f in VAH_FIELDS]


def dN_pTdpTdphidy_vah(cells, species, grid, opts):
    """calculate_dN_pTdpTdphidy_VAH_PL restated (the reference never runs it); cells: dict with VAH_FIELDS (eta optional in 2+1D)."""
    o = dict(DEFAULT_OPTS)
    o.update(opts)
    n = len(cells["tau"])
    keep = {}
    ca = _VahCells()
    for f in VAH_FIELDS:
        if cells.get(f) is not None:
            keep[f] = _f64(cells[f])
            setattr(ca, f, _p(keep[f]))
    sp = {k: _f64(species[k]) for k in ["mass", "sign", "degeneracy"]}
    npart = len(sp["mass"])
    g = {k: _f64(grid[k]) for k in ["pT", "phi", "y", "eta", "eta_w"]}
    gs = _Grid(len(g["pT"]), _p(g["pT"]), len(g["phi"]), _p(g["phi"]), len(g["y"]), _p(g["y"]), len(g["eta"]), _p(g["eta"]), _p(g["eta_w"]))
    os_ = _Opts(*[int(o[k]) for k, _ in _Opts._fields_])
    ny = 1 if o["dimension"] == 2 else len(g["y"])
    out = np.zeros(npart * len(g["pT"]) * len(g["phi"]) * ny)
    L = lib()
    L.oracle_dN_pTdpTdphidy_vah.argtypes = [C.c_long, C.c_int, _dp, _dp, _dp, C.POINTER(_VahCells), C.POINTER(_Grid), C.POINTER(_Opts), _dp]
    rc = L.oracle_dN_pTdpTdphidy_vah(n, npart, _p(sp["mass"]), _p(sp["sign"]), _p(sp["degeneracy"]), C.byref(ca), C.byref(gs), C.byref(os_), _p(out))
    if rc:
        raise RuntimeError("oracle_dN_pTdpTdphidy_vah failed rc=%d" % rc)
    return out


def vah_coefficients(tab, Lambda, aL):
    """oracle_vah_coefficients: src/cuda/deltafReader.cu:216-278 per cell.  tab: dict L, aL, c0..c4 ([n_aL][n_L]); Lambda in GeV.
    Returns (dict c0..c4, found mask): where found is False the reference leaves the cell's coefficients unset (NaN here)."""
    L = lib()
    Lg, ag = _f64(tab["L"]), _f64(tab["aL"])
    t = [_f64(tab["c%d" % k]) for k in range(5)]
    for x in t:
        assert x.shape == (len(ag), len(Lg))
    lam, al = _f64(Lambda), _f64(aL)
    n = len(lam)
    out = [np.full(n, np.nan) for _ in range(5)]
    found = np.zeros(n, dtype=np.int32)
    L.oracle_vah_coefficients.argtypes = [C.c_int, C.c_int] + [_dp] * 7 + [C.c_long, _dp, _dp] + [_dp] * 5 + [C.POINTER(C.c_int32)]
    rc = L.oracle_vah_coefficients(len(Lg), len(ag), _p(Lg), _p(ag), *[_p(x) for x in t], n, _p(lam), _p(al), *[_p(x) for x in out],
                                   found.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0
    return {"c%d" % k: out[k] for k in range(5)}, found.astype(bool)


def aL_fit(x):
    """src/cpp/arsenal.cpp:999-1028: alpha_L(PL/Peq), conformal factorisation fit, the reference's operation order."""
    x = np.asarray(x, dtype=np.float64)
    p = [x ** 0]
    for _ in range(14):
        p.append(p[-1] * x)
    x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14 = p[1:]
    num = (2.307660683188896e-22 + 1.7179667824677117e-16 * x1 + 7.2725449826862375e-12 * x2 + 4.2846163672079405e-8 * x3 + 0.00004757224421671691 * x4 +
           0.011776118846199547 * x5 + 0.7235583305942909 * x6 + 11.582755440134724 * x7 + 44.45243622597357 * x8 + 12.673594148032494 * x9 -
           33.75866652773691 * x10 + 8.04299287188939 * x11 + 1.462901772148128 * x12 - 0.6320131889637761 * x13 + 0.048528166213735346 * x14)
    den = (5.595674409987461e-19 + 8.059757191879689e-14 * x1 + 1.2033043382301483e-9 * x2 + 2.9819348588423508e-6 * x3 + 0.0015212379997299082 * x4 +
           0.18185453852532632 * x5 + 5.466199358534425 * x6 + 40.1581708710626 * x7 + 44.38310108782752 * x8 - 55.213789667214364 * x9 +
           1.5449108423263358 * x10 + 11.636087951096759 * x11 - 4.005934533735304 * x12 + 0.4703844693488544 * x13 - 0.014599143701745957 * x14)
    return num / den


def R200(aL):
    """src/cpp/arsenal.cpp:1031-1065: R200(alpha_L) = alpha_L t200(xi), xi = 1/alpha_L^2 - 1 (three branches around xi = 0)."""
    aL = np.asarray(aL, dtype=np.float64)
    x = (1.0 / (aL * aL)) - 1.0
    delta = 0.01
    t = np.full(x.shape, np.nan)
    hi = x > delta
    lo = (x < -delta) & (x > -1.0)
    mid = (x >= -delta) & (x <= delta)
    with np.errstate(invalid="ignore"):
        t[hi] = 1.0 + (1.0 + x[hi]) * np.arctan(np.sqrt(x[hi])) / np.sqrt(x[hi])
        t[lo] = 1.0 + (1.0 + x[lo]) * np.arctanh(np.sqrt(-x[lo])) / np.sqrt(-x[lo])
    xm = x[mid]
    t[mid] = 2.0 + xm * (0.6666666666666667 + xm * (-0.1333333333333333 + xm * (0.05714285714285716 + xm * (-0.031746031746031744 + xm * (0.020202020202020193 +
             xm * (-0.013986013986013984 + (0.010256410256410262 - 0.00784313725490196 * xm) * xm))))))
    return aL * t


def read_surf_VAH_PLMatch(path):
    """FO_data_reader::read_surf_VAH_PLMatch (src/cpp/readindata.cpp:813-928) in numpy: 31 columns per cell -> the VAH cell arrays
    (GeV units) with aL = aL_fit(PL/P), Lambda = T / (aL R200(aL) / 2)^(1/4) hbarc.  Cells with PL/P >= 3 make the reference exit."""
    a = np.loadtxt(path, ndmin=2)
    assert a.shape[1] == 31, a.shape
    h = 0.197327053   # src/cpp/iS3D.h:9
    names = ["tau", "x", "y", "eta", "dat", "dax", "day", "dan", "ut", "ux", "uy", "un", "E", "T", "P", "PL",
             "pitt", "pitx", "pity", "pitn", "pixx", "pixy", "pixn", "piyy", "piyn", "pinn", "Wt", "Wx", "Wy", "Wn", "bulkPi"]
    s = {n: a[:, i].copy() for i, n in enumerate(names)}
    T, P, PL = s["T"].copy(), s["P"].copy(), s["PL"].copy()          # fm^-1, fm^-4: kept for the (aL, Lambda) inference (:874-882)
    if not np.all(PL / P < 3.0):
        raise ValueError("pl is too large, stopping anisotropic variables...")
    for n in names[12:]:
        s[n] = s[n] * h
    aL = aL_fit(PL / P)
    s["aL"] = aL
    s["Lambda"] = (T / np.power(0.5 * aL * R200(aL), 0.25)) * h
    return s


def df_generator_row(pdg, gla_root, gla_weight, T, muB, with_integrals=False):
    """oracle_df_generator_row: the ten numbers the reference's coefficient generator prints for (T, muB) (deltaf_table.cpp:137-248, :296-395),
    in the order of DF_NAMES_2D.  pdg: dict with mass, gspin, baryon, sign (every entry of the list, antibaryons included);
    gla_root / gla_weight: [alpha][points] as Gauss_Laguerre::load_roots_and_weights returns them (alpha = 1..4 are used)."""
    L = lib()
    m, g, b, s_ = (_f64(pdg[k]) for k in ("mass", "gspin", "baryon", "sign"))
    r = [_f64(gla_root[a]) for a in (1, 2, 3, 4)]
    w = [_f64(gla_weight[a]) for a in (1, 2, 3, 4)]
    out, integ = np.zeros(10), np.zeros(20)
    L.oracle_df_generator_row.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_int] + [_dp] * 8 + [C.c_double, C.c_double, _dp, _dp]
    rc = L.oracle_df_generator_row(len(m), _p(m), _p(g), _p(b), _p(s_), len(r[0]), _p(r[0]), _p(w[0]), _p(r[1]), _p(w[1]), _p(r[2]), _p(w[2]),
                                   _p(r[3]), _p(w[3]), float(T), float(muB), _p(out), _p(integ))
    if rc != 0:
        raise ValueError("oracle_df_generator_row: %d" % rc)
    return (out, integ) if with_integrals else out
