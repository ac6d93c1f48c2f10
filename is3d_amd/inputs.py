"""Input bundles for tests/bench on boxes without the reference tree: grids, the mu_B = 0 rows of the
urqmd delta-f coefficient tables and the hadron list, from is3d_amd/data/inputs_urqmd.json (package data, built by
tools/make_inputs.py from the reference's data files)."""
import json
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
FIXTURE = os.path.join(_DATA, "inputs_urqmd.json")
_CACHE = None


def load_fixture():
    global _CACHE
    if _CACHE is None:
        with open(FIXTURE) as f:
            _CACHE = json.load(f)
    return _CACHE


def grid():
    """pT/phi/y/eta nodes (+ weights) of iS3D.cpp:161-167's tables."""
    g = load_fixture()["grids"]
    return dict(pT=np.array(g["pT"]["x"]), pT_w=np.array(g["pT"]["w"]), phi=np.array(g["phi"]["x"]),
                phi_w=np.array(g["phi"]["w"]), y=np.array(g["y"]["x"]), y_w=np.array(g["y"]["w"]),
                eta=np.array(g["eta"]["x"]), eta_w=np.array(g["eta"]["w"]))


def df_tables():
    d = load_fixture()["df_urqmd_muB0"]
    return {k: np.array(d[k]) for k in ["T", "c0", "c2", "F", "betabulk", "betapi"]}


DF_NAMES_2D = ["c0", "c1", "c2", "c3", "c4", "F", "G", "betabulk", "betaV", "betapi"]


def df_tables_full():
    """df_tables() plus the full (mu_B, T) grids of all ten coefficient tables (include_baryon = 1):
    keys T, muB and one [n_muB][n_T] array per name in DF_NAMES_2D under key "2d"."""
    z = np.load(os.path.join(_DATA, "df_urqmd_full.npz"))
    d = df_tables()
    assert np.array_equal(d["T"], z["T"])
    d["muB"] = z["muB"].copy()
    d["2d"] = {k: np.ascontiguousarray(z[k]) for k in DF_NAMES_2D}
    return d


def vah_df_tables():
    """The anisotropic-hydro (P_L matching) 14-moment coefficient tables deltaf_coefficients/vah/c{0..4}_vah1.dat as the CUDA tree's
    reader holds them (src/cuda/deltafReader.cu:60-82, :196-213): L [fm^-1] (80 nodes), aL (180 nodes), c0..c4 each [n_aL][n_L] in file
    units (the per-cell values are divided by hbarc^3 after the interpolation, :262-266)."""
    z = np.load(os.path.join(_DATA, "df_vah.npz"))
    return {k: np.ascontiguousarray(z[k]) for k in ("L", "aL", "c0", "c1", "c2", "c3", "c4")}


def species(which="pikp"):
    """which: 'pikp' (chosen_particles_pikp.dat) | 'urqmd' (chosen_particles_urqmd_v3.3+.dat, 305) |
    list of mc_ids.  Order = order of the chosen list (emissionfunction.cpp:336-351)."""
    fx = load_fixture()
    pdg = {int(r[0]): r for r in fx["pdg_urqmd"]}
    ids = fx["chosen_pikp"] if which == "pikp" else fx["chosen_urqmd"] if which == "urqmd" else list(which)
    rows = [pdg[int(i)] for i in ids]
    return dict(mc_id=np.array([r[0] for r in rows], dtype=np.int64), mass=np.array([r[1] for r in rows], dtype=np.float64),
                degeneracy=np.array([r[2] for r in rows], dtype=np.float64), baryon=np.array([r[3] for r in rows], dtype=np.float64),
                sign=np.array([r[4] for r in rows], dtype=np.float64))


def feqmod_tables(T_avg, deta_min=1.e-5, mass_pion0=0.138):
    """What the modified-equilibrium path (df_mode 3, 4) needs besides the coefficient tables: generalized
    Gauss-Laguerre nodes (alpha = 1, 2; tables/gla_roots_weights_32_points.txt), ALL species of the PDG file
    (the Jonah z(Pi/P), lambda(Pi/P) tables sum over them, deltafReader.cpp:249-265), the surface-averaged
    temperature as read back from average_thermodynamic_quantities.dat (15 significant digits) and the
    parameters deta_min, mass_pion0 (iS3D_parameters.dat)."""
    fx = load_fixture()
    g = fx["gla_32"]
    pdg = np.array(fx["pdg_urqmd"], dtype=np.float64)
    return dict(root1=np.array(g["root1"]), weight1=np.array(g["weight1"]), root2=np.array(g["root2"]), weight2=np.array(g["weight2"]),
                root3=np.array(g["root3"]), weight3=np.array(g["weight3"]),
                pdg_mass=pdg[:, 1].copy(), pdg_degeneracy=pdg[:, 2].copy(), pdg_sign=pdg[:, 4].copy(),
                T_avg=float("%.15g" % T_avg), deta_min=float(deta_min), mass_pion0=float(mass_pion0))


def surface_average_T(cells):
    """Surface-volume weighted temperature of readindata.cpp:422-450 (what Plasma::load_thermodynamic_averages reads back)."""
    ut = np.sqrt(1 + cells["ux"] ** 2 + cells["uy"] ** 2 + cells["tau"] ** 2 * cells["un"] ** 2)
    uds = ut * cells["dat"] + cells["ux"] * cells["dax"] + cells["uy"] * cells["day"] + cells["un"] * cells["dan"]
    dsds = cells["dat"] ** 2 - cells["dax"] ** 2 - cells["day"] ** 2 - cells["dan"] ** 2 / cells["tau"] ** 2
    mag = np.abs(uds) + np.sqrt(np.abs(uds * uds - dsds))
    return float(np.sum(cells["T"] * mag) / np.sum(mag))


def surface_averages(cells):
    """(T, E, P, muB, nB) surface-volume weighted averages as the readers accumulate them (readindata.cpp:422-466) and as
    Plasma::load_thermodynamic_averages reads them back from average_thermodynamic_quantities.dat (15 significant digits)."""
    ut = np.sqrt(1 + cells["ux"] ** 2 + cells["uy"] ** 2 + cells["tau"] ** 2 * cells["un"] ** 2)
    uds = ut * cells["dat"] + cells["ux"] * cells["dax"] + cells["uy"] * cells["day"] + cells["un"] * cells["dan"]
    dsds = cells["dat"] ** 2 - cells["dax"] ** 2 - cells["day"] ** 2 - cells["dan"] ** 2 / cells["tau"] ** 2
    mag = np.abs(uds) + np.sqrt(np.abs(uds * uds - dsds))
    out = []
    for k in ("T", "E", "P", "muB", "nB"):
        v = cells.get(k)
        out.append(float("%.15g" % (np.sum(v * mag) / np.sum(mag))) if v is not None else 0.0)
    return tuple(out)
