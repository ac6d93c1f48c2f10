#!/usr/bin/env python3
"""Build is3d_amd/data/inputs_urqmd.json (+ df_urqmd_full.npz) -- the INPUT DATA the hot path needs on a box that has no
/root/reference (GPU box, bench.py, smoke()).

It is data only: quadrature nodes/weights, the mu_B = 0 rows of the delta-f coefficient tables and
(mc_id, mass, gspin, baryon) of the hadron list, parsed from the reference's shipped data files
with small independent Python parsers that follow the reference readers' semantics:

  tables/*.dat, PDG/chosen_particles*.dat   readBlockData  (src/cpp/arsenal.cpp:406-453): a row counts
                                            only if its line is newline-terminated
  deltaf_coefficients/vh/urqmd/*.dat        Deltaf_Data::load_df_coefficient_data (src/cpp/deltafReader.cpp:120-197)
  PDG/pdg-urqmd_v3.3+.dat                   PDG_Data::read_resonances_conventional (src/cpp/readindata.cpp:1440-1568)
  deltaf_coefficients/vah/c{0..4}_vah1.dat  the anisotropic-hydro branch of load_df_coefficient_data in the CUDA tree
                                            (src/cuda/deltafReader.cu:60-82 names, :104-112 header, :196-213 scan order)

Run here (container) only:  python tools/make_inputs.py
"""
import json
import os
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "is3d_amd", "data", "inputs_urqmd.json")


def read_block(path):
    """readBlockData semantics: the last fragment (no trailing newline) is dropped."""
    with open(path, "r", newline="") as f:
        text = f.read()
    lines = text.split("\n")
    lines = lines[:-1]  # what follows the last '\n' is never pushed (arsenal.cpp:441-449)
    rows = [[float(tok) for tok in ln.split()] for ln in lines]
    ncol = len(rows[0])
    return [r[:ncol] for r in rows]


def read_df_table(path, n_keep_muB=1):
    with open(path) as f:
        nT = int(f.readline())
        nB = int(f.readline())
        f.readline()  # label line
        T, val = [], []
        for iB in range(n_keep_muB):
            for iT in range(nT):
                t, mub, v = (float(x) for x in f.readline().split())
                if iB == 0:
                    T.append(t)
                    val.append(v)
    assert nB >= 1
    return T, val


def read_df_table_full(path):
    """all mu_B rows: -> T[nT], muB[nB], values[nB][nT] (deltafReader.cpp:168-196 storage order)"""
    import numpy as np
    with open(path) as f:
        nT = int(f.readline())
        nB = int(f.readline())
        f.readline()
        a = np.array([[float(x) for x in f.readline().split()] for _ in range(nT * nB)])
    return a[:nT, 0].copy(), a[::nT, 1].copy(), a[:, 2].reshape(nB, nT).copy()


def read_vah_table(path):
    """deltaf_coefficients/vah/c*_vah1.dat: line 1 = number of Lambda nodes, line 2 = number of alpha_L nodes, one label line,
    then rows "Lambda [fm^-1]  alpha_L  value" with alpha_L OUTER and Lambda inner (src/cuda/deltafReader.cu:104-112, :196-213)
    -> L[nL], aL[naL], values[naL][nL]."""
    import numpy as np
    with open(path) as f:
        nL = int(f.readline())
        naL = int(f.readline())
        f.readline()
        a = np.array([[float(x) for x in f.readline().split()] for _ in range(nL * naL)])
    return a[(naL - 1) * nL:, 0].copy(), a[nL - 1::nL, 1].copy(), a[:, 2].reshape(naL, nL).copy()   # the arrays as the LAST rows leave them


def read_pdg(path):
    """Token stream; antibaryon appended after each baryon; sign from baryon parity."""
    with open(path) as f:
        toks = f.read().split()
    out, i = [], 0
    while i < len(toks):
        mc_id, name, mass, width, gspin, baryon, strange, charm, bottom, gisospin, charge, decays = toks[i:i + 12]
        i += 12 + 8 * int(decays)
        p = dict(mc_id=int(mc_id), name=name, mass=float(mass), gspin=int(gspin), baryon=int(baryon))
        out.append(p)
        if p["baryon"] > 0:
            out.append(dict(mc_id=-p["mc_id"], name="Anti-baryon-" + name, mass=p["mass"], gspin=p["gspin"], baryon=-p["baryon"]))
    for p in out:
        p["sign"] = -1 if p["baryon"] % 2 == 0 else 1  # readindata.cpp:1544-1545
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (run in the build container)")
    d = {"source": "derekeverett/iS3D data files (MIT), parsed by tools/make_inputs.py", "hbarC": 0.197327053}
    grids = {}
    for key, rel in [("pT", "tables/pT_gauss_legendre_table.dat"), ("phi", "tables/phi_gauss_legendre_table.dat"),
                     ("y", "tables/y_trapezoid_table_21pt.dat"), ("eta", "tables/eta/eta_trapezoid_table_241pt.dat")]:
        rows = read_block(os.path.join(REF, rel))
        grids[key] = {"file": rel, "x": [r[0] for r in rows], "w": [r[1] for r in rows]}
    d["grids"] = grids
    df = {}
    for name in ["c0", "c2", "F", "betabulk", "betapi"]:
        T, v = read_df_table(os.path.join(REF, "deltaf_coefficients/vh/urqmd", name + ".dat"))
        df["T"] = T
        df[name] = v
    d["df_urqmd_muB0"] = df
    pdg = read_pdg(os.path.join(REF, "PDG/pdg-urqmd_v3.3+.dat"))
    d["pdg_urqmd"] = [[p["mc_id"], p["mass"], p["gspin"], p["baryon"], p["sign"]] for p in pdg]
    d["pdg_urqmd_columns"] = ["mc_id", "mass", "gspin", "baryon", "sign"]
    for key, rel in [("chosen_pikp", "PDG/chosen_particles_pikp.dat"), ("chosen_urqmd", "PDG/chosen_particles_urqmd_v3.3+.dat")]:
        d[key] = [int(r[0]) for r in read_block(os.path.join(REF, rel))]
    # generalized Gauss-Laguerre roots/weights for alpha = 1, 2, 3 (Gauss_Laguerre::load_roots_and_weights,
    # src/cpp/readindata.cpp:24-53): first line "n_alpha n_points", then rows "alpha root weight"
    with open(os.path.join(REF, "tables/gla_roots_weights_32_points.txt")) as f:
        n_alpha, n_pts = (int(x) for x in f.readline().split())
        rows = [f.readline().split() for _ in range(n_alpha * n_pts)]
    gla = {}
    for al in (1, 2, 3):
        sel = rows[al * n_pts:(al + 1) * n_pts]
        assert all(int(r[0]) == al for r in sel)
        gla["root%d" % al] = [float(r[1]) for r in sel]
        gla["weight%d" % al] = [float(r[2]) for r in sel]
    d["gla_32"] = gla
    with open(OUT, "w") as f:
        json.dump(d, f, separators=(",", ":"))
        f.write("\n")
    # full (mu_B, T) tables for the include_baryon = 1 branch (bilinear interpolation): binary, 10 x 81 x 101
    import numpy as np
    full = {}
    for name in ["c0", "c1", "c2", "c3", "c4", "F", "G", "betabulk", "betaV", "betapi"]:
        T, muB, v = read_df_table_full(os.path.join(REF, "deltaf_coefficients/vh/urqmd", name + ".dat"))
        full["T"], full["muB"], full[name] = T, muB, v
    np.savez_compressed(os.path.join(os.path.dirname(OUT), "df_urqmd_full.npz"), **full)
    # anisotropic-hydro 14-moment coefficient tables (BASELINE config 5): 5 x 180 x 80
    vah = {}
    for k in range(5):
        L, aL, v = read_vah_table(os.path.join(REF, "deltaf_coefficients/vah/c%d_vah1.dat" % k))
        vah["L"], vah["aL"], vah["c%d" % k] = L, aL, v
    np.savez_compressed(os.path.join(os.path.dirname(OUT), "df_vah.npz"), **vah)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(pdg), "pdg entries;",
          {k: len(v["x"]) for k, v in grids.items()}, len(d["chosen_urqmd"]), "chosen")


if __name__ == "__main__":
    main()
