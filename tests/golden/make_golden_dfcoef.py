#!/usr/bin/env python3
"""Build tests/golden/golden_dfcoef.npz -- REFERENCE-HELD numbers for the thermal-integral layer.

The reference ships the output of its own coefficient generator
(generate_delta_f_coefficients/urqmd/df_vh_dimensionless/src/deltaf_table.cpp:137-248, :296-395):
deltaf_coefficients/vh/urqmd/{c0,c1,c2,c3,c4,F,G,betabulk,betaV,betapi}.dat, 101 T x 81 mu_B rows each, printed `fixed`
with 6 decimals.  They were computed from the same particle list the smooth path reads (the generator's pdg.dat is
PDG/pdg-urqmd_v3.3+.dat plus one blank line) with the 64-point Gauss-Laguerre rule of
generate_delta_f_coefficients/urqmd/df_vh_dimensionless/gauss_laguerre/gla_roots_weights_64_points.txt.

This script stores DATA only, parsed here with plain Python (no reader of this repository, no oracle):
  root, weight   [alpha 0..4][64]   the generator's quadrature file (alpha = 1..4 are the ones it uses, deltaf_table.cpp:84-92)
  T, muB         the grid of the shipped tables
  iT, iB         a fixed sample of grid rows: every 5th temperature x every 8th chemical potential + the four corners' neighbours
  shipped        [10][len(iB)][len(iT)]  the shipped values at the sample, in the order of `names`
  text           the same as the printed strings (what the byte-level comparison uses)
The particle list itself is already a fixture (is3d_amd/data/inputs_urqmd.json, "pdg_urqmd").

Run in the build container only:  python tests/golden/make_golden_dfcoef.py
"""
import os

import numpy as np

REF = "/root/reference"
GEN = os.path.join(REF, "generate_delta_f_coefficients/urqmd/df_vh_dimensionless")
NAMES = ["c0", "c1", "c2", "c3", "c4", "F", "G", "betabulk", "betaV", "betapi"]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_dfcoef.npz")


def read_gla(path, n_alpha_keep=5):
    tok = open(path).read().split()
    n_alpha, n_pts = int(tok[0]), int(tok[1])
    body = tok[2:]
    root, weight = np.zeros((n_alpha, n_pts)), np.zeros((n_alpha, n_pts))
    k = 0
    for a in range(n_alpha):
        for j in range(n_pts):
            assert int(body[k]) == a
            root[a, j], weight[a, j] = float(body[k + 1]), float(body[k + 2])
            k += 3
    return root[:n_alpha_keep], weight[:n_alpha_keep]


def read_table_text(path):
    with open(path) as f:
        nT, nB = int(f.readline()), int(f.readline())
        f.readline()
        rows = [f.readline().split() for _ in range(nT * nB)]
    T = np.array([float(r[0]) for r in rows[:nT]])
    B = np.array([float(rows[i * nT][1]) for i in range(nB)])
    txt = np.array([r[2] for r in rows]).reshape(nB, nT)
    return T, B, txt


def main():
    root, weight = read_gla(os.path.join(GEN, "gauss_laguerre/gla_roots_weights_64_points.txt"))
    iT = np.array(sorted(set(range(0, 101, 5)) | {1, 99}))
    iB = np.array(sorted(set(range(0, 81, 8)) | {1, 20, 50, 79}))      # mu_B = 0, 0.2 and 0.5 GeV are rows 0, 20, 50
    text, T, B = [], None, None
    for n in NAMES:
        T, B, txt = read_table_text(os.path.join(REF, "deltaf_coefficients/vh/urqmd", n + ".dat"))
        text.append(txt[np.ix_(iB, iT)])
    text = np.array(text)
    np.savez_compressed(OUT, names=np.array(NAMES), root=root, weight=weight, T=T, muB=B, iT=iT, iB=iB,
                        shipped=text.astype(np.float64), text=text)
    print("wrote", OUT, text.shape)


if __name__ == "__main__":
    main()
