#!/usr/bin/env python3
"""Generate tests/golden/golden_vah.npz (run in the build container; the output is committed).

The reference holds no fixture for the anisotropic-hydro path (its kernel is never called, src/cpp never loads the VAH tables), so -- as
for the viscous-hydro path (make_golden.py) -- the vectors pin the CPU oracle AND the device from an independent direction:

  coefficients   per-cell c0..c4 for 256 (Lambda, alpha_L) pairs inside the table grid from scipy's multilinear RegularGridInterpolator
                 (an implementation unrelated to the oracle's restatement of src/cuda/deltafReader.cu:224-278), divided by hbarc^3
  spectra        an INDEPENDENT numpy long-double (80-bit) restatement of calculate_dN_pTdpTdphidy_VAH_PL
                 (src/cpp/emissionfunction_smooth_kernels.cpp:2208-2346), vectorised over bins, for 12 seeded 3+1D cells and 4 seeded 2+1D cells
                 on a reduced momentum grid, with the scipy coefficients, regulate_deltaf on and off
"""
import os
import sys

import numpy as np
from scipy.interpolate import RegularGridInterpolator

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from is3d_amd import inputs, synth  # noqa: E402

LD = np.longdouble
HBARC = LD("0.197327053")


def coefficients_scipy(tab, Lambda_GeV, aL):
    x = np.asarray(Lambda_GeV, dtype=np.float64) / float(HBARC)
    pts = np.column_stack([np.asarray(aL, dtype=np.float64), x])
    out = {}
    for k in range(5):
        r = RegularGridInterpolator((tab["aL"], tab["L"]), tab["c%d" % k], method="linear")
        out["c%d" % k] = r(pts) / float(HBARC) ** 3
    return out


def highprec_vah(c, sp, g, dim, regulate, bulk=True, shear=True):
    """smooth_kernels.cpp:2208-2346 in long double; returns dN[iy][iphi][ipT][ipart] flattened species-fastest (:2340)."""
    pT, phi = g["pT"].astype(LD), g["phi"].astype(LD)
    if dim == 3:
        yv = g["y"].astype(LD)
    else:
        yv = np.zeros(1, dtype=LD)
        etav = g["eta"].astype(LD)
        etaw = g["eta_w"].astype(LD) * (etav[1] - etav[0])                       # :2178-2187
    pre = 1 / (8 * LD(np.pi) ** 3) / HBARC ** 3
    npart, npT, nphi, ny = len(sp["mass"]), len(pT), len(phi), len(yv)
    out = np.zeros((ny, nphi, npT, npart), dtype=LD)
    cosphi, sinphi = np.cos(phi), np.sin(phi)
    for ic in range(len(c["tau"])):
        f = {k: LD(v[ic]) for k, v in c.items()}
        tau = f["tau"]; tau2 = tau * tau
        ux, uy, un = f["ux"], f["uy"], f["un"]
        ut = np.sqrt(1 + ux * ux + uy * uy + tau2 * un * un)
        u0 = np.sqrt(1 + ux * ux + uy * uy)
        zt, zn = tau * un / u0, ut / (u0 * tau)
        Wt = (ux * f["Wx"] + uy * f["Wy"]) * ut / (u0 * u0)
        Wn = Wt * un / ut
        xiL = 1 / (f["aL"] * f["aL"]) - 1
        if dim == 3:
            etas, ws = np.array([f["eta"]], dtype=LD), np.ones(1, dtype=LD)
        else:
            etas, ws = etav, etaw
        for ip in range(npart):
            m = LD(sp["mass"][ip]); s = LD(sp["sign"][ip]); gdeg = LD(sp["degeneracy"][ip])
            mT = np.sqrt(m * m + pT * pT)                                         # [ipT]
            d = yv[:, None] - etas[None, :]                                       # [iy][ieta]
            pt = mT[None, None, :, None] * np.cosh(d)[:, None, None, :]           # [iy][1][ipT][ieta]
            t2pn = tau2 * (mT[None, None, :, None] / tau) * np.sinh(d)[:, None, None, :]
            pn = t2pn / tau2
            px = (pT[None, :] * cosphi[:, None])[None, :, :, None]                # [1][iphi][ipT][1]
            py = (pT[None, :] * sinphi[:, None])[None, :, :, None]
            pds = pt * f["dat"] + px * f["dax"] + py * f["day"] + pn * f["dan"]
            pu = pt * ut - px * ux - py * uy - t2pn * un
            pz = pt * zt - t2pn * zn
            Ea = np.sqrt(pu * pu + xiL * pz * pz)
            fa = 1 / (np.exp(Ea / f["Lambda"]) + s)
            fabar = 1 - s * fa
            df = np.zeros_like(Ea)
            if shear:
                Wp = pz * (Wt * pt - f["Wx"] * px - f["Wy"] * py - Wn * t2pn)
                pipp = (f["pitt"] * pt * pt + f["pixx"] * px * px + f["piyy"] * py * py + f["pinn"] * t2pn * t2pn
                        + 2 * (-(f["pitx"] * px + f["pity"] * py) * pt + f["pixy"] * px * py + t2pn * (f["pixn"] * px + f["piyn"] * py - f["pitn"] * pt)))
                df = df + f["c3"] * Wp + f["c4"] * pipp
            if bulk:
                df = df + (f["c0"] * m * m + f["c1"] * pz * pz + f["c2"] * pu * pu) * f["bulkPi"]
            corr = np.clip(fabar * df, -1, 1) if regulate else fabar * df
            out[:, :, :, ip] += pre * gdeg * np.sum(ws[None, None, None, :] * pds * fa * (1 + corr), axis=3)
    return out.reshape(-1).astype(np.float64)


def main():
    tab = inputs.vah_df_tables()
    rng = np.random.default_rng(20260003)
    lam = (0.6 + 0.6 * rng.random(256)) * float(HBARC)
    al = 0.2 + 1.7 * rng.random(256)
    coef = coefficients_scipy(tab, lam, al)
    g = inputs.grid()
    out = dict(coef_Lambda=lam, coef_aL=al, **{"coef_" + k: v for k, v in coef.items()})
    sp = inputs.species([211, 2212, 333])
    for dim, n, seed in ((3, 12, 31), (2, 4, 32)):
        c = synth.synth_vah_surface(n, dim, seed=seed)
        cs = coefficients_scipy(tab, c["Lambda"], c["aL"])
        c = dict(c, **cs)
        grid = dict(pT=g["pT"][::4], phi=g["phi"][::3], y=g["y"][::2], eta=g["eta"][::8], eta_w=g["eta_w"][::8])
        for k, v in c.items():
            out["cells%d_%s" % (dim, k)] = v
        for k, v in grid.items():
            out["grid%d_%s" % (dim, k)] = v
        for reg in (1, 0):
            out["dN%d_reg%d" % (dim, reg)] = highprec_vah(c, sp, grid, dim, bool(reg))
    out["species"] = np.array([211, 2212, 333])
    np.savez_compressed(os.path.join(HERE, "golden_vah.npz"), **out)
    print("wrote golden_vah.npz", {k: v.shape for k, v in out.items() if k.startswith("dN")})


if __name__ == "__main__":
    main()
