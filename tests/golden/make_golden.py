#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the build container; outputs are committed).

The reference ships no golden outputs for this path and may not be executed here, so
the vectors pin the CPU oracle from two independent directions instead:

  golden_highprec.npz  an INDEPENDENT restatement of smooth_kernels.cpp:106-349 in numpy long double
                       (80-bit), vectorised over bins, with the delta-f coefficients taken from
                       scipy's natural CubicSpline (an implementation unrelated to the oracle's
                       GSL restatement).  Inputs + spectra for a few hand-picked and seeded cells,
                       both dimensions, both df modes.
  golden_pins.json     scipy natural-spline values of the five coefficient tables at off-node
                       temperatures, and the closed-form config-1 known answers of SURVEY.md section 4.
  golden_64cell.npz    a 64-cell seeded 3+1D surface and a 16-cell 2+1D surface with full-grid
                       spectra from the CPU oracle (OMP_NUM_THREADS=1), plus their SHA-256 in
                       golden_pins.json: the regression pin the HIP path is compared against on the GPU box.
"""
import hashlib
import json
import os
import sys

import numpy as np
from scipy.interpolate import CubicSpline

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")

from is3d_amd import inputs, synth  # noqa: E402
from oracle import oracle  # noqa: E402

LD = np.longdouble
HBARC = LD("0.197327053")


def coefficients_scipy(df, df_mode, T):
    """deltafReader.cpp:337-358 with scipy's natural spline."""
    def S(name):
        return float(CubicSpline(df["T"], df[name], bc_type="natural")(T))
    T4 = T ** 4
    if df_mode == 1:
        return dict(c0=S("c0") / T4, c2=S("c2") / T4)
    return dict(F=S("F") * T, betabulk=S("betabulk") * T4, betapi=S("betapi") * T4)


def coefficients_bilinear_scipy(dff, df_mode, T, muB):
    """deltafReader.cpp:436-468 with scipy's multilinear RegularGridInterpolator (intended [muB][T] indexing)."""
    from scipy.interpolate import RegularGridInterpolator

    def S(name):
        return float(RegularGridInterpolator((dff["muB"], dff["T"]), dff["2d"][name], method="linear")([[muB, T]])[0])
    T3, T4 = T ** 3, T ** 4
    if df_mode == 1:
        return dict(c0=S("c0") / T4, c1=S("c1") / T3, c2=S("c2") / T4, c3=S("c3") / T4, c4=S("c4") / (T4 * T))
    return dict(F=S("F") * T, G=S("G"), betabulk=S("betabulk") * T4, betaV=S("betaV") * T3, betapi=S("betapi") * T4)


def highprec_spectrum(cells, sp, grid, df, opts):
    """Appendix A of SURVEY.md, long double, loops over cells only."""
    dim, dfm = opts["dimension"], opts["df_mode"]
    outflow, reg = opts.get("outflow", 1), opts.get("regulate_deltaf", 1)
    inc_bulk, inc_shear = opts.get("include_bulk_deltaf", 1), opts.get("include_shear_deltaf", 1)
    pT = grid["pT"].astype(LD)
    cosphi = np.cos(grid["phi"]).astype(LD)   # the reference forms cos/sin in double (:43-48)
    sinphi = np.sin(grid["phi"]).astype(LD)
    mass, sign, g = sp["mass"].astype(LD), sp["sign"].astype(LD), sp["degeneracy"].astype(LD)
    inc_b, inc_diff = opts.get("include_baryon", 0), opts.get("include_baryondiff_deltaf", 0)
    bar = sp["baryon"].astype(LD)
    ny = len(grid["y"]) if dim == 3 else 1
    nsp, npT, nphi = len(mass), len(pT), len(cosphi)
    out = np.zeros((ny, nphi, npT, nsp), dtype=LD)
    pref = (2 * LD(np.pi) if False else LD(2) * LD(np.pi)) * HBARC
    # the reference evaluates pow(2.0*M_PI*hbarC, -3) in double; M_PI is the double nearest to pi
    pref = LD(float(2.0 * np.pi * 0.197327053)) ** -3
    mT = np.sqrt(mass[:, None] ** 2 + pT[None, :] ** 2)                      # [s, i]
    for c in range(len(cells["tau"])):
        f = {k: LD(cells[k][c]) for k in synth.CELL_FIELDS}
        tau, tau2 = f["tau"], f["tau"] ** 2
        ux, uy, un = f["ux"], f["uy"], f["un"]
        ut = np.sqrt(1 + ux * ux + uy * uy + tau2 * un * un)
        dat, dax, day, dan = f["dat"], f["dax"], f["day"], f["dan"]
        if ut * dat + ux * dax + uy * day + un * dan <= 0:
            continue
        T, P, E = f["T"], f["P"], f["E"]
        pixx = pixy = pixn = piyy = piyn = pinn = pitn = pity = pitx = pitt = LD(0)
        if inc_shear:
            pixx, pixy, pixn, piyy, piyn = f["pixx"], f["pixy"], f["pixn"], f["piyy"], f["piyn"]
            utperp2 = 1 + ux * ux + uy * uy
            pinn = (pixx * (ux * ux - ut * ut) + piyy * (uy * uy - ut * ut) + 2 * (pixy * ux * uy + tau2 * un * (pixn * ux + piyn * uy))) / (tau2 * utperp2)
            pitn = (pixn * ux + piyn * uy + tau2 * pinn * un) / ut
            pity = (pixy * ux + piyy * uy + tau2 * piyn * un) / ut
            pitx = (pixx * ux + pixy * uy + tau2 * pixn * un) / ut
            pitt = (pitx * ux + pity * uy + tau2 * pitn * un) / ut
        bulkPi = f["bulkPi"] if inc_bulk else LD(0)
        muB = nB = Vx = Vy = Vn = Vt = LD(0)
        if inc_b and inc_diff:
            muB, nB = LD(cells["muB"][c]), LD(cells["nB"][c])
            Vx, Vy, Vn = LD(cells["Vx"][c]), LD(cells["Vy"][c]), LD(cells["Vn"][c])
            Vt = (Vx * ux + Vy * uy + tau2 * Vn * un) / ut
        alphaB, rho = muB / T, nB / (E + P)
        co = coefficients_bilinear_scipy(df, dfm, float(T), float(muB)) if inc_b else coefficients_scipy(df, dfm, float(T))
        b1 = c3 = c4 = LD(0)
        betaV = LD(1)
        if dfm == 1:
            shear = LD(0.5) / (T * T * (E + P))
            b0 = LD(co["c0"]) - LD(co["c2"])
            b2 = 4 * LD(co["c2"]) - LD(co["c0"])
            if inc_b:
                b1, c3, c4 = LD(co["c1"]), LD(co["c3"]), LD(co["c4"])
        else:
            shear = LD(0.5) / (LD(co["betapi"]) * T)
            b0 = LD(co["F"]) / (T * T * LD(co["betabulk"]))
            b2 = 1 / (3 * T * LD(co["betabulk"]))
            if inc_b:
                b1, betaV = LD(co["G"]) / LD(co["betabulk"]), LD(co["betaV"])
        if dim == 3:
            ys = grid["y"].astype(LD)
            etas, ws = np.array([f["eta"]], dtype=LD), np.array([1], dtype=LD)
        else:
            ys = np.array([0], dtype=LD)
            etas, ws = grid["eta"].astype(LD), grid["eta_w"].astype(LD)
        # axes: [y, eta, phi, pT, species]
        d = ys[:, None] - etas[None, :]
        ch, sh = np.cosh(d)[:, :, None, None, None], np.sinh(d)[:, :, None, None, None]
        w = ws[None, :, None, None, None]
        mTb = mT.T[None, None, None, :, :]
        m2 = (mass ** 2)[None, None, None, None, :]
        sg = sign[None, None, None, None, :]
        px = (pT[None, :] * cosphi[:, None])[None, None, :, :, None]
        py = (pT[None, :] * sinphi[:, None])[None, None, :, :, None]
        pt = mTb * ch
        pn = (mTb / tau) * sh
        t2pn = tau2 * pn
        pds = w * (pt * dat + px * dax + py * day + pn * dan)
        pu = pt * ut - px * ux - py * uy - t2pn * un
        bb = bar[None, None, None, None, :]
        feq = 1 / (np.exp(pu / T - bb * alphaB) + sg)
        Vp = Vt * pt - Vx * px - Vy * py - Vn * t2pn
        feqbar = 1 - sg * feq
        pipp = pitt * pt * pt + pixx * px * px + piyy * py * py + pinn * t2pn * t2pn + 2 * (
            -(pitx * px + pity * py) * pt + pixy * px * py + t2pn * (pixn * px + piyn * py - pitn * pt))
        if dfm == 1:
            dfc = feqbar * (shear * pipp + (b0 * m2 + (b1 * bb + b2 * pu) * pu) * bulkPi + (c3 * bb + c4 * pu) * Vp)
        else:
            dfc = feqbar * (shear * pipp / pu + (b0 * pu + b1 * bb + b2 * (pu - m2 / pu)) * bulkPi + (rho - bb / pu) * Vp / betaV)
        if reg:
            dfc = np.clip(dfc, -1, 1)
        term = pds * feq * (1 + dfc)
        if outflow:
            term = np.where(pds <= 0, LD(0), term)
        out += pref * g[None, None, None, :] * term.sum(axis=1)
    return out.reshape(-1)


def jonah_tables_ld(fq):
    """Deltaf_Data::compute_jonah_coefficients (deltafReader.cpp:222-297) in long double, vectorised."""
    T = LD(fq["T_avg"])
    r, w = fq["root2"].astype(LD)[None, :], fq["weight2"].astype(LD)[None, :]
    keep = fq["pdg_mass"] != 0.0
    mbar = (fq["pdg_mass"][keep].astype(LD) / T)[:, None]
    g = fq["pdg_degeneracy"][keep].astype(LD)
    sg = fq["pdg_sign"][keep].astype(LD)[:, None]
    Ebar = np.sqrt(r * r + mbar * mbar)
    feq = np.exp(r) / (np.exp(Ebar) + sg)

    def EP(lam):
        s2 = (1 + lam) ** 2
        em = np.sqrt(r * r * s2 + mbar * mbar)
        return np.sum(g * np.sum(w * em * feq, axis=1)), np.sum(g * np.sum(w * r * r * s2 / em * feq, axis=1)) / 3
    E0, P0 = EP(LD(0))
    lam = LD(-1) + np.arange(301).astype(LD) * (LD(3) / LD(300))
    z, bp = np.zeros(301, dtype=LD), np.zeros(301, dtype=LD)
    for i in range(301):
        Em, Pm = EP(lam[i])
        z[i] = E0 / Em
        bp[i] = (Pm / P0) * z[i] - 1
    return (lam * lam).astype(np.float64), z.astype(np.float64), bp.astype(np.float64)


def highprec_feqmod(cells, sp, grid, df, fq, opts):
    """calculate_dN_ptdptdphidy_feqmod (smooth_kernels.cpp:396-996), include_baryon = 0, in long double with numpy linear
    algebra and scipy splines -- independent of the oracle's C.  Cells whose feqmod breaks down (df_mode 3) or rows in the
    narrow window (3+1D, detA < 0.01) are NOT handled: the generator asserts that none occurs in the chosen cells."""
    dim, dfm = opts["dimension"], opts["df_mode"]
    outflow = opts.get("outflow", 1)
    pT = grid["pT"].astype(LD)
    cosphi, sinphi = np.cos(grid["phi"]).astype(LD), np.sin(grid["phi"]).astype(LD)
    mass, sign, gdeg = sp["mass"].astype(LD), sp["sign"].astype(LD), sp["degeneracy"].astype(LD)
    ny = len(grid["y"]) if dim == 3 else 1
    out = np.zeros((ny, len(cosphi), len(pT), len(mass)), dtype=LD)
    pref = LD(float(2.0 * np.pi * 0.197327053)) ** -3
    two_pi2_hbarC3 = LD(float(2.0 * np.pi ** 2 * 0.197327053 ** 3))
    mT = np.sqrt(mass[:, None] ** 2 + pT[None, :] ** 2)
    if dfm == 4:
        l2, zt, bpt = jonah_tables_ld(fq)
        S_l2, S_z = CubicSpline(bpt, l2, bc_type="natural"), CubicSpline(bpt, zt, bc_type="natural")
        bp_max = float(np.max(bpt))
    r1, w1, r2, w2 = (fq[k].astype(LD) for k in ("root1", "weight1", "root2", "weight2"))
    for c in range(len(cells["tau"])):
        f = {k: LD(cells[k][c]) for k in synth.CELL_FIELDS}
        tau, tau2 = f["tau"], f["tau"] ** 2
        ux, uy, un = f["ux"], f["uy"], f["un"]
        ut = np.sqrt(1 + ux * ux + uy * uy + tau2 * un * un)
        dat, dax, day, dan = f["dat"], f["dax"], f["day"], f["dan"]
        assert ut * dat + ux * dax + uy * day + un * dan > 0
        T, P, E = f["T"], f["P"], f["E"]
        pixx, pixy, pixn, piyy, piyn = f["pixx"], f["pixy"], f["pixn"], f["piyy"], f["piyn"]
        uperp, utperp = np.sqrt(ux * ux + uy * uy), np.sqrt(1 + ux * ux + uy * uy)
        pinn = (pixx * (ux * ux - ut * ut) + piyy * (uy * uy - ut * ut) + 2 * (pixy * ux * uy + tau2 * un * (pixn * ux + piyn * uy))) / (tau2 * utperp ** 2)
        pitn = (pixn * ux + piyn * uy + tau2 * pinn * un) / ut
        pity = (pixy * ux + piyy * uy + tau2 * piyn * un) / ut
        pitx = (pixx * ux + pixy * uy + tau2 * pixn * un) / ut
        pitt = (pitx * ux + pity * uy + tau2 * pitn * un) / ut
        bulkPi = f["bulkPi"]
        co = coefficients_scipy(df, 2, float(T))
        F, betabulk, betapi = LD(co["F"]), LD(co["betabulk"]), LD(co["betapi"])
        # contravariant Milne components (tau, x, y, eta) of the LRF triad, viscous_correction.cpp:8-27
        sinhL, coshL = tau * un / utperp, ut / utperp
        X = np.array([uperp * coshL, utperp * ux / uperp, utperp * uy / uperp, uperp * sinhL / tau], dtype=LD)
        Y = np.array([0, -uy / uperp, ux / uperp, 0], dtype=LD)
        Z = np.array([sinhL, 0, 0, coshL / tau], dtype=LD)
        pi = np.array([[pitt, pitx, pity, pitn], [pitx, pixx, pixy, pixn], [pity, pixy, piyy, piyn], [pitn, pixn, piyn, pinn]], dtype=LD)
        gdn = np.array([1, -1, -1, -tau2], dtype=LD)            # lowering: e_mu = g_{mu nu} e^nu
        tri = [X * gdn, Y * gdn, Z * gdn]
        piL = np.array([[a @ pi @ b for b in tri] for a in tri], dtype=LD)      # pi^{ij}_LRF = e^i_mu e^j_nu pi^{mu nu}
        if dfm == 4:
            if bulkPi < -P:
                bulkPi = -(1 - LD(1e-5)) * P
            elif bulkPi / P > bp_max:
                bulkPi = P * (LD(bp_max) - LD(1e-5))
            lam = np.sign(bulkPi) * np.sqrt(LD(float(S_l2(float(bulkPi / P)))))
            renorm, T_mod, bulk_mod = LD(float(S_z(float(bulkPi / P)))), T, lam
        else:
            T_mod, bulk_mod = T + bulkPi * F / betabulk, bulkPi / (3 * betabulk)
        A = np.eye(3, dtype=LD) * (1 + bulk_mod) + piL / (2 * betapi)
        A[2, 2] = 1 + bulk_mod - (piL[0, 0] + piL[1, 1]) / (2 * betapi)          # pizz_LRF = -(pixx_LRF + piyy_LRF), :114
        detA = np.linalg.det(A.astype(np.float64))
        Ainv = np.linalg.inv(A.astype(np.float64)).astype(LD)
        for _ in range(3):                                                        # Newton-Schulz polish in long double
            Ainv = Ainv @ (2 * np.eye(3, dtype=LD) - A @ Ainv)
        assert detA > 0.02, "generator cells must stay away from the breakdown branches"
        eta_scale = LD(detA) if (dim == 2 and fq["deta_min"] < detA < 1.0) else LD(1)
        if dim == 3:
            ys, etas, ws = grid["y"].astype(LD), np.array([f["eta"]], dtype=LD), np.array([1], dtype=LD)
        else:
            ys, etas, ws = np.array([0], dtype=LD), grid["eta"].astype(LD), grid["eta_w"].astype(LD)
        d = ys[:, None] - eta_scale * etas[None, :]
        ch, sh = np.cosh(d)[:, :, None, None, None], np.sinh(d)[:, :, None, None, None]
        w = ws[None, :, None, None, None]
        mTb = mT.T[None, None, None, :, :]
        px = (pT[None, :] * cosphi[:, None])[None, None, :, :, None]
        py = (pT[None, :] * sinphi[:, None])[None, None, :, :, None]
        pt, t2pn = mTb * ch, tau * mTb * sh
        pn = t2pn / tau2
        pds = w * (pt * dat + px * dax + py * day) + pn * dan                      # dsigma_eta outside the weight, :905
        pL = [-(e[0] * pt) + e[1] * px + e[2] * py + e[3] * t2pn for e in (X, Y, Z)]     # -X.p etc with p_mu lowered: :907-909
        pm = [Ainv[i, 0] * pL[0] + Ainv[i, 1] * pL[1] + Ainv[i, 2] * pL[2] for i in range(3)]
        Emod = np.sqrt((mass ** 2)[None, None, None, None, :] + pm[0] ** 2 + pm[1] ** 2 + pm[2] ** 2)
        if dfm == 3:
            mbar, mbm = mass / T, mass / T_mod
            neqf = T ** 3 / two_pi2_hbarC3

            def gt(kind, mb, r, wq):
                Eb = np.sqrt(r[None, :] ** 2 + mb[:, None] ** 2)
                q = np.exp(Eb) + sign[:, None]
                if kind == "neq":
                    return np.sum(wq * r * np.exp(r) / q, axis=1)
                return np.sum(wq * Eb * np.exp(r + Eb) / (q * q), axis=1)
            neq = neqf * gdeg * gt("neq", mbar, r1, w1)
            J20 = T * neqf * gdeg * gt("J20", mbar, r2, w2)
            n_lin = neq + (bulkPi / betabulk) * (neq + J20 * F / T / T)
            n_mod = (T_mod ** 3 / two_pi2_hbarC3) * gdeg * gt("neq", mbm, r1, w1)
            renorm = (n_lin / n_mod)[None, None, None, None, :]
        ren = np.abs(renorm) / (LD(detA) if dim == 3 else LD(1))
        fm = ren / (np.exp(Emod / T_mod) + sign[None, None, None, None, :])
        term = pds * fm
        if outflow:
            term = np.where(pds <= 0, LD(0), term)
        out += pref * gdeg[None, None, None, :] * term.sum(axis=1)
    return out.reshape(-1)


def hand_cells():
    """Toy cell of input/surface.dat with flow, shear and bulk switched on, plus two tilted cells."""
    h = 0.197327053
    base = dict(tau=0.5, eta=0.0, dat=1000.0, dax=0.0, day=0.0, dan=0.0, ux=0.0, uy=0.0, un=0.0, E=1.839 * h, T=0.786 * h,
                P=0.270 * h, pixx=0.0, pixy=0.0, pixn=0.0, piyy=0.0, piyn=0.0, bulkPi=0.0)
    c1 = dict(base, ux=0.4, pixx=0.01, piyy=-0.01, bulkPi=-0.002)
    c2 = dict(base, tau=3.0, eta=0.7, dat=2.0, dax=-0.5, day=0.3, dan=0.2, ux=0.3, uy=-0.2, un=0.05, T=0.145, pixx=0.004,
              pixy=-0.002, pixn=0.001, piyy=0.003, piyn=-0.0015, bulkPi=-0.001)
    c3 = dict(c2, dat=0.5, dax=1.5, eta=-1.3, T=0.159, bulkPi=0.003)   # partly inflowing: exercises the outflow cut
    cells = [base, c1, c2, c3]
    return {k: np.array([c[k] for c in cells], dtype=np.float64) for k in synth.CELL_FIELDS}


def sha256_arrays(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def main():
    g = inputs.grid()
    df = inputs.df_tables()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    sp = inputs.species("pikp")
    # ---- (a) high-precision restatement ----
    hp = {}
    hc = hand_cells()
    seeded3 = synth.synth_surface(3, 3, seed=7)
    seeded2 = synth.synth_surface(2, 2, seed=8)
    cases = []
    for name, cells, dim in [("hand3", hc, 3), ("hand2", {k: v[:2] for k, v in hc.items()}, 2), ("seed3", seeded3, 3), ("seed2", seeded2, 2)]:
        for dfm in (1, 2):
            for extra in ({}, {"outflow": 0, "regulate_deltaf": 0}):
                o = dict(dimension=dim, df_mode=dfm, **extra)
                key = "%s_df%d_%s" % (name, dfm, "raw" if extra else "std")
                res = highprec_spectrum(cells, sp, grid, df, o)
                hp[key] = res.astype(np.float64)
                cases.append(dict(key=key, cells=name, opts=o))
                chk = oracle.dN_pTdpTdphidy(cells, sp, grid, df, o)
                den = np.maximum(np.abs(hp[key]), 1e-280)
                print("%-22s oracle vs long-double restatement: max rel %.3e" % (key, np.max(np.abs(chk - hp[key]) / den)))
    # include_baryon = 1: bilinear (T, muB) coefficients, b mu_B / T in f_eq, baryon diffusion
    dff = inputs.df_tables_full()
    spb = inputs.species([211, 2212, -2212])
    sb3 = synth.synth_surface(3, 3, seed=17, baryon=True)
    sb2 = synth.synth_surface(2, 2, seed=18, baryon=True)
    for name, cells, dim in [("seedb3", sb3, 3), ("seedb2", sb2, 2)]:
        for dfm in (1, 2):
            for diff in (1, 0):
                o = dict(dimension=dim, df_mode=dfm, include_baryon=1, include_baryondiff_deltaf=diff)
                key = "%s_df%d_diff%d" % (name, dfm, diff)
                hp[key] = highprec_spectrum(cells, spb, grid, dff, o).astype(np.float64)
                cases.append(dict(key=key, cells=name, opts=o, species=[211, 2212, -2212]))
                chk = oracle.dN_pTdpTdphidy(cells, spb, grid, dff, o)
                den = np.maximum(np.abs(hp[key]), 1e-280)
                print("%-22s oracle vs long-double restatement: max rel %.3e" % (key, np.max(np.abs(chk - hp[key]) / den)))
    # modified equilibrium (df_mode 3 Mike, 4 Jonah): SURVEY.md 8f rank 3
    sf3 = synth.synth_surface(3, 3, seed=27)
    sf2 = synth.synth_surface(2, 2, seed=28)
    for name, cells, dim in [("seedf3", sf3, 3), ("seedf2", sf2, 2)]:
        fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
        for dfm in (4, 3):
            o = dict(dimension=dim, df_mode=dfm)
            key = "%s_feqmod%d" % (name, dfm)
            hp[key] = highprec_feqmod(cells, sp, grid, df, fq, o).astype(np.float64)
            cases.append(dict(key=key, cells=name, opts=o, feqmod=True))
            chk, nb = oracle.dN_pTdpTdphidy_feqmod(cells, sp, grid, df, fq, o)
            assert nb == 0
            den = np.maximum(np.abs(hp[key]), 1e-280)
            print("%-22s oracle vs long-double restatement: max rel %.3e" % (key, np.max(np.abs(chk - hp[key]) / den)))
    cells_all = [("hand3", hc), ("seed3", seeded3), ("seed2", seeded2), ("seedb3", sb3), ("seedb2", sb2), ("seedf3", sf3), ("seedf2", sf2)]
    for nm, cells in cells_all:
        for k in synth.CELL_FIELDS + (synth.BARYON_FIELDS if "muB" in cells else []):
            hp["cells_%s_%s" % (nm, k)] = cells[k]
    np.savez_compressed(os.path.join(HERE, "golden_highprec.npz"), **hp)

    # ---- (b) pins ----
    pins = {"highprec_cases": cases}
    temps = [0.1005, 0.1234, 0.1499, 0.155099063658, 0.1777, 0.19995]
    pins["spline_T"] = temps
    pins["spline_scipy_natural"] = {name: [float(CubicSpline(df["T"], df[name], bc_type="natural")(t)) for t in temps]
                                    for name in ["c0", "c2", "F", "betabulk", "betapi"]}
    pins["kat_config1"] = {  # (species index, ipT) -> [3+1D y=0, 3+1D y=2, 2+1D y=0]
        "0,0": [5.047391543445e+01, 9.930908970331e+00, 1.462829854545e+02],
        "0,15": [1.625826774482e+01, 3.646157407738e-02, 2.750583441242e+01],
        "1,0": [1.118725537687e+01, 6.095267834395e-03, 1.717785630520e+01],
        "1,15": [5.823771206295e+00, 2.946210229896e-04, 7.833067241457e+00],
        "2,0": [2.320620877593e+00, 4.865777041824e-07, 2.507196000893e+00],
        "2,15": [1.525660605278e+00, 8.037580265794e-08, 1.577221221821e+00]}
    # ---- (c) 64-cell regression pin from the oracle ----
    s3 = synth.synth_surface(64, 3, seed=20260064)
    s2 = synth.synth_surface(16, 2, seed=20260016)
    sp7 = inputs.species([211, 321, 2212, -2212, 3122, 333, 22])
    gold = {}
    for nm, cells, dim, spc in [("s3", s3, 3, sp7), ("s2", s2, 2, sp)]:
        for dfm in (1, 2):
            gold["%s_df%d" % (nm, dfm)] = oracle.dN_pTdpTdphidy(cells, spc, grid, df, dict(dimension=dim, df_mode=dfm))
    pins["golden_64cell"] = dict(seed3=20260064, seed2=20260016, species3=[211, 321, 2212, -2212, 3122, 333, 22], species2="pikp",
                                 sha256={k: sha256_arrays(v) for k, v in gold.items()})
    np.savez_compressed(os.path.join(HERE, "golden_64cell.npz"), **gold)
    with open(os.path.join(HERE, "golden_pins.json"), "w") as f:
        json.dump(pins, f, indent=1)
        f.write("\n")
    print("wrote golden_highprec.npz, golden_64cell.npz, golden_pins.json")


if __name__ == "__main__":
    main()
