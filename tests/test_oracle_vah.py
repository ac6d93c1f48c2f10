"""CPU: the anisotropic-hydro (VAH, P_L matching) smooth-kernel restatement (oracle/cf_oracle.c, BASELINE config 5).  The
reference never runs this kernel, so it is pinned through its limits: isotropic, coefficient-free cells
must give the viscous-hydro oracle's equilibrium spectrum; the response to each coefficient must be linear and match a direct
evaluation of the formula on one momentum bin."""
import numpy as np
import pytest

from conftest import relerr
from is3d_amd import synth
from oracle import oracle


def test_equilibrium_limit_is_the_viscous_hydro_oracle(fx):
    """alpha_L = 1, Lambda = T, c_i = 0: f_a = f_eq, no cut on p.dsigma -> the viscous-hydro oracle with delta-f and outflow off.
    In 2+1D the VAH kernel multiplies the eta weights by the node spacing (smooth_kernels.cpp:2180-2188)."""
    for dim in (3, 2):
        n = 8 if dim == 3 else 3
        c = synth.synth_vah_surface(n, dim, seed=5)
        c["aL"][:] = 1.0
        c["Lambda"] = c["T"].copy()
        for k in ("c0", "c1", "c2", "c3", "c4"):
            c[k][:] = 0.0
        got = oracle.dN_pTdpTdphidy_vah(c, fx["pikp"], fx["grid"], dict(dimension=dim))
        vh = synth.synth_surface(n, dim, seed=5)
        ref = oracle.dN_pTdpTdphidy(vh, fx["pikp"], fx["grid"], fx["df"], dict(dimension=dim, df_mode=1, include_bulk_deltaf=0,
                                                                               include_shear_deltaf=0, outflow=0))
        scale = 1.0 if dim == 3 else (fx["grid"]["eta"][1] - fx["grid"]["eta"][0])
        assert relerr(got, ref * scale) < 1e-13


def test_one_bin_against_the_formula(fx):
    """One cell, one species: a direct numpy evaluation of :2271-2330 at a few bins."""
    c = synth.synth_vah_surface(1, 3, seed=77)
    sp = {k: v[:1] for k, v in fx["pikp"].items()}
    g = fx["grid"]
    got = oracle.dN_pTdpTdphidy_vah(c, sp, g, dict(dimension=3, regulate_deltaf=0)).reshape(len(g["y"]), len(g["phi"]), len(g["pT"]))
    q = {k: float(v[0]) for k, v in c.items()}
    tau = q["tau"]
    ut = np.sqrt(1 + q["ux"] ** 2 + q["uy"] ** 2 + tau ** 2 * q["un"] ** 2)
    u0 = np.sqrt(1 + q["ux"] ** 2 + q["uy"] ** 2)
    zt, zn = tau * q["un"] / u0, ut / (u0 * tau)
    Wt = (q["ux"] * q["Wx"] + q["uy"] * q["Wy"]) * ut / u0 ** 2
    Wn = Wt * q["un"] / ut
    m = float(sp["mass"][0])
    for iy, iphi, ipT in ((10, 0, 3), (13, 7, 12), (6, 20, 18)):
        y, phi, pT = g["y"][iy], g["phi"][iphi], g["pT"][ipT]
        mT = np.sqrt(m * m + pT * pT)
        pt, pn = mT * np.cosh(y - q["eta"]), mT / tau * np.sinh(y - q["eta"])
        px, py, t2pn = pT * np.cos(phi), pT * np.sin(phi), tau * tau * pn
        pds = pt * q["dat"] + px * q["dax"] + py * q["day"] + pn * q["dan"]
        pu = pt * ut - px * q["ux"] - py * q["uy"] - t2pn * q["un"]
        pz = pt * zt - t2pn * zn
        Ea = np.sqrt(pu * pu + (1 / q["aL"] ** 2 - 1) * pz * pz)
        fa = 1 / (np.exp(Ea / q["Lambda"]) - 1.0)
        fabar = 1 + fa
        Wp = pz * (Wt * pt - q["Wx"] * px - q["Wy"] * py - Wn * t2pn)
        pipp = (q["pitt"] * pt * pt + q["pixx"] * px * px + q["piyy"] * py * py + q["pinn"] * t2pn * t2pn
                + 2 * (-(q["pitx"] * px + q["pity"] * py) * pt + q["pixy"] * px * py + t2pn * (q["pixn"] * px + q["piyn"] * py - q["pitn"] * pt)))
        df = q["c3"] * Wp + q["c4"] * pipp + (q["c0"] * m * m + q["c1"] * pz * pz + q["c2"] * pu * pu) * q["bulkPi"]
        want = pds * fa * (1 + fabar * df) / (8 * np.pi ** 3 * 0.197327053 ** 3)
        assert abs(got[iy, iphi, ipT] / want - 1) < 1e-12
    # regulate_deltaf clamps fabar * df to [-1, 1]: blow the shear coefficient up and every bin stays within [0, 2] f_a p.dsigma
    c["c4"] *= 1.0e4
    reg = oracle.dN_pTdpTdphidy_vah(c, sp, g, dict(dimension=3, regulate_deltaf=1))
    c["c4"][:] = 0.0
    c["c3"][:] = 0.0
    c["bulkPi"][:] = 0.0
    base = oracle.dN_pTdpTdphidy_vah(c, sp, g, dict(dimension=3))
    assert (np.abs(reg) <= 2.0 * np.abs(base) * (1 + 1e-12)).all()
    with pytest.raises(RuntimeError):
        oracle.dN_pTdpTdphidy_vah(c, sp, g, dict(dimension=4))
