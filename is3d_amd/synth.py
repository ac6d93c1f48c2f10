"""Seeded synthetic freezeout surfaces for BASELINE.json configs 2-4.

One 64-bit counter-based generator, stated here so the surfaces are reproducible anywhere:
    splitmix64(k) = mix(seed + (k + 1) * 0x9E3779B97F4A7C15)
    mix(z): z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9; z = (z ^ (z >> 27)) * 0x94D049BB133111EB; z ^ (z >> 31)
    U[0,1) = (splitmix64 >> 11) * 2**-53
Cell c draws its uniforms from counters k = 32*c + slot (slot < 32); N(0,1) by Box-Muller on two
uniforms.  All cells have u.dsigma > 0 and T inside [0.140, 0.160] GeV, so neither the skipped-cell
path nor the coefficient-table domain error of the reference can trigger.

Arrays are returned in the units the kernel consumes (GeV, GeV/fm^3; i.e. AFTER the reader's
multiplication by hbar*c, src/cpp/readindata.cpp:367-410); write_surface_dat() divides again so that
the text file is a valid mode-1 `input/surface.dat`.
"""
import numpy as np

HBARC = 0.197327053  # src/cpp/iS3D.h:9

SEED_CONFIG2 = 20260001
SEED_CONFIG3 = 20260002

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# the 18 per-cell arrays of the vhydro path, in the order the kernel boundary lists them
CELL_FIELDS = ["tau", "eta", "dat", "dax", "day", "dan", "ux", "uy", "un", "T", "P", "E",
               "pixx", "pixy", "pixn", "piyy", "piyn", "bulkPi"]


def _uniform(seed, counters):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (counters.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


BARYON_FIELDS = ["muB", "nB", "Vx", "Vy", "Vn"]


def synth_surface(n_cells, dimension, seed=None, first_cell=0, baryon=False):
    """Cells [first_cell, first_cell + n_cells) of the infinite seeded surface -> dict of float64 arrays
    (CELL_FIELDS plus x, y).  A rank of a sharded run asks for its own slice directly."""
    if dimension not in (2, 3):
        raise ValueError("dimension must be 2 or 3")
    if seed is None:
        seed = SEED_CONFIG2 if dimension == 2 else SEED_CONFIG3
    c = np.arange(first_cell, first_cell + n_cells, dtype=np.uint64) * np.uint64(32)

    def U(slot):
        return _uniform(seed, c + np.uint64(slot))

    def N(slot):
        u1, u2 = 1.0 - U(slot), U(slot + 1)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)

    s = {}
    tau = 1.0 + 9.0 * U(0)
    r = 8.0 * U(1)
    phis = 2.0 * np.pi * U(2)
    s["tau"] = tau
    s["x"] = r * np.cos(phis)
    s["y"] = r * np.sin(phis)
    rho = 0.08 * r * (1.0 + 0.1 * np.cos(2.0 * phis))
    s["ux"] = np.sinh(rho) * np.cos(phis)
    s["uy"] = np.sinh(rho) * np.sin(phis)
    dat = 0.02 * tau * (0.5 + U(3))
    s["dat"] = dat
    s["dax"] = -0.3 * dat * U(4) * np.cos(phis)
    s["day"] = -0.3 * dat * U(5) * np.sin(phis)
    T = 0.140 + 0.020 * U(6)
    P = 0.08 * (T / 0.15) ** 4
    E = 3.5 * P
    s["T"], s["P"], s["E"] = T, P, E
    s["pixx"] = 0.02 * (E + P) * N(7)
    s["pixy"] = 0.02 * (E + P) * N(9)
    s["piyy"] = 0.02 * (E + P) * N(11)
    s["bulkPi"] = -0.02 * P * U(13)
    if dimension == 2:
        z = np.zeros(n_cells)
        s["eta"], s["un"], s["dan"], s["pixn"], s["piyn"] = z, z.copy(), z.copy(), z.copy(), z.copy()
    else:
        eta = -4.0 + 8.0 * U(14)
        s["eta"] = eta
        s["un"] = 0.05 * np.tanh(eta) / tau
        s["dan"] = 0.1 * dat * (2.0 * U(15) - 1.0)
        s["pixn"] = 0.02 * (E + P) * N(16) / tau
        s["piyn"] = 0.02 * (E + P) * N(18) / tau
    if baryon:
        # include_baryon = 1 extras: mu_B inside the [0, 0.8] GeV table, net baryon density, diffusion current V^mu
        s["muB"] = 0.05 + 0.35 * U(20)
        s["nB"] = 0.02 + 0.08 * U(21)
        s["Vx"] = 0.002 * N(22)
        s["Vy"] = 0.002 * N(24)
        s["Vn"] = (0.002 * N(26) / tau) if dimension == 3 else np.zeros(n_cells)
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in s.items()}


VAH_FIELDS = ["tau", "eta", "ux", "uy", "un", "dat", "dax", "day", "dan", "T", "pitt", "pitx", "pity", "pitn", "pixx", "pixy", "pixn",
              "piyy", "piyn", "pinn", "bulkPi", "Wx", "Wy", "Lambda", "aL", "c0", "c1", "c2", "c3", "c4"]


def synth_vah_surface(n_cells, dimension, seed=None, first_cell=0):
    """Anisotropic-hydro (P_L matching) cells for the VAH smooth kernel (smooth_kernels.cpp:2140-2393): the viscous-hydro
    synthetic surface plus all ten pi_perp^{mu nu} components (orthogonal to u, traceless, as the kernel reconstructs them on
    the viscous-hydro path), W^x, W^y, the anisotropic variables Lambda ~ T and alpha_L in [0.7, 1.2], and per-cell 14-moment
    coefficients c0..c4 of the magnitude the viscous-hydro tables give (delta-f of order 0.1)."""
    s = synth_surface(n_cells, dimension, seed=seed, first_cell=first_cell)
    if seed is None:
        seed = SEED_CONFIG2 if dimension == 2 else SEED_CONFIG3
    c = np.arange(first_cell, first_cell + n_cells, dtype=np.uint64) * np.uint64(32)

    def U(slot):
        return _uniform(seed ^ 0x5A5A5A5A, c + np.uint64(slot))

    tau, ux, uy, un = s["tau"], s["ux"], s["uy"], s["un"]
    tau2 = tau * tau
    ut = np.sqrt(1.0 + ux * ux + uy * uy + tau2 * un * un)
    utperp2 = 1.0 + ux * ux + uy * uy
    pixx, pixy, pixn, piyy, piyn = s["pixx"], s["pixy"], s["pixn"], s["piyy"], s["piyn"]
    pinn = (pixx * (ux * ux - ut * ut) + piyy * (uy * uy - ut * ut) + 2.0 * (pixy * ux * uy + tau2 * un * (pixn * ux + piyn * uy))) / (tau2 * utperp2)
    pitn = (pixn * ux + piyn * uy + tau2 * pinn * un) / ut
    pity = (pixy * ux + piyy * uy + tau2 * piyn * un) / ut
    pitx = (pixx * ux + pixy * uy + tau2 * pixn * un) / ut
    pitt = (pitx * ux + pity * uy + tau2 * pitn * un) / ut
    T, P, E = s["T"], s["P"], s["E"]
    v = {k: s[k] for k in ["tau", "eta", "ux", "uy", "un", "dat", "dax", "day", "dan", "T", "pixx", "pixy", "pixn", "piyy", "piyn", "bulkPi"]}
    v.update(pitt=pitt, pitx=pitx, pity=pity, pitn=pitn, pinn=pinn)
    v["Wx"] = 0.01 * (E + P) * (2.0 * U(1) - 1.0)
    v["Wy"] = 0.01 * (E + P) * (2.0 * U(2) - 1.0)
    v["Lambda"] = T * (0.9 + 0.25 * U(3))
    v["aL"] = 0.7 + 0.5 * U(4)
    shear = 0.5 / (T * T * (E + P))
    v["c4"] = shear * (0.8 + 0.4 * U(5))
    v["c3"] = shear * (2.0 * U(6) - 1.0)
    v["c0"] = 4.0 / T ** 4 * (0.5 + U(7))
    v["c1"] = -8.0 / T ** 4 * (0.5 + U(8))
    v["c2"] = 6.0 / T ** 4 * (0.5 + U(9))
    v["x"], v["y"] = s["x"], s["y"]
    # what a mode-2 surface file carries besides (write_surface_vah_dat): E, P and a longitudinal pressure with PL/P in [0.5, 1.5] --
    # the reader infers (alpha_L, Lambda) from (T, P, PL), so a surface read back from such a file has ITS OWN Lambda, aL, not the
    # ones drawn above
    v["E"], v["P"] = E, P
    v["PL"] = P * (0.5 + U(10))
    return {k: np.ascontiguousarray(a, dtype=np.float64) for k, a in v.items()}


def write_surface_vah_dat(path, s):
    """Mode-2 (`read_surf_VAH_PLMatch`, src/cpp/readindata.cpp:813-928) text surface: 31 columns -- tau x y eta | dat dax day dan |
    ut ux uy un | E T P PL | ten pi_perp^{mu nu} | Wt Wx Wy Wn | bulkPi -- 17 significant digits, thermodynamic, viscous and W
    columns in fm^-n (divided by hbar*c).  W^tau and W^eta as the kernel reconstructs them (smooth_kernels.cpp:2244-2245)."""
    h = HBARC
    tau, ux, uy, un = s["tau"], s["ux"], s["uy"], s["un"]
    ut = np.sqrt(1.0 + ux * ux + uy * uy + tau * tau * un * un)
    Wt = (ux * s["Wx"] + uy * s["Wy"]) * ut / (1.0 + ux * ux + uy * uy)
    Wn = Wt * un / ut
    cols = [tau, s["x"], s["y"], s["eta"], s["dat"], s["dax"], s["day"], s["dan"], ut, ux, uy, un,
            s["E"] / h, s["T"] / h, s["P"] / h, s["PL"] / h] + [s[k] / h for k in ("pitt", "pitx", "pity", "pitn", "pixx", "pixy", "pixn", "piyy", "piyn", "pinn")] + \
           [Wt / h, s["Wx"] / h, s["Wy"] / h, Wn / h, s["bulkPi"] / h]
    np.savetxt(path, np.column_stack(cols), fmt="%.17e", delimiter=" ")


def write_surface_dat(path, s):
    """Mode-1 (`read_surf_VH`, src/cpp/readindata.cpp:320-420) text surface, 20 columns, 17 significant
    digits, thermodynamic and viscous columns in fm^-n (divided by hbar*c)."""
    h = HBARC
    cols = [s["tau"], s["x"], s["y"], s["eta"], s["dat"], s["dax"], s["day"], s["dan"], s["ux"], s["uy"], s["un"],
            s["E"] / h, s["T"] / h, s["P"] / h, s["pixx"] / h, s["pixy"] / h, s["pixn"] / h, s["piyy"] / h,
            s["piyn"] / h, s["bulkPi"] / h]
    if "muB" in s:   # include_baryon: + muB [fm^-1]; include_baryondiff_deltaf: + nB Vx Vy Vn (readindata.cpp:399-420)
        cols += [s["muB"] / h, s["nB"], s["Vx"], s["Vy"], s["Vn"]]
    np.savetxt(path, np.column_stack(cols), fmt="%.17e", delimiter=" ")
