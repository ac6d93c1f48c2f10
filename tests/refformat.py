"""Test helper: write input files in the reference's on-disk formats from the JSON
fixture, so that reader / CLI tests run on boxes that have no /root/reference.  The quirks of the shipped
files are reproduced on purpose: CRLF PDG records with decay lines, leading tabs, a phi table that ends
in "\\n\\t" (a fragment readBlockData drops), coefficient tables with the three header lines."""
import os

import numpy as np

from is3d_amd import inputs, synth

PARAMS_TEMPLATE = """operation	                = {operation}	 # operation
mode       		      	= {mode} 	 # mode for reading in freeze out information
hrg_eos				= {hrg_eos}	 # HRG equation of state
set_FO_temperature		= {set_FO_temperature}      # sampler fast mode: T_switch replaces the surface average
T_switch			= 0.151
dimension  		     	= {dimension}      # 2: boost invariant, 3: full 3+1D
df_mode		                = {df_mode}     # 1: 14-moment, 2: Chapman-Enskog
include_baryon            	= {include_baryon}
Include_Bulk_Deltaf       	= {include_bulk_deltaf}     # names are case-insensitive
include_shear_deltaf      	= {include_shear_deltaf}
include_baryondiff_deltaf 	= {include_baryondiff_deltaf}
regulate_deltaf           	= {regulate_deltaf}
outflow 			= {outflow}	             # Theta(p.dsigma)

deta_min 			= 1.e-5
mass_pion0			= 0.138  # lightest pion mass for the feqmod breakdown test
oversample			= {oversample}
fast				= {fast}
y_cut				= 0.7
min_num_hadrons			= {min_num_hadrons}
max_num_samples			= 1000
sampler_seed			= {sampler_seed}
test_sampler			= {test_sampler}
pT_lower_cut			= 0.0
pT_upper_cut			= 3.0
pT_bins				= 30
y_bins 				= 14
eta_cut 			= 6
eta_bins 			= 24
tau_min				= 0.0
tau_max				= 12.0
tau_bins			= 12
r_min				= 0.0
r_max				= 10.0
r_bins				= 10
group_particles                 = 0
"""


def write_table(path, x, w, leading_tab=False, dangling_fragment=False):
    with open(path, "w") as f:
        for a, b in zip(x, w):
            f.write(("\t" if leading_tab else "") + repr(float(a)) + "\t" + repr(float(b)) + "\n")
        if dangling_fragment:
            f.write("\t")


def write_pdg(path, rows, trailing_blank=True):
    """rows: [mc_id, mass, gspin, baryon, sign] for PARTICLES only (baryon >= 0); antibaryons are synthesised by the reader."""
    with open(path, "w", newline="") as f:
        for mc_id, mass, gspin, baryon, _ in rows:
            f.write("%8d  %-20s %9.5f %9.5f %2d %2d  0  0  0  1  0  2\r\n" % (mc_id, "P%d" % mc_id, mass, 0.0, gspin, baryon))
            f.write("%8d  2  0.600 %13d %7d       0       0       0\r\n" % (mc_id, 211, -211))
            f.write("%8d  1  0.400 %13d       0       0       0       0\r\n" % (mc_id, mc_id))
        if trailing_blank:
            f.write("\r\n")


# hrg_eos = 3 (PDG/pdg_box.dat, read_resonances_smash_box): "name mass width parity id [id...]" lines, '#' comments.  A few hadrons of the shipped list
# (name, mass, ids): enough for pi / K / p runs and for the digit rules (eta: its own antiparticle; K: a meson with one; Delta: four charge states)
BOX_ROWS = [("π", 0.138, [111, 211]), ("η", 0.548, [221]), ("K", 0.494, [311, 321]), ("ρ", 0.776, [113, 213]), ("N", 0.938, [2112, 2212]),
            ("Δ", 1.232, [1114, 2114, 2214, 2224]), ("Λ", 1.116, [3122]), ("φ", 1.019, [333]), ("f₂", 1.275, [225])]


def box_rows_for(species_ids):
    have = {abs(i) for row in BOX_ROWS for i in row[2]}
    missing = [i for i in species_ids if abs(int(i)) not in have]
    assert not missing, "species %s are not in refformat.BOX_ROWS" % missing
    return BOX_ROWS


def write_pdg_box(path, rows):
    with open(path, "w", encoding="utf-8") as f:
        f.write("# NAME MASS[GEV] WIDTH[GEV] PARITY PDG\n\n########## a few hadrons ##########\n\n")
        for k, (name, mass, ids) in enumerate(rows):
            f.write("%-14s %7.3f   %-8s %s  %s%s\n" % (name, mass, "1.5e-3" if k % 2 else "0", "-" if k % 3 else "+", " ".join("%8d" % i for i in ids),
                                                     "     # a comment behind the ids" if k % 4 == 1 else ""))
        f.write("   \n# end\n")


def box_entries(rows):
    """What read_resonances_smash_box + read_mcid make of such rows (readindata.cpp:1571-1685, :1201-1418), restated for the tests:
    (mc_id, mass, gspin, baryon, sign) per entry, the antiparticle behind its particle."""
    out = []
    for _, mass, ids in rows:
        for i in ids:
            d = [(i // 10 ** k) % 10 for k in range(10)]
            nJ, nq3, nq2, nq1 = d[0] + d[7], d[1], d[2], d[3]
            assert nq3 != 0 and nq2 != 0 and nJ > 0
            b = 1 if nq1 != 0 else 0
            sign = 1.0 if b else -1.0
            out.append((i, mass, float(nJ), float(b), sign))
            if b or nq2 != nq3:
                out.append((-i, mass, float(nJ), float(-b), sign))
    return out


def write_df_table(path, T, values, label, n_muB=2):
    with open(path, "w") as f:
        f.write("%d\n%d\n" % (len(T), n_muB))
        f.write("T [GeV]\t\tmuB [GeV]\t\t%s\n" % label)
        for ib in range(n_muB):
            for t, v in zip(T, values):
                f.write("%.6f\t\t%.6f\t\t%s\n" % (t, 0.01 * ib, repr(float(v) * (1.0 + 0.5 * ib))))


def write_vah_df_tables(directory, tab, header_pad=0):
    """deltaf_coefficients/vah/c{0..4}_vah1.dat in the shipped layout (src/cuda/deltafReader.cu:104-127, :196-213): "n_L\\nn_aL\\n",
    one label line, rows "L\\t\\taL\\t\\tvalue" with alpha_L outer and Lambda inner."""
    os.makedirs(directory, exist_ok=True)
    for k in range(5):
        with open(os.path.join(directory, "c%d_vah1.dat" % k), "w") as f:
            f.write("%d\n%d\n" % (len(tab["L"]), len(tab["aL"])))
            f.write("L [fm^-1]\t\taL\t\tc%d_vah1 [...]%s\n" % (k, " x" * header_pad))
            for i2, al in enumerate(tab["aL"]):
                for i1, lam in enumerate(tab["L"]):
                    f.write("%s\t\t%s\t\t%s\n" % (repr(float(lam)), repr(float(al)), repr(float(tab["c%d" % k][i2, i1]))))


def make_run_dir(root, cells, species_ids, params):
    """An iS3D-style run directory: iS3D_parameters.dat, input/, PDG/, tables/, deltaf_coefficients/, results/."""
    fx = inputs.load_fixture()
    g = inputs.grid()
    df = inputs.df_tables()
    hrg_eos = int(params.get("hrg_eos", 1))   # 1: urqmd, 2: smash (same file format), 3: smash box (the line-oriented list) -- readindata.h:217-219
    df_dir = {1: "urqmd", 2: "smash", 3: "smash_box"}[hrg_eos]
    for d in ("input", "PDG", "tables/eta", "deltaf_coefficients/vh/" + df_dir, "results/vn_continuous", "results/dN_dy", "results/dN_deta",
              "results/momentum_distribution", "results/vn", "results/spacetime_distribution"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    p = dict(operation=1, mode=1, dimension=3, df_mode=1, include_bulk_deltaf=1, include_shear_deltaf=1, regulate_deltaf=1, outflow=1,
             include_baryon=0, include_baryondiff_deltaf=0, oversample=0, min_num_hadrons=2000, sampler_seed=17, fast=0, test_sampler=0,
             set_FO_temperature=1, hrg_eos=1)
    p.update(params)
    with open(os.path.join(root, "iS3D_parameters.dat"), "w") as f:
        f.write(PARAMS_TEMPLATE.format(**p))
    synth.write_surface_dat(os.path.join(root, "input", "surface.dat"), cells)
    write_table(os.path.join(root, "tables", "pT_gauss_legendre_table.dat"), g["pT"], g["pT_w"])
    write_table(os.path.join(root, "tables", "phi_gauss_legendre_table.dat"), g["phi"], g["phi_w"], leading_tab=True, dangling_fragment=True)
    write_table(os.path.join(root, "tables", "y_trapezoid_table_21pt.dat"), g["y"], g["y_w"])
    write_table(os.path.join(root, "tables", "eta", "eta_trapezoid_table_241pt.dat"), g["eta"], g["eta_w"])
    if p["include_baryon"]:
        dff = inputs.df_tables_full()   # all ten tables with every mu_B row, as the shipped files have them
        for name in inputs.DF_NAMES_2D:
            with open(os.path.join(root, "deltaf_coefficients", "vh", df_dir, name + ".dat"), "w") as f:
                f.write("%d\n%d\nT [GeV]\t\tmuB [GeV]\t\t%s\n" % (len(dff["T"]), len(dff["muB"]), name))
                for ib, mub in enumerate(dff["muB"]):
                    for it, t in enumerate(dff["T"]):
                        f.write("%s\t\t%s\t\t%s\n" % (repr(float(t)), repr(float(mub)), repr(float(dff["2d"][name][ib, it]))))
    else:
        for name in ("c0", "c2", "F", "betabulk", "betapi"):
            write_df_table(os.path.join(root, "deltaf_coefficients", "vh", df_dir, name + ".dat"), df["T"], df[name], name)
    # tables/gla_roots_weights_32_points.txt layout: "n_alpha\tn_points", rows "alpha\troot\tweight"; alpha = 0 is never read
    gla = fx["gla_32"]
    with open(os.path.join(root, "tables", "gla_roots_weights_32_points.txt"), "w") as f:
        f.write("4\t%d\n" % len(gla["root1"]))
        for al, (rk, wk) in enumerate([("root1", "weight1"), ("root1", "weight1"), ("root2", "weight2"), ("root3", "weight3")]):
            for r, w in zip(gla[rk], gla[wk]):
                f.write("%d\t%s\t%s\n" % (al, repr(float(r) * (7.0 if al == 0 else 1.0)), repr(float(w))))
    particles = [r for r in fx["pdg_urqmd"] if r[3] >= 0]
    if hrg_eos == 3:
        write_pdg_box(os.path.join(root, "PDG", "pdg_box.dat"), box_rows_for(species_ids))
    else:
        write_pdg(os.path.join(root, "PDG", "pdg-urqmd_v3.3+.dat" if hrg_eos == 1 else "pdg_smash.dat"), particles)
    with open(os.path.join(root, "PDG", "chosen_particles.dat"), "w") as f:
        for i in species_ids:
            f.write("\t%d\n" % i)
    return root


def write_surface_mode(path, s, mode, include_baryon=0, include_baryondiff=0):
    """A surface.dat in one of the other viscous-hydro formats (readindata.cpp:148-318, :552-810, :1059-1196).  The
    columns the readers drop (u^tau, pi^{tau mu}, pi^{eta eta}, mu_S, mu_C, V^tau) get recognisable junk."""
    h, n = synth.HBARC, len(s["tau"])
    tau = s["tau"]
    ut = np.sqrt(1 + s["ux"] ** 2 + s["uy"] ** 2 + tau ** 2 * s["un"] ** 2)
    junk = np.full(n, 7.0)
    muB = s.get("muB", np.zeros(n))
    if mode == 0:
        cols = [tau, s["x"], s["y"], s["eta"], s["dat"], s["dax"], s["day"], s["dan"], ut, s["ux"], s["uy"], s["un"],
                s["E"] / h, s["T"] / h, s["P"] / h, junk, junk, junk, junk, s["pixx"] / h, s["pixy"] / h, s["pixn"] / h,
                s["piyy"] / h, s["piyn"] / h, junk, s["bulkPi"] / h]
        if include_baryon:
            cols.append(muB / h)
        if include_baryondiff:
            cols += [s["nB"], junk, s["Vx"], s["Vy"], s["Vn"]]
    elif mode == 5:      # gpu-vh + thermal vorticity (readindata.cpp:470-551): mode-1 columns, V^tau in the diffusion block, six w^{mu nu}
        cols = [tau, s["x"], s["y"], s["eta"], s["dat"], s["dax"], s["day"], s["dan"], s["ux"], s["uy"], s["un"],
                s["E"] / h, s["T"] / h, s["P"] / h, s["pixx"] / h, s["pixy"] / h, s["pixn"] / h, s["piyy"] / h, s["piyn"] / h, s["bulkPi"] / h]
        if include_baryon:
            cols.append(muB / h)
        if include_baryondiff:
            cols += [s["nB"], junk, s["Vx"], s["Vy"], s["Vn"]]
        cols += [junk * k for k in (1, 2, 3, 4, 5, 6)]
    elif mode in (4, 6):
        ent = (s["E"] + s["P"]) / s["T"]                       # entropy density, fm^-3: p = T s - e
        cols = [tau, s["x"], s["y"], junk, s["dat"] / tau, s["dax"] / tau, s["day"] / tau, s["dan"] / tau, ut, s["ux"], s["uy"],
                s["un"] * tau, s["E"] / h, s["T"] / h, muB / h]
        if mode == 6:
            cols += [junk, junk]
        cols += [ent, junk, junk, junk, junk, s["pixx"] / h, s["pixy"] / h, s["pixn"] * tau / h, s["piyy"] / h,
                 s["piyn"] * tau / h, junk, s["bulkPi"] / h]
    elif mode == 7:
        cols = [tau, s["x"], s["y"], junk, s["dat"] / tau, s["dax"] / tau, s["day"] / tau, junk, s["ux"] / ut, s["uy"] / ut, junk,
                junk, junk, junk, junk, s["pixx"], s["pixy"], s["pixn"] * tau, s["piyy"], s["piyn"] * tau, junk, s["bulkPi"],
                s["T"], s["E"], s["P"], muB]
    else:
        raise ValueError(mode)
    np.savetxt(path, np.column_stack(cols), fmt="%.17e", delimiter=" ")


def read_surface_like_reference(path):
    """What read_surf_VH does to a mode-1 file (readindata.cpp:343-410), in numpy: -> dict of the 18 arrays."""
    a = np.loadtxt(path, ndmin=2)
    h = synth.HBARC
    names = ["tau", "x", "y", "eta", "dat", "dax", "day", "dan", "ux", "uy", "un", "E", "T", "P", "pixx", "pixy", "pixn", "piyy", "piyn", "bulkPi"]
    if a.shape[1] >= 25:   # + muB [fm^-1], nB, Vx, Vy, Vn (include_baryon && include_baryondiff_deltaf)
        names = names + ["muB", "nB", "Vx", "Vy", "Vn"]
    s = {n: a[:, i].copy() for i, n in enumerate(names)}
    for n in ("E", "T", "P", "pixx", "pixy", "pixn", "piyy", "piyn", "bulkPi", "muB"):
        if n in s:
            s[n] = s[n] * h
    return s
