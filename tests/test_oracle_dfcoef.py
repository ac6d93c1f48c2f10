"""CPU: the thermal-integral layer against numbers the REFERENCE ITSELF holds.

The reference ships the output of its own coefficient generator (generate_delta_f_coefficients/urqmd/df_vh_dimensionless/src/
deltaf_table.cpp:137-248, :296-395): deltaf_coefficients/vh/urqmd/{c0..c4, F, G, betabulk, betaV, betapi}.dat -- 81 810 numbers computed
from the same particle list the smooth path reads, with 64-point Gauss-Laguerre sums of the same integrands (thermal_integrands.cpp) that
the sampler densities, the df_mode-3 renormalisation, Jonah's table and calculate_total_yield use (src/cpp/gaussThermal.cpp).  Recomputing
them from
  * the particle list as THIS repository's reader returns it (is3d_pdg_read: 327 entries with the synthesised antibaryons, gspin as the
    degeneracy, sign from the baryon number, the reference's count - 1),
  * the Gauss-Laguerre file as is3d_gla_read returns it,
  * the oracle's integrand functions (oracle_df_generator_row calls neq_int, J10_int, J20_int, J11_int, J30_int, J31_int, E_mod_int,
    P_mod_int -- the functions behind oracle_total_yield, the sampler and the feqmod restatement -- plus J21, J40, J41, J32)
and comparing with the shipped PRINTED text pins those pieces on reference-held values: all 81 810 numbers come out digit for digit
(`python tests/test_oracle_dfcoef.py --all`, 35 s; the tests below take a sample).  The device kernels that evaluate the same integrands
(cf_sampler_density, cf_feqmod_renorm, cf_yield) are tied to these functions by tests/test_gpu_sampler.py, test_gpu_feqmod.py and
test_total_yield_matches_the_oracle.  What this does NOT pin: the Cooper-Frye integrand of row a1 itself -- the reference holds no
spectra: parity of the spectra stays unpinned."""
import os
import sys

import numpy as np
import pytest

from is3d_amd import api, inputs
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
GEN = os.path.join(REFERENCE, "generate_delta_f_coefficients/urqmd/df_vh_dimensionless")
HBARC = 0.197327053


def fixture():
    z = np.load(os.path.join(HERE, "golden", "golden_dfcoef.npz"))
    return {k: z[k] for k in z.files}


def pdg_fixture():
    a = np.array(inputs.load_fixture()["pdg_urqmd"], dtype=np.float64)        # columns: mc_id, mass, gspin, baryon, sign
    return dict(mass=a[:, 1].copy(), gspin=a[:, 2].copy(), baryon=a[:, 3].copy(), sign=a[:, 4].copy())


def printed(v):
    """what `ofstream << fixed << value` prints (deltaf_table.cpp:240-244, :387-391): six decimals; the sign of a rounded zero is noise of the
    baryon / antibaryon cancellation at mu_B = 0"""
    s = "%.6f" % v
    return "0.000000" if s == "-0.000000" else s


def mismatches(pdg, root, weight, fx, rows=None):
    bad, worst = [], 0.0
    for jB, iB in enumerate(fx["iB"]):
        for jT, iT in enumerate(fx["iT"]):
            if rows is not None and (jB, jT) not in rows:
                continue
            out = oracle.df_generator_row(pdg, root, weight, fx["T"][iT], fx["muB"][iB])
            for k, name in enumerate(fx["names"]):
                worst = max(worst, abs(out[k] - fx["shipped"][k, jB, jT]))
                if printed(out[k]) != printed(float(fx["text"][k, jB, jT])):
                    bad.append((str(name), float(fx["T"][iT]), float(fx["muB"][iB]), out[k], str(fx["text"][k, jB, jT])))
    return bad, worst


def test_shipped_coefficient_tables_are_reproduced_digit_for_digit():
    """345 (T, mu_B) rows x 10 tables (mu_B = 0, 0.2, 0.5 GeV among them; the grid's corners and their neighbours): every printed digit.  5e-7 is half
    a unit of the last printed decimal -- for betapi/T^4 ~ 30-120 that is 4e-9 to 2e-8 relative, for c4 T^5 ~ 2e-4 three significant digits:
    the digit-for-digit comparison is as sharp as the reference's own output allows."""
    fx = fixture()
    assert list(fx["names"]) == oracle.DF_NAMES_2D and fx["text"].shape == (10, 15, 23)
    assert {0, 20, 50} <= set(fx["iB"].tolist()) and abs(fx["muB"][20] - 0.2) < 1e-12 and abs(fx["muB"][50] - 0.5) < 1e-12
    bad, worst = mismatches(pdg_fixture(), fx["root"], fx["weight"], fx)
    assert not bad, bad[:5]
    assert worst <= 5.0e-7 * (1 + 1e-9)


def test_the_comparison_has_teeth():
    """The same comparison fails as soon as one ingredient deviates from the reference's semantics: the list without its last entry (the
    reference's count - 1 taken twice), with it doubled (no count - 1: the stream's failed read leaves a copy), a spin degeneracy that includes
    isospin, Boltzmann statistics for the pions, the alpha = 2 rule where alpha = 1 belongs."""
    fx = fixture()
    rows = {(0, 0), (0, 11), (6, 11), (14, 22), (4, 5)}
    pdg = pdg_fixture()
    bad, _ = mismatches(pdg, fx["root"], fx["weight"], fx, rows)
    assert not bad

    def variant(**kw):
        p = {k: v.copy() for k, v in pdg.items()}
        p.update(kw)
        return p
    n_bad = {}
    n_bad["last entry dropped"] = len(mismatches({k: v[:-1] for k, v in pdg.items()}, fx["root"], fx["weight"], fx, rows)[0])
    n_bad["last entry doubled"] = len(mismatches({k: np.append(v, v[-1]) for k, v in pdg.items()}, fx["root"], fx["weight"], fx, rows)[0])
    g = pdg["gspin"].copy()
    g[1] = 3.0                                                                # the pi+ entry carrying the isospin triplet
    n_bad["degeneracy"] = len(mismatches(variant(gspin=g), fx["root"], fx["weight"], fx, rows)[0])
    s = pdg["sign"].copy()
    s[1:4] = 0.0
    n_bad["pion statistics"] = len(mismatches(variant(sign=s), fx["root"], fx["weight"], fx, rows)[0])
    r, w = fx["root"].copy(), fx["weight"].copy()
    r[1], w[1] = r[2], w[2]
    n_bad["alpha"] = len(mismatches(pdg, r, w, fx, rows)[0])
    assert all(v > 0 for v in n_bad.values()), n_bad
    assert n_bad["degeneracy"] >= 20 and n_bad["pion statistics"] >= 20 and n_bad["alpha"] >= 10, n_bad


def test_generator_integrals_are_the_yield_restatements_densities():
    """The tie to what the sampler / yield use: n_B of the generator (deltaf_table.cpp:345, nB_int = b * neq_int) is the sum of b_i n_eq,i over the
    species densities oracle_total_yield forms (Deltaf_Data::compute_particle_densities, deltafReader.cpp:553-585) when both run on the same
    64-point rule; e + p and the pressure follow the same way from E_mod_int / P_mod_int at lambda = 0 (Jonah's table, deltafReader.cpp:247-266)."""
    fx = fixture()
    pdg = pdg_fixture()
    T, muB = 0.15, 0.3
    out, integ = oracle.df_generator_row(pdg, fx["root"], fx["weight"], T, muB, with_integrals=True)
    nB_gen = integ[12]
    keep = pdg["mass"] > 0
    sp = dict(mass=pdg["mass"][keep], sign=pdg["sign"][keep], degeneracy=pdg["gspin"][keep], baryon=pdg["baryon"][keep])
    gla = dict(root1=fx["root"][1], weight1=fx["weight"][1], root2=fx["root"][2], weight2=fx["weight"][2], root3=fx["root"][3], weight3=fx["weight"][3])
    z = np.zeros(2)
    cells = dict(tau=z + 1, eta=z, dat=z + 1, dax=z, day=z, dan=z, ux=z, uy=z, un=z, T=z + T, P=z + 0.08, E=z + 0.3, pixx=z, pixy=z, pixn=z, piyy=z,
                 piyn=z, bulkPi=z, muB=z + muB, nB=z + 0.05, Vx=z, Vy=z, Vn=z)
    avg5 = np.array([T, 0.3, 0.08, muB, 0.05])
    _, dens = oracle.total_yield(cells, sp, inputs.df_tables_full(), gla, avg5, dict(dimension=3, df_mode=2, include_baryon=1, include_baryondiff_deltaf=1))
    assert abs(np.sum(sp["baryon"] * dens[0]) / nB_gen - 1) < 1e-12
    # and the generator's own consistency at mu_B = 0: betaV = M11 (n_B = 0), G = 0, c1 = c4 = 0 up to the cancellation noise
    out0, i0 = oracle.df_generator_row(pdg, fx["root"], fx["weight"], T, 0.0, with_integrals=True)
    assert abs(i0[12]) < 1e-15 * i0[18] and abs(out0[6]) < 1e-12 and abs(out0[1]) < 1e-12 and abs(out0[4]) < 1e-12
    assert abs(out0[8] * T ** 3 / i0[19] - 1) < 1e-12


@pytest.mark.reference
def test_readers_and_oracle_on_the_references_own_files():
    """Container only: the same comparison with every input read through the LIBRARY's readers from the reference's files -- is3d_pdg_read on
    PDG/pdg-urqmd_v3.3+.dat and on the generator's own pdg.dat (one more blank line: the same 327 entries), is3d_gla_read on the generator's
    64-point file (21 alphas), is3d_df_table_read_full on the ten shipped tables -- on every 4th temperature x every 5th chemical potential;
    and the fixtures are what those readers return."""
    pdg = api.pdg_read(os.path.join(REFERENCE, "PDG/pdg-urqmd_v3.3+.dat"))
    gen = api.pdg_read(os.path.join(GEN, "pdg.dat"))
    assert len(pdg["mass"]) == 327
    for k in pdg:
        assert np.array_equal(pdg[k], gen[k]), k
    pf = pdg_fixture()
    for k in pf:
        assert np.array_equal(pf[k], pdg[k]), k
    root, weight = api.gla_read(os.path.join(GEN, "gauss_laguerre/gla_roots_weights_64_points.txt"))
    assert root.shape == (21, 64)
    fx = fixture()
    assert np.array_equal(root[:5], fx["root"]) and np.array_equal(weight[:5], fx["weight"])
    tabs = {}
    for n in oracle.DF_NAMES_2D:
        T, B, v = api.df_table_read_full(os.path.join(REFERENCE, "deltaf_coefficients/vh/urqmd", n + ".dat"))
        assert v.shape == (81, 101) and np.array_equal(T, fx["T"]) and np.array_equal(B, fx["muB"])
        assert np.array_equal(v[np.ix_(fx["iB"], fx["iT"])], fx["shipped"][oracle.DF_NAMES_2D.index(n)])
        tabs[n] = v
    bad = []
    for iB in range(0, 81, 5):
        for iT in range(0, 101, 4):
            out = oracle.df_generator_row(pdg, root, weight, T[iT], B[iB])
            for k, n in enumerate(oracle.DF_NAMES_2D):
                if printed(out[k]) != printed(tabs[n][iB, iT]) or abs(out[k] - tabs[n][iB, iT]) > 5.0e-7 * (1 + 1e-9):
                    bad.append((n, T[iT], B[iB], out[k], tabs[n][iB, iT]))
    assert not bad, bad[:5]


@pytest.mark.reference
@pytest.mark.parametrize("which", ["smash", "smash_box"])
def test_smash_tables_of_the_reference_pin_the_other_two_particle_list_readers(which):
    """hrg_eos = 2 (the reference's shipped default: PDG/pdg_smash.dat through read_resonances_conventional, 493 entries) and hrg_eos = 3
    (PDG/pdg_box.dat through read_resonances_smash_box / read_mcid, 400 entries): deltaf_coefficients/vh/{smash, smash_box}/*.dat are the same
    generator's output for those lists (generate_delta_f_coefficients/{smash, smash_box}/df_vh_dimensionless/src: deltaf_table.cpp differs from the
    urqmd one in file names only).  The lists as is3d_pdg_read / is3d_pdg_read_box return them reproduce every sampled printed value."""
    if which == "smash":
        pdg = api.pdg_read(os.path.join(REFERENCE, "PDG/pdg_smash.dat"))
        assert len(pdg["mass"]) == 493
    else:
        pdg = api.pdg_read(os.path.join(REFERENCE, "PDG/pdg_box.dat"), box=True)
        assert len(pdg["mass"]) == 400
    root, weight = api.gla_read(os.path.join(REFERENCE, "generate_delta_f_coefficients", which, "df_vh_dimensionless/gauss_laguerre/gla_roots_weights_64_points.txt"))
    tabs = {n: api.df_table_read_full(os.path.join(REFERENCE, "deltaf_coefficients/vh", which, n + ".dat")) for n in oracle.DF_NAMES_2D}
    T, B, _ = tabs["c0"]
    bad = []
    for iB in range(0, len(B), 8):
        for iT in range(0, len(T), 10):
            out = oracle.df_generator_row(pdg, root, weight, T[iT], B[iB])
            for k, n in enumerate(oracle.DF_NAMES_2D):
                if printed(out[k]) != printed(tabs[n][2][iB, iT]):
                    bad.append((n, T[iT], B[iB], out[k], tabs[n][2][iB, iT]))
    assert not bad, bad[:5]


if __name__ == "__main__" and "--all" in sys.argv:
    # every row of every shipped table (container only, ~35 s)
    pdg = api.pdg_read(os.path.join(REFERENCE, "PDG/pdg-urqmd_v3.3+.dat"))
    root, weight = api.gla_read(os.path.join(GEN, "gauss_laguerre/gla_roots_weights_64_points.txt"))
    tabs = {n: api.df_table_read_full(os.path.join(REFERENCE, "deltaf_coefficients/vh/urqmd", n + ".dat")) for n in oracle.DF_NAMES_2D}
    T, B, _ = tabs["c0"]
    n_bad, worst = 0, 0.0
    for iB in range(81):
        for iT in range(101):
            out = oracle.df_generator_row(pdg, root, weight, T[iT], B[iB])
            for k, n in enumerate(oracle.DF_NAMES_2D):
                worst = max(worst, abs(out[k] - tabs[n][2][iB, iT]))
                n_bad += printed(out[k]) != printed(tabs[n][2][iB, iT])
    print("81 x 101 rows x 10 tables: %d printed values differ, max |difference| %.3e" % (n_bad, worst))
