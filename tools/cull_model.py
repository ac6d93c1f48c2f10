#!/usr/bin/env python3
"""tools/cull_model.py -- host model of the row-culling rules of cf_main_tile3e on one cell chunk of BASELINE config 3.

Not part of the product or of any test: a design tool.  For a chunk of the seeded config-3 surface it forms, per (lane, cell,
phi tile, y row), the exponent bound earg = pT Dmax - mT C'_k the kernel tests, an estimate of the chunk's final accumulators
acc[lane][j][k] (equilibrium integrand, u = 1/2), and counts the (wave, cell, tile, row) quadruples each rule keeps:

  A  the kernel's rule: one threshold per (lane, tile) from the minimum over the tile's JT x R accumulators
  B  one threshold per (lane, tile, row): minimum over the row's JT accumulators
  C  per (lane, j, k): a row is live if any of its JT terms can reach half an ulp of ITS accumulator (needs E2_j per lane)

Thresholds use the final accumulators of the chunk (the best any stale running minimum can do), so the live fractions are
lower bounds; rule A's number is compared with what the device reports (status.n_wave_rows_culled).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from is3d_amd import inputs, synth  # noqa: E402


def unit_list_model():
    """under zero_skip = 3 (static threshold floors) cf_cull_floor could also emit, per (chunk, phi tile, row block,
    workgroup), the list of units that are live for at least one of the workgroup's lane-waves, and the main kernel could stage only those.
    What that can return is bounded by the cycle accounting of the kernel as it is (profiles/r02_cycle_accounting.log, culling on): the phases
    that scale with the number of units that pass through LDS.  Pure arithmetic on the committed counters -- build only if >= 6 % of the wave cycles."""
    import re
    log = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_cycle_accounting.log")).readline()
    f = {}
    for k, v in re.findall(r"(record part of stage|vmcnt part of wait|stage|wait|dead units|live units) (\d\.\d+)", log):
        f.setdefault(k, float(v))
    dead_n, live_n = [int(x) for x in re.findall(r"(?:dead|live) units (\d+) \(", log)]
    dead_frac = dead_n / (dead_n + live_n)
    # a unit can be left out of a workgroup's stream only if it is dead for BOTH of its lane-waves; neighbouring lane-waves (m_T-sorted) agree on
    # most units -- take the per-wave dead fraction as the upper bound
    save_dead = f["dead units"]                       # the dead units' own test: two LDS reads, three instructions, a vote
    save_stage = f["stage"] * dead_frac               # staging scales with the units staged
    # what the indirection costs: the records are 928 B, the staging pieces 1 KiB -- a gathered unit costs one whole piece for its record (today
    # 0.906 pieces per unit on average) and an address per unit instead of per four pieces; the E2 table (2 KiB per unit) is piece-aligned
    cost_gather = f["record part of stage"] * (1.0 - dead_frac) * (1.0 / 0.906 - 1.0) + f["stage"] * (1.0 - dead_frac) * 0.25
    net = save_dead + save_stage - cost_gather
    print("unit-list model (zero_skip = 3 only): staged units %.0f M, dead on arrival %.1f %%" % ((dead_n + live_n) / 1e6, 100 * dead_frac))
    print("  upper bound of what is saved: dead-unit tests %.2f %% + staging of dead units %.2f %% of the wave cycles = %.2f %%" % (
        100 * save_dead, 100 * save_stage, 100 * (save_dead + save_stage)))
    print("  gathered staging costs back ~%.2f %% (whole pieces per record, one address per unit)" % (100 * cost_gather))
    print("  net <= %.2f %% of the wave cycles; the barrier wait (%.1f %%) is the two waves' imbalance, not a per-unit cost" % (100 * net, 100 * f["wait"]))
    print("  threshold for building it: 6 %% -> %s" % ("build" if net >= 0.06 else "not built"))


def main():
    if "--unit-list-model" in sys.argv:
        unit_list_model()
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=4525)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--JT", type=int, default=8)
    ap.add_argument("--R", type=int, default=7)
    ap.add_argument("--order", default="mT", choices=["mT", "pT", "pTpad", "band8", "band4", "band16"],
                    help="lane order: sorted by mT (the plan's), pT-major with the classes sorted by mass inside, or pT-major with "
                         "every pT padded to whole waves")
    ap.add_argument("--eta-bin", type=float, default=None,
                    help="eta-binned chunk (VERDICT round 2, item 5): take only cells with |eta - ETA_BIN| < half the y spacing, i.e. what a chunk "
                         "would hold if cf_prep binned the cells by eta")
    ap.add_argument("--origin", default="0", help="row-block origin: an integer row, or 'eta': the row nearest the bin's eta minus R // 2, so that the "
                                                  "live window of the chunk's cells sits in one block (blocks then cover rows origin + R m, clipped to the grid)")
    a = ap.parse_args()
    g = inputs.grid()
    sp = inputs.species("urqmd")
    pT, phi, yv = g["pT"], g["phi"], g["y"]
    cls = sorted(set(zip(sp["mass"].tolist(), sp["sign"].tolist())))
    mass = np.array([c[0] for c in cls])
    mT = np.sqrt(mass[:, None] ** 2 + pT[None, :] ** 2).ravel()
    pTl = np.broadcast_to(pT[None, :], (len(cls), len(pT))).ravel()
    if a.order == "mT":
        order = np.argsort(mT, kind="stable")
    elif a.order.startswith("band"):
        bw = int(a.order[4:])          # bands of bw adjacent pT values, lanes sorted by mT inside a band
        ipT = np.broadcast_to(np.arange(len(pT))[None, :], (len(cls), len(pT))).ravel()
        order = np.lexsort((mT, ipT // bw))
    else:
        order = np.lexsort((mT, pTl))      # pT-major, then mass
    mT, pTl = mT[order], pTl[order]
    if a.order == "pTpad":
        ncl = len(cls)
        npad = (ncl + 63) // 64 * 64
        m2 = np.zeros((len(pT), npad)); p2 = np.zeros((len(pT), npad))
        m2[:, :ncl] = mT.reshape(len(pT), ncl); p2[:, :ncl] = pTl.reshape(len(pT), ncl)
        m2[:, ncl:] = m2[:, ncl - 1:ncl]; p2[:, ncl:] = p2[:, ncl - 1:ncl]      # duplicates of the heaviest class: never change a vote
        mT, pTl = m2.ravel(), p2.ravel()
    L = len(mT)
    nw = (L + 63) // 64
    mTmax, pTmax = mT.max(), pTl.max()
    pe = np.ceil(np.log2(np.maximum(mT / mTmax, pTl / pTmax))).astype(int)
    pe = np.minimum(pe, 0)

    if a.eta_bin is None:
        s = synth.synth_surface(a.cells, 3, first_cell=a.first)
    else:
        hw = 0.5 * (yv[1] - yv[0])
        big = synth.synth_surface(a.cells * 24, 3, first_cell=a.first)
        sel = np.nonzero(np.abs(big["eta"] - a.eta_bin) < hw)[0][:a.cells]
        s = {k: v[sel] for k, v in big.items()}
        a.cells = len(sel)
    tau, eta, T = s["tau"], s["eta"], s["T"]
    ut = np.sqrt(1 + s["ux"] ** 2 + s["uy"] ** 2 + tau ** 2 * s["un"] ** 2)
    dy = yv[None, :] - eta[:, None]                       # [c][k]
    ch, sh = np.cosh(dy), np.sinh(dy)
    Cp = (ch * ut[:, None] - sh * (tau * s["un"])[:, None]) / T[:, None]
    Ak = ch * s["dat"][:, None] + sh * (s["dan"] / tau)[:, None]
    Dp = (np.cos(phi)[None, :] * s["ux"][:, None] + np.sin(phi)[None, :] * s["uy"][:, None]) / T[:, None]   # [c][j]
    Bj = np.cos(phi)[None, :] * s["dax"][:, None] + np.sin(phi)[None, :] * s["day"][:, None]
    kmin, kmax = yv.min(), yv.max()
    gmax = np.cosh(np.maximum(np.abs(kmin - eta), np.abs(kmax - eta)))
    bound = ((mTmax * (np.abs(s["dat"]) + np.abs(s["dan"] / tau)) + pTmax * (np.abs(s["dax"]) + np.abs(s["day"]))) * gmax).max()
    e_sc = int(np.frexp(bound)[1])
    psc = 2.0 ** -e_sc
    J, K, JT, R = len(phi), len(yv), a.JT, a.R
    if a.origin == "eta":
        center = int(round(((a.eta_bin if a.eta_bin is not None else 0.0) - yv[0]) / (yv[1] - yv[0])))
        origin = (center - R // 2) % R
        origin = origin - R if origin > 0 else origin          # first block starts at or before row 0
    else:
        origin = -(int(a.origin) % R) if int(a.origin) % R else 0
    blocks = [(max(o, 0), min(o + R, K)) for o in range(origin, K, R) if min(o + R, K) > max(o, 0)]
    jt_n, kt_n = (J + JT - 1) // JT, len(blocks)
    C = a.cells

    # accumulators acc[l][j][k] (equilibrium, u = 1/2) and the exponent bounds
    acc = np.zeros((L, J, K))
    cb = 64
    for c0 in range(0, C, cb):
        c1 = min(C, c0 + cb)
        x = mT[:, None, None, None] * Cp[None, c0:c1, None, :] - pTl[:, None, None, None] * Dp[None, c0:c1, :, None]   # [l][c][j][k]
        pds = np.maximum(mT[:, None, None, None] * Ak[None, c0:c1, None, :] + pTl[:, None, None, None] * Bj[None, c0:c1, :, None], 0.0) * psc
        acc += (pds * np.exp(-x)).sum(axis=1) * 0.5
    with np.errstate(divide="ignore"):
        e_acc = np.where(acc > 1e-290, np.floor(np.log2(np.maximum(acc, 1e-300))) + 1, -np.inf)    # frexp exponent
    ln2 = 0.6931471805599453

    def thr_from(e):   # e: frexp exponent(s) of a lower bound on the accumulators a row adds to
        return np.maximum(-745.2, (e - 58 - pe.reshape((-1,) + (1,) * (e.ndim - 1))) * ln2)

    live = {k: 0 for k in "ABC"}
    unit_live = {k: 0 for k in "ABC"}
    lane_live = {k: 0 for k in "AB"}
    per_kt = {k: np.zeros(kt_n) for k in "AB"}
    total_rows = 0
    for jt in range(jt_n):
        js = slice(jt * JT, min(J, (jt + 1) * JT))
        Dmax = Dp[:, js].max(axis=1)                                     # [c]
        for kt in range(kt_n):
            ks = slice(blocks[kt][0], blocks[kt][1])
            earg = pTl[:, None, None] * Dmax[None, :, None] - mT[:, None, None] * Cp[None, :, ks]      # [l][c][r]
            eA = e_acc[:, js, ks].min(axis=(1, 2))                        # [l]
            eB = e_acc[:, js, ks].min(axis=1)                             # [l][r]
            thrA = thr_from(eA)[:, None, None]
            thrB = thr_from(eB)[:, None, :]
            lvA = earg >= thrA
            lvB = earg >= thrB
            # C: exists j: earg + log E2_j >= thr(acc[j][k]);  log E2_j = pT (Dp_j - Dmax)
            lvC = np.zeros_like(lvA)
            for j in range(js.start, js.stop):
                thrC = thr_from(e_acc[:, j, ks])[:, None, :]
                lE2 = pTl[:, None] * (Dp[None, :, j] - Dmax[None, :])
                lvC |= (earg + lE2[:, :, None]) >= thrC
            nr = ks.stop - ks.start
            total_rows += nw * C * nr
            for key, lv in (("A", lvA), ("B", lvB), ("C", lvC)):
                pad = np.zeros((nw * 64 - L,) + lv.shape[1:], dtype=bool)
                w = np.concatenate([lv, pad]).reshape(nw, 64, C, nr).any(axis=1)    # [w][c][r]
                live[key] += int(w.sum())
                unit_live[key] += int(w.any(axis=2).sum())
                if key in per_kt:
                    per_kt[key][kt] += w.sum() * R / nr
                    lane_live[key] += int(lv.sum())
    units = nw * C * jt_n * kt_n
    print("cells %d  lanes %d (%d waves)  tiles %d x %d  row blocks %s  scale 2^-%d" % (C, L, nw, jt_n, kt_n, blocks, e_sc))
    for key in "ABC":
        print("rule %s: live wave-rows %.4f  live units %.4f  rows per live unit %.2f" % (
            key, live[key] / total_rows, unit_live[key] / units, live[key] / max(unit_live[key], 1)))
    # issued fp64-VALU instructions per (wave, cell, phi tile) under rule A, priced with the instruction counts of cf_main_tile3e (DESIGN.md
    # section 5): a live row 8 x 14 + 25 (operands, row exponential), a live unit's header 59 (24 multiplications, 21 row-test instructions,
    # 14 v_readfirstlane), a dead row inside a live unit 3, a unit culled whole 3
    lr, lu = live["A"], unit_live["A"]
    dead_rows_in_live_units = lu * R - lr
    instr = lr * (8 * 14 + 25) + lu * 59 + max(dead_rows_in_live_units, 0) * 3 + (units - lu) * 3
    print("rule A: modelled issued instructions per (wave, cell, phi tile): %.1f   (units per cell and tile: %d)" % (instr / (nw * C * jt_n), kt_n))
    for key in "AB":
        print("rule %s: live (lane, row) pairs %.4f;  live wave-rows by row block: %s" % (
            key, lane_live[key] / (L * C * jt_n * K), np.round(per_kt[key] / (nw * C * jt_n * R), 4)))


if __name__ == "__main__":
    main()
