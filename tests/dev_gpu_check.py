#!/usr/bin/env python3
"""(Lives under tests/ because it calls the oracle.)  Developer check on a GPU box: parity of every kernel variant against the CPU oracle on small seeded
surfaces, then timing of a mid-size run through the device-resident plan.  (The judged artefacts are
tests/ and bench.py; this is the quick loop.)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402
from oracle import oracle  # noqa: E402  (checker only)


def relerr(a, b, floor=1e-280):
    den = np.maximum(np.abs(b), floor)
    return np.max(np.abs(a - b) / den)


def parity(ncell=96):
    g = inputs.grid()
    df = inputs.df_tables()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    worst = 0.0
    for dim in (3, 2):
        cells = synth.synth_surface(ncell if dim == 3 else max(8, ncell // 8), dim)
        sp = inputs.species("pikp") if dim == 2 else inputs.species([211, 321, 2212, -2212, 3122, 333, 22])
        for dfm in (1, 2):
            for extra in (dict(), dict(outflow=0, regulate_deltaf=0), dict(outflow=0), dict(regulate_deltaf=0)):
                o = dict(dimension=dim, df_mode=dfm)
                o.update(extra)
                t0 = time.time()
                ref = oracle.dN_pTdpTdphidy(cells, sp, grid, df, o)
                t1 = time.time()
                for var in (1, 2, 3, 4):
                    oo = dict(o)
                    oo["kernel_variant"] = var
                    out, st = api.smooth_spectra(cells, sp, grid, df, oo)
                    e = relerr(out, ref)
                    worst = max(worst, e)
                    print("dim=%d df=%d %-40s var=%d  max rel err %.3e   (oracle %.2fs, gpu main %.3f ms, classes %d)" % (
                        dim, dfm, str(extra), var, e, t1 - t0, st["ms_main"], st["n_classes"]), flush=True)
    print("WORST", worst)
    return worst


def timing(ncell, species, dim, dfm, variant, reps=2):
    import torch
    g = inputs.grid()
    df = inputs.df_tables()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    sp = inputs.species(species)
    cells = synth.synth_surface(ncell, dim)
    dev = torch.device("cuda:0")
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    plan = api.Plan(sp, grid, df, dict(dimension=dim, df_mode=dfm, kernel_variant=variant), max_cells=ncell)
    plan.set_timing(True)
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream
    nbins = len(grid["pT"]) * len(grid["phi"]) * (len(grid["y"]) if dim == 3 else 1)
    evals = ncell * nbins * len(sp["mass"])
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        st = plan.execute(ncell, ptrs, out.data_ptr(), stream)
        torch.cuda.synchronize()
        t1 = time.time()
        tm = plan.timings()
        inner = evals * (len(grid["eta"]) if dim == 2 else 1)
        print("timing dim=%d df=%d %s ncell=%d var=%d: wall %.1f ms  prep %.2f main %.2f fin %.2f ms  -> %.3e evals/s (%.3e inner/s) classes=%d ws=%.2f GB" % (
            dim, dfm, species, ncell, variant, (t1 - t0) * 1e3, tm["ms_prep"], tm["ms_main"], tm["ms_finalize"],
            evals / (t1 - t0), inner / (t1 - t0), st["n_classes"], plan.workspace_bytes / 1e9), flush=True)
    res = out.cpu().numpy()
    plan.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--ncell3", type=int, default=100000)
    ap.add_argument("--ncell2", type=int, default=20000)
    a = ap.parse_args()
    print(api.load().is3d_version(), "devices:", api.load().is3d_device_count())
    if not a.skip_parity:
        parity()
    for var in (1, 2, 3, 4):
        timing(a.ncell3, "urqmd", 3, 2, var)
        timing(a.ncell2, "pikp", 2, 1, var)
