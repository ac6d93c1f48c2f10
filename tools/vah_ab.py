#!/usr/bin/env python3
"""A/B of the anisotropic-hydro main kernel on one resident surface, interleaved rounds in one process: kernel_variant 2 (round-1
kernel, 6 x 7 tile, expanded quadratic forms) against 3 (cf_main_vah3: factored exponent, 8 x 7 tile, row / unit lower-bound culls,
exp_p9), with and without culling; checks the spectra against each other."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=1000000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dim", type=int, default=3, choices=[2, 3], help="2: the 2+1D kernels (eta quadrature; use --species pikp --cells 100000 for config 2's shape)")
    ap.add_argument("--species", default="urqmd")
    ap.add_argument("--sets", default="variant=2;variant=3;variant=2,zero_skip=2;variant=3,zero_skip=2")
    a = ap.parse_args()
    import torch
    dev = torch.device("cuda:0")
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    sp, tab = inputs.species(a.species), inputs.vah_df_tables()
    cells = synth.synth_vah_surface(a.cells, a.dim)
    fields = [f for f in api.VAH_FIELDS[:25] if f != "T"]
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in fields}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream
    sets = []
    for s in a.sets.split(";"):
        o = dict(dimension=a.dim)
        for kv in s.split(","):
            k, v = kv.split("=")
            o["kernel_variant" if k == "variant" else k] = int(v)
        sets.append((s, o))
    plans, outs = [], []
    for s, o in sets:
        p = api.VahPlan(sp, grid, o, tab=tab, max_cells=a.cells)
        p.set_timing(True)
        plans.append(p)
        outs.append(torch.zeros(p.output_size, dtype=torch.float64, device=dev))
    times, preps = [[] for _ in sets], [[] for _ in sets]
    for r in range(a.rounds + 1):
        for i, p in enumerate(plans):
            p.execute(a.cells, ptrs, outs[i].data_ptr(), stream, want_status=False)
            t = p.timings()
            if r:
                times[i].append(t["ms_main"]); preps[i].append(t["ms_prep"])
    ref = outs[0].cpu().numpy()
    nev = a.cells * len(sp["mass"]) * len(grid["pT"]) * len(grid["phi"]) * (len(grid["y"]) if a.dim == 3 else 1)
    for i, (s, o) in enumerate(sets):
        st = plans[i].execute(a.cells, ptrs, outs[i].data_ptr(), stream)
        got = outs[i].cpu().numpy()
        err = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * np.abs(ref).max())))
        print("%-26s %-13s tile=%s main ms: median %.2f min %.2f  prep ms %.2f  culled %.4f  workspace %.1f GB -> %.3e evals/s  max rel diff vs first %.2e  bitwise %s" % (
            s, plans[i].main_kernel_name, plans[i].tile_shape, np.median(times[i]), min(times[i]), np.median(preps[i]),
            st["n_wave_rows_culled"] / max(st["n_wave_rows"], 1), plans[i].workspace_bytes / 1e9, nev / (np.median(times[i]) * 1e-3), err,
            bool(np.array_equal(got, ref))), flush=True)


if __name__ == "__main__":
    main()
