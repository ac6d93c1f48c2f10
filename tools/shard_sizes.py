#!/usr/bin/env python3
"""Strong-scaling prediction on ONE GPU: BASELINE config 4 is config 3's 1e6-cell surface in 8 shards of
125 000 cells.  Times the shard sizes 1e6 / 5e5 / 2.5e5 / 1.25e5 (the first shard of the surface each) in one process, interleaved
rounds: kernel ms (HIP events), step ms (host clock around execute + sync), cull fraction, the one-shot host entry, and for the
smallest shard a few cell-chunk counts.  compute-side efficiency at N = t(1e6) / (N t(1e6 / N)).  Writes JSON to stdout."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--total", type=int, default=1000000)
    ap.add_argument("--chunks", default="0,36,72,144", help="cell_chunks values tried on the smallest shard (0 = the plan's default)")
    a = ap.parse_args()
    import torch
    dev = torch.device("cuda:0")
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df, sp = inputs.df_tables(), inputs.species("urqmd")
    cells = synth.synth_surface(a.total, 3)
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream
    sizes = [a.total, a.total // 2, a.total // 4, a.total // 8]
    cases = [(n, 0) for n in sizes] + [(sizes[-1], int(c)) for c in a.chunks.split(",") if int(c)]
    plans, outs = [], []
    for n, ch in cases:
        p = api.Plan(sp, grid, df, dict(dimension=3, df_mode=2, cell_chunks=ch), max_cells=n)
        p.set_timing(True)
        plans.append(p)
        outs.append(torch.zeros(p.output_size, dtype=torch.float64, device=dev))
    rec = [dict(cells=n, cell_chunks=ch, main=[], prep=[], fin=[], step=[]) for n, ch in cases]
    for r in range(a.rounds + 1):
        for i, (n, ch) in enumerate(cases):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plans[i].execute(n, ptrs, outs[i].data_ptr(), stream, want_status=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            t = plans[i].timings()
            if r:
                rec[i]["main"].append(t["ms_main"]); rec[i]["prep"].append(t["ms_prep"]); rec[i]["fin"].append(t["ms_finalize"]); rec[i]["step"].append(dt)
    res = []
    for i, (n, ch) in enumerate(cases):
        st = plans[i].execute(n, ptrs, outs[i].data_ptr(), stream)
        torch.cuda.synchronize()
        d = dict(cells=n, cell_chunks_requested=ch, kernel_ms=dict(prep=float(np.median(rec[i]["prep"])), main=float(np.median(rec[i]["main"])),
                 finalize=float(np.median(rec[i]["fin"]))), step_ms=float(np.median(rec[i]["step"])),
                 wave_rows_culled_frac=st["n_wave_rows_culled"] / max(st["n_wave_rows"], 1), workspace_GB=plans[i].workspace_bytes / 1e9)
        if ch == 0:
            sub = {k: v[:n] for k, v in cells.items()}
            t0 = time.perf_counter()
            _, sh = api.smooth_spectra(sub, sp, grid, df, dict(dimension=3, df_mode=2))
            d["host_entry_ms"] = (time.perf_counter() - t0) * 1e3
            d["host_entry_kernels_ms"] = sh["ms_prep"] + sh["ms_main"] + sh["ms_finalize"]
        res.append(d)
        print(json.dumps(d), file=sys.stderr, flush=True)
    base = res[0]
    eff = {}
    for k, d in enumerate(res[:4]):
        nsh = a.total // d["cells"]
        eff["N=%d" % nsh] = dict(kernels=base["kernel_ms"]["main"] / (nsh * d["kernel_ms"]["main"]), step=base["step_ms"] / (nsh * d["step_ms"]))
    # the sum of the shard spectra is the whole spectrum (first two halves only: cheap)
    print(json.dumps(dict(what="strong-scaling shard sizes of BASELINE config 3's surface on one MI355X (first shard of each size)", rounds=a.rounds,
                          cases=res, compute_side_efficiency=eff)))
    for p in plans:
        p.close()


if __name__ == "__main__":
    main()
