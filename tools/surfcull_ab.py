#!/usr/bin/env python3
"""A/B of the row-cull rules of cf_main_tile3e in one process on BASELINE config 3: zero_skip 0 (accumulator-relative, bitwise) against
zero_skip 3 (surface-relative floors from the chunks that ran first); prints the main kernel's HIP-event times, the culled fraction and the
largest relative difference of the spectra."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    import torch
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df = inputs.df_tables()
    sp = inputs.species("urqmd")
    cells = synth.synth_surface(n, 3)
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    plans, outs, t, st = {}, {}, {}, {}
    for zs in (0, 3):
        plans[zs] = api.Plan(sp, grid, df, dict(dimension=3, df_mode=2, zero_skip=zs), max_cells=n)
        plans[zs].set_timing(True)
        outs[zs] = torch.zeros(plans[zs].output_size, dtype=torch.float64, device=dev)
        t[zs] = []
    for r in range(4):
        for zs in (0, 3):
            st[zs] = plans[zs].execute(n, ptrs, outs[zs].data_ptr(), 0, want_status=True)
            if r:
                t[zs].append(plans[zs].timings()["ms_main"])
    a, b = outs[0].cpu().numpy(), outs[3].cpu().numpy()
    rel = np.abs(b - a) / np.maximum(np.abs(a), 1e-300)
    for zs in (0, 3):
        print("zero_skip %d: main %.2f ms (min %.2f)  wave-rows culled %.4f" % (zs, np.median(t[zs]), min(t[zs]),
              st[zs]["n_wave_rows_culled"] / max(st[zs]["n_wave_rows"], 1)), flush=True)
    print("spectra: max relative difference %.3e (one-sided: %s), bins that differ %d of %d" % (rel.max(), bool((b <= a).all()), int((a != b).sum()), a.size))


if __name__ == "__main__":
    main()
