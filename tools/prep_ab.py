#!/usr/bin/env python3
"""A/B of cf_prep's record writer in one process: IS3D_PREP_PAIR = 0 / 1 / 3 alternated between executes of one plan (the switch is read
at every launch), BASELINE config 3 and config 2 surfaces; prints the prep kernel's HIP-event times and checks the spectra bitwise."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    import torch
    dev = torch.device("cuda:0")
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df = inputs.df_tables()
    for dim, n, species, dfm in ((3, 1000000, "urqmd", 2), (2, 100000, "pikp", 1)):
        sp = inputs.species(species)
        cells = synth.synth_surface(n, dim)
        tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
        ptrs = {k: v.data_ptr() for k, v in tens.items()}
        plan = api.Plan(sp, grid, df, dict(dimension=dim, df_mode=dfm), max_cells=n)
        plan.set_timing(True)
        outs = [torch.zeros(plan.output_size, dtype=torch.float64, device=dev) for _ in range(4)]
        t = {0: [], 1: [], 2: [], 3: []}
        for r in range(5):
            for pair in (0, 1, 3):
                os.environ["IS3D_PREP_PAIR"] = str(pair)
                plan.execute(n, ptrs, outs[pair].data_ptr(), 0, want_status=False)
                ms = plan.timings()["ms_prep"]
                if r:
                    t[pair].append(ms)
        print("dim %d, %d cells: prep ms  one element per lane %.3f (min %.3f)   two per lane %.3f (min %.3f)   duo / rows writer %.3f (min %.3f)   bitwise %s %s" % (
            dim, n, np.median(t[0]), min(t[0]), np.median(t[1]), min(t[1]), np.median(t[3]), min(t[3]), bool(torch.equal(outs[0], outs[1])),
            bool(torch.equal(outs[0], outs[3]))), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
