import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from is3d_amd import api, inputs, synth
dev = torch.device("cuda:0")
g = inputs.grid(); grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
df = inputs.df_tables(); sp = inputs.species("urqmd"); n = 1000000
cells = synth.synth_surface(n, 3)
tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}; ptrs = {k: v.data_ptr() for k, v in tens.items()}
plan = api.Plan(sp, grid, df, dict(dimension=3, df_mode=2), max_cells=n); plan.set_timing(True)
out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
for skip in (0, 1, 2, 3, 4, 7, 0):
    os.environ["IS3D_PREP_SKIP"] = str(skip)
    t = []
    for r in range(4):
        plan.execute(n, ptrs, out.data_ptr(), 0, want_status=False)
        t.append(plan.timings()["ms_prep"])
    print("skip mask %d: prep %.3f ms" % (skip, min(t[1:])), flush=True)
