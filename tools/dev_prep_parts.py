import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from is3d_amd import api, inputs, synth
dev = torch.device("cuda:0")
g = inputs.grid(); grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3
df = inputs.df_tables(); sp = inputs.species("urqmd" if dim == 3 else "pikp"); n = 1000000 if dim == 3 else 100000
cells = synth.synth_surface(n, dim)
tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}; ptrs = {k: v.data_ptr() for k, v in tens.items()}
plan = api.Plan(sp, grid, df, dict(dimension=dim, df_mode=2 if dim == 3 else 1), max_cells=n); plan.set_timing(True)
out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
for skip in (0, 16, 1, 17, 0, 16, 1, 17):
    os.environ["IS3D_PREP_SKIP"] = str(skip)
    t = []
    for r in range(4):
        plan.execute(n, ptrs, out.data_ptr(), 0, want_status=False)
        t.append(plan.timings()["ms_prep"])
    print("writer %s  skip mask %d (16 = non-temporal record stores): prep %.3f ms" % (os.environ.get("IS3D_PREP_PAIR", "default"), skip, min(t[1:])), flush=True)
