#!/usr/bin/env python3
"""dev: the one-shot host entry (is3d_smooth_spectra) in a FRESH process, config 3: wall time of the single call a drop-in caller makes,
for a given workspace cap (argv[1], bytes; 0 = the default).  Run it once per setting: the runtime's allocation cache must be cold."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from is3d_amd import api, inputs, synth
ws = int(float(sys.argv[1])) if len(sys.argv) > 1 else 0
g = inputs.grid(); grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
df = inputs.df_tables(); sp = inputs.species("urqmd"); n = 1000000
cells = synth.synth_surface(n, 3)
api.load()
import torch
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()      # the HIP context exists (the CLI warms it while it parses)
t0 = time.perf_counter()
out, st = api.smooth_spectra(cells, sp, grid, df, dict(dimension=3, df_mode=2, workspace_bytes=ws))
t1 = time.perf_counter()
print("workspace cap %.1f GB: one call %.1f ms  (kernels %.1f ms, passes %d, h2d %.1f, d2h %.1f)  checksum %.17g" % (
    ws / 1e9, (t1 - t0) * 1e3, st["ms_prep"] + st["ms_main"] + st["ms_finalize"], st["n_passes"], st["ms_h2d"], st["ms_d2h"], float(np.sum(out))), flush=True)
