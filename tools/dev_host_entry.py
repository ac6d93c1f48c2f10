#!/usr/bin/env python3
"""dev: where does the one-shot host entry's time go?  Plan creation (workspace + partial hipMalloc) with and without another plan alive."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from is3d_amd import api, inputs, synth
g = inputs.grid(); grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
df = inputs.df_tables(); sp = inputs.species("urqmd"); n = 1000000
cells = synth.synth_surface(n, 3)
o = dict(dimension=3, df_mode=2)
for rep in range(3):
    t0 = time.perf_counter(); p = api.Plan(sp, grid, df, o, max_cells=n); t1 = time.perf_counter(); p.close(); t2 = time.perf_counter()
    print("plan create %.1f ms, destroy %.1f ms (workspace %.1f GB)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, p.workspace_bytes / 1e9), flush=True)
keep = api.Plan(sp, grid, df, o, max_cells=n)
for rep in range(3):
    t0 = time.perf_counter(); p = api.Plan(sp, grid, df, o, max_cells=n); t1 = time.perf_counter(); p.close(); t2 = time.perf_counter()
    print("with another plan alive: create %.1f ms, destroy %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
for rep in range(3):
    t0 = time.perf_counter(); _, st = api.smooth_spectra(cells, sp, grid, df, o); t1 = time.perf_counter()
    print("one-shot entry %.1f ms (kernels %.1f, h2d %.1f, d2h %.1f)" % ((t1 - t0) * 1e3, st["ms_prep"] + st["ms_main"] + st["ms_finalize"], st["ms_h2d"], st["ms_d2h"]), flush=True)
for cc in (219,):
    for rep in range(2):
        t0 = time.perf_counter(); _, st = api.smooth_spectra(cells, sp, grid, df, dict(o, cell_chunks=cc)); t1 = time.perf_counter()
        print("one-shot entry, cell_chunks %d: %.1f ms (kernels %.1f)" % (cc, (t1 - t0) * 1e3, st["ms_prep"] + st["ms_main"] + st["ms_finalize"]), flush=True)
