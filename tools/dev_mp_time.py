import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from is3d_amd import api, inputs, synth
g = inputs.grid()
grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
df, sp = inputs.df_tables(), inputs.species("urqmd")
o = dict(dimension=3, df_mode=2)
for n in (125000, 1000000):
    cells = synth.synth_surface(n, 3)
    for devs in ([0, 0], [0]):
        for rep in range(2):
            t0 = time.perf_counter(); a, st, _ = api.smooth_spectra_multi(cells, sp, grid, df, o, devices=devs); t1 = time.perf_counter()
            print(n, devs, "one-shot %.1f ms (kernels %.1f, h2d %.1f, d2h %.1f)" % ((t1 - t0) * 1e3, st["ms_prep"] + st["ms_main"] + st["ms_finalize"], st["ms_h2d"], st["ms_d2h"]), flush=True)
        t0 = time.perf_counter(); mp = api.MultiPlan(sp, grid, df, o, devices=devs, max_cells=n); t1 = time.perf_counter()
        print(n, devs, "create %.1f ms" % ((t1 - t0) * 1e3))
        for rep in range(3):
            t0 = time.perf_counter(); b, st, _ = mp.execute(cells); t1 = time.perf_counter()
            print(n, devs, "execute %.1f ms (kernels %.1f, h2d %.1f, d2h %.1f) bitwise %s" % ((t1 - t0) * 1e3, st["ms_prep"] + st["ms_main"] + st["ms_finalize"], st["ms_h2d"], st["ms_d2h"], np.array_equal(a, b)), flush=True)
        mp.close()
