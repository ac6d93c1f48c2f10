#!/usr/bin/env python3
"""Developer stress test (GPU box): the accumulator-relative row / unit cull against the same kernels with culling off, bit for
bit, on surfaces pushed away from the benchmark's statistics (skipped cells, strong viscous corrections, inflow cells, narrow
and wide rapidity ranges, few and many cells per chunk, all tile variants, baryon records, modified equilibrium)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from is3d_amd import api, inputs, synth

g = inputs.grid(); grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
df = inputs.df_tables(); dff = inputs.df_tables_full()
rng = np.random.default_rng(7)
bad = 0; n = 0
for trial in range(24):
    dim = 3 if trial % 3 else 2
    nc = int(rng.integers(40, 2500)) if dim == 3 else int(rng.integers(8, 120))
    baryon = (trial % 4 == 1)
    cells = synth.synth_surface(nc, dim, seed=1000 + trial, baryon=baryon)
    cells = {k: v.copy() for k, v in cells.items()}
    cells["eta"] *= rng.choice([0.1, 1.0, 1.5])
    for k in ("pixx", "pixy", "pixn", "piyy", "piyn"):
        cells[k] *= rng.choice([0.0, 1.0, 8.0])
    cells["bulkPi"] *= rng.choice([0.0, 1.0, 20.0])
    flip = rng.random(nc) < 0.15                                  # inflow-ish cells: spatial dsigma dominates
    for k in ("dax", "day"):
        cells[k][flip] *= 40.0
    skip = rng.random(nc) < 0.05
    for k in ("dat", "dax", "day", "dan"):
        cells[k][skip] *= -1.0
    sp = inputs.species("urqmd" if trial % 2 else [211, 321, 2212, -2212, 3122, 333])
    df_mode = int(rng.choice([1, 2]))
    o = dict(dimension=dim, df_mode=df_mode, cell_chunks=int(rng.choice([0, 1, 5])), kernel_variant=int(rng.choice([0, 2, 3, 4])))
    tabs = df
    if baryon:
        o.update(include_baryon=1, include_baryondiff_deltaf=1); tabs = dff
    a, sa = api.smooth_spectra(cells, sp, grid, tabs, dict(o, zero_skip=0))
    b, sb = api.smooth_spectra(cells, sp, grid, tabs, dict(o, zero_skip=2))
    ok = np.array_equal(a, b)
    n += 1; bad += (not ok)
    print("trial %2d dim %d cells %4d df %d baryon %d var %d chunks %d: culled %.3f  %s" % (trial, dim, nc, df_mode, baryon, o["kernel_variant"], o["cell_chunks"],
          sa["n_wave_rows_culled"] / max(sa["n_wave_rows"], 1), "identical" if ok else "DIFFERENT max rel %.3e" % np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))), flush=True)
    if trial % 3 == 0 and not baryon:
        fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
        for m in (3, 4):
            of = dict(dimension=dim, df_mode=m, cell_chunks=o["cell_chunks"])
            try:
                a, sa = api.smooth_spectra(cells, sp, grid, df, dict(of, zero_skip=0), fq=fq)
                b, sb = api.smooth_spectra(cells, sp, grid, df, dict(of, zero_skip=2), fq=fq)
            except api.Is3dError as e:
                print("   feqmod %d: %s" % (m, str(e)[:80])); continue
            ok = np.array_equal(a, b, equal_nan=True)
            n += 1; bad += (not ok)
            print("   feqmod %d: culled %.3f  %s" % (m, sa["n_wave_rows_culled"] / max(sa["n_wave_rows"], 1), "identical" if ok else "DIFFERENT"), flush=True)
print("cases %d, different %d" % (n, bad))
sys.exit(1 if bad else 0)
