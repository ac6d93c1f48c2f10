#!/usr/bin/env python3
"""(Lives under tests/ because its CPU leg calls the oracle, which only tests, smoke() and bench.py's cpu_baseline may do.)
Throughput of the device particle sampler (is3d_sample_particles) on the BASELINE config-3 surface (1e6 synthetic 3+1D
cells, 305 urqmd species, Chapman-Enskog delta-f) next to the CPU restatement (serial over cells, like the reference's
sample_dN_pTdpTdphidy) on a bounded slice of the same surface.  Not the BASELINE metric (that is bench.py); prints one JSON
line for profiles/."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=1000000)
    ap.add_argument("--events", type=int, default=20)
    ap.add_argument("--cpu-cells", type=int, default=50000)
    a = ap.parse_args()
    from is3d_amd import api, inputs, synth
    from oracle import oracle  # CPU baseline leg only
    sp = inputs.species("urqmd")
    df = inputs.df_tables()
    gla = inputs.feqmod_tables(0.15)
    o = dict(dimension=3, df_mode=2)
    cells = synth.synth_surface(a.cells, 3)
    api.sample_particles({k: v[:1000] for k, v in cells.items()}, sp, df, gla, o, n_events=1, seed=1)   # context + code objects
    t0 = time.perf_counter()
    p, st = api.sample_particles(cells, sp, df, gla, o, n_events=a.events, seed=20260002)
    wall = time.perf_counter() - t0          # two calls inside: count-only, then fill (the binding sizes the buffer first)
    dev_ms = st["ms_prep"] + st["ms_count"] + st["ms_fill"]
    nc = min(a.cpu_cells, a.cells)
    sub = {k: v[:nc] for k, v in cells.items()}
    t0 = time.perf_counter()
    ref, rst = oracle.sample_particles(sub, sp, df, gla, o, n_events=a.events, seed=20260002)
    cpu_s = time.perf_counter() - t0
    sel = p["cell"] < nc
    same = int(sel.sum()) == len(ref["E"]) and bool(np.array_equal(p["species"][sel], ref["species"]))
    res = dict(what="particle sampler, config-3 surface", cells=a.cells, species=len(sp["mass"]), events=a.events,
               particles=int(st["n_particles"]), hadrons_drawn=int(st["n_hadrons_drawn"]),
               momentum_sampling_efficiency=st["n_acceptances"] / max(st["n_momentum_samples"], 1),
               device_ms=dict(h2d=st["ms_h2d"], prep=st["ms_prep"], count=st["ms_count"], fill=st["ms_fill"]),
               particles_per_s_device=st["n_particles"] / (dev_ms * 1e-3), cell_events_per_s_device=a.cells * a.events / (dev_ms * 1e-3),
               wall_s_two_calls_with_transfers=wall,
               cpu=dict(kind="port", cores=1, cells=nc, seconds=cpu_s, particles=int(rst["n_kept"]), particles_per_s=rst["n_kept"] / cpu_s,
                        cell_events_per_s=nc * a.events / cpu_s, same_list_on_the_slice=same))
    res["device_over_cpu_cell_events"] = res["cell_events_per_s_device"] / res["cpu"]["cell_events_per_s"]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
