#!/usr/bin/env python3
"""Developer A/B on a GPU box: time the main kernel for several option sets on one resident surface, interleaved
rounds in one process (cdna_hip_programming.md rule 24), and check that the spectra agree."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=200000)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--df", type=int, default=2)
    ap.add_argument("--species", default="urqmd")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--baryon", action="store_true", help="include_baryon = 1 with baryon diffusion (full (T, mu_B) tables, muB/nB/V^mu cell arrays)")
    ap.add_argument("--sets", default="variant=3;variant=5;variant=3,zero_skip=2;variant=5,zero_skip=2")
    a = ap.parse_args()
    import torch
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df = inputs.df_tables_full() if a.baryon else inputs.df_tables()
    sp = inputs.species(a.species)
    cells = synth.synth_surface(a.cells, a.dim, baryon=a.baryon)
    dev = torch.device("cuda:0")
    fields = synth.CELL_FIELDS + (synth.BARYON_FIELDS if a.baryon else [])
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in fields}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream
    sets = []
    for s in a.sets.split(";"):
        o = dict(dimension=a.dim, df_mode=a.df)
        if a.baryon:
            o.update(include_baryon=1, include_baryondiff_deltaf=1)
        for kv in s.split(","):
            k, v = kv.split("=")
            o["kernel_variant" if k == "variant" else k] = int(v)
        sets.append((s, o))
    plans, outs, times, preps = [], [], [[] for _ in sets], [[] for _ in sets]
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells)) if a.df in (3, 4) else None   # modified equilibrium: Gauss-Laguerre nodes, PDG list, T_avg
    culled = [None] * len(sets)
    for s, o in sets:
        p = api.Plan(sp, grid, df, o, max_cells=a.cells, fq=fq)
        p.set_timing(True)
        plans.append(p)
        outs.append(torch.zeros(p.output_size, dtype=torch.float64, device=dev))
    for r in range(a.rounds + 1):
        for i, p in enumerate(plans):
            p.execute(a.cells, ptrs, outs[i].data_ptr(), stream, want_status=False)
            t = p.timings()
            if r > 0:
                times[i].append(t["ms_main"])
                preps[i].append(t["ms_prep"])
    for i, p in enumerate(plans):   # one more execute each with the status read back: the culled fraction of the wave-rows
        st = p.execute(a.cells, ptrs, outs[i].data_ptr(), stream)
        culled[i] = st["n_wave_rows_culled"] / max(st["n_wave_rows"], 1)
    ref = outs[0].cpu().numpy()
    nb = len(grid["pT"]) * len(grid["phi"]) * (len(grid["y"]) if a.dim == 3 else 1) * len(sp["mass"])
    for i, (s, o) in enumerate(sets):
        got = outs[i].cpu().numpy()
        err = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-280)))
        ms = np.array(times[i])
        print("%-40s %s tile=%s main ms: median %.2f min %.2f  prep ms %.2f  workspace %.1f GB -> %.3e evals/s   wave-rows culled %.4f   max rel diff vs first %.2e  bitwise %s" % (
            s, plans[i].main_kernel_name, plans[i].tile_shape, np.median(ms), ms.min(), np.median(preps[i]), plans[i].workspace_bytes / 1e9,
            a.cells * nb / (np.median(ms) * 1e-3), culled[i], err, bool(np.array_equal(got, ref))), flush=True)


if __name__ == "__main__":
    main()
