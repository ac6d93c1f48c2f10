#!/usr/bin/env python3
"""Cycle accounting of cf_main_feqmod (df_mode 4, 8 x 7 tile) on the config-3 surface: the PROF instantiation of the developer build
(`make -C is3d_amd/csrc DEV=1`, run with IS3D_USE_DEV_LIB=1 IS3D_DEV_PROF=1) prints, per launch, where the wave cycles go -- staging issue,
barrier wait, dead units, live units (bounds + row tests | evaluated rows) -- for the three ways of walking a unit's rows (kernel_variant 5:
pipelined, 6: row mask + exact row thresholds, 3: row mask from the unit threshold only, the default; and 3 with one-wave workgroups), culling on and off.  stderr -> profiles/r04_cycle_accounting_feqmod.log"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["IS3D_USE_DEV_LIB"] = "1"
os.environ["IS3D_DEV_PROF"] = "1"
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    assert api.DEV_LIB
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df = inputs.df_tables()
    sp = inputs.species("urqmd")
    cells = synth.synth_surface(n, 3)
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
    dev = torch.device("cuda:0")
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream
    for variant, wpg in ((5, 0), (6, 0), (3, 0), (3, 1)):
        for zs in (0, 2):
            p = api.Plan(sp, grid, df, dict(dimension=3, df_mode=4, kernel_variant=variant, zero_skip=zs, waves_per_group=wpg), max_cells=n, fq=fq)
            p.set_timing(True)
            out = torch.zeros(p.output_size, dtype=torch.float64, device=dev)
            st = p.execute(n, ptrs, out.data_ptr(), stream)
            t = p.timings()
            print("# variant %d waves_per_group %d zero_skip %d: main %.2f ms (with the accounting's own overhead), wave-rows culled %.4f" % (
                variant, wpg, zs, t["ms_main"], st["n_wave_rows_culled"] / max(st["n_wave_rows"], 1)), file=sys.stderr, flush=True)
            p.close()
            del out


if __name__ == "__main__":
    main()
