#!/usr/bin/env python3
"""Child of tests/test_gpu_parity.py::test_prep_record_writers_write_the_same_streams, run with IS3D_USE_DEV_LIB=1 (the developer build of
the library, `make -C is3d_amd/csrc DEV=1`): cf_prep's alternative record writers (IS3D_PREP_PAIR = 0 | 1, read at every launch) and
non-temporal record stores (IS3D_PREP_SKIP bit 4) must give bitwise the spectra and cull counts of the default writer -- cell counts that are
not multiples of the batch, a workspace that forces several passes, grids that are not multiples of the tiles."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    assert api.DEV_LIB and b"DEV BUILD" in api.load().is3d_version(), "needs the developer build (IS3D_USE_DEV_LIB=1)"
    g = inputs.grid()
    grid0 = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df = inputs.df_tables()
    rng = np.random.default_rng(5)
    odd = dict(pT=np.array([0.1, 0.7, 2.5, 3.1, 4.4]), phi=np.sort(rng.random(7) * 2 * np.pi), y=np.linspace(-3.5, 3.5, 29), eta=np.linspace(-2.0, 2.0, 37),
               eta_w=np.full(37, 4.0 / 36))
    cases = [(3, 2, "urqmd", 1237, grid0, {}), (3, 1, "pikp", 333, odd, {}), (3, 2, "pikp", 1000, grid0, dict(workspace_bytes=1 << 22)),
             (2, 1, "pikp", 203, grid0, {}), (2, 2, "pikp", 37, odd, {}), (3, 2, "pikp", 100, grid0, dict(kernel_variant=4)),
             (2, 1, "pikp", 50, grid0, dict(kernel_variant=2))]
    for dim, dfm, species, n, grid, extra in cases:
        cells = synth.synth_surface(n, dim, seed=100 + n)
        sp = inputs.species(species)
        o = dict(dimension=dim, df_mode=dfm, **extra)
        for k in ("IS3D_PREP_PAIR", "IS3D_PREP_SKIP"):
            os.environ.pop(k, None)
        ref, st = api.smooth_spectra(cells, sp, grid, df, o)
        for env in (dict(IS3D_PREP_PAIR="0"), dict(IS3D_PREP_PAIR="1"), dict(IS3D_PREP_SKIP="16"), dict(IS3D_PREP_PAIR="1", IS3D_PREP_SKIP="16")):
            for k in ("IS3D_PREP_PAIR", "IS3D_PREP_SKIP"):
                os.environ.pop(k, None)
            os.environ.update(env)
            got, s2 = api.smooth_spectra(cells, sp, grid, df, o)
            assert np.array_equal(got, ref), (dim, dfm, species, n, env)
            assert s2["n_wave_rows_culled"] == st["n_wave_rows_culled"]
    print("writers agree on %d cases" % len(cases))


if __name__ == "__main__":
    main()
