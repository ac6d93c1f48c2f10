#!/usr/bin/env python3
"""One-off: the random-configuration parity test of tests/test_gpu_fuzz.py on seeds beyond the suite's (cases 60 .. 60 + N), against the shipped library or,
with IS3D_USE_DEV_LIB=1, the developer build (all kernel variants)."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from is3d_amd import inputs  # noqa: E402
import test_gpu_fuzz as tf  # noqa: E402

g = inputs.grid()
fx = dict(grid=dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"]), grid_w=g, df=inputs.df_tables(), pikp=inputs.species("pikp"),
          urqmd=inputs.species("urqmd"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for case in range(60, 60 + n):
    try:
        tf.test_random_configuration_matches_the_oracle(fx, case)
    except Exception:
        bad += 1
        print("case", case, "FAILED")
        traceback.print_exc()
    if case % 20 == 0:
        print("case", case, "ok so far, failures:", bad, flush=True)
for case in range(16, 16 + n // 4):
    try:
        tf.test_random_sampler_configuration_gives_the_oracles_list(fx, case)
    except Exception:
        bad += 1
        print("sampler case", case, "FAILED")
        traceback.print_exc()
print("done: %d failures" % bad)
sys.exit(1 if bad else 0)
