#!/usr/bin/env python3
"""Developer timing of the host-side text I/O at BASELINE config-3 size, no GPU needed: a 1e6-cell mode-1 surface.dat (0.49 GB) through the
two-call reader (is3d_surface_read: two reads + two parses), through is3d_surface_open without / with the binary sidecar, and the 305-species
results/ text (0.62 GB x 2 files) through is3d_write_results.  Scratch under /tmp."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from is3d_amd import api, inputs, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    root = tempfile.mkdtemp(prefix="is3d_io_", dir="/tmp")
    path = os.path.join(root, "surface.dat")
    cells = synth.synth_surface(n, 3)
    synth.write_surface_dat(path, cells)
    L = api.load()
    import ctypes as C

    def t(f):
        t0 = time.perf_counter()
        r = f()
        return time.perf_counter() - t0, r

    dt, _ = t(lambda: api.surface_read(path, 1, 0, 0, 3))
    print("two-call reader (count + fill, + numpy allocation): %.3f s" % dt, flush=True)
    for label, cache in (("open, cache off", 0), ("open, cache on (first: parse + write sidecar)", 1), ("open, cache on (second: sidecar)", 1), ("open, whole-file hash (sidecar)", 2)):
        h = C.c_void_p()
        L.is3d_surface_open.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.is3d_surface_source.argtypes = [C.c_void_p]
        L.is3d_surface_close.argtypes = [C.c_void_p]
        L.is3d_surface_close.restype = None
        t0 = time.perf_counter()
        rc = L.is3d_surface_open(path.encode(), 1, 0, 0, 3, cache, C.byref(h))
        t1 = time.perf_counter()
        src = L.is3d_surface_source(h)
        t2 = time.perf_counter()
        L.is3d_surface_close(h)
        print("%-50s rc %d  open %.3f s  (+ %.3f s until the sidecar writer has finished)  source %d" % (label, rc, t1 - t0, t2 - t1, src), flush=True)
    g = inputs.grid()
    sp = inputs.species("urqmd")
    ids = list(inputs.load_fixture()["chosen_urqmd"])
    rng = np.random.default_rng(0)
    dN = rng.random(len(ids) * 32 * 24 * 21) * 1e-3
    res = os.path.join(root, "results")
    os.makedirs(os.path.join(res, "vn_continuous"))
    dt, _ = t(lambda: api.write_results(res, 3, ids, g["pT"], g["pT_w"], g["phi"], g["phi_w"], g["y"], dN))
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(res) for f in fs)
    print("is3d_write_results, %d species: %.3f s for %.1f MB" % (len(ids), dt, size / 1e6), flush=True)
    shutil.rmtree(root)


if __name__ == "__main__":
    main()
