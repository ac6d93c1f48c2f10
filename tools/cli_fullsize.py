#!/usr/bin/env python3
"""End-to-end run of the command line driver at BASELINE config-3 size on the GPU box: a 1e6-cell surface.dat (~0.5 GB of
text) in the reference's run-directory layout -> iS3D_amd -> results/ (0.6 GB of text); prints the driver's own wall-time
split and spot-checks the written spectrum against the library called directly.  Scratch files live under /tmp."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import refformat
    from is3d_amd import api, inputs, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    fx = inputs.load_fixture()
    ids = fx["chosen_urqmd"]
    root = tempfile.mkdtemp(prefix="is3d_full_", dir="/tmp")
    t0 = time.time()
    cells = synth.synth_surface(n, 3)
    refformat.make_run_dir(root, cells, ids, dict(operation=1, dimension=3, df_mode=2))
    t_make = time.time() - t0
    size_in = os.path.getsize(os.path.join(root, "input", "surface.dat"))
    import filecmp
    runs = []
    for k in range(2):     # first run: parses the text and writes input/surface.dat.is3dcache; second run: loads the sidecar
        if k == 1:
            shutil.move(os.path.join(root, "results"), os.path.join(root, "results_first"))
            os.makedirs(os.path.join(root, "results", "vn_continuous"))
        t0 = time.time()
        r = subprocess.run([api.CLI_PATH], cwd=root, capture_output=True, text=True, timeout=3000)
        t_cli = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.split("\n") if ln.startswith("wall:") or ln.startswith("device time:") or "classes" in ln or ln.startswith("surface:")]
        runs.append(dict(cli_wall_s=t_cli, cli_report=lines))
    # the two runs' results/ must be the same bytes (the sidecar holds exactly what the text parse produced)
    names = sorted(os.path.relpath(os.path.join(d, f), os.path.join(root, "results")) for d, _, fs in os.walk(os.path.join(root, "results")) for f in fs)
    match, mismatch, errors = filecmp.cmpfiles(os.path.join(root, "results_first"), os.path.join(root, "results"), names, shallow=False)
    assert not mismatch and not errors and len(match) == len(names) > 900, (len(match), mismatch[:3], errors[:3])
    lines = runs[0]["cli_report"]
    t_cli = runs[0]["cli_wall_s"]
    size_out = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(root, "results")) for f in fs)
    # spot check: first species block of the concatenated file against the library on the parsed surface
    parsed = refformat.read_surface_like_reference(os.path.join(root, "input", "surface.dat")) if n <= 200000 else None
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    sp = inputs.species(ids)
    want, _ = api.smooth_spectra(parsed if parsed is not None else cells, sp, grid, inputs.df_tables(), dict(dimension=3, df_mode=2))
    want = want.reshape(21, 24, 32, len(ids))
    got = []
    with open(os.path.join(root, "results", "dN_pTdpTdphidy_211.dat")) as f:
        f.readline()
        for ln in f:
            if ln.strip():
                got.append(float(ln.split("\t")[3]))
    got = np.array(got).reshape(21, 24, 32)
    i211 = ids.index(211)
    err = float(np.max(np.abs(got - want[:, :, :, i211]) / np.maximum(np.abs(want[:, :, :, i211]), 1e-250)))
    print(json.dumps(dict(cells=n, species=len(ids), surface_MB=size_in / 1e6, results_MB=size_out / 1e6, make_inputs_s=t_make, cli_wall_s=t_cli,
                          cli_report=lines, second_run_with_sidecar=runs[1], results_files_identical=len(match),
                          sidecar_MB=os.path.getsize(os.path.join(root, "input", "surface.dat.is3dcache")) / 1e6,
                          file_vs_library_relerr_pi_plus=err)), flush=True)
    assert err < 3e-8     # the files carry 9 significant digits; the surface text carries 17
    shutil.rmtree(root)


if __name__ == "__main__":
    main()
