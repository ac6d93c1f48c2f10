import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"   # present only in the build container; never read by -m gpu tests


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: reads data files under /root/reference (container only)")
    config.addinivalue_line("markers", "devlib: has work for the developer build of the library too (A/B kernel variants): tests/test_gpu_devlib.py re-runs it there")
    # a fresh checkout has no built library (build products are git-ignored): build it once, as __graft_entry__.build() does
    lib = os.path.join(ROOT, "is3d_amd", "lib", "libis3d_amd.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "is3d_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "reference" in item.keywords and not os.path.isdir(REFERENCE):
            item.add_marker(skip_ref)


def honoured(kind, dim, variants, baryon=False):
    """The kernel variants of `variants` that the library loaded in THIS process runs as requested.  The shipped library (is3d_amd/lib) holds the
    kernels the plan's defaults reach -- delta-f: 6 (E2 table stream) and 3 (8 x 7 without it, no baryon slots) in 3+1D, 7 in 2+1D; modified
    equilibrium ("fq"): 3 in 3+1D, 7 in 2+1D; anisotropic hydro ("vah"): 3 -- and maps every other request onto them; the A/B forms of rounds 1-5
    exist in the developer build (is3d_amd/lib_dev, IS3D_USE_DEV_LIB=1), where tests/test_gpu_devlib.py re-runs the tests marked `devlib`."""
    from is3d_amd import api
    if api.DEV_LIB:
        return list(variants)
    if kind == "vah":
        ok = {0, 3}
    elif kind == "fq":
        ok = {0, 3} if dim == 3 else {0, 7}
    else:
        ok = ({0, 6} | (set() if baryon else {3})) if dim == 3 else {0, 7}
    return [v for v in variants if v in ok]


def relerr(a, b, floor=1e-280):
    """max |a-b| / max(|b|, floor): relative error with an absolute floor for the exactly-zero / denormal tails
    (pT up to 40 GeV, |y - eta| up to 9: values down to 1e-300 and exact zeros)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


@pytest.fixture(scope="session")
def fx():
    from is3d_amd import inputs
    g = inputs.grid()
    return dict(grid=dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"]), grid_w=g,
                df=inputs.df_tables(), pikp=inputs.species("pikp"), urqmd=inputs.species("urqmd"))


@pytest.fixture(scope="session")
def pins():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden_pins.json")) as f:
        return json.load(f)
