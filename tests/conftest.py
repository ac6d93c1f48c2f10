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
