"""GPU (-m gpu): the A/B kernel variants of rounds 1-5 live in the DEVELOPER build of the library only (make -C is3d_amd/csrc DEV=1 ->
is3d_amd/lib_dev; __graft_entry__.build() makes it): 1 direct, 2 / 4 other tile shapes, 5 hand-pipelined rows, 8 register-staged copy, 9 scalar
path for the delta-f kernels; the modified-equilibrium row walks 5, 6 and its 61-row tiles; the round-1 anisotropic-hydro kernel.  The shipped
library maps a request for one of them onto its default (status.kernel_variant says what ran).  Their parity tests are the tests marked `devlib`
(conftest.honoured gives every test the variants the library it runs on holds): this module re-runs those tests in ONE child process on the
developer build -- a library is chosen at import, and nothing that initialised the GPU may exec."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from is3d_amd import api

pytestmark = pytest.mark.gpu


def test_shipped_library_maps_ab_variants_onto_its_defaults(fx):
    if api.DEV_LIB:
        pytest.skip("this process runs the developer build")
    from is3d_amd import inputs, synth
    cells = synth.synth_surface(50, 3, seed=5)
    ref, st = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=2))
    assert st["kernel_variant"] == 6
    for v in (1, 2, 4, 5, 8, 9):
        got, s = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=2, kernel_variant=v))
        assert s["kernel_variant"] == 6 and (got == ref).all(), v
    _, s = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=2, kernel_variant=3))
    assert s["kernel_variant"] == 3                      # the 8 x 7 tile without the E2 stream: also the default for pT grids of more than 32 values
    c2 = synth.synth_surface(6, 2, seed=5)
    for v in (1, 2, 3, 4, 8):
        _, s = api.smooth_spectra(c2, fx["pikp"], fx["grid"], fx["df"], dict(dimension=2, df_mode=1, kernel_variant=v))
        assert s["kernel_variant"] == 7, v
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
    for v in (2, 4, 5, 6):
        _, s = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=4, kernel_variant=v), fq=fq)
        assert s["kernel_variant"] == 3, v
    vc = synth.synth_vah_surface(20, 3, seed=2)
    _, s = api.smooth_spectra_vah(vc, fx["pikp"], fx["grid"], dict(dimension=3, kernel_variant=2))
    assert s["kernel_variant"] == 3


def test_ab_variants_on_the_developer_build():
    if api.DEV_LIB:
        pytest.skip("already the developer build (this is the child)")
    dev = os.path.join(ROOT, "is3d_amd", "lib_dev", "libis3d_amd.so")
    if not os.path.exists(dev):
        pytest.skip("developer build absent (make -C is3d_amd/csrc DEV=1)")
    env = dict(os.environ, IS3D_USE_DEV_LIB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-x", "-m", "gpu and devlib", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    last = [ln for ln in r.stdout.splitlines() if " passed" in ln]
    assert last and " failed" not in last[-1], tail
    print("developer build:", last[-1].strip())
