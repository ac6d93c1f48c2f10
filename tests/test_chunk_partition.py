"""CPU: the cell partition of the main kernels' grids (is3d::chunk_cells, is3d_amd/csrc/cf_device.h) -- a host-only build of the same inline
function the kernels call.  The chunks tile [0, n_cells) without gaps; with a tapered tail the last nch_small chunks hold a quarter of the
cells of the others; nch_small = 0 is the partition of rounds 1-3 (c0 = chunk n_cells / nch)."""
import os
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("chunk") / "chunk_cells_main")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "chunk_cells_main.hip"), "-o", out],
                   check=True, capture_output=True, timeout=600)
    return out


def _partition(prog, n, nch, small):
    rows = subprocess.run([prog, str(n), str(nch), str(small)], check=True, capture_output=True, text=True, timeout=60).stdout.split("\n")
    return [tuple(int(x) for x in r.split()) for r in rows if r]


@pytest.mark.parametrize("n,nch,small", [(1000000, 887, 24), (125000, 162, 24), (6144, 24, 24), (1000003, 100, 0), (125000, 144, 0), (7, 3, 0)])
def test_chunks_tile_the_cells(prog, n, nch, small):
    part = _partition(prog, n, nch, small)
    assert [p[0] for p in part] == list(range(nch))
    assert part[0][1] == 0 and part[-1][2] == n
    assert all(a[2] == b[1] for a, b in zip(part, part[1:]))             # no gap, no overlap
    sizes = [c1 - c0 for _, c0, c1 in part]
    assert min(sizes) >= 0
    nbig = nch - small
    if small == 0:
        assert all(c0 == (c * n) // nch for c, c0, _ in part)           # the uniform partition of rounds 1-3
        assert max(sizes) - min(sizes) <= 1
    else:
        big, tail = sizes[:nbig], sizes[nbig:]
        if big:
            assert max(big) - min(big) <= 1
            assert all(abs(4 * t - big[0]) <= 4 for t in tail)          # quarter-size chunks at the end
        assert max(tail) - min(tail) <= 1
