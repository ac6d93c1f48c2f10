"""CPU: the C-ABI library loads, exports every symbol include/is3d_amd.h declares, validates its arguments
and refuses to compute without a GPU (no CPU fallback).  No compute call succeeds on a box without a device."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from is3d_amd import api, inputs, synth


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "is3d_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(is3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = api.load()
    names = declared_symbols()
    assert len(names) >= 18 and set(names) == set(api.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.is3d_version().startswith(b"is3d_amd")
    assert lib.is3d_device_count() >= 0


def test_shared_object_is_pure_c_abi():
    """Only is3d_* functions are exported with C linkage from the boundary; no torch / pybind symbols."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH], text=True)
    syms = [ln.split()[-1] for ln in out.splitlines() if " T " in ln]
    c_syms = [s for s in syms if not s.startswith("_Z")]
    assert set(api.EXPORTS) <= set(c_syms)
    assert not [s for s in syms if "torch" in s.lower() or "pybind" in s.lower()]


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors == what a C compiler makes of include/is3d_amd.h (sizes and a few offsets)."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "is3d_amd.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(is3d_options), offsetof(is3d_options, workspace_bytes), sizeof(is3d_status), offsetof(is3d_status, ms_prep),'
                   'sizeof(is3d_cells), sizeof(is3d_grid), sizeof(is3d_species), sizeof(is3d_df_tables), offsetof(is3d_status, bad_cell),'
                   'sizeof(is3d_feqmod_tables), offsetof(is3d_feqmod_tables, T_avg), offsetof(is3d_status, n_cells_narrow));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    c = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    py = [ctypes.sizeof(api.Options), api.Options.workspace_bytes.offset, ctypes.sizeof(api.Status), api.Status.ms_prep.offset,
          ctypes.sizeof(api.Cells), ctypes.sizeof(api.Grid), ctypes.sizeof(api.Species), ctypes.sizeof(api.DfTables), api.Status.bad_cell.offset,
          ctypes.sizeof(api.FeqmodTables), api.FeqmodTables.T_avg.offset, api.Status.n_cells_narrow.offset]
    assert c == py


def test_argument_validation_precedes_device_use(fx):
    """Unsupported options are refused with IS3D_EINVAL and a message (reference: printf + exit(-1))."""
    sp, g, df = fx["pikp"], fx["grid"], fx["df"]
    for bad in (dict(dimension=4), dict(df_mode=3), dict(df_mode=4), dict(include_baryon=1)):
        with pytest.raises(api.Is3dError) as e:
            api.Plan(sp, g, df, bad)
        assert e.value.code == api.IS3D_EINVAL, bad
    with pytest.raises(api.Is3dError) as e:
        api.Plan(sp, g, dict(df, T=df["T"][::-1].copy()), dict(df_mode=1))
    assert e.value.code == api.IS3D_EINVAL
    with pytest.raises(api.Is3dError):
        api.Plan(dict(sp, mass=sp["mass"][:0], sign=sp["sign"][:0], degeneracy=sp["degeneracy"][:0], baryon=sp["baryon"][:0]), g, df)


def test_no_cpu_fallback(fx):
    """Without a HIP device the product path fails loudly with IS3D_ENODEVICE instead of computing on the host."""
    if api.load().is3d_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu tests")
    cells = synth.synth_surface(4, 3)
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=1))
    assert e.value.code == api.IS3D_ENODEVICE
    assert "no CPU path" in str(e.value)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under is3d_amd/ or include/ may import, link or name it."""
    pat = re.compile(r"oracle", re.I)
    for base in ("is3d_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if "__pycache__" in dp or os.sep + "lib" in dp or os.sep + "bin" in dp:
                continue
            for f in fn:
                if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                    assert not pat.search(open(os.path.join(dp, f), errors="ignore").read()), os.path.join(dp, f)
    ldd = subprocess.check_output(["ldd", api.LIB_PATH], text=True)
    assert "oracle" not in ldd
    # the same for helper scripts: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg call the checker
    imp = re.compile(r"^\s*(from\s+oracle|import\s+oracle)", re.M)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            assert not imp.search(open(os.path.join(ROOT, "tools", f), errors="ignore").read()), f


def test_shard_bounds_and_multi_entry_argument_checks(fx):
    """is3d_shard_bounds is the split bench.py / is3d_amd.dist use; the multi-device entry validates before touching a device
    and, like every compute entry, has no CPU path."""
    from is3d_amd import dist as idist
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            assert [api.shard_bounds(n, r, w) for r in range(w)] == [idist.shard_bounds(n, r, w) for r in range(w)]
    with pytest.raises(api.Is3dError):
        api.shard_bounds(10, 2, 2)
    cells = synth.synth_surface(4, 3)
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=1), devices=[0], reduce=7)
    assert e.value.code == api.IS3D_EINVAL
    if api.load().is3d_device_count() == 0:
        with pytest.raises(api.Is3dError) as e:
            api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=1), devices=[0, 0])
        assert e.value.code == api.IS3D_ENODEVICE
    else:
        with pytest.raises(api.Is3dError) as e:
            api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], dict(dimension=3, df_mode=1), devices=[0, 99])
        assert e.value.code == api.IS3D_EINVAL
