"""ctypes binding of include/is3d_amd.h (lib/libis3d_amd.so) -- plumbing, not the product.

The compute entry points run HIP kernels only; there is no CPU fallback here or in the library:
without the built extension `load()` raises, without a GPU the library returns IS3D_ENODEVICE.
PyTorch is used by callers for device memory and streams; this module itself needs only ctypes+numpy.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IS3D_USE_DEV_LIB=1: the developer build (`make -C is3d_amd/csrc DEV=1`: A/B switches and cycle-accounting kernels, csrc/errors.h) for tools/
DEV_LIB = os.environ.get("IS3D_USE_DEV_LIB", "") == "1"
LIB_PATH = os.path.join(_HERE, "lib_dev" if DEV_LIB else "lib", "libis3d_amd.so")
CLI_PATH = os.path.join(_HERE, "bin_dev" if DEV_LIB else "bin", "iS3D_amd")

IS3D_OK, IS3D_EINVAL, IS3D_ENODEVICE, IS3D_EDOMAIN, IS3D_ENOMEM, IS3D_EIO = 0, -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)

CELL_FIELDS = ["tau", "eta", "dat", "dax", "day", "dan", "ux", "uy", "un", "T", "P", "E",
               "pixx", "pixy", "pixn", "piyy", "piyn", "bulkPi", "muB", "nB", "Vx", "Vy", "Vn"]
# order of the cell_arrays23 argument of is3d_surface_read_vh
SURFACE_READ_ORDER = ["T", "P", "E", "tau", "eta", "ux", "uy", "un", "dat", "dax", "day", "dan",
                      "pixx", "pixy", "pixn", "piyy", "piyn", "bulkPi", "muB", "nB", "Vx", "Vy", "Vn"]


class Cells(C.Structure):
    _fields_ = [("n_cells", C.c_int64)] + [(n, C.c_void_p) for n in CELL_FIELDS]


class Species(C.Structure):
    _fields_ = [("n", C.c_int32), ("mass", _dp), ("sign", _dp), ("degeneracy", _dp), ("baryon", _dp)]


class Grid(C.Structure):
    _fields_ = [("n_pT", C.c_int32), ("pT", _dp), ("n_phi", C.c_int32), ("phi", _dp), ("n_y", C.c_int32), ("y", _dp),
                ("n_eta", C.c_int32), ("eta", _dp), ("eta_w", _dp)]


class DfTables(C.Structure):
    _fields_ = [("n_T", C.c_int32), ("T", _dp), ("n_muB", C.c_int32), ("muB", _dp)] + \
               [(n, _dp) for n in ["c0", "c1", "c2", "c3", "c4", "F", "G", "betabulk", "betaV", "betapi"]]


class FeqmodTables(C.Structure):
    _fields_ = [("n_gla", C.c_int32), ("root1", _dp), ("weight1", _dp), ("root2", _dp), ("weight2", _dp), ("n_pdg", C.c_int32),
                ("pdg_mass", _dp), ("pdg_degeneracy", _dp), ("pdg_sign", _dp), ("T_avg", C.c_double), ("deta_min", C.c_double),
                ("mass_pion0", C.c_double)]


class Particle(C.Structure):
    _fields_ = [("cell", C.c_int64), ("event", C.c_int32), ("species", C.c_int32)] + \
               [(n, C.c_double) for n in ["tau", "x", "y", "eta", "t", "z", "E", "px", "py", "pz"]]


PARTICLE_DTYPE = np.dtype([("cell", "<i8"), ("event", "<i4"), ("species", "<i4")] +
                          [(n, "<f8") for n in ["tau", "x", "y", "eta", "t", "z", "E", "px", "py", "pz"]])


class SamplerInputs(C.Structure):
    _fields_ = [("n_events", C.c_int32), ("n_gla", C.c_int32), ("seed", C.c_uint64), ("y_cut", C.c_double), ("first_cell", C.c_int64),
                ("x", _dp), ("y", _dp), ("root1", _dp), ("weight1", _dp), ("feqmod", C.POINTER(FeqmodTables)), ("fast", C.c_int32),
                ("batch_events", C.c_int32), ("T_avg", C.c_double), ("T_avg_switch", C.c_double), ("muB_avg", C.c_double)]


class SamplerStats(C.Structure):
    _fields_ = [("n_cells_skipped", C.c_int64), ("n_hadrons_drawn", C.c_int64), ("n_momentum_samples", C.c_int64),
                ("n_acceptances", C.c_int64), ("n_classes", C.c_int32), ("reserved", C.c_int32), ("n_cells_breakdown", C.c_int64),
                ("ms_h2d", C.c_double),
                ("ms_prep", C.c_double), ("ms_count", C.c_double), ("ms_fill", C.c_double), ("ms_density", C.c_double), ("ms_poisson", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Options(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ["dimension", "df_mode", "include_baryon", "include_bulk_deltaf",
                                         "include_shear_deltaf", "include_baryondiff_deltaf", "regulate_deltaf", "outflow",
                                         "accumulate", "device", "kernel_variant", "cell_chunks"]] + \
               [("workspace_bytes", C.c_int64), ("collapse_species", C.c_int32), ("zero_skip", C.c_int32), ("waves_per_group", C.c_int32), ("reference_bilinear_indexing", C.c_int32),
                ("reserved", C.c_int32 * 4)]


class Status(C.Structure):
    _fields_ = [("code", C.c_int32), ("n_classes", C.c_int32), ("n_cells_skipped", C.c_int64), ("bad_cell", C.c_int64),
                ("n_passes", C.c_int32), ("kernel_variant", C.c_int32), ("ms_prep", C.c_double), ("ms_main", C.c_double),
                ("ms_finalize", C.c_double), ("ms_h2d", C.c_double), ("ms_d2h", C.c_double), ("n_wave_rows", C.c_int64),
                ("n_wave_rows_culled", C.c_int64), ("n_cells_breakdown", C.c_int64), ("n_cells_narrow", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTS = ["is3d_last_error", "is3d_version", "is3d_device_count", "is3d_smooth_spectra", "is3d_smooth_spectra_feqmod", "is3d_plan_create",
           "is3d_plan_create_feqmod",
           "is3d_probe_shader_clock", "is3d_plan_output_size", "is3d_plan_execute", "is3d_plan_set_timing", "is3d_plan_timings", "is3d_plan_observables",
           "is3d_plan_main_kernel_name", "is3d_plan_tile_shape", "is3d_plan_workspace_bytes", "is3d_plan_destroy", "is3d_param_get",
           "is3d_table_read", "is3d_surface_read_vh", "is3d_surface_read", "is3d_pdg_read", "is3d_pdg_read_box", "is3d_df_table_read", "is3d_df_table_read_full",
           "is3d_gla_read", "is3d_write_results", "is3d_sample_particles", "is3d_write_particle_list_osc",
           "is3d_run_particlization", "is3d_run_result_free", "is3d_write_sampler_tests", "is3d_smooth_spectra_vah",
           "is3d_smooth_spectra_multi", "is3d_shard_bounds", "is3d_comm_unique_id", "is3d_comm_create", "is3d_comm_rank",
           "is3d_comm_allreduce", "is3d_comm_destroy", "is3d_plan_execute_allreduce", "is3d_run_particlization_on", "is3d_total_yield", "is3d_plan_check", "is3d_sample_particles_multi",
           "is3d_comm_check", "is3d_comm_abort", "is3d_comm_timings", "is3d_comm_set_timeout", "is3d_comm_synchronize", "is3d_surface_open", "is3d_surface_cells", "is3d_surface_source", "is3d_surface_from_sidecar",
           "is3d_surface_arrays", "is3d_surface_close", "is3d_sampler_plan_create", "is3d_sampler_plan_execute", "is3d_sampler_plan_destroy", "is3d_multi_plan_create", "is3d_multi_plan_execute",
           "is3d_multi_plan_shards", "is3d_multi_plan_output_size", "is3d_multi_plan_destroy",
           "is3d_vah_df_read", "is3d_vah_coefficients", "is3d_smooth_spectra_vah_df", "is3d_vah_plan_create", "is3d_vah_plan_output_size",
           "is3d_vah_plan_workspace_bytes", "is3d_vah_plan_execute", "is3d_vah_plan_set_timing", "is3d_vah_plan_timings",
           "is3d_vah_plan_tile_shape", "is3d_vah_plan_destroy", "is3d_surface_read_vah", "is3d_vah_plan_main_kernel_name", "is3d_math_probe", "is3d_resource_counters"]

REDUCE_ORDERED, REDUCE_RCCL = 0, 1
IS3D_EPEER = -6
COMM_ID_BYTES = 128


class Is3dError(RuntimeError):
    def __init__(self, code, msg, bad_cell=None):
        super().__init__("is3d_amd error %d: %s" % (code, msg))
        self.code = code
        self.bad_cell = bad_cell


_LIB = None


def build(verbose=False):
    """Compile the HIP library and the CLI in-tree (hipcc --offload-arch=gfx950)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], stdout=out)
    return LIB_PATH


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C is3d_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    try:
        # torch bundles its own libamdhip64.so.7; if our library pulled in /opt/rocm's copy first, the
        # process would hold two HIP runtimes and the second one finds no GPU.  Importing torch first
        # makes the loader resolve our DT_NEEDED libamdhip64.so.7 to the copy that is already mapped.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.is3d_last_error.restype = C.c_char_p
    L.is3d_version.restype = C.c_char_p
    L.is3d_plan_main_kernel_name.restype = C.c_char_p
    L.is3d_plan_main_kernel_name.argtypes = [C.c_void_p]
    L.is3d_smooth_spectra.argtypes = [C.POINTER(Cells), C.POINTER(Species), C.POINTER(Grid), C.POINTER(DfTables),
                                      C.POINTER(Options), _dp, C.POINTER(Status)]
    L.is3d_plan_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Species), C.POINTER(Grid), C.POINTER(DfTables),
                                   C.POINTER(Options), C.c_int64]
    L.is3d_smooth_spectra_feqmod.argtypes = [C.POINTER(Cells), C.POINTER(Species), C.POINTER(Grid), C.POINTER(DfTables),
                                             C.POINTER(FeqmodTables), C.POINTER(Options), _dp, C.POINTER(Status)]
    L.is3d_plan_create_feqmod.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Species), C.POINTER(Grid), C.POINTER(DfTables),
                                          C.POINTER(FeqmodTables), C.POINTER(Options), C.c_int64]
    L.is3d_probe_shader_clock.argtypes = [C.c_int32, C.c_double, C.POINTER(C.c_double)]
    L.is3d_plan_output_size.restype = C.c_int64
    L.is3d_plan_output_size.argtypes = [C.c_void_p]
    L.is3d_plan_workspace_bytes.restype = C.c_int64
    L.is3d_plan_workspace_bytes.argtypes = [C.c_void_p]
    L.is3d_plan_tile_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.is3d_plan_execute.argtypes = [C.c_void_p, C.POINTER(Cells), C.c_void_p, C.c_void_p, C.POINTER(Status)]
    L.is3d_plan_observables.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.is3d_plan_set_timing.argtypes = [C.c_void_p, C.c_int32]
    L.is3d_plan_timings.argtypes = [C.c_void_p, C.POINTER(Status)]
    L.is3d_plan_destroy.argtypes = [C.c_void_p]
    L.is3d_plan_destroy.restype = None
    L.is3d_param_get.argtypes = [C.c_char_p, C.c_char_p, _dp]
    L.is3d_table_read.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _dp, C.c_int64]
    L.is3d_surface_read_vh.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64),
                                       C.POINTER(_dp), _dp]
    L.is3d_surface_read.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64),
                                    C.POINTER(_dp), _dp]
    L.is3d_pdg_read.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), _dp, _dp, _dp, _dp, C.c_int32]
    L.is3d_df_table_read.argtypes = [C.c_char_p, C.POINTER(C.c_int32), _dp, _dp, C.c_int32]
    L.is3d_df_table_read_full.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp, C.c_int64]
    L.is3d_sample_particles.argtypes = [C.POINTER(Cells), C.POINTER(Species), C.POINTER(DfTables), C.POINTER(SamplerInputs),
                                        C.POINTER(Options), C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(SamplerStats)]
    L.is3d_write_particle_list_osc.argtypes = [C.c_char_p, C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
    L.is3d_gla_read.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, C.c_int64]
    L.is3d_write_results.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int32, _dp, _dp,
                                     C.c_int32, _dp, _dp, C.c_int32, _dp, _dp]
    L.is3d_smooth_spectra_multi.argtypes = [C.POINTER(Cells), C.POINTER(Species), C.POINTER(Grid), C.POINTER(DfTables),
                                            C.POINTER(FeqmodTables), C.POINTER(Options), C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                            _dp, C.POINTER(Status), C.POINTER(Status)]
    L.is3d_shard_bounds.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.is3d_comm_unique_id.argtypes = [C.c_char_p]
    L.is3d_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
    L.is3d_comm_rank.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.is3d_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.is3d_comm_destroy.argtypes = [C.c_void_p]
    L.is3d_comm_destroy.restype = None
    L.is3d_plan_execute_allreduce.argtypes = [C.c_void_p, C.POINTER(Cells), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Status)]
    L.is3d_comm_check.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.is3d_comm_abort.argtypes = [C.c_void_p]
    L.is3d_comm_timings.argtypes = [C.c_void_p, _dp]
    L.is3d_plan_check.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.is3d_multi_plan_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Species), C.POINTER(Grid), C.POINTER(DfTables),
                                         C.POINTER(FeqmodTables), C.POINTER(Options), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int64]
    L.is3d_multi_plan_execute.argtypes = [C.c_void_p, C.POINTER(Cells), _dp, C.POINTER(Status), C.POINTER(Status)]
    L.is3d_multi_plan_shards.argtypes = [C.c_void_p]
    L.is3d_multi_plan_output_size.argtypes = [C.c_void_p]
    L.is3d_multi_plan_output_size.restype = C.c_int64
    L.is3d_multi_plan_destroy.argtypes = [C.c_void_p]
    L.is3d_multi_plan_destroy.restype = None
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise Is3dError(rc, load().is3d_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


DF_NAMES_2D = ["c0", "c1", "c2", "c3", "c4", "F", "G", "betabulk", "betaV", "betapi"]

DEFAULT_OPTS = dict(dimension=3, df_mode=1, include_baryon=0, include_bulk_deltaf=1, include_shear_deltaf=1,
                    include_baryondiff_deltaf=0, regulate_deltaf=1, outflow=1, accumulate=0, device=-1,
                    kernel_variant=0, cell_chunks=0, workspace_bytes=0, collapse_species=0, zero_skip=0, waves_per_group=0,
                    reference_bilinear_indexing=0)


def _pack_common(species, grid, df, opts):
    o = dict(DEFAULT_OPTS)
    o.update(opts or {})
    keep = {}
    sp = {k: _f64(species[k]) for k in ["mass", "sign", "degeneracy", "baryon"]}
    g = {k: _f64(grid[k]) for k in ["pT", "phi", "y", "eta", "eta_w"]}
    d = {k: _f64(df[k]) for k in ["T", "c0", "c2", "F", "betabulk", "betapi"]}
    keep.update(sp=sp, g=g, d=d)
    full = None
    if "2d" in df:   # full (mu_B, T) grids: needed (and used) only with include_baryon = 1
        full = {k: _f64(df["2d"][k]) for k in DF_NAMES_2D}
        full["muB"] = _f64(df["muB"])
        keep["full"] = full
    sps = Species(len(sp["mass"]), _p(sp["mass"]), _p(sp["sign"]), _p(sp["degeneracy"]), _p(sp["baryon"]))
    gs = Grid(len(g["pT"]), _p(g["pT"]), len(g["phi"]), _p(g["phi"]), len(g["y"]), _p(g["y"]), len(g["eta"]),
              _p(g["eta"]), _p(g["eta_w"]))
    ds = DfTables()
    ds.n_T, ds.T = len(d["T"]), _p(d["T"])
    if full is not None:
        ds.n_muB, ds.muB = len(full["muB"]), _p(full["muB"])
        for k in DF_NAMES_2D:
            assert full[k].shape == (ds.n_muB, ds.n_T), k
            setattr(ds, k, _p(full[k]))
    else:
        ds.n_muB = 1
        for k in ["c0", "c2", "F", "betabulk", "betapi"]:
            setattr(ds, k, _p(d[k]))
    os_ = Options()
    for k, v in o.items():
        setattr(os_, k, int(v))
    ny_eff = 1 if o["dimension"] == 2 else len(g["y"])
    nout = len(sp["mass"]) * len(g["pT"]) * len(g["phi"]) * ny_eff
    return sps, gs, ds, os_, nout, keep


VAH_FIELDS = ["tau", "eta", "ux", "uy", "un", "dat", "dax", "day", "dan", "T", "pitt", "pitx", "pity", "pitn", "pixx", "pixy", "pixn",
              "piyy", "piyn", "pinn", "bulkPi", "Wx", "Wy", "Lambda", "aL", "c0", "c1", "c2", "c3", "c4"]


class VahCells(C.Structure):
    _fields_ = [("n_cells", C.c_int64)] + [(f, _dp) for f in VAH_FIELDS]


class VahDfTables(C.Structure):
    _fields_ = [("n_L", C.c_int32), ("n_aL", C.c_int32), ("L", _dp), ("aL", _dp), ("c0", _dp), ("c1", _dp), ("c2", _dp), ("c3", _dp), ("c4", _dp)]


def _pack_vah_tables(tab, keep):
    """is3d_vah_df_tables from the dict of is3d_amd.inputs.vah_df_tables() / vah_df_read()."""
    a = {k: _f64(tab[k]) for k in ("L", "aL", "c0", "c1", "c2", "c3", "c4")}
    for k in ("c0", "c1", "c2", "c3", "c4"):
        assert a[k].shape == (len(a["aL"]), len(a["L"])), k
    keep["vah_tab"] = a
    return VahDfTables(len(a["L"]), len(a["aL"]), _p(a["L"]), _p(a["aL"]), _p(a["c0"]), _p(a["c1"]), _p(a["c2"]), _p(a["c3"]), _p(a["c4"]))


_VAH_DUMMY_DF = dict(T=[0.1, 0.15, 0.2], c0=[0, 0, 0], c2=[0, 0, 0], F=[0, 0, 0], betabulk=[1, 1, 1], betapi=[1, 1, 1])


def _vah_cells_struct(cells, held, device=False):
    cs = VahCells()
    if device:
        cs.n_cells = int(cells["n_cells"])
        for f in VAH_FIELDS:
            p = cells.get(f)
            if p:
                setattr(cs, f, C.cast(C.c_void_p(int(p)), _dp))
        return cs
    n = len(cells["tau"])
    cs.n_cells = n
    for f in VAH_FIELDS:
        a = cells.get(f)
        if a is not None:
            a = _f64(a)
            assert a.shape == (n,), f
            held.append(a)
            setattr(cs, f, _p(a))
    return cs


def smooth_spectra_vah(cells, species, grid, opts=None, out=None, tab=None):
    """is3d_smooth_spectra_vah (the drop-in for calculate_dN_pTdpTdphidy_VAH_PL).  cells: dict of host arrays per VAH_FIELDS.
    tab (dict L, aL, c0..c4): is3d_smooth_spectra_vah_df -- the cells' c0..c4 are ignored, the coefficients come from the
    (Lambda, alpha_L) tables (src/cuda/deltafReader.cu:224-278)."""
    L = load()
    sps, gs, _, os_, nout, keep = _pack_common(species, grid, _VAH_DUMMY_DF, opts)
    held = []
    if tab is not None:
        cells = {k: v for k, v in cells.items() if k not in ("c0", "c1", "c2", "c3", "c4")}
    cs = _vah_cells_struct(cells, held)
    if out is None:
        out = np.zeros(nout)
    st = Status()
    L.is3d_smooth_spectra_vah.argtypes = [C.POINTER(VahCells), C.POINTER(Species), C.POINTER(Grid), C.POINTER(Options), _dp, C.POINTER(Status)]
    L.is3d_smooth_spectra_vah_df.argtypes = [C.POINTER(VahCells), C.POINTER(Species), C.POINTER(Grid), C.POINTER(VahDfTables), C.POINTER(Options), _dp,
                                             C.POINTER(Status)]
    if tab is not None:
        ts = _pack_vah_tables(tab, keep)
        rc = L.is3d_smooth_spectra_vah_df(C.byref(cs), C.byref(sps), C.byref(gs), C.byref(ts), C.byref(os_), _p(out), C.byref(st))
    else:
        rc = L.is3d_smooth_spectra_vah(C.byref(cs), C.byref(sps), C.byref(gs), C.byref(os_), _p(out), C.byref(st))
    if rc != 0:
        raise Is3dError(rc, L.is3d_last_error().decode(), bad_cell=st.bad_cell)
    return out, st.as_dict()


def vah_coefficients(tab, Lambda, aL, device=-1):
    """is3d_vah_coefficients: per-cell c0..c4 (divided by hbarc^3) from the tables, evaluated on the device.  Lambda in GeV.
    Raises Is3dError(IS3D_EDOMAIN) with .bad_cell and .values (zeros at the offending cells) for a cell beyond the last node."""
    L = load()
    keep = {}
    ts = _pack_vah_tables(tab, keep)
    lam, al = _f64(Lambda), _f64(aL)
    n = len(lam)
    out = [np.zeros(n) for _ in range(5)]
    bad = C.c_int64(-1)
    L.is3d_vah_coefficients.argtypes = [C.POINTER(VahDfTables), C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int32, C.POINTER(C.c_int64)]
    rc = L.is3d_vah_coefficients(C.byref(ts), n, _p(lam), _p(al), *[_p(x) for x in out], int(device), C.byref(bad))
    vals = {"c%d" % k: out[k] for k in range(5)}
    if rc != 0:
        e = Is3dError(rc, L.is3d_last_error().decode(), bad_cell=bad.value)
        e.values = vals
        raise e
    return vals


def vah_df_read(directory):
    """is3d_vah_df_read: <dir>/c{0..4}_vah1.dat -> dict L, aL, c0..c4 ([n_aL][n_L])."""
    L = load()
    nL, naL = C.c_int32(), C.c_int32()
    L.is3d_vah_df_read.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp, C.c_int64]
    _check(L.is3d_vah_df_read(directory.encode(), C.byref(nL), C.byref(naL), None, None, None, 0))
    Lg, ag, c = np.zeros(nL.value), np.zeros(naL.value), np.zeros((5, naL.value, nL.value))
    _check(L.is3d_vah_df_read(directory.encode(), C.byref(nL), C.byref(naL), _p(Lg), _p(ag), _p(c), c.size))
    d = dict(L=Lg, aL=ag)
    for k in range(5):
        d["c%d" % k] = c[k].copy()
    return d


VAH_SURFACE_ORDER = VAH_FIELDS[:25] + ["E", "P", "PL", "Wt", "Wn", "x", "y"]


def surface_read_vah(path, dimension=3):
    """is3d_surface_read_vah (mode 2, read_surf_VAH_PLMatch) -> dict of the 32 arrays of VAH_SURFACE_ORDER."""
    L = load()
    n = C.c_int64(0)
    L.is3d_surface_read_vah.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(_dp)]
    _check(L.is3d_surface_read_vah(path.encode(), int(dimension), C.byref(n), None))
    arrs = {f: np.zeros(n.value) for f in VAH_SURFACE_ORDER}
    ptrs = (_dp * 32)(*[_p(arrs[f]) for f in VAH_SURFACE_ORDER])
    if n.value > 0:
        _check(L.is3d_surface_read_vah(path.encode(), int(dimension), C.byref(n), ptrs))
    return arrs


class VahPlan:
    """Device-resident VAH plan (is3d_vah_plan_*): cell arrays and the output are device pointers (ints); with `tab` the
    coefficients are interpolated on the device from (Lambda, aL) and c0..c4 need not be given."""

    def __init__(self, species, grid, opts=None, tab=None, max_cells=1):
        L = load()
        sps, gs, _, os_, self.output_size, self._keep = _pack_common(species, grid, _VAH_DUMMY_DF, opts)
        ts = _pack_vah_tables(tab, self._keep) if tab is not None else None
        L.is3d_vah_plan_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Species), C.POINTER(Grid), C.POINTER(VahDfTables), C.POINTER(Options), C.c_int64]
        L.is3d_vah_plan_execute.argtypes = [C.c_void_p, C.POINTER(VahCells), C.c_void_p, C.c_void_p, C.POINTER(Status)]
        L.is3d_vah_plan_timings.argtypes = [C.c_void_p, C.POINTER(Status)]
        L.is3d_vah_plan_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.is3d_vah_plan_tile_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.is3d_vah_plan_output_size.argtypes = [C.c_void_p]
        L.is3d_vah_plan_output_size.restype = C.c_int64
        L.is3d_vah_plan_workspace_bytes.argtypes = [C.c_void_p]
        L.is3d_vah_plan_workspace_bytes.restype = C.c_int64
        L.is3d_vah_plan_destroy.argtypes = [C.c_void_p]
        L.is3d_vah_plan_destroy.restype = None
        self._h = C.c_void_p()
        _check(L.is3d_vah_plan_create(C.byref(self._h), C.byref(sps), C.byref(gs), C.byref(ts) if ts is not None else None, C.byref(os_), int(max_cells)))
        assert L.is3d_vah_plan_output_size(self._h) == self.output_size
        self.workspace_bytes = L.is3d_vah_plan_workspace_bytes(self._h)
        jt, r = C.c_int32(), C.c_int32()
        L.is3d_vah_plan_tile_shape(self._h, C.byref(jt), C.byref(r))
        self.tile_shape = (jt.value, r.value)
        L.is3d_vah_plan_main_kernel_name.argtypes = [C.c_void_p]
        L.is3d_vah_plan_main_kernel_name.restype = C.c_char_p
        self.main_kernel_name = L.is3d_vah_plan_main_kernel_name(self._h).decode()

    def set_timing(self, enable=True):
        _check(load().is3d_vah_plan_set_timing(self._h, 1 if enable else 0))

    def execute(self, n_cells, cell_ptrs, out_ptr, stream=0, want_status=True):
        cs = _vah_cells_struct(dict(cell_ptrs, n_cells=n_cells), None, device=True)
        st = Status()
        rc = load().is3d_vah_plan_execute(self._h, C.byref(cs), C.c_void_p(int(out_ptr)), C.c_void_p(int(stream or 0)), C.byref(st) if want_status else None)
        if rc != 0:
            raise Is3dError(rc, load().is3d_last_error().decode(), bad_cell=st.bad_cell)
        return st.as_dict() if want_status else None

    def timings(self):
        st = Status()
        _check(load().is3d_vah_plan_timings(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            load().is3d_vah_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _pack_feqmod(fq, keep):
    """is3d_feqmod_tables from the dict of is3d_amd.inputs.feqmod_tables()."""
    a = {k: _f64(fq[k]) for k in ["root1", "weight1", "root2", "weight2", "pdg_mass", "pdg_degeneracy", "pdg_sign"]}
    keep["fq"] = a
    return FeqmodTables(len(a["root1"]), _p(a["root1"]), _p(a["weight1"]), _p(a["root2"]), _p(a["weight2"]), len(a["pdg_mass"]),
                        _p(a["pdg_mass"]), _p(a["pdg_degeneracy"]), _p(a["pdg_sign"]), float(fq["T_avg"]), float(fq["deta_min"]),
                        float(fq["mass_pion0"]))


def smooth_spectra(cells, species, grid, df, opts=None, out=None, fq=None):
    """Host-pointer entry is3d_smooth_spectra (the drop-in for calculate_dN_pTdpTdphidy); with fq (df_mode 3, 4)
    is3d_smooth_spectra_feqmod (the drop-in for calculate_dN_ptdptdphidy_feqmod).
    cells: dict of numpy arrays (host).  Returns (dN flat numpy array, status dict)."""
    L = load()
    sps, gs, ds, os_, nout, keep = _pack_common(species, grid, df, opts)
    n = len(cells["tau"])
    cs = Cells()
    cs.n_cells = n
    held = []
    for f in CELL_FIELDS:
        a = cells.get(f)
        if a is not None:
            a = _f64(a)
            assert a.shape == (n,), f
            held.append(a)
            setattr(cs, f, a.ctypes.data)
    if out is None:
        out = np.zeros(nout)
    assert out.dtype == np.float64 and out.size == nout and out.flags.c_contiguous
    st = Status()
    if fq is not None:
        fqs = _pack_feqmod(fq, keep)
        rc = L.is3d_smooth_spectra_feqmod(C.byref(cs), C.byref(sps), C.byref(gs), C.byref(ds), C.byref(fqs), C.byref(os_), _p(out), C.byref(st))
    else:
        rc = L.is3d_smooth_spectra(C.byref(cs), C.byref(sps), C.byref(gs), C.byref(ds), C.byref(os_), _p(out), C.byref(st))
    _check(rc)
    return out, st.as_dict()


def shard_bounds(n_cells, rank, n_ranks):
    """is3d_shard_bounds: the contiguous cell shard [lo, hi) of `rank` (what is3d_smooth_spectra_multi uses)."""
    lo, hi = C.c_int64(), C.c_int64()
    _check(load().is3d_shard_bounds(int(n_cells), int(rank), int(n_ranks), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def smooth_spectra_multi(cells, species, grid, df, opts=None, devices=None, reduce=REDUCE_ORDERED, out=None, fq=None):
    """is3d_smooth_spectra_multi: the host entry with the cells sharded over `devices` (list of HIP ordinals, one shard per
    entry, an ordinal may repeat; None = every visible device).  Returns (dN, aggregate status dict, [per-shard status dicts])."""
    L = load()
    sps, gs, ds, os_, nout, keep = _pack_common(species, grid, df, opts)
    n = len(cells["tau"])
    cs = Cells()
    cs.n_cells = n
    held = []
    for f in CELL_FIELDS:
        a = cells.get(f)
        if a is not None:
            a = _f64(a)
            assert a.shape == (n,), f
            held.append(a)
            setattr(cs, f, a.ctypes.data)
    if out is None:
        out = np.zeros(nout)
    assert out.dtype == np.float64 and out.size == nout and out.flags.c_contiguous
    nd = len(devices) if devices is not None else 0
    dv = (C.c_int32 * nd)(*[int(d) for d in devices]) if nd else None
    st = Status()
    sst = (Status * max(nd, L.is3d_device_count(), 1))()
    fqs = _pack_feqmod(fq, keep) if fq is not None else None
    rc = L.is3d_smooth_spectra_multi(C.byref(cs), C.byref(sps), C.byref(gs), C.byref(ds), C.byref(fqs) if fqs is not None else None,
                                     C.byref(os_), dv, nd, int(reduce), _p(out), C.byref(st), sst)
    _check(rc)
    return out, st.as_dict(), [sst[i].as_dict() for i in range(nd or L.is3d_device_count())]


class MultiPlan:
    """is3d_multi_plan_*: the persistent form of smooth_spectra_multi -- per-shard plans, workspaces, streams, pinned staging and the
    communicator set are created once; execute(cells) only uploads, runs and sums."""

    def __init__(self, species, grid, df, opts=None, devices=None, reduce=REDUCE_ORDERED, max_cells=1, fq=None):
        L = load()
        sps, gs, ds, os_, self.output_size, self._keep = _pack_common(species, grid, df, opts)
        nd = len(devices) if devices is not None else 0
        dv = (C.c_int32 * nd)(*[int(d) for d in devices]) if nd else None
        fqs = _pack_feqmod(fq, self._keep) if fq is not None else None
        self._h = C.c_void_p()
        _check(L.is3d_multi_plan_create(C.byref(self._h), C.byref(sps), C.byref(gs), C.byref(ds), C.byref(fqs) if fqs is not None else None,
                                        C.byref(os_), dv, nd, int(reduce), int(max_cells)))
        self.n_shards = L.is3d_multi_plan_shards(self._h)
        assert L.is3d_multi_plan_output_size(self._h) == self.output_size

    def execute(self, cells, out=None):
        """cells: dict of host numpy arrays.  Returns (dN, aggregate status dict, [per-shard status dicts])."""
        n = len(cells["tau"])
        cs = Cells()
        cs.n_cells = n
        held = []
        for f in CELL_FIELDS:
            a = cells.get(f)
            if a is not None:
                a = _f64(a)
                assert a.shape == (n,), f
                held.append(a)
                setattr(cs, f, a.ctypes.data)
        if out is None:
            out = np.zeros(self.output_size)
        assert out.dtype == np.float64 and out.size == self.output_size and out.flags.c_contiguous
        st = Status()
        sst = (Status * self.n_shards)()
        _check(load().is3d_multi_plan_execute(self._h, C.byref(cs), _p(out), C.byref(st), sst))
        return out, st.as_dict(), [sst[i].as_dict() for i in range(self.n_shards)]

    def close(self):
        if self._h:
            load().is3d_multi_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """is3d_comm: the library's RCCL communicator for one-process-per-GPU hosts.  Rank 0 makes the id (Comm.unique_id()),
    the host ships the 128 bytes to the other ranks, every rank constructs Comm(id, n_ranks, rank, device)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(load().is3d_comm_unique_id(buf))
        return buf.raw

    def __init__(self, uid, n_ranks, rank, device=-1):
        assert len(uid) == COMM_ID_BYTES
        self._h = C.c_void_p()
        _check(load().is3d_comm_create(C.byref(self._h), C.create_string_buffer(uid, COMM_ID_BYTES), int(n_ranks), int(rank), int(device)))
        self.n_ranks, self.rank = int(n_ranks), int(rank)

    def allreduce(self, dev_ptr, n, stream=0):
        _check(load().is3d_comm_allreduce(self._h, C.c_void_p(int(dev_ptr)), int(n), C.c_void_p(int(stream or 0))))

    def rank_seen(self):
        """(rank, n_ranks) as the library's communicator reports them (is3d_comm_rank)."""
        r, n = C.c_int32(-1), C.c_int32(-1)
        _check(load().is3d_comm_rank(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def check(self, stream=0):
        """is3d_comm_check: raises Is3dError(IS3D_EPEER) if a rank's execute had failed before one of the all-reduces since the
        last check (its n_failed attribute says how many); synchronises the stream."""
        k = C.c_int32(0)
        rc = load().is3d_comm_check(self._h, C.c_void_p(int(stream or 0)), C.byref(k))
        if rc != 0:
            e = Is3dError(rc, load().is3d_last_error().decode())
            e.n_failed = k.value
            raise e

    def abort(self):
        _check(load().is3d_comm_abort(self._h))

    def set_timeout(self, seconds):
        """Deadline of the library's host-side waits behind this communicator's collectives (is3d_comm_set_timeout)."""
        L = load()
        L.is3d_comm_set_timeout.argtypes = [C.c_void_p, C.c_double]
        _check(L.is3d_comm_set_timeout(self._h, float(seconds)))

    def synchronize(self, stream=0):
        """hipStreamSynchronize with that deadline (is3d_comm_synchronize): raises Is3dError(IS3D_ENODEVICE) instead of blocking for ever
        behind a collective a peer never joined."""
        _check(load().is3d_comm_synchronize(self._h, C.c_void_p(int(stream or 0))))

    def allreduce_ms(self):
        """Device time of the last collective on this rank (is3d_comm_timings)."""
        ms = C.c_double(0.0)
        _check(load().is3d_comm_timings(self._h, C.byref(ms)))
        return ms.value

    def close(self):
        if self._h:
            load().is3d_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MATH_FUNCS = dict(exp_full=0, exp_p9=1, exp_p9_sat=2, exp_full_sat=3, sqrt_g1=4, sqrt_nr=5, rcp_nr1=6, rcp_nr=7, exp_p9_x32=8)


def resource_counters():
    """is3d_resource_counters: (plans created, device allocations made) by the library in this process so far."""
    L = load()
    a, b = C.c_int64(0), C.c_int64(0)
    L.is3d_resource_counters.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    _check(L.is3d_resource_counters(C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def math_probe(which, x, device=-1):
    """is3d_math_probe: the device's elementary function `which` (a key of MATH_FUNCS) at the host array x."""
    xs = _f64(x)
    y = np.zeros_like(xs)
    L = load()
    L.is3d_math_probe.argtypes = [C.c_int32, C.c_int64, _dp, _dp, C.c_int32]
    _check(L.is3d_math_probe(MATH_FUNCS[which], xs.size, _p(xs), _p(y), int(device)))
    return y


def probe_shader_clock(seconds=0.3, device=0):
    """Shader clock in GHz averaged over `seconds`, sampled by idle waves on a private stream (0.0: the device's two counters
    tick at the same rate, nothing to measure).  Call it from a second thread while a kernel runs to price that kernel."""
    ghz = C.c_double(0.0)
    _check(load().is3d_probe_shader_clock(int(device), float(seconds), C.byref(ghz)))
    return ghz.value


class Plan:
    """Device-resident plan (is3d_plan_*).  Cell arrays and the output are device pointers (ints),
    e.g. torch tensors' data_ptr(); `stream` is a hipStream_t handle (torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, species, grid, df, opts=None, max_cells=1, fq=None):
        L = load()
        sps, gs, ds, os_, nout, keep = _pack_common(species, grid, df, opts)
        self._h = C.c_void_p()
        if fq is not None:
            fqs = _pack_feqmod(fq, keep)
            _check(L.is3d_plan_create_feqmod(C.byref(self._h), C.byref(sps), C.byref(gs), C.byref(ds), C.byref(fqs), C.byref(os_), int(max_cells)))
        else:
            _check(L.is3d_plan_create(C.byref(self._h), C.byref(sps), C.byref(gs), C.byref(ds), C.byref(os_), int(max_cells)))
        self.output_size = int(L.is3d_plan_output_size(self._h))
        assert self.output_size == nout
        self.workspace_bytes = int(L.is3d_plan_workspace_bytes(self._h))
        self.main_kernel_name = L.is3d_plan_main_kernel_name(self._h).decode()
        jt, r = C.c_int32(), C.c_int32()
        _check(L.is3d_plan_tile_shape(self._h, C.byref(jt), C.byref(r)))
        self.tile_shape = (jt.value, r.value)

    def set_timing(self, enable=True):
        _check(load().is3d_plan_set_timing(self._h, 1 if enable else 0))

    def execute(self, n_cells, cell_ptrs, out_ptr, stream=0, want_status=True):
        """cell_ptrs: dict field -> device pointer (int)."""
        cs = Cells()
        cs.n_cells = int(n_cells)
        for f in CELL_FIELDS:
            p = cell_ptrs.get(f)
            if p:
                setattr(cs, f, int(p))
        st = Status()
        rc = load().is3d_plan_execute(self._h, C.byref(cs), C.c_void_p(int(out_ptr)), C.c_void_p(int(stream or 0)),
                                      C.byref(st) if want_status else None)
        _check(rc)
        return st.as_dict() if want_status else None

    def execute_allreduce(self, n_cells, cell_ptrs, out_ptr, comm=None, stream=0, want_status=True):
        """is3d_plan_execute_allreduce: execute on this rank's shard, then the RCCL all-reduce of the spectrum over `comm`
        (a Comm, or None for a single rank) on the same stream."""
        cs = Cells()
        cs.n_cells = int(n_cells)
        for f in CELL_FIELDS:
            p = cell_ptrs.get(f)
            if p:
                setattr(cs, f, int(p))
        st = Status()
        rc = load().is3d_plan_execute_allreduce(self._h, C.byref(cs), C.c_void_p(int(out_ptr)), comm._h if comm is not None else None,
                                                C.c_void_p(int(stream or 0)), C.byref(st) if want_status else None)
        _check(rc)
        return st.as_dict() if want_status else None

    def observables(self, dN_ptr, pT_w, phi_w, dndy_ptr=0, spec2pi_ptr=0, vn_ptr=0, stream=0):
        """is3d_plan_observables: device pointers (ints) in and out, host weight arrays."""
        pw, fw = _f64(pT_w), _f64(phi_w)
        _check(load().is3d_plan_observables(self._h, C.c_void_p(int(dN_ptr)), _p(pw), _p(fw), C.c_void_p(int(dndy_ptr or 0)),
                                            C.c_void_p(int(spec2pi_ptr or 0)), C.c_void_p(int(vn_ptr or 0)), C.c_void_p(int(stream or 0))))

    def check(self, stream=0):
        """is3d_plan_check: raises Is3dError(IS3D_EDOMAIN) if an execute since the last check (status-less ones included) met a
        cell outside the coefficient table; synchronises the stream."""
        bad = C.c_int64(-1)
        rc = load().is3d_plan_check(self._h, C.c_void_p(int(stream or 0)), C.byref(bad))
        if rc != 0:
            raise Is3dError(rc, load().is3d_last_error().decode(), bad_cell=bad.value)

    def timings(self):
        st = Status()
        _check(load().is3d_plan_timings(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            load().is3d_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host I/O wrappers -------------------------------------------------------------------------
def param_get(path, name):
    v = C.c_double()
    _check(load().is3d_param_get(path.encode(), name.encode(), C.byref(v)))
    return v.value


def table_read(path):
    L = load()
    rows, cols = C.c_int64(), C.c_int32()
    _check(L.is3d_table_read(path.encode(), C.byref(rows), C.byref(cols), None, 0))
    data = np.zeros((rows.value, cols.value))
    _check(L.is3d_table_read(path.encode(), C.byref(rows), C.byref(cols), _p(data), data.size))
    return data


def surface_read_vh(path, include_baryon=0, include_baryondiff_deltaf=0, dimension=3):
    L = load()
    n = C.c_int64(0)
    _check(L.is3d_surface_read_vh(path.encode(), include_baryon, include_baryondiff_deltaf, dimension, C.byref(n), None, None))
    arrs = {f: np.zeros(n.value) for f in SURFACE_READ_ORDER}
    ptrs = (_dp * 23)(*[_p(arrs[f]) for f in SURFACE_READ_ORDER])
    avg = np.zeros(5)
    if n.value > 0:
        _check(L.is3d_surface_read_vh(path.encode(), include_baryon, include_baryondiff_deltaf, dimension, C.byref(n), ptrs, _p(avg)))
    return arrs, avg


def surface_read(path, mode, include_baryon=0, include_baryondiff_deltaf=0, dimension=3):
    """is3d_surface_read: modes 0, 1, 4, 5, 6, 7 -> (dict of the 23 arrays, averages)."""
    L = load()
    n = C.c_int64(0)
    _check(L.is3d_surface_read(path.encode(), mode, include_baryon, include_baryondiff_deltaf, dimension, C.byref(n), None, None))
    arrs = {f: np.zeros(n.value) for f in SURFACE_READ_ORDER}
    ptrs = (_dp * 23)(*[_p(arrs[f]) for f in SURFACE_READ_ORDER])
    avg = np.zeros(5)
    if n.value > 0:
        _check(L.is3d_surface_read(path.encode(), mode, include_baryon, include_baryondiff_deltaf, dimension, C.byref(n), ptrs, _p(avg)))
    return arrs, avg


def surface_open(path, mode=1, include_baryon=0, include_baryondiff_deltaf=0, dimension=3, cache=1):
    """is3d_surface_open / _arrays / _source / _close: one read and one parse of the text, or the binary sidecar `<path>.is3dcache` when it
    matches -> (dict of arrays [copies], averages or None, source) with source 0 text parsed | 1 text parsed + sidecar written | 2 sidecar.
    Modes 0, 1, 4, 5, 6, 7: the 23 arrays of SURFACE_READ_ORDER (absent ones None) + x, y; mode 2: the 32 arrays of VAH_SURFACE_ORDER."""
    L = load()
    L.is3d_surface_open.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.is3d_surface_cells.argtypes = [C.c_void_p]
    L.is3d_surface_cells.restype = C.c_int64
    L.is3d_surface_source.argtypes = [C.c_void_p]
    L.is3d_surface_arrays.argtypes = [C.c_void_p, C.POINTER(_dp), C.c_int32, _dp]
    L.is3d_surface_close.argtypes = [C.c_void_p]
    L.is3d_surface_close.restype = None
    h = C.c_void_p()
    _check(L.is3d_surface_open(path.encode(), int(mode), int(include_baryon), int(include_baryondiff_deltaf), int(dimension), int(cache), C.byref(h)))
    try:
        n = L.is3d_surface_cells(h)
        names = list(VAH_SURFACE_ORDER) if mode == 2 else list(SURFACE_READ_ORDER) + ["x", "y"]
        ptrs = (_dp * len(names))()
        avg = np.zeros(5)
        _check(L.is3d_surface_arrays(h, ptrs, len(names), _p(avg)))
        if n > 0:
            arrs = {f: (np.ctypeslib.as_array(ptrs[i], shape=(n,)).copy() if ptrs[i] else None) for i, f in enumerate(names)}
        else:
            arrs = {f: np.zeros(0) for f in names}
        L.is3d_surface_from_sidecar.argtypes = [C.c_void_p]
        early = L.is3d_surface_from_sidecar(h)                 # known at open, never waits for the writer
        source = L.is3d_surface_source(h)
        assert early == (1 if source == 2 else 0)
    finally:
        L.is3d_surface_close(h)
    return arrs, (None if mode == 2 else avg), source


def pdg_read(path, box=False):
    """is3d_pdg_read (hrg_eos 1, 2: the conventional token-stream files) | box = True: is3d_pdg_read_box (hrg_eos 3: PDG/pdg_box.dat)."""
    L = load()
    f = L.is3d_pdg_read_box if box else L.is3d_pdg_read
    n = C.c_int32(0)
    _check(f(path.encode(), C.byref(n), None, None, None, None, None, 0))
    ids = np.zeros(n.value, dtype=np.int64)
    mass, gspin, baryon, sign = (np.zeros(n.value) for _ in range(4))
    _check(f(path.encode(), C.byref(n), ids.ctypes.data_as(C.POINTER(C.c_int64)), _p(mass), _p(gspin), _p(baryon), _p(sign), n.value))
    return dict(mc_id=ids, mass=mass, gspin=gspin, baryon=baryon, sign=sign)


def df_table_read(path):
    L = load()
    n = C.c_int32(0)
    _check(L.is3d_df_table_read(path.encode(), C.byref(n), None, None, 0))
    T, v = np.zeros(n.value), np.zeros(n.value)
    _check(L.is3d_df_table_read(path.encode(), C.byref(n), _p(T), _p(v), n.value))
    return T, v


def df_table_read_full(path):
    L = load()
    nT, nB = C.c_int32(0), C.c_int32(0)
    _check(L.is3d_df_table_read_full(path.encode(), C.byref(nT), C.byref(nB), None, None, None, 0))
    T, B, v = np.zeros(nT.value), np.zeros(nB.value), np.zeros((nB.value, nT.value))
    _check(L.is3d_df_table_read_full(path.encode(), C.byref(nT), C.byref(nB), _p(T), _p(B), _p(v), v.size))
    return T, B, v


def sample_particles(cells, species, df, gla, opts=None, n_events=1, seed=1, y_cut=0.5, first_cell=0, capacity=None, fq=None, fast=0,
                     T_avg=0.0, T_avg_switch=0.0, batch_events=0, muB_avg=0.0, devices=None):
    """is3d_sample_particles (the drop-in for sample_dN_pTdpTdphidy, df_mode 1-4).  cells: dict of host arrays (x, y optional);
    gla: dict with root1, weight1; fq: the feqmod tables (df_mode 3, 4; fast mode with df_mode 2).  Returns (numpy structured array of PARTICLE_DTYPE, stats dict); capacity = None sizes the
    buffer from a count-only first call."""
    L = load()
    grid_dummy = dict(pT=[1.0], phi=[0.0], y=[0.0], eta=[0.0], eta_w=[1.0])
    sps, _, ds, os_, _, keep = _pack_common(species, grid_dummy, df, opts)
    n = len(cells["tau"])
    cs = Cells()
    cs.n_cells = n
    held = []
    for f in CELL_FIELDS:
        a = cells.get(f)
        if a is not None:
            a = _f64(a)
            assert a.shape == (n,), f
            held.append(a)
            setattr(cs, f, a.ctypes.data)
    r1, w1 = _f64(gla["root1"]), _f64(gla["weight1"])
    xs = _f64(cells["x"]) if cells.get("x") is not None else None
    ys = _f64(cells["y"]) if cells.get("y") is not None else None
    fqs = _pack_feqmod(fq, keep) if fq is not None else None
    si = SamplerInputs(int(n_events), len(r1), int(seed), float(y_cut), int(first_cell), _p(xs) if xs is not None else None,
                       _p(ys) if ys is not None else None, _p(r1), _p(w1), C.pointer(fqs) if fqs is not None else None, int(fast), int(batch_events),
                       float(T_avg), float(T_avg_switch), float(muB_avg))
    st = SamplerStats()
    cnt = C.c_int64(0)
    if devices is not None:   # is3d_sample_particles_multi: one cell shard per listed device (an ordinal may repeat)
        dv = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        L.is3d_sample_particles_multi.argtypes = [C.POINTER(Cells), C.POINTER(Species), C.POINTER(DfTables), C.POINTER(SamplerInputs),
                                                  C.POINTER(Options), C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_int64,
                                                  C.POINTER(C.c_int64), C.POINTER(SamplerStats)]

        def call(buf, cap):
            return L.is3d_sample_particles_multi(C.byref(cs), C.byref(sps), C.byref(ds), C.byref(si), C.byref(os_), dv, len(devices), buf, cap,
                                                 C.byref(cnt), C.byref(st))
    else:
        def call(buf, cap):
            return L.is3d_sample_particles(C.byref(cs), C.byref(sps), C.byref(ds), C.byref(si), C.byref(os_), buf, cap, C.byref(cnt), C.byref(st))
    if capacity is None:
        _check(call(None, 0))
        capacity = int(cnt.value)
    out = np.zeros(max(int(capacity), 1), dtype=PARTICLE_DTYPE)
    assert out.dtype.itemsize == C.sizeof(Particle)
    rc = call(out.ctypes.data, int(capacity))
    _check(rc)
    d = st.as_dict()
    d["n_particles"] = int(cnt.value)
    return out[:min(int(cnt.value), int(capacity))], d


class SamplerPlan:
    """is3d_sampler_plan_*: the device-resident, persistent form of is3d_sample_particles.  Tables, species classes and (after the first
    execute of a shape) the workspaces live on the device; execute takes DEVICE pointers for the cell arrays (dict field -> int), an optional
    DEVICE particle buffer (PARTICLE_DTYPE entries) and returns (n_particles, stats).  The list is the one is3d_sample_particles gives."""

    def __init__(self, species, df, gla, opts=None, max_cells=1, fq=None, fast=0, T_avg=0.0, T_avg_switch=0.0, muB_avg=0.0, y_cut=0.5):
        L = load()
        grid_dummy = dict(pT=[1.0], phi=[0.0], y=[0.0], eta=[0.0], eta_w=[1.0])
        sps, _, ds, os_, _, self._keep = _pack_common(species, grid_dummy, df, opts)
        r1, w1 = _f64(gla["root1"]), _f64(gla["weight1"])
        fqs = _pack_feqmod(fq, self._keep) if fq is not None else None
        si = SamplerInputs(1, len(r1), 0, float(y_cut), 0, None, None, _p(r1), _p(w1), C.pointer(fqs) if fqs is not None else None, int(fast), 0,
                           float(T_avg), float(T_avg_switch), float(muB_avg))
        L.is3d_sampler_plan_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Species), C.POINTER(DfTables), C.POINTER(SamplerInputs), C.POINTER(Options), C.c_int64]
        L.is3d_sampler_plan_execute.argtypes = [C.c_void_p, C.POINTER(Cells), C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_int64, C.c_int32, C.c_void_p,
                                                C.c_int64, C.POINTER(C.c_int64), C.POINTER(SamplerStats)]
        L.is3d_sampler_plan_destroy.argtypes = [C.c_void_p]
        L.is3d_sampler_plan_destroy.restype = None
        self._h = C.c_void_p()
        _check(L.is3d_sampler_plan_create(C.byref(self._h), C.byref(sps), C.byref(ds), C.byref(si), C.byref(os_), int(max_cells)))

    def execute(self, n_cells, dev_ptrs, n_events, seed, particles_ptr=0, capacity=0, x_ptr=0, y_ptr=0, first_cell=0, batch_events=0):
        cs = Cells()
        cs.n_cells = int(n_cells)
        for f in CELL_FIELDS:
            if dev_ptrs.get(f):
                setattr(cs, f, int(dev_ptrs[f]))
        cnt, st = C.c_int64(0), SamplerStats()
        rc = load().is3d_sampler_plan_execute(self._h, C.byref(cs), C.c_void_p(int(x_ptr) or None), C.c_void_p(int(y_ptr) or None), int(n_events), int(seed),
                                              int(first_cell), int(batch_events), C.c_void_p(int(particles_ptr) or None), int(capacity), C.byref(cnt), C.byref(st))
        _check(rc)
        d = st.as_dict()
        d["n_particles"] = int(cnt.value)
        return int(cnt.value), d

    def close(self):
        if self._h:
            load().is3d_sampler_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class YieldInputs(C.Structure):
    _fields_ = [(n, C.c_double) for n in ["T", "E", "P", "muB", "nB"]] + [("root3", _dp), ("weight3", _dp)]


def total_yield(cells, species, df, gla, avg5, opts=None, y_cut=0.5, fq=None):
    """is3d_total_yield (the drop-in for calculate_total_yield): mean particle yield of the surface.  gla: dict with root1, weight1,
    root2, weight2 (+ root3, weight3 for df_mode 1), e.g. is3d_amd.inputs.feqmod_tables(); avg5 = (T, E, P, muB, nB) surface
    averages; fq: the feqmod tables for df_mode 4 (defaults to gla).  Returns (yield, densities[3][n_species])."""
    L = load()
    grid_dummy = dict(pT=[1.0], phi=[0.0], y=[0.0], eta=[0.0], eta_w=[1.0])
    sps, _, ds, os_, _, keep = _pack_common(species, grid_dummy, df, opts)
    n = len(cells["tau"])
    cs = Cells()
    cs.n_cells = n
    held = []
    for f in CELL_FIELDS:
        a = cells.get(f)
        if a is not None:
            a = _f64(a)
            assert a.shape == (n,), f
            held.append(a)
            setattr(cs, f, a.ctypes.data)
    r1, w1 = _f64(gla["root1"]), _f64(gla["weight1"])
    fqs = _pack_feqmod(fq if fq is not None else dict(gla, T_avg=gla.get("T_avg", avg5[0]), deta_min=gla.get("deta_min", 1e-5),
                                                      mass_pion0=gla.get("mass_pion0", 0.138)), keep)
    si = SamplerInputs(1, len(r1), 0, float(y_cut), 0, None, None, _p(r1), _p(w1), C.pointer(fqs), 0, 0, 0.0, 0.0, 0.0)
    r3 = _f64(gla["root3"]) if "root3" in gla else None
    w3 = _f64(gla["weight3"]) if "weight3" in gla else None
    yi = YieldInputs(float(avg5[0]), float(avg5[1]), float(avg5[2]), float(avg5[3]), float(avg5[4]), _p(r3) if r3 is not None else None,
                     _p(w3) if w3 is not None else None)
    out = C.c_double(0.0)
    dens = np.zeros((3, sps.n))
    L.is3d_total_yield.argtypes = [C.POINTER(Cells), C.POINTER(Species), C.POINTER(DfTables), C.POINTER(SamplerInputs), C.POINTER(YieldInputs),
                                   C.POINTER(Options), C.POINTER(C.c_double), _dp]
    _check(L.is3d_total_yield(C.byref(cs), C.byref(sps), C.byref(ds), C.byref(si), C.byref(yi), C.byref(os_), C.byref(out), _p(dens)))
    return out.value, dens


def write_particle_list_osc(path, n_events, particles, mc_id):
    particles = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
    ids = np.ascontiguousarray(mc_id, dtype=np.int64)
    _check(load().is3d_write_particle_list_osc(path.encode(), int(n_events), len(particles), particles.ctypes.data,
                                               ids.ctypes.data_as(C.POINTER(C.c_int64))))


class SamplerTestBins(C.Structure):
    _fields_ = [(n, C.c_double) for n in ["y_cut", "eta_cut", "pT_lower_cut", "pT_upper_cut", "tau_min", "tau_max", "r_min", "r_max"]] + \
               [(n, C.c_int32) for n in ["y_bins", "eta_bins", "pT_bins", "tau_bins", "r_bins", "reserved"]]


def write_sampler_tests(results_dir, bins, n_events, mc_id, particles, mean_yield=0.0):
    """is3d_write_sampler_tests: the test_sampler = 1 binned outputs from a particle list."""
    particles = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
    ids = np.ascontiguousarray(mc_id, dtype=np.int64)
    b = SamplerTestBins()
    for k, v in bins.items():
        setattr(b, k, v)
    L = load()
    L.is3d_write_sampler_tests.argtypes = [C.c_char_p, C.POINTER(SamplerTestBins), C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int64,
                                           C.c_void_p, C.c_double]
    _check(L.is3d_write_sampler_tests(results_dir.encode(), C.byref(b), int(n_events), len(ids), ids.ctypes.data_as(C.POINTER(C.c_int64)),
                                      len(particles), particles.ctypes.data, float(mean_yield)))


def gla_read(path):
    """is3d_gla_read -> (root[n_alpha][n_points], weight[n_alpha][n_points])"""
    L = load()
    na, npts = C.c_int32(), C.c_int32()
    _check(L.is3d_gla_read(path.encode(), C.byref(na), C.byref(npts), None, None, 0))
    r, w = np.zeros((na.value, npts.value)), np.zeros((na.value, npts.value))
    _check(L.is3d_gla_read(path.encode(), C.byref(na), C.byref(npts), _p(r), _p(w), r.size))
    return r, w


def write_results(results_dir, dimension, mc_id, pT, pT_w, phi, phi_w, y, dN):
    L = load()
    mc = np.ascontiguousarray(mc_id, dtype=np.int64)
    pT, pT_w, phi, phi_w, y, dN = (_f64(a) for a in (pT, pT_w, phi, phi_w, y, dN))
    _check(L.is3d_write_results(results_dir.encode(), dimension, len(mc), mc.ctypes.data_as(C.POINTER(C.c_int64)), len(pT),
                                _p(pT), _p(pT_w), len(phi), _p(phi), _p(phi_w), len(y), _p(y), _p(dN)))
