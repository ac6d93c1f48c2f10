"""GPU (-m gpu): the bench.py contract on small surfaces -- one JSON line on stdout with the contract's fields, the `roofline` and
`roofline_valu` objects, for every workload bench.py knows (BASELINE configs 3, 2 and the smooth leg of 5).  Sizes are tiny: this guards the
plumbing, the numbers come from the full-size runs under profiles/."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,kernel,extra", [("config3", "cf_main_tile3e", []), ("config2", "cf_main_tile", []), ("config5", "cf_main_vah3", []),
                                                   ("config3", "cf_main_feqmod", ["--df-mode", "4"]),
                                                   ("config3", "cf_main_tile3e", ["--include-baryon"]),      # SURVEY.md 8f rank 1 at the bench
                                                   ("config3", "cf_main_tile3e", ["--df-mode", "1"]),        # config 1's physics (14-moment) on config 3's surface
                                                   ("config2", "cf_main_tile", ["--include-baryon"])])
def test_bench_line(workload, kernel, extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--cells", "6000", "--steps", "2", "--warmup", "1",
                        "--cpu-baseline-seconds", "1", "--no-clock-probe"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "roofline_valu", "kernel_ms", "executed_evals_per_s"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 0
    assert d["config"]["kernel"] == kernel and d["config"]["cells_total"] == 6000 and d["config"]["spectrum_finite"]
    assert d["config"]["culled_rows_change_no_bit"] is True
    ro, rv = d["roofline"], d["roofline_valu"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert rv is not None and rv["bound"] == "fp64_valu" and 20 <= rv["executed_flop_per_eval"] <= 70 and 0 < rv["frac"] < 1
    assert d["kernel_ms"]["main"] > 0 and d["kernel_ms"]["prep"] > 0
    # executed integrands against reference-equivalent integrands, like with like (2+1D: both count the eta nodes)
    assert 0 < d["executed_fraction_of_value"] <= 1, d["executed_fraction_of_value"]
    assert d["transfers_included"] is False and 0 < d["value_incl_transfers"] <= d["value"] * 1.05 and d["ms_per_step_incl_transfers"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and 1 <= cb["cores"] <= cb["cores_available"] and cb["cpu_model"] and cb["unit"] == "evals/s"
    keys = list(d)
    assert keys.index("gpu_over_cpu_executed") < keys.index("gpu_over_cpu")       # the like-for-like ratio leads
    assert d["config"]["workspace_bytes_per_cell"] > 0
    if "--include-baryon" in extra:
        # baryon number joins the class key (p / pbar, Lambda / Lambdabar ... are different classes), five more cell arrays in the algorithmic bytes
        assert d["config"]["species_classes_evaluated"] == (124 if workload == "config3" else 3)
        n_arr = 23 if workload == "config3" else 22
        assert ro["algorithmic_bytes"] == 8.0 * (n_arr * 6000 + d["config"]["species"] * d["config"]["bins"])
        assert "include_baryon = 1" in d["config"]["workload"]
    if extra == ["--df-mode", "1"]:
        assert "df_mode overridden to 1" in d["config"]["workload"] and rv["executed_flop_per_eval"] < 22


def test_bench_line_sampler_leg():
    """bench.py --workload config5-sampler: the particle-sampler leg of BASELINE config 5 through the device-resident is3d_sampler_plan -- the
    contract's fields, per-kernel device times, the HBM-form roofline, the fp64-VALU roofline of the density kernel, a serial CPU baseline on a
    slice whose list is the device's, and a second execute of the same shape that allocates nothing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config5-sampler", "--cells", "20000", "--events", "5", "--steps", "2",
                        "--warmup", "1", "--cpu-baseline-seconds", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "roofline_valu", "kernel_ms", "cpu_baseline", "particles_per_s"):
        assert key in d, key
    assert d["unit"] == "cell-events/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["particles_per_step"] > 1000
    assert d["config"]["cells_total"] == 20000 and d["config"]["events"] == 5 and d["config"]["species_classes_evaluated"] == 75
    assert 0.3 < d["momentum_sampling_efficiency"] < 1.0
    km = d["kernel_ms"]
    assert all(km[k] > 0 for k in ("prep", "density", "count", "poisson", "fill")) and km["density"] < km["prep"] and km["poisson"] < km["count"]
    assert d["device_allocations_during_timed_steps"] == 0            # the plan's workspaces were sized by the first (count-only) execute
    assert d["same_list_as_host_entry"] is True and d["cpu_baseline"]["same_list_on_the_slice"] is True and d["cpu_baseline"]["cores"] == 1
    ro, rv = d["roofline"], d["roofline_valu"]
    assert ro["bound"] == "hbm" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12 and ro["algorithmic_bytes"] == 8.0 * 18 * 20000 + 96.0 * d["particles_per_step"]
    assert rv["bound"] == "fp64_valu" and rv["kernel"] == "cf_sampler_density" and 0 < rv["frac"] < 1
