"""GPU (-m gpu): the multi-device entries of the C ABI on a one-GPU box -- the same device listed several times gives
several cell-axis shards (own plan, own stream, own host thread each) whose spectra are summed in shard order on the
device (IS3D_REDUCE_ORDERED), which is the code path an 8-GPU node runs with eight different ordinals.  The RCCL
calls (ncclCommInitRank / ncclCommInitAll / ncclAllReduce) are executed with one rank.
Reference semantics: the spectrum is a plain sum over cells (smooth_kernels.cpp:363-375)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import relerr
from is3d_amd import api, inputs, synth
from oracle import oracle  # the checker

pytestmark = pytest.mark.gpu
TOL = 2e-9


def test_one_device_is_the_single_gpu_entry(fx):
    cells = synth.synth_surface(300, 3, seed=31)
    o = dict(dimension=3, df_mode=2)
    one, st1 = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o)
    got, st, sh = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0])
    assert np.array_equal(got, one) and len(sh) == 1 and st["n_classes"] == st1["n_classes"]
    # devices=None: every visible device (one here)
    got2, _, _ = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o)
    if api.load().is3d_device_count() == 1:
        assert np.array_equal(got2, one)
    else:
        assert relerr(got2, one) < 1e-13


@pytest.mark.parametrize("dim,df_mode,n,shards", [(3, 2, 1000, 2), (3, 1, 333, 3), (2, 1, 41, 2), (3, 2, 2, 3), (3, 2, 0, 2)])
def test_same_device_twice_two_shards(fx, dim, df_mode, n, shards):
    """n_devices = 2 with the same ordinal: two shards on one GPU, ordered device sum == the single-shard spectrum to 1e-13
    (rounding of the different summation order only), and == the oracle."""
    cells = synth.synth_surface(n, dim, seed=70 + n)
    sp = fx["pikp"] if dim == 2 else inputs.species([211, 321, 2212, -2212, 3122, 333, 22])
    o = dict(dimension=dim, df_mode=df_mode)
    one, _ = api.smooth_spectra(cells, sp, fx["grid"], fx["df"], o)
    got, st, sh = api.smooth_spectra_multi(cells, sp, fx["grid"], fx["df"], o, devices=[0] * shards)
    assert relerr(got, one) < 1e-13
    assert len(sh) == shards and st["code"] == 0
    if n:
        ref = oracle.dN_pTdpTdphidy(cells, sp, fx["grid"], fx["df"], o)
        assert relerr(got, ref) < TOL
    else:
        assert not got.any()
    # bitwise reproducible for a given shard count
    again, _, _ = api.smooth_spectra_multi(cells, sp, fx["grid"], fx["df"], o, devices=[0] * shards)
    assert np.array_equal(again, got)
    # shard bounds are the library's
    sizes = [api.shard_bounds(n, r, shards) for r in range(shards)]
    assert sizes[0][0] == 0 and sizes[-1][1] == n and all(sizes[i][1] == sizes[i + 1][0] for i in range(shards - 1))


def test_multi_feqmod_and_accumulate(fx):
    cells = synth.synth_surface(500, 3, seed=5)
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
    o = dict(dimension=3, df_mode=4)
    one, st1 = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o, fq=fq)
    got, st, _ = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0, 0], fq=fq)
    assert relerr(got, one) < 1e-13 and st["n_cells_narrow"] == st1["n_cells_narrow"]
    # accumulate = 1: dN_out += result (smooth_kernels.cpp:375)
    acc = one.copy()
    api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], dict(o, accumulate=1), devices=[0, 0], out=acc, fq=fq)
    assert relerr(acc, 2.0 * one) < 1e-13


def test_multi_status_is_aggregated_and_domain_errors_carry_the_global_index(fx):
    cells = synth.synth_surface(200, 3, seed=9)
    cells["dat"][[3, 150, 151]] *= -1.0         # u.dsigma <= 0: skipped cells in both shards
    cells["dax"][[3, 150, 151]] = 0.0
    cells["day"][[3, 150, 151]] = 0.0
    cells["dan"][[3, 150, 151]] = 0.0
    o = dict(dimension=3, df_mode=1)
    one, st1 = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o)
    got, st, sh = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0, 0])
    assert st1["n_cells_skipped"] == 3 and st["n_cells_skipped"] == 3
    assert [s["n_cells_skipped"] for s in sh] == [1, 2]
    assert relerr(got, one) < 1e-13
    cells["T"][170] = 0.25                      # outside the coefficient table: the reference aborts in gsl_spline_eval
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0, 0])
    assert e.value.code == api.IS3D_EDOMAIN and "shard 1" in str(e.value) and "cell 70" in str(e.value)


def test_rccl_calls_execute_with_one_rank(fx, tmp_path):
    """The library's RCCL binding on a one-GPU box: ncclGetUniqueId, ncclCommInitRank (1 rank), ncclAllReduce in place
    through is3d_plan_execute_allreduce; and the in-process form (ncclCommInitAll over [0]) through
    is3d_smooth_spectra_multi(reduce = RCCL).  Two shards on one device are refused for RCCL (one rank per GPU)."""
    import torch
    dev = torch.device("cuda:0")
    cells = synth.synth_surface(400, 3, seed=77)
    o = dict(dimension=3, df_mode=2)
    one, _ = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o)
    uid = api.Comm.unique_id()
    assert len(uid) == api.COMM_ID_BYTES and any(uid)
    comm = api.Comm(uid, 1, 0, 0)
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    plan = api.Plan(fx["pikp"], fx["grid"], fx["df"], o, max_cells=400)
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    st = plan.execute_allreduce(400, {k: v.data_ptr() for k, v in tens.items()}, out.data_ptr(), comm, stream)
    torch.cuda.synchronize()
    assert st["code"] == 0 and np.array_equal(out.cpu().numpy(), one)
    # comm = None: plain execute
    out.zero_()
    plan.execute_allreduce(400, {k: v.data_ptr() for k, v in tens.items()}, out.data_ptr(), None, stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), one)
    plan.close()
    comm.close()
    got, _, _ = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0], reduce=api.REDUCE_RCCL)
    assert np.array_equal(got, one)
    with pytest.raises(api.Is3dError) as e:
        api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0, 0], reduce=api.REDUCE_RCCL)
    assert e.value.code == api.IS3D_EINVAL and "distinct" in str(e.value)


def test_sampler_over_several_shards_gives_the_single_device_list(fx):
    """is3d_sample_particles_multi: three cell shards on one GPU (own threads, count + fill each), merged per event: the same
    hadrons in the same order as one device gives (the streams are keyed by the global cell index)."""
    cells = synth.synth_surface(30000, 3, seed=61)
    sp = inputs.species([211, 321, 2212, -2212])
    gla = inputs.feqmod_tables(0.15)
    o = dict(dimension=3, df_mode=2)
    one, st1 = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=7, seed=99)
    three, st3 = api.sample_particles(cells, sp, fx["df"], gla, o, n_events=7, seed=99, devices=[0, 0, 0])
    assert len(one) == len(three) > 1000 and st3["n_particles"] == st1["n_particles"]
    for f in one.dtype.names:
        assert np.array_equal(one[f], three[f]), f
    assert st3["n_hadrons_drawn"] == st1["n_hadrons_drawn"] and st3["n_momentum_samples"] == st1["n_momentum_samples"]
    with pytest.raises(api.Is3dError) as e:                                   # buffer too small: full count reported, IS3D_ENOMEM
        api.sample_particles(cells, sp, fx["df"], gla, o, n_events=7, seed=99, devices=[0, 0], capacity=10)
    assert e.value.code == api.IS3D_ENOMEM


def test_persistent_multi_plan_is_bitwise_the_one_shot_and_cheaper_per_execute(fx):
    """is3d_multi_plan_*: plans, workspaces, streams created once.  Two shards on one GPU through the persistent plan ==
    is3d_smooth_spectra_multi bitwise, and -- the structural fact behind "cheaper per execute" -- a second execute creates no plan and makes
    no device allocation (is3d_resource_counters), whereas every one-shot call creates its two plans and allocates their workspaces anew.
    The wall-clock times of both are printed for the log, not asserted (a shared box's jitter is not a defect)."""
    import time
    N = 125000                                   # one shard of BASELINE config 4
    cells = synth.synth_surface(N, 3)
    o = dict(dimension=3, df_mode=2)
    api.smooth_spectra_multi(cells, fx["urqmd"], fx["grid"], fx["df"], o, devices=[0, 0])    # warm (contexts, code objects)
    t_one_shot = 1e9
    for _ in range(3):      # per call: two plans, two workspace hipMallocs, streams, events -- and the work
        p0, a0 = api.resource_counters()
        t0 = time.perf_counter()
        one_shot, _, _ = api.smooth_spectra_multi(cells, fx["urqmd"], fx["grid"], fx["df"], o, devices=[0, 0])
        t_one_shot = min(t_one_shot, time.perf_counter() - t0)
        p1, a1 = api.resource_counters()
        assert p1 - p0 == 2 and a1 - a0 >= 2 * 3, (p1 - p0, a1 - a0)   # two shard plans; per shard at least workspace + cells + spectrum
    mp = api.MultiPlan(fx["urqmd"], fx["grid"], fx["df"], o, devices=[0, 0], max_cells=N)
    first, st, sh = mp.execute(cells)
    assert np.array_equal(first, one_shot) and mp.n_shards == 2 and len(sh) == 2 and st["code"] == 0
    t_second = 1e9
    for _ in range(3):
        p0, a0 = api.resource_counters()
        t0 = time.perf_counter()
        second, st2, _ = mp.execute(cells)
        t_second = min(t_second, time.perf_counter() - t0)
        assert api.resource_counters() == (p0, a0)                   # nothing created, nothing allocated
        assert np.array_equal(second, one_shot)
    print("multi plan, 2 x 62 500 cells on one GPU (best of 3): one-shot entry %.1f ms, persistent execute %.1f ms (h2d %.1f, d2h + sum %.1f ms)" % (
        t_one_shot * 1e3, t_second * 1e3, st2["ms_h2d"], st2["ms_d2h"]))
    cells = {k: v[:20000] for k, v in cells.items()}
    # fewer cells than max_cells, an odd count, an empty surface: the same plan
    for n in (7777, 1, 0):
        sub = {k: v[:n] for k, v in cells.items()}
        got, _, _ = mp.execute(sub)
        ref, _, _ = api.smooth_spectra_multi(sub, fx["urqmd"], fx["grid"], fx["df"], o, devices=[0, 0])
        assert np.array_equal(got, ref), n
    with pytest.raises(api.Is3dError) as e:
        mp.execute(synth.synth_surface(N + 1, 3, seed=1))
    assert e.value.code == api.IS3D_EINVAL
    mp.close()


@pytest.mark.parametrize("shards", [3, 4, 5, 8])
def test_tree_sum_is_reproducible_and_matches_one_shard(fx, shards):
    """IS3D_REDUCE_ORDERED is a fixed pairwise tree: bitwise reproducible for a given shard count, equal to the single-shard spectrum up
    to the rounding of the different association, and equal to the host's own pairwise sum of the shard spectra in that order."""
    cells = synth.synth_surface(shards * 130 + 3, 3, seed=40 + shards)
    o = dict(dimension=3, df_mode=2)
    one, _ = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o)
    got, _, sh = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0] * shards)
    again, _, _ = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0] * shards)
    assert np.array_equal(got, again) and relerr(got, one) < 1e-13 and len(sh) == shards
    parts = []
    for r in range(shards):
        lo, hi = api.shard_bounds(len(cells["tau"]), r, shards)
        parts.append(api.smooth_spectra({k: v[lo:hi] for k, v in cells.items()}, fx["pikp"], fx["grid"], fx["df"], o)[0])
    stride = 1
    while stride < shards:
        for i in range(0, shards - stride, 2 * stride):
            parts[i] = parts[i] + parts[i + stride]
        stride *= 2
    assert np.array_equal(got, parts[0])


def test_allreduce_error_word_and_refusals(fx):
    """is3d_plan_execute_allreduce never leaves a peer waiting: a rank with an argument error still joins (zeros + error word), the
    error word reaches is3d_comm_check, accumulate = 1 is refused in front of a collective, and an aborted communicator refuses
    further use.  One rank here (a one-GPU box holds one RCCL rank); the first run with N > 1 RCCL ranks is the driver's SCALE run."""
    import torch
    dev = torch.device("cuda:0")
    cells = synth.synth_surface(300, 3, seed=3)
    o = dict(dimension=3, df_mode=2)
    one, _ = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o)
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream
    comm = api.Comm(api.Comm.unique_id(), 1, 0, 0)
    assert comm.rank_seen() == (0, 1)
    plan = api.Plan(fx["pikp"], fx["grid"], fx["df"], o, max_cells=300)
    out = torch.full((plan.output_size,), 7.0, dtype=torch.float64, device=dev)
    plan.execute_allreduce(300, ptrs, out.data_ptr(), comm, stream, want_status=False)
    comm.check(stream)                                   # clean
    assert np.array_equal(out.cpu().numpy(), one) and comm.allreduce_ms() >= 0.0
    # more cells than the plan holds: IS3D_EINVAL locally, the collective still happens with a neutral contribution
    out.fill_(7.0)
    with pytest.raises(api.Is3dError) as e:
        plan.execute_allreduce(301, ptrs, out.data_ptr(), comm, stream, want_status=False)
    assert e.value.code == api.IS3D_EINVAL
    torch.cuda.synchronize()
    assert not out.cpu().numpy().any()
    with pytest.raises(api.Is3dError) as e:
        comm.check(stream)
    assert e.value.code == api.IS3D_EPEER and e.value.n_failed == 1
    comm.check(stream)                                   # cleared
    # a domain error: the rank joins with what it could evaluate and returns its own code
    tens["T"][17] = 0.3
    with pytest.raises(api.Is3dError) as e:
        plan.execute_allreduce(300, ptrs, out.data_ptr(), comm, stream)
    assert e.value.code == api.IS3D_EDOMAIN and "cell 17" in str(e.value)
    with pytest.raises(api.Is3dError) as e:
        comm.check(stream)
    assert e.value.code == api.IS3D_EPEER
    tens["T"][17] = 0.15
    plan.close()
    # accumulate = 1 with a communicator
    plan_acc = api.Plan(fx["pikp"], fx["grid"], fx["df"], dict(o, accumulate=1), max_cells=300)
    with pytest.raises(api.Is3dError) as e:
        plan_acc.execute_allreduce(300, ptrs, out.data_ptr(), comm, stream)
    assert e.value.code == api.IS3D_EINVAL and "accumulate" in str(e.value)
    plan_acc.execute_allreduce(300, ptrs, out.data_ptr(), None, stream)   # fine without one
    plan_acc.close()
    try:
        comm.check(stream)
    except api.Is3dError:
        pass
    # abort: the communicator refuses further use
    plan = api.Plan(fx["pikp"], fx["grid"], fx["df"], o, max_cells=300)
    comm.abort()
    with pytest.raises(api.Is3dError) as e:
        plan.execute_allreduce(300, ptrs, out.data_ptr(), comm, stream)
    assert e.value.code == api.IS3D_ENODEVICE and "aborted" in str(e.value)
    plan.close()
    comm.close()


def _check_multi_rank_line(d):
    """What an N > 1 line must carry to be self-sufficient (north_star: the CPU path timed on the node's own cores in the same run, beside
    the GPU numbers): cpu_baseline with the cores and the CPU model, the contract's roofline, every rank's own fp64-VALU roofline, and the
    transfer-inclusive step -- present and sane; the magnitudes belong to the full-size runs under profiles/."""
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and 1 <= cb["cores"] <= cb["cores_available"] and cb["cpu_model"] and cb["kind"] == "port"
    assert d["roofline"]["frac"] > 0 and d["roofline"]["bound"] == "hbm" and d["roofline_valu"]["frac"] > 0
    for x in d["ranks"]:
        rv = x["roofline_valu"]
        assert rv["bound"] == "fp64_valu" and 0 < rv["frac"] < 1 and rv["kernel_ms"] == x["kernel_ms"]["main"]
        assert 0 <= rv["wave_rows_culled_frac"] < 1 and rv["integrands_executed"] == x["integrands_executed"]
    assert d["roofline_valu"]["shader_clock_ghz"] is None or 1.0 < d["roofline_valu"]["shader_clock_ghz"] < 3.0   # rank 0's probe runs for any N
    assert 0 < d["executed_fraction_of_value"] <= 1 and d["value_incl_transfers"] > 0 and d["ms_per_step_incl_transfers"] > 0


def test_waits_behind_a_collective_have_a_deadline(fx):
    """RCCL enqueues an all-reduce and returns; a rank whose peer never joins would block for ever in hipStreamSynchronize.  The library's
    own host-side waits (is3d_comm_check, is3d_comm_synchronize, is3d_comm_timings, the status read-back of is3d_plan_execute_allreduce) poll
    the stream against the communicator's deadline instead and abort THIS rank's communicator when it passes.  One real RCCL rank here, so
    nothing can actually hang: the deadline is set far below the time the enqueued kernels need (a 40 000-cell urqmd surface: tens of ms),
    which takes the same code path as a peer that never arrives."""
    import torch
    dev = torch.device("cuda:0")
    n = 40000
    cells = synth.synth_surface(n, 3, seed=8)
    o = dict(dimension=3, df_mode=2)
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in synth.CELL_FIELDS}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream
    plan = api.Plan(fx["urqmd"], fx["grid"], fx["df"], o, max_cells=n)
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    comm = api.Comm(api.Comm.unique_id(), 1, 0, 0)
    plan.execute_allreduce(n, ptrs, out.data_ptr(), comm, stream, want_status=False)
    comm.synchronize(stream)                                            # default deadline (300 s): finishes
    ref = out.cpu().numpy().copy()
    # round 5: with one collective in flight the clock starts when the stream REACHES it -- a timeout below the run time of this rank's own
    # kernels (~15 ms here) no longer poisons the communicator (round 4 aborted it after 2.5 ms of local compute) ...
    comm.set_timeout(2.5e-3)
    plan.execute_allreduce(n, ptrs, out.data_ptr(), comm, stream, want_status=False)
    comm.synchronize(stream)
    assert np.array_equal(out.cpu().numpy(), ref)
    # ... and the backstop (20 x the timeout from entry while the collective has not been reached) still ends a wait behind work that never finishes
    comm.set_timeout(1e-4)
    plan.execute_allreduce(n, ptrs, out.data_ptr(), comm, stream, want_status=False)
    with pytest.raises(api.Is3dError) as e:
        comm.synchronize(stream)
    assert e.value.code == api.IS3D_ENODEVICE and "waited" in str(e.value) and "aborted" in str(e.value)
    torch.cuda.synchronize()                                            # the kernels themselves were fine
    assert np.array_equal(out.cpu().numpy(), ref)
    with pytest.raises(api.Is3dError) as e:                             # the communicator is gone, as after any abort
        plan.execute_allreduce(n, ptrs, out.data_ptr(), comm, stream, want_status=False)
    assert e.value.code == api.IS3D_ENODEVICE
    with pytest.raises(api.Is3dError):
        comm.set_timeout(0.0)
    comm.close()
    # the same deadline inside is3d_comm_check
    comm = api.Comm(api.Comm.unique_id(), 1, 0, 0)
    comm.set_timeout(1e-4)
    plan.execute_allreduce(n, ptrs, out.data_ptr(), comm, stream, want_status=False)
    with pytest.raises(api.Is3dError) as e:
        comm.check(stream)
    assert e.value.code == api.IS3D_ENODEVICE and "is3d_comm_check" in str(e.value)
    torch.cuda.synchronize()
    comm.close()
    plan.close()


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment: bench.py starts the two rank processes itself (before it
    imports torch), they share the one GPU through --backend gloo, and exactly one JSON line comes back -- strong scaling of ONE
    surface (BASELINE config 4's shape: 2 x 1000 cells of a 2000-cell surface)."""
    import json
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cells", "2000", "--steps", "1",
                        "--warmup", "0", "--cpu-baseline-seconds", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["cells_total"] == 2000 and d["config"]["cells_per_gpu"] == 1000
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and [x["cells"] for x in d["ranks"]] == [1000, 1000]
    assert d["value"] > 0 and d["config"]["spectrum_finite"]
    _check_multi_rank_line(d)


def test_persistent_multi_plan_other_modes(fx):
    """The persistent plan for the modified-equilibrium kernel (df_mode 4), for the 2+1D kernel, with accumulate = 1, and with the RCCL
    reduction over the one device of a test box: each equals the one-shot entry bitwise."""
    cells = synth.synth_surface(900, 3, seed=15)
    fq = inputs.feqmod_tables(inputs.surface_average_T(cells))
    o = dict(dimension=3, df_mode=4)
    one, st1, _ = api.smooth_spectra_multi(cells, fx["pikp"], fx["grid"], fx["df"], o, devices=[0, 0, 0], fq=fq)
    mp = api.MultiPlan(fx["pikp"], fx["grid"], fx["df"], o, devices=[0, 0, 0], max_cells=1000, fq=fq)
    got, st, sh = mp.execute(cells)
    assert np.array_equal(got, one) and st["n_cells_narrow"] == st1["n_cells_narrow"] and len(sh) == 3
    mp.close()
    c2 = synth.synth_surface(50, 2, seed=16)
    o2 = dict(dimension=2, df_mode=1)
    one2, _, _ = api.smooth_spectra_multi(c2, fx["pikp"], fx["grid"], fx["df"], o2, devices=[0, 0])
    mp = api.MultiPlan(fx["pikp"], fx["grid"], fx["df"], dict(o2, accumulate=1), devices=[0, 0], max_cells=50)
    acc = one2.copy()
    mp.execute(c2, out=acc)
    mp.execute(c2, out=acc)
    assert relerr(acc, 3.0 * one2) < 1e-14
    mp.close()
    o3 = dict(dimension=3, df_mode=2)
    ref, _ = api.smooth_spectra(cells, fx["pikp"], fx["grid"], fx["df"], o3)
    mp = api.MultiPlan(fx["pikp"], fx["grid"], fx["df"], o3, devices=[0], reduce=api.REDUCE_RCCL, max_cells=900)
    got, _, _ = mp.execute(cells)
    assert np.array_equal(got, ref)
    mp.close()
    with pytest.raises(api.Is3dError) as e:
        api.MultiPlan(fx["pikp"], fx["grid"], fx["df"], o3, devices=[0, 0], reduce=api.REDUCE_RCCL, max_cells=900)
    assert e.value.code == api.IS3D_EINVAL and "distinct" in str(e.value)
    with pytest.raises(api.Is3dError) as e:
        api.MultiPlan(fx["pikp"], fx["grid"], fx["df"], o3, devices=[0, 7], max_cells=900)
    assert e.value.code == api.IS3D_EINVAL


def test_two_ranks_through_a_stub_communicator(tmp_path):
    """is3d_plan_execute_allreduce with TWO ranks.  RCCL refuses two ranks on one device, and a test box has one GPU, so the library's
    communicator is pointed (IS3D_RCCL_LIBRARY) at a process-level test double of the nine RCCL entry points it binds
    (tests/cpp/fake_rccl.cpp: shared-memory slots, a barrier, an abort flag).  What this exercises for real is the library's own
    multi-rank logic: the sum over ranks, the error word (a failed rank joins with zeros, every rank learns), IS3D_EPEER on the
    synchronous path, survival of the communicator.  Step 5 (rank 1 aborts, rank 0's collective fails) shows BEHAVIOUR OF THE TEST DOUBLE
    ONLY: its host-synchronous all-reduce sees a shared abort flag; a real RCCL all-reduce returns at once and a peer's local abort does not
    unblock it -- there the library's deadline (test_waits_behind_a_collective_have_a_deadline) is what ends the wait.  It is not a
    test of RCCL; the first run with more than one real RCCL rank is the driver's SCALE run."""
    import json
    import sys
    from conftest import ROOT
    so = str(tmp_path / "libfakerccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread"])
    env = dict(os.environ, IS3D_RCCL_LIBRARY=so)
    uid = str(tmp_path / "uid.bin")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi_rank_worker.py"), str(r), "2", uid], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    res = []
    for p, (so_, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
        line = [ln for ln in so_.splitlines() if ln.startswith("RESULT ")]
        assert len(line) == 1, so_ + se[-2000:]
        res.append(json.loads(line[0][7:]))
    r0, r1 = sorted(res, key=lambda d: d["rank"])
    assert r0["seen"] == [0, 2] and r1["seen"] == [1, 2]
    for r in (r0, r1):
        assert r["clean_check"] == 0 and r["clean_relerr"] < 1e-13          # 1: the sum over two ranks is the whole surface
        assert r["s2_check"] == api.IS3D_EPEER and r["s2_sum_is_rank0_only"]  # 2: everybody learns; the failed rank contributed zeros
        assert r["s4_check"] == 0 and r["s4_relerr"] < 1e-13                # 4: the communicator survived
    assert r0["s2_exec"] == 0 and r1["s2_exec"] == api.IS3D_EINVAL
    assert r0["s3_exec"] == api.IS3D_EDOMAIN and r0["s3_text_has_cell"] and r1["s3_exec"] == api.IS3D_EPEER   # 3
    assert r0["s5_exec"] == api.IS3D_ENODEVICE and "abort" in r0["s5_text"]        # 5: fails (behaviour of the test double; see the docstring)
    print("s5: rank 0's collective failed %.2f s after rank 1's abort" % r0["s5_seconds"])


def test_bench_two_ranks_through_the_library_communicator(tmp_path):
    """bench.py's N > 1 path end to end with two self-started ranks on the one GPU: process group on gloo (rendezvous and barrier only), the
    data-path collective through the LIBRARY's communicator (is3d_plan_execute_allreduce) backed by the test double of RCCL -- the code path
    the driver's SCALE run takes with real RCCL, minus RCCL."""
    import json
    import sys
    from conftest import ROOT
    so = str(tmp_path / "libfakerccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IS3D_RCCL_LIBRARY"] = so
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cells", "3000", "--steps", "2",
                        "--warmup", "1", "--cpu-baseline-seconds", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["allreduce"].startswith("is3d_plan_execute_allreduce")
    assert d["ranks_seen"] == [0, 1] and [x["comm_rank_seen"] for x in d["ranks"]] == [[0, 2], [1, 2]]
    assert all(x["allreduce_ms"] is not None and x["allreduce_ms"] >= 0 for x in d["ranks"]) and d["config"]["spectrum_finite"]
    _check_multi_rank_line(d)


def test_bench_four_ranks_uneven_shards_through_the_library_communicator(tmp_path):
    """The same path with FOUR self-started ranks and a cell count that does not divide (3 001 cells: shards of 751, 750, 750, 750), so that the first
    real N > 2 line is not also the first line with more than two `ranks[]` entries.  Four, not eight: a GPU box of this pool lets at most six
    processes use its card at once (this test process is one of them), so the N = 8 shape of BASELINE config 4 cannot be rehearsed on one GPU;
    nothing in bench.py depends on N beyond is3d_shard_bounds and the length of `ranks`."""
    import json
    import sys
    from conftest import ROOT
    so = str(tmp_path / "libfakerccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IS3D_RCCL_LIBRARY"] = so
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--cells", "3001", "--steps", "2",
                        "--warmup", "1", "--cpu-baseline-seconds", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["allreduce"].startswith("is3d_plan_execute_allreduce")
    assert d["config"]["cells_total"] == 3001 and [x["cells"] for x in d["ranks"]] == [751, 750, 750, 750]
    assert [x["first_cell"] for x in d["ranks"]] == [0, 751, 1501, 2251] and [x["rank"] for x in d["ranks"]] == [0, 1, 2, 3]
    assert d["ranks_seen"] == [0, 1, 2, 3] and [x["comm_rank_seen"] for x in d["ranks"]] == [[i, 4] for i in range(4)]
    assert all(x["allreduce_ms"] is not None and x["allreduce_ms"] >= 0 for x in d["ranks"]) and d["config"]["spectrum_finite"]
    assert d["value"] == pytest.approx(3001.0 * d["config"]["bins"] * d["config"]["species"] / (d["ms_per_step"] * 1e-3), rel=1e-9)
    _check_multi_rank_line(d)
