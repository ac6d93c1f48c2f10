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
