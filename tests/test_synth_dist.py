"""CPU: seeded synthetic surfaces and the N > 1 path (cell shards + one all-reduce) on gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr
from is3d_amd import dist as idist
from is3d_amd import inputs, synth


@pytest.mark.parametrize("dim", [2, 3])
def test_surface_properties(dim):
    s = synth.synth_surface(5000, dim)
    ut = np.sqrt(1 + s["ux"] ** 2 + s["uy"] ** 2 + s["tau"] ** 2 * s["un"] ** 2)
    uds = ut * s["dat"] + s["ux"] * s["dax"] + s["uy"] * s["day"] + s["un"] * s["dan"]
    assert (uds > 0).all()                                        # no skipped cells
    assert s["T"].min() >= 0.140 and s["T"].max() <= 0.160        # inside the [0.1, 0.2] GeV coefficient tables
    assert (s["tau"] >= 1).all() and (s["tau"] <= 10).all()
    if dim == 2:
        assert not s["dan"].any() and not s["un"].any() and not s["eta"].any() and not s["pixn"].any()
    else:
        assert np.abs(s["eta"]).max() <= 4.0 and s["dan"].any()
    assert abs(np.mean(s["pixx"] / (0.02 * (s["E"] + s["P"])))) < 0.05 and abs(np.std(s["pixx"] / (0.02 * (s["E"] + s["P"]))) - 1) < 0.05


def test_surface_is_counter_based():
    """A rank can generate its own slice: cells depend only on (seed, global index)."""
    a = synth.synth_surface(1000, 3)
    b = synth.synth_surface(300, 3, first_cell=450)
    for k in synth.CELL_FIELDS:
        assert np.array_equal(a[k][450:750], b[k])
    assert not np.array_equal(synth.synth_surface(10, 3, seed=1)["tau"], synth.synth_surface(10, 3, seed=2)["tau"])
    assert synth.SEED_CONFIG2 == 20260001 and synth.SEED_CONFIG3 == 20260002


def test_shard_bounds_cover_the_surface():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = [idist.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        idist.shard_bounds(10, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle  # the CPU checker stands in for the GPU kernel here: this test is about the sharding
    r, w, _ = idist.init_process_group("gloo")
    g = inputs.grid()
    grid = dict(pT=g["pT"][::4], phi=g["phi"][::4], y=g["y"][::3], eta=g["eta"], eta_w=g["eta_w"])
    n = 37
    lo, hi = idist.shard_bounds(n, r, w)
    cells = synth.synth_surface(hi - lo, 3, seed=123, first_cell=lo)
    part = oracle.dN_pTdpTdphidy(cells, inputs.species("pikp"), grid, inputs.df_tables(), dict(dimension=3, df_mode=2))
    t = torch.from_numpy(part.copy())
    idist.allreduce_spectrum(t)
    dist.barrier()
    # df_mode 4 needs ONE surface-average temperature for all ranks; the sampler shards by global cell index
    T_glob = idist.surface_average_T_global(cells)
    nn = 300
    lo2, hi2 = idist.shard_bounds(nn, r, w)
    shard = synth.synth_surface(hi2 - lo2, 3, seed=321, first_cell=lo2)
    gla = inputs.feqmod_tables(0.15)
    pl, _ = oracle.sample_particles(shard, inputs.species("pikp"), inputs.df_tables(), gla, dict(dimension=3, df_mode=2), n_events=30, seed=9,
                                    first_cell=lo2)
    import numpy as np
    from is3d_amd import api
    arr = np.zeros(len(pl["E"]), dtype=api.PARTICLE_DTYPE)
    for f in arr.dtype.names:
        arr[f] = pl[f]
    merged = idist.gather_particles(arr)
    if r == 0:
        whole = oracle.dN_pTdpTdphidy(synth.synth_surface(n, 3, seed=123), inputs.species("pikp"), grid, inputs.df_tables(),
                                      dict(dimension=3, df_mode=2))
        T_whole = inputs.surface_average_T(synth.synth_surface(n, 3, seed=123))
        ref, _ = oracle.sample_particles(synth.synth_surface(nn, 3, seed=321), inputs.species("pikp"), inputs.df_tables(), gla,
                                         dict(dimension=3, df_mode=2), n_events=30, seed=9)
        same = len(merged) == len(ref["E"]) and all(np.array_equal(merged[f], ref[f]) for f in merged.dtype.names)
        q.put((t.numpy().copy(), whole, T_glob, T_whole, bool(same), len(merged)))
    else:
        assert merged is None
    dist.destroy_process_group()


def _worker8(rank, world, port, q):
    """BASELINE config 4's shape on the host side: eight ranks, eight contiguous shards of ONE surface, one sum of the spectrum, the line's
    `ranks[]` gathered as bench.py gathers it (all_gather_object) -- the oracle stands in for the kernel, gloo for RCCL."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    oracle.set_num_threads(1)
    r, w, _ = idist.init_process_group("gloo")
    g = inputs.grid()
    grid = dict(pT=g["pT"][::8], phi=g["phi"][::6], y=g["y"][::5], eta=g["eta"], eta_w=g["eta_w"])
    n = 83                                                       # 83 = 8 x 10 + 3: shards of 11, 11, 11, 10, 10, 10, 10, 10
    lo, hi = idist.shard_bounds(n, r, w)
    cells = synth.synth_surface(hi - lo, 3, seed=77, first_cell=lo)
    part = oracle.dN_pTdpTdphidy(cells, inputs.species("pikp"), grid, inputs.df_tables(), dict(dimension=3, df_mode=2))
    t = torch.from_numpy(part.copy())
    idist.allreduce_spectrum(t)
    ranks = [None] * w
    dist.all_gather_object(ranks, dict(rank=r, cells=hi - lo, first_cell=lo))
    tt = torch.tensor([float(r)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)                    # bench.py's MAX over ranks of the step time
    dist.barrier()
    if r == 0:
        whole = oracle.dN_pTdpTdphidy(synth.synth_surface(n, 3, seed=77), inputs.species("pikp"), grid, inputs.df_tables(), dict(dimension=3, df_mode=2))
        q.put((t.numpy().copy(), whole, ranks, float(tt.item())))
    dist.destroy_process_group()


def test_eight_rank_shards_allreduce_to_the_whole_spectrum():
    """world_size 8 on gloo (CPU): what can be rehearsed of BASELINE config 4 without eight GPUs -- a GPU box lets at most six processes use its
    card, so the eight-rank shape runs here, on the host side (tests/test_gpu_multi.py has two and four ranks through the library's communicator)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got, whole, ranks, tmax = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert relerr(got, whole) < 1e-13
    assert [x["rank"] for x in ranks] == list(range(8)) and [x["cells"] for x in ranks] == [11, 11, 11, 10, 10, 10, 10, 10]
    assert [x["first_cell"] for x in ranks] == [0, 11, 22, 33, 43, 53, 63, 73] and tmax == 7.0


def test_two_rank_shards_allreduce_to_the_whole_spectrum():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, whole, T_glob, T_whole, same, n_sampled = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert relerr(got, whole) < 1e-13
    assert abs(T_glob / T_whole - 1) < 1e-14
    assert same and n_sampled > 20          # the two shards' particle lists, merged, are the one-process list
