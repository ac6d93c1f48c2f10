"""Worker of tests/test_gpu_multi.py::test_two_ranks_through_a_stub_communicator: one RANK of a two-process job that shares the box's one GPU.
The library's communicator is backed by the process-level test double tests/cpp/fake_rccl.cpp (IS3D_RCCL_LIBRARY), so that
is3d_plan_execute_allreduce runs its multi-rank control flow for real.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, n_ranks, uid_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch
    from is3d_amd import api, inputs, synth
    dev = torch.device("cuda:0")
    if rank == 0:
        uid = api.Comm.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    comm = api.Comm(uid, n_ranks, rank, 0)
    g = inputs.grid()
    grid = dict(pT=g["pT"][::2], phi=g["phi"][::2], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df, sp = inputs.df_tables(), inputs.species("pikp")
    n = 1001
    cells = synth.synth_surface(n, 3, seed=77)
    lo, hi = api.shard_bounds(n, rank, n_ranks)
    o = dict(dimension=3, df_mode=2)
    tens = {k: torch.from_numpy(cells[k][lo:hi].copy()).to(dev) for k in synth.CELL_FIELDS}
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    plan = api.Plan(sp, grid, df, o, max_cells=hi - lo)
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    whole, _ = api.smooth_spectra(cells, sp, grid, df, o)
    mine, _ = api.smooth_spectra({k: v[lo:hi] for k, v in cells.items()}, sp, grid, df, o)
    res = dict(rank=rank, seen=list(comm.rank_seen()))

    def code(fn):
        try:
            fn()
            return 0, ""
        except api.Is3dError as e:
            return e.code, str(e)

    # 1. clean step: every rank ends with the whole surface's spectrum
    plan.execute_allreduce(hi - lo, ptrs, out.data_ptr(), comm, stream, want_status=False)
    res["clean_check"] = code(lambda: comm.check(stream))[0]
    got = out.cpu().numpy()
    res["clean_relerr"] = float(np.max(np.abs(got - whole) / np.maximum(np.abs(whole), 1e-280)))
    # 2. rank 1 asks for more cells than its plan holds (asynchronous call): it joins with zeros, everybody learns at the check
    out.fill_(5.0)
    res["s2_exec"] = code(lambda: plan.execute_allreduce(hi - lo + (1 if rank == 1 else 0), ptrs, out.data_ptr(), comm, stream, want_status=False))[0]
    c2 = code(lambda: comm.check(stream))
    res["s2_check"] = c2[0]
    got = out.cpu().numpy()
    only0, _ = api.smooth_spectra({k: v[:api.shard_bounds(n, 0, n_ranks)[1]] for k, v in cells.items()}, sp, grid, df, o)
    res["s2_sum_is_rank0_only"] = bool(np.max(np.abs(got - only0) / np.maximum(np.abs(only0), 1e-280)) < 1e-13)
    # 3. rank 0 meets a cell outside the coefficient table (synchronous call): its own code there, IS3D_EPEER on the other rank
    t3 = float(tens["T"][3].item())
    if rank == 0:
        tens["T"][3] = 0.3
    c3 = code(lambda: plan.execute_allreduce(hi - lo, ptrs, out.data_ptr(), comm, stream))
    res["s3_exec"], res["s3_text_has_cell"] = c3[0], ("cell 3" in c3[1])
    if rank == 0:
        tens["T"][3] = t3
    comm_state = code(lambda: comm.check(stream))[0]   # the synchronous call already consumed the word on the peer; rank 0 still holds it
    res["s3_check_after"] = comm_state
    # 4. a clean step again: the communicator survived both failures
    plan.execute_allreduce(hi - lo, ptrs, out.data_ptr(), comm, stream, want_status=False)
    res["s4_check"] = code(lambda: comm.check(stream))[0]
    got = out.cpu().numpy()
    res["s4_relerr"] = float(np.max(np.abs(got - whole) / np.maximum(np.abs(whole), 1e-280)))
    # 5. rank 1 leaves the job: ncclCommAbort; rank 0's collective fails instead of waiting for ever
    if rank == 1:
        time.sleep(0.5)
        comm.abort()
        res["s5_exec"] = None
    else:
        t0 = time.time()
        c5 = code(lambda: plan.execute_allreduce(hi - lo, ptrs, out.data_ptr(), comm, stream))
        res["s5_exec"], res["s5_seconds"], res["s5_text"] = c5[0], time.time() - t0, c5[1][-160:]
    res["mine_ok"] = bool(np.isfinite(mine).all())
    plan.close()
    comm.close()
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
