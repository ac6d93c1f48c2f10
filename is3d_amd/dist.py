"""Multi-GPU plumbing for the smooth-spectra path: one process per GPU, the freezeout-cell axis in
contiguous shards, no data-path collective, one all-reduce (RCCL over xGMI; gloo on CPU in tests) of
the per-bin spectrum at the end.  The reference has no distributed code."""
import os


def shard_bounds(n_cells, rank, world_size):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one cell."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(int(n_cells), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """Join the job torchrun started (MASTER_ADDR/PORT, RANK, WORLD_SIZE in the env).  backend defaults to
    nccl (= RCCL on ROCm) when a GPU is visible, else gloo."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world == 1 or dist.is_initialized():
        return rank, world, local
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend == "nccl":
        torch.cuda.set_device(local)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        except TypeError:   # a torch without the device_id argument
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def library_comm(device):
    """The library's own RCCL communicator (is3d_comm_*, include/is3d_amd.h) over the ranks of the job torchrun started:
    rank 0 makes the ncclUniqueId through the C ABI, the 128 bytes travel through the already-initialised torch.distributed
    group (any backend: this is rendezvous plumbing, not the data path), every rank joins with its HIP device.  After this
    the spectrum all-reduce is is3d_plan_execute_allreduce / Comm.allreduce -- RCCL called by the library, no torch involved."""
    import torch.distributed as dist
    from . import api
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [api.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return api.Comm(box[0], world, rank, device)


def allreduce_spectrum(spectrum):
    """In-place sum of the flat fp64 spectrum over all ranks (a no-op for a single process).
    All terms are >= 0 when outflow = 1, so the reduction order only moves the last bits."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(spectrum, op=dist.ReduceOp.SUM)
    return spectrum


def surface_average_T_global(cells):
    """Surface-volume weighted temperature over ALL ranks' shards (what the reference reads back from
    average_thermodynamic_quantities.dat, readindata.cpp:422-450): df_mode 4 builds its lambda(Pi/P), z(Pi/P) tables at this one
    temperature, so every rank must use the same value.  One all-reduce of two scalars."""
    import numpy as np
    import torch
    import torch.distributed as dist
    ut = np.sqrt(1 + cells["ux"] ** 2 + cells["uy"] ** 2 + cells["tau"] ** 2 * cells["un"] ** 2)
    uds = ut * cells["dat"] + cells["ux"] * cells["dax"] + cells["uy"] * cells["day"] + cells["un"] * cells["dan"]
    dsds = cells["dat"] ** 2 - cells["dax"] ** 2 - cells["day"] ** 2 - cells["dan"] ** 2 / cells["tau"] ** 2
    mag = np.abs(uds) + np.sqrt(np.abs(uds * uds - dsds))
    t = torch.tensor([float(np.sum(cells["T"] * mag)), float(np.sum(mag))], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0].item() / t[1].item())


def gather_particles(particles, dst=0):
    """Sampled particle lists of all ranks (numpy structured arrays, each ordered by (event, cell, draw) with GLOBAL cell
    indices, see is3d_sampler_inputs.first_cell) merged on rank `dst` into the order one GPU would have produced.  Other
    ranks get None.  The sampler itself needs no collective: the streams are keyed by the global cell index."""
    import numpy as np
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return particles
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(particles, parts, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged = np.concatenate(parts)
    return merged[np.lexsort((merged["cell"], merged["event"]))]     # stable: the draw order within a (event, cell) survives
