#!/usr/bin/env python3
"""bench.py -- throughput of the smooth Cooper-Frye spectra path on N GPUs of one node.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N ...            (no launcher: starts N rank processes itself, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Metric (BASELINE.json): FO-cell x momentum-bin x species evals/sec.  A "step" is one pass of the hot path
(prep -> main -> finalize, plus the spectrum all-reduce when N > 1) over one synthetic surface whose cell
arrays are already resident in HBM.  Default workload = BASELINE config 3 (the configuration north_star
quotes its target on): 1e6-cell seeded 3+1D surface, Chapman-Enskog delta-f, the 305-species urqmd list,
32 x 24 x 21 momentum bins = 4.919e12 evals per step.  N > 1 is BASELINE config 4: the SAME 1e6-cell surface,
its cell axis in N contiguous shards (8 x 125 000 cells at N = 8) -- strong scaling, `"scaling": "strong"` -- and one
RCCL all-reduce of the 39 MB spectrum ends the step (the sum over chunks of cells of
/root/reference/src/cpp/emissionfunction_smooth_kernels.cpp:363-375 is what makes the split exact).  `--scaling weak` gives
every rank its own 1e6-cell slice of the (infinite, counter-based) seeded surface instead.  `--workload config2` gives
the 1e5-cell 2+1D case, `--workload config5` the smooth leg of BASELINE config 5: the anisotropic-hydro (VAH) kernel on a 1e6-cell
surface with its 14-moment coefficients interpolated on the device from the deltaf_coefficients/vah tables.

One JSON line on stdout (rank 0).  Besides the contract's fields it carries
  roofline        the contract's object for the dominant kernel, HBM view: algorithmic bytes / kernel time
  roofline_valu   the roofline that actually binds this kernel: fp64 VALU
  cpu_baseline    the CPU oracle (a port of the reference loop) timed on this host's cores (all of them: `cores`, `cores_available`, `cpu_model`),
                  bounded sample, in the same run for EVERY N (rank 0, after the timed region, the other ranks parked at the closing barrier)
  executed_evals_per_s        `value` counts evals as the reference would execute them (every species, every row); the kernel
                              evaluates one representative per distinct (mass, sign) class (75 of 305) and skips rows that
                              cannot change a bit of the result -- this is the rate of integrands actually executed
  value_incl_transfers, ms_per_step_incl_transfers   the same steps with the H->D upload of the cell arrays (every rank its shard) and the
                              D->H download of the spectrum inside the timed region (SURVEY.md 8d's t_kernel; `value` itself follows the bench
                              contract: inputs resident in HBM when the timed region starts), any N
  kernel_ms.main_no_cull      the dominant kernel with all culling off (zero_skip = 2): the data-independent floor
  kernel_ms.surface_cull      config 3 only: the dominant kernel with the opt-in surface-relative cull (zero_skip = 3), and how far its spectrum is from the default's
  ranks           N > 1: one entry per rank -- device, cells, kernel_ms, allreduce_ms (device time of the collective on that rank,
                  the wait for the slowest rank included), the rank / size its library communicator reports (is3d_comm_rank), and the
                  rank's own roofline_valu (its culled fraction, its kernel time)
N > 1: the all-reduce is the library's own RCCL call (is3d_plan_execute_allreduce over an is3d_comm); torch.distributed
only launches the ranks, ships the ncclUniqueId and provides the barrier.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# the cpu_baseline leg runs an OpenMP team on every core the process may use; where the affinity mask is wider than the CPU share (a container),
# idle team members must sleep, not spin the share away (read by libgomp when it is first loaded: before numpy / torch)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def workload(name):
    if name == "config3":
        return dict(name="config3", dimension=3, df_mode=2, species="urqmd", cells=1000000,
                    text="BASELINE config 3: 1e6-cell synthetic 3+1D surface (seed 20260002), Chapman-Enskog delta-f, "
                         "305-species pdg-urqmd_v3.3+ list, 32x24x21 (pT,phi,y) bins")
    if name == "config5":
        return dict(name="config5", dimension=3, df_mode=4, species="urqmd", cells=1000000,
                    text="BASELINE config 5 (smooth leg): 1e6-cell synthetic 3+1D anisotropic-hydro (VAH, P_L matching) surface (seed 20260002), "
                         "14-moment delta-f with c0..c4 interpolated on the device from the deltaf_coefficients/vah tables, "
                         "305-species pdg-urqmd_v3.3+ list, 32x24x21 (pT,phi,y) bins")
    if name == "config2":
        return dict(name="config2", dimension=2, df_mode=1, species="pikp", cells=100000,
                    text="BASELINE config 2: 1e5-cell synthetic 2+1D boost-invariant surface (seed 20260001), 14-moment "
                         "delta-f, pi/K/p, 32x24 (pT,phi) bins x 241-point eta quadrature")
    raise SystemExit("unknown workload %s" % name)


def isa_counts(kernel_name, wl, JT_R, variant=0, baryon=False):
    p = os.path.join(ROOT, "is3d_amd", "csrc", "isa_counts.json")
    if not os.path.exists(p):
        return None
    d = json.load(open(p))
    ce, d3 = int(wl["df_mode"] == 2), int(wl["dimension"] == 3)
    if kernel_name == "cf_main_feqmod":
        rows = (0 if variant == 5 else 1 if variant == 6 else 2) if d3 else (3 if variant == 7 else 0)   # how the kernel walks a unit's rows (cf_feqmod.hip)
        key = "cf_main_feqmod:DIM3=%d,OUTFLOW=1,MODE3=%d,JT=%d,R=%d,ROWS=%d" % (d3, int(wl["df_mode"] == 3), JT_R[0], JT_R[1], rows)
    elif kernel_name == "cf_main_vah":
        key = "cf_main_vah:DIM3=%d,REG=1,JT=%d,R=%d" % (d3, JT_R[0], JT_R[1])
    elif kernel_name == "cf_main_vah3":
        key = "cf_main_vah3:DIM3=%d,REG=1,JT=%d,R=%d" % (d3, JT_R[0], JT_R[1])
    elif kernel_name == "cf_main_tile3e":
        key = "cf_main_tile3e%s:CE=%d,OUTFLOW=1,REG=1,JT=%d,R=%d,MODE=%d" % ("_baryon" if baryon else "", ce, JT_R[0], JT_R[1], 0 if variant == 5 else 1)
    elif kernel_name == "cf_main_tile":
        key = "cf_main_tile:CE=%d,DIM3=%d,OUTFLOW=1,REG=1,BARYON=%d,JT=%d,R=%d" % (ce, d3, int(baryon), JT_R[0], JT_R[1])
    else:
        key = "cf_main_direct:CE=%d,DIM3=%d,OUTFLOW=1,REG=1,KT=%d" % (ce, d3, JT_R[1])
    return d.get(key)


def host_cores():
    """Cores this process may actually use: its affinity mask, capped by the cgroup CPU quota (a 1-GPU box of the pool sees all 256 hardware
    threads in its mask but owns a 16-core share: 256 OpenMP threads on that share ran the oracle 7x SLOWER than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                                 # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n


def host_cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_mem_available():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return 1024.0 * float(ln.split()[1])
    except OSError:
        pass
    return 16e9


def pmc_traffic(workload_name, n_cells):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE need their
    own passes and cannot be collected inside this run): the NEWEST profiles/rNN_pmc_traffic.json (by round number, found by glob -- a new
    round's profile is picked up without touching this file) that holds this workload at this shard size; `shards` maps a cell count
    (a rank's shard at N = 1, 2, 4, 8) to its counter set.  The source file is named in the line (`traffic_source`): a kernel changed
    since that round's passes is visible there."""
    import glob
    import re
    found = []
    for tp in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        m = re.match(r"r(\d+)_pmc_traffic\.json$", os.path.basename(tp))
        if m:
            found.append((int(m.group(1)), tp))
    for _, tp in sorted(found, reverse=True):
        try:
            tj = json.load(open(tp)).get(workload_name)
        except (OSError, ValueError):
            continue
        if not tj:
            continue
        hit = tj if tj.get("cells") == n_cells else (tj.get("shards") or {}).get(str(n_cells))
        if hit and hit.get("hbm_bytes_per_launch"):
            return hit["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc passes of this command at %d cells, not measured in this run)" % (os.path.basename(tp), n_cells)
    return None, None


def cpu_baseline(wl, sp, grid, df, seconds_budget=25.0, fq=None):
    """The oracle (variant B, a scratch-free port of smooth_kernels.cpp:106-349; OpenMP over cells) on the first
    cells of the same surface, all host cores.  Also the reference-shaped variant A (chunk + scratch + reduce)."""
    from is3d_amd import synth
    from oracle import oracle  # checker doubling as the reported CPU baseline
    nbins = len(grid["pT"]) * len(grid["phi"]) * (len(grid["y"]) if wl["dimension"] == 3 else 1)
    per_cell = nbins * len(sp["mass"])
    opts = dict(dimension=wl["dimension"], df_mode=wl["df_mode"])
    if wl.get("baryon"):
        opts.update(include_baryon=1, include_baryondiff_deltaf=1)
    vah_tab = df if wl["name"] == "config5" else None   # config5: `df` is the VAH (Lambda, alpha_L) table set
    # every host core this process may run on ("all host cores, count and model stated"); the oracle keeps one
    # partial spectrum per thread (39 MB for config 3), so the count is bounded by a quarter of the free host memory, nothing else
    avail = host_cores()
    per_thread = 8.0 * per_cell + 1e6
    tmax = max(1, min(avail, int(0.25 * host_mem_available() / per_thread)))
    make = (lambda n, d: synth.synth_surface(n, d, baryon=True)) if wl.get("baryon") else synth.synth_surface
    if vah_tab is not None:
        make = synth.synth_vah_surface

        def run(c, **kw):   # the reference's order: coefficients into the surface (deltafReader.cu:224-278), then the kernel
            coef, found = oracle.vah_coefficients(vah_tab, c["Lambda"], c["aL"])
            assert found.all()
            return oracle.dN_pTdpTdphidy_vah(dict(c, **coef), sp, grid, dict(dimension=wl["dimension"]))
    elif fq is not None:
        run = lambda c, **kw: oracle.dN_pTdpTdphidy_feqmod(c, sp, grid, df, fq, opts)
    else:
        run = lambda c, **kw: oracle.dN_pTdpTdphidy(c, sp, grid, df, opts, **kw)
    # How many threads: the affinity mask can be wider than what the process may use (a 1-GPU box of the pool shows 256 hardware threads and
    # owns a 16-core share that no cgroup file here reveals), and an oversubscribed OpenMP team ran this oracle 7x slower than one that fits.
    # So the usable parallelism is MEASURED first: a cheap run (3 species instead of the workload's list) on all tmax threads, CPU time of the
    # process over wall time = the cores it was really given; the team is that many threads (all of them when the ratio is within 20 % of tmax).
    import resource
    from is3d_amd import inputs as _inputs
    sp_probe = _inputs.species("pikp")
    oracle.set_num_threads(tmax)
    pc = synth.synth_surface(64 * tmax, wl["dimension"])
    po = dict(dimension=wl["dimension"], df_mode=2 if wl["dimension"] == 3 else 1)
    pdf = _inputs.df_tables()
    oracle.dN_pTdpTdphidy({k: v[:tmax] for k, v in pc.items()}, sp_probe, grid, pdf, po)   # starts the team
    r0, w0 = resource.getrusage(resource.RUSAGE_SELF), time.time()
    oracle.dN_pTdpTdphidy(pc, sp_probe, grid, pdf, po)
    r1, w1 = resource.getrusage(resource.RUSAGE_SELF), time.time()
    eff = ((r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)) / max(w1 - w0, 1e-6)
    threads = tmax if eff > 0.8 * tmax else max(1, min(tmax, int(round(eff))))
    threads = oracle.set_num_threads(threads)
    probe = make(2 * threads, wl["dimension"])
    t0 = time.time()
    run(probe)
    t_probe = max(time.time() - t0, 1e-3)
    n = int(max(2 * threads, min(65536, 2 * threads * (0.5 * seconds_budget / t_probe))))
    n -= n % threads
    cells = make(n, wl["dimension"])
    t0 = time.time()
    run(cells)
    tb = time.time() - t0
    res = dict(value=n * per_cell / tb, unit="evals/s", cores=threads, cores_available=avail, cores_effective_measured=eff,
               cpu_model=host_cpu_model(), kind="port",
               sample="first %d cells of the workload surface x all %d species x %d bins, oracle %s, %.1f s" % (
                   n, len(sp["mass"]), nbins, "VAH restatement (coefficients + kernel)" if vah_tab is not None else
                   "feqmod restatement" if fq is not None else "variant B (no scratch)", tb))
    if fq is not None or vah_tab is not None:
        return res
    # variant A: the reference's own structure; scratch = npart*chunk*bins*8 B must fit
    na = n
    while na * per_cell * 8 > 6e9 and na > threads:
        na //= 2
    ca = {k: v[:na] for k, v in cells.items()}
    t0 = time.time()
    oracle.dN_pTdpTdphidy(ca, sp, grid, df, opts, chunked=True, FO_chunk=na)
    ta = time.time() - t0
    res["reference_shaped"] = dict(value=na * per_cell / ta, unit="evals/s", cores=threads,
                                   sample="%d cells, oracle variant A (10000-cell-chunk scratch + collapse(4) reduction as in the reference), %.1f s" % (na, ta))
    return res


def launcher_command(argv, n_ranks, port):
    """The child a launcher-less `python bench.py --gpus N` starts: torch.distributed.run with N fresh rank processes of this
    script, same arguments, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this process becomes the launcher.  It has not imported
    torch and has made no HIP call -- the ranks are fresh child processes, nothing that initialised the GPU is ever re-exec'd --
    it relays rank 0's one JSON line and returns the children's exit code."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(argv, n_ranks, port)
    print("bench.py: no launcher in the environment, starting %d ranks: %s" % (n_ranks, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    out, _ = proc.communicate()
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    for ln in lines[-1:]:
        sys.stdout.write(ln + "\n")
    sys.stdout.flush()
    if proc.returncode != 0:
        print("bench.py: the rank processes exited with code %d" % proc.returncode, file=sys.stderr, flush=True)
        return proc.returncode if 0 < proc.returncode < 256 else 1
    if not lines:
        print("bench.py: the rank processes printed no result line", file=sys.stderr, flush=True)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config3", choices=["config3", "config2", "config5", "config5-sampler"])
    ap.add_argument("--dimension", type=int, default=0, choices=[0, 2, 3],
                    help="--workload config5 only: 2 = the 2+1D anisotropic-hydro kernel on a 1e5-cell boost-invariant surface, pi/K/p (config 2's shape)")
    ap.add_argument("--events", type=int, default=20, help="--workload config5-sampler: events sampled per step")
    ap.add_argument("--df-mode", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="override the workload's df_mode (3, 4: modified-equilibrium kernel; not the BASELINE metric's configuration)")
    ap.add_argument("--include-baryon", action="store_true",
                    help="config3 / config2 with df_mode 1, 2: include_baryon = 1 and include_baryondiff_deltaf = 1 -- mu_B, n_B, V^mu per cell, bilinear (T, mu_B) "
                         "coefficients, 124 species classes (SURVEY.md 8f rank 1; smooth_kernels.cpp:186-197, :297, :303-321); not the BASELINE metric's configuration")
    ap.add_argument("--cells", type=int, default=0, help="override the surface size: total cells (strong scaling) / cells per GPU (weak)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--zero-skip", type=int, default=0, choices=[0, 1, 2, 3], help="dev: culling mode of the main kernel (2 = off; 3 = surface-relative floors, bounded instead of bitwise: include/is3d_amd.h)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (default) = BASELINE config 4, ONE surface in N shards; weak = every rank its own surface")
    ap.add_argument("--cell-chunks", type=int, default=0, help="dev: override the number of cell chunks of the main kernel's grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=25.0, help="budget of CPU work for the cpu_baseline sample (variant B; variant A takes about as long again)")
    ap.add_argument("--no-clock-probe", action="store_true")
    ap.add_argument("--no-cull-check", action="store_true", help="skip the untimed bitwise comparison with culling off")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="dev, --gpus 1 only: run the N > 1 code path (process group, library RCCL communicator, is3d_plan_execute_allreduce, "
                         "barriers, MAX over ranks) with a single rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo lets several ranks share one GPU (rehearsal on a 1-GPU box)")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))   # before torch is imported or HIP touched in this process

    # stdout carries ONE JSON line and nothing else: native libraries print there too (RCCL's version banner when a communicator
    # is created, gloo's connection notice), so file descriptor 1 is pointed at stderr for the run and the line goes to the saved one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from is3d_amd import api, inputs, synth
    from is3d_amd import dist as idist

    rank, world, local = idist.env_rank_world()
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    local = local % torch.cuda.device_count()   # ranks may share a GPU only in a --backend gloo rehearsal
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm, allreduce_by = None, None
    multi = world > 1 or a.rehearse_comm      # the distributed code path (a one-rank rehearsal takes it too)
    if multi:
        if world == 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group(a.backend, rank=0, world_size=1)
        else:
            idist.init_process_group(a.backend)
            import torch.distributed as dist
        if a.backend == "nccl" or os.environ.get("IS3D_RCCL_LIBRARY"):
            # the data-path collective is the library's: an is3d_comm (RCCL) built from an id made through the C ABI
            # (with --backend gloo only when IS3D_RCCL_LIBRARY names the communicator library: the test suite's two-ranks-on-one-GPU rehearsal)
            try:
                comm = idist.library_comm(local)
                allreduce_by = "is3d_plan_execute_allreduce (librccl ncclAllReduce called by libis3d_amd.so)"
            except Exception as e:   # keep the scaling run alive: torch's RCCL group does the same sum
                print("bench.py rank %d: library communicator unavailable (%s); falling back to torch.distributed.all_reduce" % (rank, e),
                      file=sys.stderr, flush=True)
        if comm is not None:
            # RCCL builds its rings / connections lazily at the first collective: pay that here, not in a timed step (--warmup 0 included)
            try:
                prime = torch.zeros(1024, dtype=torch.float64, device=dev)
                comm.set_timeout(120.0)   # a communicator that cannot complete its first collective is given up within two minutes (deadline wait, not torch.cuda.synchronize)
                comm.allreduce(prime.data_ptr(), prime.numel(), torch.cuda.current_stream().cuda_stream)
                comm.synchronize(torch.cuda.current_stream().cuda_stream)
                comm.set_timeout(300.0)
            except Exception as e:
                print("bench.py rank %d: priming all-reduce failed (%s); falling back to torch.distributed.all_reduce" % (rank, e), file=sys.stderr, flush=True)
                comm.close()
                comm = None
        ok = torch.tensor([1 if comm is not None else 0], device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
            comm = None
            allreduce_by = "torch.distributed.all_reduce (%s)" % a.backend

    if a.workload == "config5-sampler":
        if comm is not None:
            comm.close()   # the sampler shards need no collective: the streams are keyed by the global cell index
        bench_sampler(a, rank, world, local, dev, multi, json_fd)
        if multi:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    wl = workload(a.workload)
    if a.dimension and a.workload != "config5":
        raise SystemExit("--dimension applies to --workload config5 only")
    if a.workload == "config5" and a.dimension == 2:
        wl.update(dimension=2, species="pikp", cells=100000,
                  text="anisotropic-hydro (VAH, P_L matching) kernel in 2+1D: 1e5-cell synthetic boost-invariant surface (seed 20260001), 14-moment delta-f with "
                       "c0..c4 interpolated on the device from the deltaf_coefficients/vah tables, pi/K/p, 32x24 (pT,phi) bins x 241-point eta quadrature "
                       "(config 2's shape on config 5's kernel; not a BASELINE configuration)")
    if a.df_mode:
        wl["text"] += " -- df_mode overridden to %d" % a.df_mode
        wl["df_mode"] = a.df_mode
    if a.include_baryon:
        if a.workload not in ("config3", "config2") or wl["df_mode"] not in (1, 2):
            raise SystemExit("--include-baryon applies to --workload config3 / config2 with df_mode 1 or 2")
        wl["baryon"] = True
        wl["text"] += " -- include_baryon = 1, include_baryondiff_deltaf = 1: + mu_B, n_B, V^x, V^y, V^eta per cell, bilinear (T, mu_B) coefficient tables"
    g = inputs.grid()
    grid = dict(pT=g["pT"], phi=g["phi"], y=g["y"], eta=g["eta"], eta_w=g["eta_w"])
    df = inputs.df_tables_full() if wl.get("baryon") else inputs.df_tables()
    sp = inputs.species(wl["species"])
    n_total = (a.cells or wl["cells"]) * (world if a.scaling == "weak" else 1)
    lo, hi = idist.shard_bounds(n_total, rank, world)
    n_loc = hi - lo
    vah = wl["name"] == "config5"
    opts = dict(dimension=wl["dimension"], df_mode=wl["df_mode"], kernel_variant=a.variant, device=local, zero_skip=a.zero_skip, cell_chunks=a.cell_chunks)
    if wl.get("baryon"):
        opts.update(include_baryon=1, include_baryondiff_deltaf=1)
    fq = None
    if vah:
        # anisotropic hydro: 24 cell arrays (no T, no c0..c4: the coefficients are interpolated on the device from the tables)
        if a.df_mode:
            raise SystemExit("--df-mode does not apply to --workload config5")
        cells = synth.synth_vah_surface(n_loc, wl["dimension"], first_cell=lo)
        cell_fields = [f for f in api.VAH_FIELDS[:25] if f != "T"]
        df = inputs.vah_df_tables()
        opts = dict(dimension=wl["dimension"], device=local, zero_skip=a.zero_skip, cell_chunks=a.cell_chunks, kernel_variant=a.variant)
        tens = {k: torch.from_numpy(cells[k]).to(dev) for k in cell_fields}   # resident in HBM before timing
        make_plan = lambda o: api.VahPlan(sp, grid, o, tab=df, max_cells=max(n_loc, 1))
    else:
        cells = synth.synth_surface(n_loc, wl["dimension"], first_cell=lo, baryon=bool(wl.get("baryon")))
        cell_fields = list(synth.CELL_FIELDS) + (list(synth.BARYON_FIELDS) if wl.get("baryon") else [])
        tens = {k: torch.from_numpy(cells[k]).to(dev) for k in cell_fields}   # resident in HBM before timing
        if wl["df_mode"] in (3, 4):   # modified equilibrium: Gauss-Laguerre nodes, PDG list, the surface-average temperature (all ranks' cells)
            fq = inputs.feqmod_tables(idist.surface_average_T_global(cells))
        make_plan = lambda o: api.Plan(sp, grid, df, o, max_cells=max(n_loc, 1), fq=fq)
    plan = make_plan(opts)
    plan.set_timing(True)
    out = torch.zeros(plan.output_size, dtype=torch.float64, device=dev)
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    stream = torch.cuda.current_stream().cuda_stream

    ms = dict(prep=[], main=[], finalize=[], allreduce=[])

    def step(record):
        if comm is not None and vah:
            plan.execute(n_loc, ptrs, out.data_ptr(), stream, want_status=False)
            comm.allreduce(out.data_ptr(), plan.output_size, stream)   # is3d_comm_allreduce: the VAH plan has no fused form
        elif comm is not None:
            plan.execute_allreduce(n_loc, ptrs, out.data_ptr(), comm, stream, want_status=False)
        else:
            plan.execute(n_loc, ptrs, out.data_ptr(), stream, want_status=False)
            if multi:
                if record:
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                idist.allreduce_spectrum(out)
                if record:
                    torch.cuda.synchronize()
                    ms["allreduce"].append((time.perf_counter() - ta) * 1e3)   # host clock: the torch / gloo fallback of a rehearsal
        if record:
            t = plan.timings()   # HIP events recorded on `stream` around each kernel of this step
            ms["prep"].append(t["ms_prep"])
            ms["main"].append(t["ms_main"])
            ms["finalize"].append(t["ms_finalize"])
            if comm is not None:
                ms["allreduce"].append(comm.allreduce_ms())   # HIP events on `stream` around the library's RCCL group

    def fence():
        if comm is not None:
            comm.synchronize(stream)   # hipStreamSynchronize with a deadline: a rank whose peer died raises (exits non-zero) instead of hanging the node
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if not vah:
        plan.check(stream)   # the timed steps ran without status read-back: any cell outside the coefficient table surfaces here
    if comm is not None and not vah:
        comm.check(stream)   # ... and so does any rank whose execute failed before a collective (the error word summed beside the spectrum)
    st = plan.execute(n_loc, ptrs, out.data_ptr(), stream)   # untimed: status (classes, skipped cells) + sanity
    torch.cuda.synchronize()
    nbins = len(grid["pT"]) * len(grid["phi"]) * (len(grid["y"]) if wl["dimension"] == 3 else 1)
    culled = (st["n_wave_rows_culled"] / st["n_wave_rows"]) if st.get("n_wave_rows") else 0.0
    unique_evals = float(n_loc) * nbins * st["n_classes"] * (len(grid["eta"]) if wl["dimension"] == 2 else 1)
    executed_evals = unique_evals * (1.0 - culled)   # rows the kernel proved unable to change a bit for a whole wave are not executed
    ms_main = float(np.mean(ms["main"]))
    # ---- the binding roofline, per rank: fp64 VALU (flops and instructions per integrand from the emitted ISA, this rank's own culled fraction and kernel time)
    ic = isa_counts(plan.main_kernel_name, wl, plan.tile_shape, st["kernel_variant"], baryon=bool(wl.get("baryon")))

    def valu_roofline(clock_ghz=None):
        if not ic:
            return None
        tf = executed_evals * ic["flop_per_eval"] / (ms_main * 1e-3) / 1e12
        return dict(bound="fp64_valu", achieved=tf, peak=FP64_VALU_PEAK_TF, unit="TFLOP/s", frac=tf / FP64_VALU_PEAK_TF,
                    executed_flop_per_eval=ic["flop_per_eval"], fp64_valu_instr_per_eval=ic["valu_f64_instr_per_eval"],
                    issue_cycles_per_eval=ic["issue_cycles_per_eval"],
                    issue_bound_frac_at_2p4GHz=executed_evals / 64.0 * ic["issue_cycles_per_eval"] / (1024 * 2.4e9 * ms_main * 1e-3),
                    shader_clock_ghz=clock_ghz,
                    frac_at_shader_clock=(tf / (FP64_VALU_PEAK_TF * clock_ghz / 2.4)) if clock_ghz else None,
                    integrands_per_launch=unique_evals, integrands_executed=executed_evals, wave_rows_culled_frac=culled,
                    kernel=plan.main_kernel_name, kernel_ms=ms_main,
                    note="flops and issue cycles count executed integrands only: rows whose exp(-p.u/T) is exactly +0 for a "
                         "whole wave are skipped (bitwise-identical result)")

    # untimed: the shader clock the main kernel runs at (idle probe waves on a private stream while one more step executes); rank 0, any N
    clock_ghz, clock_err = None, None
    if rank == 0 and not a.no_clock_probe:
        import threading
        box = {}

        def probe():
            try:
                time.sleep(float(np.mean(ms["prep"])) * 1e-3 + 0.002)   # let cf_prep pass: the probe should see cf_main only
                box["ghz"] = api.probe_shader_clock(min(0.25, 0.6 * np.mean(ms["main"]) * 1e-3), local)
            except Exception as e:   # diagnostic only
                box["err"] = str(e)

        th = threading.Thread(target=probe)
        plan.execute(n_loc, ptrs, out.data_ptr(), stream, want_status=False)   # enqueue, returns before the kernels finish
        th.start()
        th.join()
        torch.cuda.synchronize()
        clock_ghz, clock_err = box.get("ghz") or None, box.get("err")
    rv = valu_roofline(clock_ghz)
    ranks = None
    if multi:
        mine = dict(rank=rank, device=local, cells=n_loc, first_cell=lo,
                    kernel_ms=dict(prep=float(np.mean(ms["prep"])), main=ms_main, finalize=float(np.mean(ms["finalize"]))),
                    allreduce_ms=float(np.mean(ms["allreduce"])) if ms["allreduce"] else None,
                    comm_rank_seen=list(comm.rank_seen()) if comm is not None else None,
                    wave_rows_culled_frac=culled, integrands_executed=executed_evals, roofline_valu=rv)
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    # SURVEY.md 8d's t_kernel: the same steps with the upload of the cell arrays (every rank its shard) and the download of the (summed)
    # spectrum inside the timed region -- pinned host buffers, as a host that cares would hold them; barrier + MAX over ranks as for `value`
    used = [k for k in cell_fields if k in ptrs and (k != "eta" or wl["dimension"] == 3)]
    hpin = {k: torch.from_numpy(cells[k]).pin_memory() for k in used}
    hout = torch.empty(plan.output_size, dtype=torch.float64).pin_memory()

    def step_incl():
        for k in used:
            tens[k].copy_(hpin[k], non_blocking=True)
        step(False)
        hout.copy_(out, non_blocking=True)
        torch.cuda.synchronize()   # the spectrum is resident on the host

    step_incl()
    fence()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        step_incl()
    fence()
    incl = (time.perf_counter() - t1) / a.steps
    if multi:
        tt = torch.tensor([incl], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        incl = float(tt.item())
    del hpin, hout
    host_entry = None
    if world == 1:
        # and the one-shot host entry a maintainer would call from calculate_dN_pTdpTdphidy: pageable host arrays, plan creation and
        # workspace allocation included
        # (twice: on a box whose driver has never handed out a second workspace of this size the first call's hipMalloc alone took 0.57 s --
        # profiles/r04_bench.json of the closing run: 913 ms against 340 ms for the same call later on the same box)
        ms_calls = []
        for _ in range(2):
            t1 = time.perf_counter()
            if vah:
                _, st_host = api.smooth_spectra_vah(cells, sp, grid, opts, tab=df)
            else:
                _, st_host = api.smooth_spectra(cells, sp, grid, df, opts, fq=fq)
            ms_calls.append((time.perf_counter() - t1) * 1e3)
        host_entry = dict(ms=ms_calls[1], ms_first_call=ms_calls[0], ms_h2d=st_host["ms_h2d"], ms_d2h=st_host["ms_d2h"],
                          ms_kernels=st_host["ms_prep"] + st_host["ms_main"] + st_host["ms_finalize"],
                          note="%s, one call: plan creation + workspace hipMalloc + pageable H->D + kernels + D->H (the second of two calls; "
                               "ms_first_call: the first, with the driver's first allocation of a second workspace on this box)" % (
                              "is3d_smooth_spectra_vah_df" if vah else "is3d_smooth_spectra"))
    # untimed: the row / unit culling skips only work that cannot change a bit of the spectrum -- check it here against the same
    # kernels with culling off (zero_skip = 2) on the same resident surface
    cull_identical, ms_no_cull, surf_cull = None, None, None
    if world == 1 and not a.no_cull_check:
        plan2 = make_plan(dict(opts, zero_skip=2))
        plan2.set_timing(True)
        out2 = torch.zeros(plan2.output_size, dtype=torch.float64, device=dev)
        plan2.execute(n_loc, ptrs, out2.data_ptr(), stream, want_status=False)
        torch.cuda.synchronize()
        ms_no_cull = plan2.timings()["ms_main"]   # untimed w.r.t. the metric: the data-independent floor of the dominant kernel
        cull_identical = bool(torch.equal(out, out2))
        plan2.close()
        if not vah and wl["dimension"] == 3 and wl["df_mode"] in (1, 2) and a.zero_skip == 0:
            # untimed w.r.t. the metric: the opt-in surface-relative cull (zero_skip = 3; bounded, not bitwise) on the same surface
            plan3 = make_plan(dict(opts, zero_skip=3))
            plan3.set_timing(True)
            for _ in range(2):
                plan3.execute(n_loc, ptrs, out2.data_ptr(), stream, want_status=False)
            torch.cuda.synchronize()
            d = (out2 - out).abs() / out.abs().clamp_min(1e-300)
            surf_cull = dict(main_ms=plan3.timings()["ms_main"], max_rel_diff=float(d.max().item()), bins_that_differ=int((out2 != out).sum().item()))
            plan3.close()
        del out2
    spectrum_ok = bool(torch.isfinite(out).all().item())

    if rank == 0:
        nsp = len(sp["mass"])
        evals_step = float(n_total) * nbins * nsp
        value = evals_step * a.steps / elapsed
        # ---- roofline (contract form, HBM): algorithmic bytes of the launch = cell arrays read once + spectrum written once
        ncell_arrays = len(cell_fields) - (0 if wl["dimension"] == 3 else 1)   # 18 (17 in 2+1D: no eta; 23 / 22 with the baryon arrays); VAH: 24
        b_alg = 8.0 * (ncell_arrays * n_loc + nsp * nbins)
        tkey = wl["name"] + ("_feqmod%d" % a.df_mode if a.df_mode in (3, 4) else "_df%d" % a.df_mode if a.df_mode in (1, 2) and a.df_mode != workload(a.workload)["df_mode"] else "") \
            + ("_baryon" if wl.get("baryon") else "") + ("_dim2" if (vah and wl["dimension"] == 2) else "")
        traffic, traffic_source = (None, None) if a.variant else pmc_traffic(tkey, n_loc)
        roofline = dict(bound="hbm", achieved=b_alg / (ms_main * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=b_alg / (ms_main * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_source, kernel=plan.main_kernel_name,
                        kernel_ms=ms_main, algorithmic_bytes=b_alg,
                        note="rank 0's launch (its shard of %d cells); not the binding roofline: 1e6 flop/byte; see roofline_valu and DESIGN.md section 5" % n_loc)
        # integrands the kernels actually executed per second, summed over the ranks' own counters
        executed_all = sum(r["integrands_executed"] for r in ranks) if ranks else executed_evals
        executed_per_s = executed_all * a.steps / elapsed
        inner = len(grid["eta"]) if wl["dimension"] == 2 else 1   # 2+1D: an eval is a sum over the eta table; integrands are its terms
        res = dict(metric="FO-cell x momentum-bin x species evals/sec", value=value, unit="evals/s", n_gpus=world, steps=a.steps,
                   warmup=a.warmup, ms_per_step=elapsed / a.steps * 1e3, higher_is_better=True, scaling=a.scaling, vs_baseline=None,
                   dtype="f64", data="synthetic",
                   value_note="reference-equivalent evals (cells x bins x all %d species, every row) per second, cell arrays resident in HBM when the "
                              "timed region starts (the bench contract); SURVEY.md 8d's t_kernel also contains the H->D upload of the cell arrays and the "
                              "D->H download of the spectrum: value_incl_transfers / ms_per_step_incl_transfers.  The kernel executes %d species "
                              "classes and %.1f %% of their rows: executed_evals_per_s (2+1D: integrands = evals x %d eta nodes)" % (
                                  nsp, st["n_classes"], 100.0 * (1.0 - culled), inner),
                   executed_evals_per_s=executed_per_s, executed_fraction_of_value=executed_per_s / (value * inner),
                   transfers_included=False,
                   value_incl_transfers=evals_step / incl,
                   ms_per_step_incl_transfers=incl * 1e3,
                   host_entry=host_entry,
                   allreduce=allreduce_by,
                   allreduce_ms=(max(r["allreduce_ms"] for r in ranks) if ranks and all(r["allreduce_ms"] is not None for r in ranks) else None),
                   ranks_seen=(sorted(r["comm_rank_seen"][0] for r in ranks) if ranks and all(r["comm_rank_seen"] for r in ranks) else None),
                   ranks=ranks,
                   config=dict(workload=wl["text"], cells_total=n_total, cells_per_gpu=n_loc, species=nsp, species_classes_evaluated=st["n_classes"],
                               bins=nbins, evals_per_step=evals_step, kernel=plan.main_kernel_name, kernel_variant=st["kernel_variant"],
                               parallelism=("cell-axis shards x%d of %s, one all-reduce of the spectrum" % (
                                   world, "one surface (BASELINE config 4 at 8)" if a.scaling == "strong" else "a surface that grows with N")) if world > 1 else "1 GPU",
                               workspace_GB=plan.workspace_bytes / 1e9, workspace_bytes_per_cell=plan.workspace_bytes / float(max(n_loc, 1)),
                               spectrum_finite=spectrum_ok,
                               culled_rows_change_no_bit=cull_identical),
                   kernel_ms=dict(prep=float(np.mean(ms["prep"])), main=ms_main, finalize=float(np.mean(ms["finalize"])), main_no_cull=ms_no_cull, surface_cull=surf_cull),
                   roofline=roofline, roofline_valu=rv)
        if clock_err:
            res["clock_probe_error"] = clock_err
        if not a.no_cpu_baseline:
            # every N: the CPU path on this node's own host cores in the same run (north_star); rank 0 computes it after the timed region while
            # the other ranks wait at the closing barrier
            res["cpu_baseline"] = cpu_baseline(wl, sp, grid, df, seconds_budget=a.cpu_baseline_seconds, fq=fq)
            # like for like first: integrands the GPU executed against the evals the CPU port executed (it evaluates every species and every row);
            # gpu_over_cpu divides reference-equivalent evals (305 species, every row) by the same CPU rate -- neither is a measure of kernel quality
            res["gpu_over_cpu_executed"] = executed_per_s / inner / res["cpu_baseline"]["value"]
            res["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    plan.close()
    if comm is not None:
        comm.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def sampler_cpu_baseline(cells, sp, df, gla, opts, n_events, seed, seconds_budget):
    """The serial CPU restatement of sample_dN_pTdpTdphidy (the reference IS serial over cells, emissionfunction_sampling_kernels.cpp:878: five
    std:: engines consumed in cell order) on a bounded slice of the same surface, one core."""
    from oracle import oracle  # checker doubling as the reported CPU baseline
    n_all = len(cells["tau"])
    nc = min(2000, n_all)
    sub = {k: v[:nc] for k, v in cells.items()}
    t0 = time.time()
    oracle.sample_particles(sub, sp, df, gla, opts, n_events=n_events, seed=seed)
    t_probe = max(time.time() - t0, 1e-3)
    nc = int(max(nc, min(n_all, nc * 0.8 * seconds_budget / t_probe)))
    sub = {k: v[:nc] for k, v in cells.items()}
    t0 = time.time()
    ref, rst = oracle.sample_particles(sub, sp, df, gla, opts, n_events=n_events, seed=seed)
    dt = time.time() - t0
    return dict(value=nc * n_events / dt, unit="cell-events/s", cores=1, cores_available=host_cores(), cpu_model=host_cpu_model(), kind="port",
                particles_per_s=rst["n_kept"] / dt,
                sample="first %d cells of the workload surface x %d events, oracle sampler (serial over cells like the reference), %.1f s" % (nc, n_events, dt)), ref, nc


def bench_sampler(a, rank, world, local, dev, multi, json_fd):
    """--workload config5-sampler: the Monte Carlo particle-sampler leg of BASELINE config 5 (sample_dN_pTdpTdphidy,
    /root/reference/src/cpp/emissionfunction_sampling_kernels.cpp:833-1225) on the 1e6-cell surface: a step = `--events` events over the rank's
    shard through the device-resident is3d_sampler_plan (cell arrays and the particle buffer in HBM; densities + cell records, Poisson pass +
    compaction, count pass + scan, fill pass).  The shards need no collective (streams keyed by the global cell index)."""
    import torch
    from is3d_amd import api, inputs, synth
    from is3d_amd import dist as idist
    if multi:
        import torch.distributed as dist
    df_mode = a.df_mode or 2
    n_total = (a.cells or 1000000) * (world if a.scaling == "weak" else 1)
    lo, hi = idist.shard_bounds(n_total, rank, world)
    n_loc = hi - lo
    sp = inputs.species("urqmd")
    df = inputs.df_tables()
    cells = synth.synth_surface(n_loc, 3, first_cell=lo)
    T_avg = idist.surface_average_T_global(cells) if df_mode in (3, 4) else 0.15
    gla = inputs.feqmod_tables(T_avg)
    opts = dict(dimension=3, df_mode=df_mode, device=local)
    seed = 20260002
    fq = gla if df_mode in (3, 4) else None
    tens = {k: torch.from_numpy(cells[k]).to(dev) for k in list(synth.CELL_FIELDS) + ["x", "y"]}     # resident in HBM before timing
    ptrs = {k: v.data_ptr() for k, v in tens.items()}
    xy = dict(x_ptr=ptrs["x"], y_ptr=ptrs["y"])
    plan = api.SamplerPlan(sp, df, gla, opts, max_cells=max(n_loc, 1), fq=fq)
    count, st0 = plan.execute(n_loc, ptrs, a.events, seed, first_cell=lo, **xy)    # count-only: sizes the particle buffer (and the workspaces)
    buf = torch.zeros(max(count, 1) * api.PARTICLE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    keys = ("ms_prep", "ms_density", "ms_count", "ms_poisson", "ms_fill")
    ms = {k: [] for k in keys}
    last = {}

    def step(record):
        n, st = plan.execute(n_loc, ptrs, a.events, seed, particles_ptr=buf.data_ptr(), capacity=count, first_cell=lo, **xy)
        assert n == count
        if record:
            for k in keys:
                ms[k].append(st[k])
            last.update(st)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    p0, a0 = api.resource_counters()
    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    allocs_in_steps = api.resource_counters()[1] - a0
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # the one-shot host entry (pageable arrays in, particle list out, plan + workspaces created inside): what operation = 2 of the driver calls
    host_entry = None
    if world == 1:
        t1 = time.perf_counter()
        plist, sth = api.sample_particles(cells, sp, df, gla, opts, n_events=a.events, seed=seed, fq=fq)
        host_entry = dict(ms=(time.perf_counter() - t1) * 1e3, ms_h2d=sth["ms_h2d"], ms_device=sth["ms_prep"] + sth["ms_count"] + sth["ms_fill"],
                          particles=int(sth["n_particles"]),
                          note="is3d_sample_particles twice (count-only, then fill): plan + workspaces + pageable H->D + kernels + D->H of the list")
        got = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=api.PARTICLE_DTYPE)[:count]
        same_as_host_entry = bool(len(plist) == count and all(np.array_equal(got[f], plist[f]) for f in got.dtype.names))
    mine = dict(rank=rank, device=local, cells=n_loc, first_cell=lo, particles=count, hadrons_drawn=int(last.get("n_hadrons_drawn", 0)),
                kernel_ms={k[3:]: float(np.mean(v)) for k, v in ms.items()})
    ranks = None
    if multi:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    if rank != 0:
        plan.close()
        return
    particles_all = sum(r["particles"] for r in ranks) if ranks else count
    dev_ms = float(np.mean(ms["ms_prep"]) + np.mean(ms["ms_count"]) + np.mean(ms["ms_fill"]))
    kms = {k[3:]: float(np.mean(v)) for k, v in ms.items()}
    kms["cells_records"] = kms["prep"] - kms["density"]
    kms["count_and_scan"] = kms["count"] - kms["poisson"]
    value = float(n_total) * a.events * a.steps / elapsed
    n_used_arrays = 18
    b_alg = 8.0 * n_used_arrays * n_loc + 96.0 * count
    # fp64-VALU view of the Gauss-Laguerre density kernel (flops per node from the emitted ISA, tools/count_isa.py)
    rv = None
    icp = os.path.join(ROOT, "is3d_amd", "csrc", "isa_counts.json")
    ic = json.load(open(icp)).get("cf_sampler_density") if os.path.exists(icp) else None
    if ic and kms["density"] > 0:
        nodes = float(n_loc) * last["n_classes"] * len(gla["root1"])
        tf = nodes * ic["flop_per_node"] / (kms["density"] * 1e-3) / 1e12
        rv = dict(bound="fp64_valu", kernel="cf_sampler_density", kernel_ms=kms["density"], achieved=tf, peak=FP64_VALU_PEAK_TF, unit="TFLOP/s",
                  frac=tf / FP64_VALU_PEAK_TF, quadrature_nodes_per_launch=nodes, flop_per_node=ic["flop_per_node"],
                  fp64_valu_instr_per_node=ic["valu_f64_instr_per_node"])
    div = None
    import glob
    pps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_sampler.json")))   # newest round's counters
    if pps:
        div = json.load(open(pps[-1]))
        div.setdefault("file", "profiles/" + os.path.basename(pps[-1]))
    res = dict(metric="FO-cell x event samples/sec (Monte Carlo particle sampler, the second leg of BASELINE config 5)", value=value, unit="cell-events/s",
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=elapsed / a.steps * 1e3, higher_is_better=True, scaling=a.scaling, vs_baseline=None,
               dtype="f64", data="synthetic",
               particles_per_s=float(particles_all) * a.steps / elapsed, particles_per_step=particles_all,
               hadrons_drawn_per_step=int(last["n_hadrons_drawn"]) if not ranks else sum(r["hadrons_drawn"] for r in ranks),
               momentum_sampling_efficiency=last["n_acceptances"] / max(last["n_momentum_samples"], 1),
               device_ms_per_step=dev_ms, device_allocations_during_timed_steps=allocs_in_steps,
               value_note="(cell, event) pairs sampled per second, wall clock of the steps; cell arrays and the particle buffer resident in HBM; a step also "
                          "reads back two counters per event batch (the number of emitting pairs, the batch's particle count), nothing else crosses PCIe",
               host_entry=host_entry, same_list_as_host_entry=(same_as_host_entry if world == 1 else None), ranks=ranks,
               config=dict(workload="BASELINE config 5, sampler leg: 1e6-cell synthetic 3+1D surface (seed 20260002), %s, 305-species pdg-urqmd_v3.3+ list, "
                                    "%d events per step, regular (not fast) mode; the reference's VAH sampler is an empty stub "
                                    "(emissionfunction_sampling_kernels.cpp:1231-1239), so the viscous-hydro sampler is what exists to be measured" % (
                                        {1: "14-moment delta-f", 2: "Chapman-Enskog delta-f", 3: "modified equilibrium (Mike)", 4: "modified equilibrium (Jonah)"}[df_mode], a.events),
                           cells_total=n_total, cells_per_gpu=n_loc, species=len(sp["mass"]), species_classes_evaluated=last["n_classes"], events=a.events,
                           df_mode=df_mode, parallelism=("cell-axis shards x%d, no collective" % world) if world > 1 else "1 GPU"),
               kernel_ms=kms,
               roofline=dict(bound="hbm", achieved=b_alg / (dev_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=b_alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             traffic=(div or {}).get("hbm_bytes_per_step"), traffic_source=(div or {}).get("source"),
                             kernel="the sampler's kernels of one step (density, cell records, Poisson, compaction, count, scan, fill)", kernel_ms=dev_ms,
                             algorithmic_bytes=b_alg,
                             note="algorithmic bytes = 18 cell arrays read once + 96 B per kept particle written; not the binding bound: the time goes to "
                                  "fp64 quadrature (density), Philox + rejection loops (count / fill, divergent) -- roofline_valu, divergence"),
               roofline_valu=rv, divergence=div)
    if not a.no_cpu_baseline:
        cb, ref, nc = sampler_cpu_baseline(cells, sp, df, gla, dict(dimension=3, df_mode=df_mode), a.events, seed, a.cpu_baseline_seconds)
        res["cpu_baseline"] = cb
        res["gpu_over_cpu"] = value / cb["value"]
        if world == 1:   # and the checker's list on that slice is the device's list (same hadrons, same order)
            got = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=api.PARTICLE_DTYPE)[:count]
            sel = got["cell"] < nc
            res["cpu_baseline"]["same_list_on_the_slice"] = bool(int(sel.sum()) == len(ref["E"]) and np.array_equal(got["species"][sel], ref["species"]))
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(res) + "\n").encode())
    plan.close()



if __name__ == "__main__":
    main()
