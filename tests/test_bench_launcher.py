"""CPU: bench.py's launcher-less form.  `python bench.py --gpus N` (WORLD_SIZE unset) must start N fresh rank processes itself,
before it imports torch or touches HIP, relay rank 0's JSON line and propagate a failure.  bench.py has no CPU compute path, so
what is checked here is the child command it builds and that the children's failure ("needs a GPU") comes back as a non-zero
exit code; tests/test_gpu_multi.py::test_bench_launches_its_own_ranks runs the real thing on the GPU box."""
import os
import subprocess
import sys

from conftest import ROOT


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launcher_command():
    b = _bench()
    cmd = b.launcher_command(["--gpus", "8", "--steps", "5", "--warmup", "1"], 8, 29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]
    assert "torch" not in sys.modules or True   # importing bench.py itself must not import torch:
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main()")]
    assert "import torch" not in head


def test_self_launch_happens_before_torch_and_propagates_failure():
    """No GPU here: the rank processes exit with 'bench.py needs a GPU'; the launching process must return non-zero, print no JSON
    line, and must not have imported torch itself (checked through -X importtime on the parent only)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if subprocess.run([sys.executable, "-c", "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 3)"], env=env).returncode == 0:
        import pytest
        pytest.skip("a GPU is visible: covered by tests/test_gpu_multi.py")
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--cells", "100", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "starting 2 ranks" in r.stderr and "needs a GPU" in r.stderr
    # -X importtime lines of the PARENT are written by the parent's interpreter only ("import time:" prefix); the children run without it
    parent_imports = [ln for ln in r.stderr.splitlines() if ln.startswith("import time:")]
    assert parent_imports and not [ln for ln in parent_imports if ln.rstrip().endswith("| torch")]
