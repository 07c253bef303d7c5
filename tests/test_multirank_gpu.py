"""N > 1 ranks on the 1-GPU box (all ranks on device 0, gloo instead of RCCL): bench.py's self-spawn + N-rank code path
with its per-step exchange, and a DDP training step of the sparse detector."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, CNRMA_BENCH_BACKEND="gloo", CNRMA_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    return env


def test_bench_spawns_its_ranks_and_exchanges_detections():
    """`python bench.py --gpus 2` as a plain command: the parent starts two fresh ranks before touching the GPU"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1",
           "--scenes", "3", "--slots", "2", "--windows", "2", "--window-s", "0.05", "--no-secondary", "--no-cpu-baseline",
           "--no-profile"]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["value"] > 0 and r["plan_violations"] == 0
    assert r["scaling"] == "weak" and len(r["windows_scenes_per_s"]) == 2
    assert abs(r["value"] - 2 * 3 * r["scenes_per_step"] / r["window_s"]) < 1e-6 * r["value"]


def test_ddp_gradient_allreduce():
    port = 29600 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py")]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert p.stdout.count("identical_across_ranks True") == 2
