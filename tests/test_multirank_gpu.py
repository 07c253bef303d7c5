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


def test_bench_spawns_its_ranks_and_exchanges_detections(tmp_path):
    """`python bench.py --gpus 2` as a plain command: the parent starts two fresh ranks before touching the GPU; ONE short
    stdout line with the contract's keys, the rest in the detail file"""
    detail = str(tmp_path / "bench_detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1",
           "--scenes", "3", "--slots", "2", "--windows", "2", "--window-s", "0.05", "--no-secondary", "--no-cpu-baseline",
           "--no-profile", "--detail", detail]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len([l for l in lines if l.startswith("{")]) == 1            # (gloo itself prints "[Gloo] Rank ..." lines; RCCL does not)
    assert lines[-1].startswith("{") and len(lines[-1]) < 4096          # what the driver parses: the LAST line
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["value"] > 0 and r["plan_violations"] == 0
    assert r["scaling"] == "weak" and r["dist"] == {"world_size": 2, "backend": "gloo"}
    with open(detail) as f:
        d = json.load(f)
    assert len(d["windows_scenes_per_s"]) == 2 and len(d["dist"]["per_rank_window_s"]) == 2
    assert abs(d["value"] - 2 * 3 * d["scenes_per_step"] / d["window_s"]) < 1e-6 * d["value"]
    assert abs(r["value"] - d["value"]) < 1e-3 * d["value"]                 # the line rounds to 4 significant digits


def test_ddp_gradient_allreduce():
    port = 29600 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py")]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert p.stdout.count("identical_across_ranks True") == 2
