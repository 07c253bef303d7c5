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


def test_eight_rank_rehearsal_gathers_what_eight_single_ranks_compute(tmp_path):
    """BASELINE configs[3] rehearsed on what exists: `bench.py --gpus 8` on ONE device over gloo --
    rank r works on the scenes of seed block r -- against eight single-process runs of the same blocks: the gathered
    [8, S, K, W] block of the last step equals the eight single-rank results row for row (bit for bit: the same graphs, the
    same seeds), and the line's `dist` block reports world size 8.  No scaling number: all ranks share one GPU."""
    import numpy as np
    common = ["--workload", "tiny", "--steps", "2", "--warmup", "1", "--scenes", "2", "--slots", "2", "--windows", "1",
              "--scenes-per-step", "2", "--no-secondary", "--no-cpu-baseline", "--no-profile"]
    multi = tmp_path / "multi"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dump-exchange", str(multi),
           "--detail", str(tmp_path / "detail8.json")] + common
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 8 and r["dist"] == {"world_size": 8, "backend": "gloo"} and r["plan_violations"] == 0
    g = np.load(multi / "gathered.npz")
    det_all, valid_all = g["det_all"], g["valid_all"]
    assert det_all.shape[0] == 8 and det_all.shape[1] == 2 and valid_all.shape[:2] == (8, 2)
    single = tmp_path / "single"
    procs = []
    for rk in range(8):                                       # four at a time: each is a fresh process with its own graphs
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--as-rank", str(rk), "--dump-exchange", str(single),
               "--detail", str(tmp_path / f"detail1_{rk}.json")] + common
        procs.append(subprocess.Popen(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
        if len(procs) == 4 or rk == 7:
            for q in procs:
                out, err = q.communicate(timeout=900)
                assert q.returncode == 0, err[-3000:]
            procs = []
    some = 0
    for rk in range(8):
        own = np.load(multi / f"rank{rk}_of8.npz")             # what rank rk itself held when the exchange ran
        one = np.load(single / f"rank{rk}_of1.npz")            # the same scenes in a process that knows no other rank
        # (bit patterns: rows behind the live ones are whatever the static buffers held, NaN patterns included)
        assert np.array_equal(det_all[rk].view(np.int32), own["det"].view(np.int32)) and np.array_equal(valid_all[rk], own["valid"])
        assert np.array_equal(valid_all[rk], one["valid"]) and np.array_equal(own["sizes"], one["sizes"]), rk
        for sc in range(det_all.shape[1]):                     # live rows of every level block (rows behind them are padding)
            r0 = 0
            for k, v in zip(one["sizes"], valid_all[rk, sc]):
                assert np.array_equal(det_all[rk, sc, r0:r0 + v].view(np.int32), one["det"][sc, r0:r0 + v].view(np.int32)), (rk, sc)
                r0 += int(k)
        some += int(valid_all[rk].sum())
    assert some > 0                                           # the blocks hold detections, not padding only
    assert len({det_all[rk].tobytes() for rk in range(8)}) == 8      # eight different scene blocks


def test_ddp_gradient_allreduce():
    port = 29600 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py")]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert p.stdout.count("identical_across_ranks True") == 2
