"""scripts/dense_l2sim.cpp -- the replay simulation DESIGN.md's bound for the dense kernel rests on -- builds and gives the
qualitative result it is quoted for (4 of the 40 views, one sweep: seconds on the CPU): the brick-ordered traversal
beats the z-fastest order, and an XCD working on one 32^3 brick in lockstep beats both."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hit(out):
    m = re.search(r"L2 hit ([0-9.]+)%\s+fabric reads ([0-9.]+) GB", out)
    assert m, out
    return float(m.group(1)), float(m.group(2))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_l2_replay_simulation_builds_and_ranks_the_schedules(tmp_path):
    exe = str(tmp_path / "l2sim")
    subprocess.run(["g++", "-O2", "-fopenmp", "-o", exe, os.path.join(ROOT, "scripts", "dense_l2sim.cpp")], check=True)
    res = {}
    for name, args in (("linear", ["linear"]), ("brick", ["brick", "R=128"]),
                       ("lockstep", ["brick", "R=128", "sx=32", "sy=32", "sz=32", "lockstep=1"])):
        out = subprocess.run([exe] + args + ["V=4"], check=True, capture_output=True, text=True, timeout=600).stdout
        res[name] = _hit(out)
    assert res["linear"][0] < res["brick"][0] < res["lockstep"][0], res
    assert res["lockstep"][1] < res["brick"][1] < res["linear"][1], res
