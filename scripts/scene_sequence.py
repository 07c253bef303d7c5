"""the launch sequence of ONE replayed scene graph, from a rocprofv3 --kernel-trace run of bench.py with --slots 1:
    python scripts/scene_sequence.py <rocprof output dir> <nodes per scene>
prints the last scene's kernels in start order with their durations and the gap to the previous kernel's end"""
import csv
import glob
import os
import re
import sys
src, n = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a scene graph's first kernels are the march tables / the march; its dense kernel is unique per scene: take the last complete scene =
# the n dispatches that END with the last select_decode / topk of a replay, found from the last dense kernel backwards
dense = [i for i, r in enumerate(rows) if "backproject_accum" in r[2]]
assert dense, "no dense kernel in the trace"
# the dense kernel sits at a fixed position inside a scene's sequence: align on the previous scene's dense kernel
period = dense[-1] - dense[-2] if len(dense) > 1 else n
start = dense[-1] - (dense[-1] - dense[-2]) if len(dense) > 1 else max(0, dense[-1] - 8)
# scene k spans [first kernel after scene k-1's last, ...]: print one period that starts right behind the previous dense kernel's scene,
# i.e. the `period` dispatches ending where the last scene ends (dense[-1] + the same tail length as before)
tail = period - 1
last = rows[dense[-2] + 1: dense[-1] + 1] if len(dense) > 1 else rows[-n:]
print(f"# {len(last)} dispatches from behind the dense kernel of one scene to the dense kernel of the next (a cyclic cut of one scene's "
      f"{period} launches; graph nodes reported by bench.py: {n})")
prev = None
tot = gaps = 0.0
for s, e, name in last:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    short = re.match(r"[\w:]+(<[^(]*>)?", name)
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {(short.group(0) if short else name)[:110]}")
    tot += (e - s) / 1e3
    gaps += max(gap, 0.0)
    prev = e
print(f"{n} kernels: {tot / 1e3:.3f} ms of kernels + {gaps / 1e3:.3f} ms of gaps = {(last[-1][1] - last[0][0]) / 1e6:.3f} ms span")
