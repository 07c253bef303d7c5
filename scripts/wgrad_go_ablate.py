"""where the time of the gather-once weight gradient goes: the kernel on a synthetic compact surface (Morton rows), whole and
with phases switched off (cnrma_debug_conv_tuning ablate bits 256 no compute, 512 no LDS stores, 1024 no row loads, 2048 LDS
reads without MFMAs -- timing only, the results are wrong on purpose)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cnrma_amd import sparse as S
from cnrma_amd.sparse import call, ptr, stream

dev = torch.device("cuda:0")
n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rng = np.random.RandomState(0)
xy = rng.uniform(0, 12.0, size=(n_pts, 2))
z = 1.5 + 0.8 * np.sin(xy[:, 0]) * np.cos(0.7 * xy[:, 1]) + rng.uniform(-0.06, 0.06, size=n_pts)
pts = torch.from_numpy(np.concatenate((xy, z[:, None]), axis=1).astype(np.float32)).to(dev)
x, _ = S.voxelize(pts, torch.zeros(n_pts, 1, device=dev), 0.04)
cs = x.cs
n = cs.n
F = torch.randn(n, Cin, device=dev)
G = torch.randn(n, Cout, device=dev)
nbr = cs.neighbours(cs, 3, cs.stride)
tu = S.tile_union(cs, cs, 3, cs.stride)
per = 2 * ((Cin + 63) // 64) * ((Cout + 63) // 64)
parts = max(1, min((n + 63) // 64, S.WGRAD_GO_BLOCKS // per))
slabs = torch.empty((parts, 27, Cin, Cout), device=dev)
print(f"{n} rows, {Cin} -> {Cout}, {parts} parts, neighbours per row {(nbr >= 0).float().sum(1).mean().item():.1f}")


def run(ablate, reps=20):
    S.conv_tuning(ablate=ablate)
    for _ in range(3):
        call("cnrma_sparse_conv_wgrad_go_bf16", ptr(F), Cin, ptr(tu), ptr(G), Cout, n, None, parts, ptr(slabs), stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call("cnrma_sparse_conv_wgrad_go_bf16", ptr(F), Cin, ptr(tu), ptr(G), Cout, n, None, parts, ptr(slabs), stream())
    e1.record(); e1.synchronize()
    S.conv_tuning()
    return e0.elapsed_time(e1) / reps * 1e3


for name, ab in (("whole", 0), ("no compute", 256), ("LDS reads, no MFMAs", 2048), ("no LDS stores", 512), ("no row loads", 1024),
                 ("no row loads, no stores", 1536), ("no compute, no row loads, no stores", 256 + 1536)):
    print(f"{name:40s} {run(ab):8.1f} us")
