"""tile-union builder alone: kernel time per coordinate set of a scene (20 launches back to back), with the builder's phases
switched off one at a time (cnrma_debug_conv_tuning ablate bits 32 / 64 / 128: insertion, compaction, ranking)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, synth, _lib
from cnrma_amd import sparse as S

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "S"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
sets = []
orig = S.tile_union


def rec(in_cs, out_set, ks, st):
    if all(in_cs is not s_ for s_ in sets):
        sets.append(in_cs)
    return orig(in_cs, out_set, ks, st)


S.tile_union = rec
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
S.tile_union = orig
for cs in sets:
    nbr = cs.neighbours(cs, 3, cs.stride)
    tu = torch.empty(_lib.load().cnrma_sparse_tile_union_bytes(cs.n), dtype=torch.uint8, device=dev)
    res = []
    for abl in (0, 32, 64, 128, 32 | 64 | 128):
        S.conv_tuning(ablate=abl)
        for _ in range(3):
            _lib.call("cnrma_sparse_tile_union_build", _lib.ptr(nbr), cs.n, None, 27, _lib.ptr(tu), S.stream())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            _lib.call("cnrma_sparse_tile_union_build", _lib.ptr(nbr), cs.n, None, 27, _lib.ptr(tu), S.stream())
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20 * 1e3)
    S.conv_tuning()
    _lib.call("cnrma_sparse_tile_union_build", _lib.ptr(nbr), cs.n, None, 27, _lib.ptr(tu), S.stream())
    n_t = (cs.n + 63) // 64
    hdr = tu[:n_t * 84 * 4].view(torch.int32).view(n_t, 84)
    g = hdr[:, 0].float()
    un = hdr[:, 3].float()
    print(f"rows={cs.n:7d} stride={cs.stride:3d} tiles={n_t:5d} groups mean {float(g.mean()):.2f} max {int(g.max())} union(first group) mean {float(un.mean()):.0f} | "
          f"full {res[0]:6.1f} us  no-insert {res[1]:6.1f}  no-compaction {res[2]:6.1f}  no-rank {res[3]:6.1f}  none {res[4]:6.1f}", flush=True)
