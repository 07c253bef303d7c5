"""Executed MFMA instructions per 3x3x3 stride-1 convolution, next to the algorithmic count (VERDICT round 5, next #1c).
Run under rocprofv3 with a counter pass (scripts/profile_mfma.sh): one warm eager scene, then one recorded eager scene whose
gather-once convolutions are listed, in launch order, in <out.json> (rows, channels, pairs, what a 64-row-tile offset mask
executes).  scripts/mfma_join.py joins that list with the per-dispatch counters.
    python scripts/mfma_count.py S|NS out.json [f16x3|f32]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import _lib, pipeline, synth
from cnrma_amd import sparse as S

dev = torch.device("cuda:0")
wl, out_path = sys.argv[1], sys.argv[2]
S.CONV_PRECISION = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev, channels_last=True)
feat, proj, tsdf = sc["features"][:, 0], sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
layers = []
orig_conv, orig_call = S.conv, _lib.call
state = {"names": None}


def call(name, *a):
    if state["names"] is not None and name in bench.CONV_CALLS:
        state["names"].append(name)
    return orig_call(name, *a)


def conv(x, weight, kernel_size=3, stride=1, *a, **k):
    state["names"] = []
    y = orig_conv(x, weight, kernel_size, stride, *a, **k)
    names, state["names"] = state["names"], None
    K = kernel_size ** 3
    rec = dict(K=K, stride=stride, Cin=x.F.shape[1], Cout=y.F.shape[1], n_out=y.cs.n, entry=names[0] if names else None)
    if K == 27 and stride == 1:
        v = x.cs.neighbours(y.cs, 3, x.cs.stride)[:y.cs.n] >= 0
        n = y.cs.n
        nt = (n + 63) // 64
        v64 = torch.cat((v, torch.zeros((nt * 64 - n, 27), dtype=torch.bool, device=dev))).view(nt, 64, 27)
        rec.update(pairs=int(v.sum()), tile_offsets=int(v64.any(dim=1).sum()), tiles=nt)
    layers.append(rec)
    return y


with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
    torch.cuda.synchronize()
    for mod in ("cnrma_amd._lib", "cnrma_amd.rma", "cnrma_amd.sparse"):
        sys.modules[mod].call = call
    S.conv = conv
    sys.modules["cnrma_amd.nn"].S.conv = conv
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
    torch.cuda.synchronize()
with open(out_path, "w") as f:
    json.dump(dict(workload=wl, precision=S.CONV_PRECISION, layers=layers), f, indent=1)
print(f"{len(layers)} convolutions, {sum(1 for L in layers if L['entry'] and '_go_' in L['entry'])} on the gather-once kernels")
