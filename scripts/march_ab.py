"""A/B of the march kernels (per-step sigmoid vs table-driven, free-space skipping off / on) at a workload shape: HIP-event
times (table builds included), interleaved; counts / sums / kept-sample records compared bit for bit"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnrma_amd import rma, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "NS"
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene((V, 8, H, W, dims, stride), seed=0, boxes=3)
feat = rma.to_nhwc(sc["features"][:, 0].to(dev))
pinv = rma.projection_inverse(sc["projection"][:, 0], stride).to(dev)
tsdf = sc["tsdf"][0, 0].to(dev)
m = rma._March(feat, pinv, tsdf, dims, 0.04, (0, 0, 0), 300, 0.05, "neus", 0)
MODES = {"per-step sigmoid": (False, False), "table": (True, False), "table + free-space skip": (True, True)}
res = {}
for rep in range(4):
    for name, (table, skip) in MODES.items():
        rma.SIGMOID_TABLE, rma.MARCH_SKIP = table, skip
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = m.march(); b.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(a.elapsed_time(b))
        res[("out", name)] = out
for name in MODES:
    print(wl, f"{name:26s} ms:", [round(x, 3) for x in res[name]])
c0, w0, k0, _ = res[("out", "per-step sigmoid")]
for name in list(MODES)[1:]:
    c1, w1, k1, _ = res[("out", name)]
    live = torch.arange(k0.shape[1], device=dev)[None, :] < c0[:, None].clamp(max=k0.shape[1])
    print(name, ": counts equal", torch.equal(c0, c1), "wsum equal", torch.equal(w0, w1), "records equal", bool((k0[live] == k1[live]).all()),
          "rays", c0.numel(), "kept", int(c0.sum()))
if getattr(m, "_skip", None) is not None:
    nb = 1
    for d in dims:
        nb *= -(-d // 4)
    r = m._skip[:nb]
    print("skip radii (blocks of 4^3 voxels):", {int(v): int((r == v).sum()) for v in (0, 4, 8, 12, 16)})
