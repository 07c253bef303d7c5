"""A/B of the march kernels (per-step sigmoid vs table-driven) at a workload shape: HIP-event times, interleaved"""
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
res = {}
for rep in range(4):
    for mode in (False, True):          # (the image-row variant of the table kernel was removed with its environment switch)
        rma.SIGMOID_TABLE = bool(mode)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = m.march(); b.record(); torch.cuda.synchronize()
        res.setdefault(mode, []).append(a.elapsed_time(b))
        res[("out", mode)] = out
print(wl, "per-step sigmoid kernel ms:", [round(x, 3) for x in res[False]])
print(wl, "table kernel, 8x8-pixel tiles per wave (+table build) ms:", [round(x, 3) for x in res[True]])
c0, w0, k0, _ = res[("out", False)]
for mode in (True,):
    c1, w1, k1, _ = res[("out", mode)]
    live = torch.arange(k0.shape[1], device=dev)[None, :] < c0[:, None].clamp(max=k0.shape[1])
    print(mode, "counts equal", torch.equal(c0, c1), "wsum equal", torch.equal(w0, w1), "records equal", bool((k0[live] == k1[live]).all()),
          "rays", c0.numel(), "kept", int(c0.sum()))
