"""A/B of sparse-conv kernel variants on one scene (per-layer HIP-event times): CNRMA_CONV_DB=0/1 -- diagnostics"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, synth
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "S"
var = sys.argv[2] if len(sys.argv) > 2 else "CNRMA_CONV_DB"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, device=dev)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
res = {}
modes = ("0", "1") if len(sys.argv) <= 3 else tuple(sys.argv[3:])
for rep in range(3):
    for mode in modes:
        os.environ[var] = mode
        if var == "PAIR_CONV":
            from cnrma_amd import sparse as S
            S.PAIR_CONV = mode == "1"
        out = pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf)
        prof = bench.KernelProfile(); prof.install()
        try:
            out = pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf)
        finally:
            prof.uninstall()
        kern, layers = prof.summary(1)
        res.setdefault(mode, []).append(layers)
        res[("out", mode)] = out
for mode in modes:
    print(var, mode, "conv total ms per rep:", [round(sum(L["ms"] for L in layers), 3) for layers in res[mode]])
a, b = res[modes[0]][-1], res[modes[-1]][-1]
for La, Lb in zip(a, b):
    if La["ms"] > 0.05:
        print(f"rows={La['n_out']:7d} Cin={La['Cin']:4d} Cout={La['Cout']:4d} K={La['K']:2d}  {La['ms']:.4f} -> {Lb['ms']:.4f} ms  ({Lb['ms'] / La['ms']:.2f}x)")
oa, ob = res[("out", modes[0])], res[("out", modes[-1])]
print("detections bit-identical:", torch.equal(oa["bboxes"], ob["bboxes"]) and torch.equal(oa["scores"], ob["scores"]))
