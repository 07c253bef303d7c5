"""per-layer convolution times of one eager scene (bench.KernelProfile: HIP events around every C-ABI call) for a workload and a
convolution precision:  python scripts/conv_layers.py NS f32"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, synth
from cnrma_amd import sparse as S

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "S"
S.CONV_PRECISION = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev, channels_last=True)
feat, proj, tsdf = sc["features"][:, 0], sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
    prof = bench.KernelProfile()
    prof.install()
    try:
        for _ in range(2):
            pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
    finally:
        prof.uninstall()
kern, layers = prof.summary(2)
tot = 0.0
for i, L in enumerate(layers):
    fl = 2.0 * L["pairs"] * L["Cin"] * L["Cout"]
    tot += L["ms"]
    print(f'{i:2d} K={L["K"]:2d} {L["Cin"]:4d}->{L["Cout"]:4d} rows={L["n_out"]:7d} pairs={L["pairs"]:9d} {L["ms"] * 1e3:8.1f} us {fl / L["ms"] / 1e9:7.1f} TF/s')
print(f"{wl} {S.CONV_PRECISION}: convolutions {tot:.3f} ms per scene; entry points:",
      {k: (round(v["ms_per_scene"], 3), v["launches_per_scene"]) for k, v in kern.items() if "conv" in k})
