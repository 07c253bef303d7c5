"""one north-star (NS) scene through the hot path, stage by stage, with HIP-event times and the sizes that matter"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "NS"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES[wl]
t0 = time.time()
sc = synth.make_scene(wl, seed=0, device=dev)
feat = sc["features"][:, 0]
proj, tsdf = sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
torch.cuda.synchronize()
print(f"scene generated in {time.time() - t0:.1f}s; features {feat.numel() * 4 / 1e9:.2f} GB", flush=True)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
for i in range(reps):
    t0 = time.time()
    out = pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, timing=True)
    torch.cuda.synchronize()
    print(f"rep {i}: wall {1e3 * (time.time() - t0):.1f} ms; stage_ms", {k: round(v, 3) for k, v in out["stage_ms"].items()}, flush=True)
print({k: out[k] for k in ("M", "M_selected", "M_unique", "level_rows", "head_rows")})
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
