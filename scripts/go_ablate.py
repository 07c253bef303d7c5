"""gather-once kernel on the big layers with parts switched off (cnrma_debug_conv_tuning ablate bits: 1 MFMAs + fragment reads,
2 union-row loads, 4 weight loads, 8 LDS stores of the union, 16 epilogue stores, 32 local-index load)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, synth
from cnrma_amd import sparse as S

REPS = 20
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "S"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
calls, seen = [], set()
orig_conv = S.conv


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    key = (x.cs.n, x.F.shape[1], y.F.shape[1])
    if kernel_size == 3 and stride == 1 and x.F.shape[1] % 32 == 0 and y.F.shape[1] >= 64 and x.cs.compact and x.cs.n >= 30000 and key not in seen:
        seen.add(key)
        calls.append(dict(x=x, weight=weight, scale=scale, shift=shift, residual=residual, act=act))
    return y


S.conv = rec_conv
sys.modules["cnrma_amd.nn"].S.conv = rec_conv
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
S.conv = orig_conv
sys.modules["cnrma_amd.nn"].S.conv = orig_conv
del feat


def timed(c):
    def run():
        return orig_conv(c["x"], c["weight"], 3, 1, c["scale"], c["shift"], c["residual"], c["act"])
    run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3


MASKS = [(0, "full"), (1, "-mfma"), (4, "-W"), (5, "-mfma-W"), (7, "-mfma-W-Aload"), (15, "-mfma-W-Aload-lds"), (31, "..-epilogue"),
         (63, "..-lidx"), (2, "-Aload"), (16, "-epilogue"), (8, "-lds")]
for c in calls:
    x = c["x"]
    res = []
    for m, name in MASKS:
        S.conv_tuning(ablate=m)
        res.append(f"{name} {timed(c):.0f}")
    S.conv_tuning()
    print(f"rows={x.cs.n:7d} Cin={x.F.shape[1]:4d} Cout={c['weight'].shape[-1]:4d} | " + "  ".join(res), flush=True)
