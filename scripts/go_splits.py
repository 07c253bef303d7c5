"""gather-once kernel: forced channel-slice splits per layer (launch + reduce timed together)"""
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
S.GO_WS_ROWS = 1 << 30


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    key = (x.cs.n, x.F.shape[1], y.F.shape[1], residual is not None)
    if kernel_size == 3 and stride == 1 and x.F.shape[1] % 32 == 0 and y.F.shape[1] >= 64 and x.cs.compact and key not in seen:
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


for c in calls:
    x = c["x"]
    ns = x.F.shape[1] // 32
    res = []
    for sp in (-1, 1, 2, 4, 8, 16):
        if sp > ns:
            continue
        S.conv_tuning(splits=sp)
        res.append(f"splits {sp if sp > 0 else 'auto'}: {timed(c):6.1f}")
    S.conv_tuning()
    tiles = (x.cs.n + 63) // 64
    print(f"rows={x.cs.n:7d} tiles={tiles:5d} Cin={x.F.shape[1]:4d} Cout={c['weight'].shape[-1]:4d} res={int(c['residual'] is not None)} | " + "  ".join(res), flush=True)
