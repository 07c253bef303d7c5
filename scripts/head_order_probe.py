"""What would Morton row order on the head levels buy?  Every 3x3x3 stride-1 convolution of a scene is replayed on its own
tensor (row order as the network produces it: child-slice order after the generative transpose) and on the SAME tensor with
the rows sorted by the Morton code of their coordinates; stage kernel and gather-once kernel each."""
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
calls = []
orig_conv = S.conv


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    if kernel_size == 3 and stride == 1 and x.F.shape[1] % 32 == 0 and y.F.shape[1] >= 64 and y.cs.n >= 30000:
        calls.append(dict(x=x, weight=weight, scale=scale, shift=shift, act=act))
    return y


S.conv = rec_conv
sys.modules["cnrma_amd.nn"].S.conv = rec_conv
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
S.conv = orig_conv
sys.modules["cnrma_amd.nn"].S.conv = orig_conv
del feat


def spread(v):
    v = v.long() & 0xFFFF
    v = (v | (v << 32)) & 0x00FF00000000FFFF
    v = (v | (v << 16)) & 0x00FF0000FF0000FF
    v = (v | (v << 8)) & 0xF00F00F00F00F00F
    v = (v | (v << 4)) & 0x30C30C30C30C30C3
    v = (v | (v << 2)) & 0x9249249249249249
    return v


def timed(x, c):
    def run():
        return orig_conv(x, c["weight"], 3, 1, c["scale"], c["shift"], None, c["act"])
    out = run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        out = run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3, out


for c in calls:
    x = c["x"]
    C4 = x.cs.C
    s_ = x.cs.stride
    key = (spread((C4[:, 1] + 32768)) << 2) | (spread((C4[:, 2] + 32768)) << 1) | spread((C4[:, 3] + 32768))
    order = torch.argsort(key)
    sorted_already = bool((order == torch.arange(len(order), device=dev)).all())
    xm = S.SparseTensor(x.F.index_select(0, order).contiguous(), S.CoordSet(C4.index_select(0, order).contiguous(), s_), None, None)
    res = []
    for go in (False, True):
        S.GO_CONV = go
        t0, o0 = timed(x, c)
        t1, o1 = timed(xm, c)
        err = float((o1.F - o0.F.index_select(0, order)).abs().max() / (o0.F.abs().max() + 1e-30))
        res.append((t0, t1, err))
    S.GO_CONV = False
    tu = S.tile_union(xm.cs, xm.cs, 3, s_)
    n_t = (xm.cs.n + 63) // 64
    groups = tu[:n_t * 84 * 4].view(torch.int32).view(n_t, 84)[:, 0].float()
    print(f"rows={x.cs.n:7d} Cin={x.F.shape[1]:4d} Cout={c['weight'].shape[-1]:4d} morton-already={int(sorted_already)} | stage: as-is {res[0][0]:7.1f} us, "
          f"morton {res[0][1]:7.1f} us | gather-once: as-is {res[1][0]:7.1f}, morton {res[1][1]:7.1f} (groups/tile {float(groups.mean()):.2f}) | rel.err {res[0][2]:.1e} {res[1][2]:.1e}",
          flush=True)
