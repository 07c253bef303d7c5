"""gather-once convolution vs the stage kernel, layer by layer on one scene (HIP events over REPS back-to-back launches;
the tile unions / neighbour tables are cached, i.e. excluded; the union build is timed separately)"""
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
    calls.append(dict(x=x, weight=weight, ks=kernel_size, stride=stride, scale=scale, shift=shift, residual=residual, act=act, n_out=y.cs.n))
    return y


S.conv = rec_conv
sys.modules["cnrma_amd.nn"].S.conv = rec_conv
S.GO_CONV = False
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
S.conv = orig_conv
sys.modules["cnrma_amd.nn"].S.conv = orig_conv
del feat


def timed(c):
    def run():
        return orig_conv(c["x"], c["weight"], c["ks"], c["stride"], c["scale"], c["shift"], c["residual"], c["act"]).F
    out = run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        out = run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3, out


tot = [0.0, 0.0]
for c in calls:
    K, Cin, Cout = c["ks"] ** 3, c["x"].F.shape[1], c["weight"].shape[-1]
    if not (K == 27 and c["stride"] == 1 and Cin % 32 == 0 and Cout >= 64):
        continue
    S.GO_CONV = False
    t0, ref = timed(c)
    S.GO_CONV = True
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c["x"].cs._union.clear()
    e0.record()
    tu = S.tile_union(c["x"].cs, c["x"].cs, 3, c["x"].cs.stride)
    e1.record()
    torch.cuda.synchronize()
    n_t = (c["n_out"] + 63) // 64
    hdr = tu[:n_t * 84 * 4].view(torch.int32).view(n_t, 84)
    groups = hdr[:, 0].float()
    t1, out = timed(c)
    err = float((out - ref).abs().max() / (ref.abs().max() + 1e-30))
    abl = []
    for mask_ in (1, 4, 5):
        S.conv_tuning(None, -1, -1, mask_)
        abl.append(timed(c)[0])
    S.conv_tuning(pf=3)                      # Cout 64: the 2 x 2 waves-over-rows-x-columns form (every weight fragment fetched twice)
    t_old = timed(c)[0]
    S.conv_tuning()
    tot[0] += t0
    tot[1] += t1
    print(f"rows={c['n_out']:7d} Cin={Cin:4d} Cout={Cout:4d} res={int(c['residual'] is not None)}  stage {t0:7.1f} us  gather-once {t1:7.1f} us "
          f"({t0 / t1:.2f}x; 2x2-wave form {t_old:7.1f})  union build {e0.elapsed_time(e1) * 1e3:6.1f} us, groups/tile mean {float(groups.mean()):.2f} max {int(groups.max())}"
          f"  rel.err {err:.1e}  ablate noMFMA/noB/neither {abl[0]:.0f}/{abl[1]:.0f}/{abl[2]:.0f}", flush=True)
print(f"sum over the {wl} scene's 3x3x3 stride-1 convolutions: stage {tot[0]:.0f} us, gather-once {tot[1]:.0f} us")
