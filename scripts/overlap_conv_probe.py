"""do the dense unprojection (L2 <-> fabric bound, its waves mostly waiting for memory) and the sparse convolutions (matrix-pipe
bound) of ANOTHER scene co-execute when they are launched on two streams -- with the convolutions on a high-priority stream, so that
their workgroups take the slots dense workgroups free up?  NS shape; wall time of {dense, R x all convolutions of a scene, both}.
    python scripts/overlap_conv_probe.py [NS|S] [repeats of the convolution chain]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, rma, synth
from cnrma_amd import sparse as S
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "NS"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev, channels_last=True)
feat_nchw, proj3, tsdf = sc["features"][:, 0], sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
calls = []
orig_conv = S.conv


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    calls.append((x, weight, kernel_size, stride, scale, shift, residual, act))
    return y


S.conv = rec_conv
sys.modules["cnrma_amd.nn"].S.conv = rec_conv
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat_nchw, proj3, tsdf, dense=False)
S.conv = orig_conv
sys.modules["cnrma_amd.nn"].S.conv = orig_conv
feat = rma.to_nhwc(feat_nchw)
proj = rma.scale_projection(proj3, stride).to(dev)
print(f"{wl}: {len(calls)} convolutions recorded")


def dense():
    return rma.backproject_accum(feat, None, dims, 0.04, (0, 0, 0), stride, proj_scaled=proj)


def convs():
    out = None
    for _ in range(R):
        for x, w, k, s, sc_, sh, res, act in calls:
            out = orig_conv(x, w, k, s, sc_, sh, res, act)
    return out


def timed(fns, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = []
    for f, s in zip(fns, streams):
        with torch.cuda.stream(s), torch.no_grad():
            keep.append(f())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


lo, hi, eq = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
for name, fns, st in (("dense alone", [dense], [lo]), (f"convs x{R} alone", [convs], [hi]),
                      ("dense (low prio) + convs (high prio)", [dense, convs], [lo, hi]),
                      ("convs (high prio) launched first + dense", [convs, dense], [hi, lo]),
                      ("dense + convs, equal priority", [dense, convs], [lo, eq])):
    ts = [timed(fns, st) for _ in range(4)]
    print(f"{name:44s} ms {[round(t, 2) for t in ts[1:]]}", flush=True)
