"""do two big kernels of the path overlap when launched on two streams?  (NS shape; HIP events + wall)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnrma_amd import rma, synth
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES["NS"]
sc = synth.make_scene("NS", seed=0, device=dev)
nchw = sc["features"][:, 0]
feat = rma.to_nhwc(nchw)
proj = rma.scale_projection(sc["projection"][:, 0], stride).to(dev)
pinv = rma.projection_inverse(sc["projection"][:, 0], stride).to(dev)
tsdf = sc["tsdf"][0, 0].to(dev)
m = rma._March(feat, pinv, tsdf, dims, 0.04, (0, 0, 0), 300, 0.05, "neus", 0)
out_nhwc = torch.empty_like(feat)

def dense():
    return rma.backproject_accum(feat, None, dims, 0.04, (0, 0, 0), stride, proj_scaled=proj)
def march():
    return m.march()
def nhwc():
    return rma.to_nhwc(nchw, out=out_nhwc)

def timed(fns, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = []
    for f, s in zip(fns, streams):
        with torch.cuda.stream(s):
            keep.append(f())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ref = dense()
# usage: overlap_probe.py [dense-tuning spec ...]   e.g.  variant=1  variant=1,pipe=2  variant=0
for spec in sys.argv[1:] or ["default"]:
    kw = {} if spec == "default" else {k: int(v) for k, v in (kv.split("=") for kv in spec.split(","))}
    rma.dense_tuning(**kw)
    v = dense()
    torch.cuda.synchronize()
    assert torch.equal(v[0], ref[0]) and torch.equal(v[1], ref[1])
    del v
    for name, fns in (("dense", [dense]), ("march", [march]), ("nhwc", [nhwc]), ("dense+march", [dense, march]),
                      ("dense+nhwc", [dense, nhwc]), ("nhwc+dense", [nhwc, dense]), ("march+nhwc", [march, nhwc]),
                      ("dense+march+nhwc", [dense, march, nhwc])):
        ts = [timed(fns, [s1, s2, s3]) for _ in range(4)]
        print(f"{spec:22s} {name:18s} ms {[round(t, 2) for t in ts[1:]]}", flush=True)
rma.dense_tuning()
