"""dense unprojection kernel at a workload shape under different schedule switches (rma.dense_tuning): HIP-event times
and bit equality with the first configuration.

    python scripts/dense_ab.py NS variant=0 variant=1 variant=1,epi=1 variant=1,st=32,lockstep=1 ...
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnrma_amd import rma, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "NS"
dev = torch.device("cuda:0")
if "," in wl:                       # custom shape V,C,H,W,X,Y,Z,stride
    n = [int(x) for x in wl.split(",")]
    shape = (n[0], n[1], n[2], n[3], (n[4], n[5], n[6]), n[7])
else:
    shape = synth.SHAPES[wl]
V, C, H, W, dims, stride = shape
sc = synth.make_scene(shape, seed=0, device=dev)
feat = rma.to_nhwc(sc["features"][:, 0])
del sc["features"]
proj = rma.scale_projection(sc["projection"][:, 0], stride).to(dev)
ref = None
for spec in sys.argv[2:] or ["default"]:
    kw = {} if spec == "default" else {k: int(v) for k, v in (kv.split("=") for kv in spec.split(","))}
    rma.dense_tuning(**kw)
    ts = []
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); vol, cnt = rma.backproject_accum(feat, None, dims, 0.04, (0, 0, 0), stride, proj_scaled=proj); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    if ref is None:
        ref = (vol.clone(), cnt.clone())
    print(wl, spec, "ms", [round(t, 3) for t in ts], "same", torch.equal(vol, ref[0]) and torch.equal(cnt, ref[1]), flush=True)
    del vol, cnt
rma.dense_tuning()
