"""dense unprojection kernel at a workload shape under different block orders (CNRMA_DENSE_CHUNK): HIP-event times"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnrma_amd import rma, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "NS"
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, device=dev)
feat = rma.to_nhwc(sc["features"][:, 0])
del sc["features"]
proj = rma.scale_projection(sc["projection"][:, 0], stride).to(dev)
ref = None
for chunk in sys.argv[2:]:
    if chunk.startswith("persist"):
        os.environ["CNRMA_DENSE_PERSIST"] = chunk[7:]
    elif chunk.startswith("lpv"):
        os.environ["CNRMA_DENSE_LPV"] = chunk[3:]
        os.environ.pop("CNRMA_DENSE_CHUNK", None)
    elif chunk == "default":
        os.environ.pop("CNRMA_DENSE_CHUNK", None)
        os.environ.pop("CNRMA_DENSE_SLAB", None)
        for k in ("CNRMA_SLAB_Z", "CNRMA_SLAB_S", "CNRMA_SLAB_T", "CNRMA_SLAB_ZI"):
            os.environ.pop(k, None)
    elif chunk == "linear":
        os.environ.pop("CNRMA_DENSE_CHUNK", None)
        os.environ["CNRMA_DENSE_SLAB"] = "0"
    elif chunk.startswith("slab"):            # slab[:Z:S:T]
        os.environ.pop("CNRMA_DENSE_CHUNK", None)
        os.environ["CNRMA_DENSE_SLAB"] = "1"
        parts = chunk.split(":")[1:]
        os.environ.pop("CNRMA_SLAB_ZI", None)
        for name, val in zip(("CNRMA_SLAB_Z", "CNRMA_SLAB_S", "CNRMA_SLAB_T", "CNRMA_SLAB_ZI"), parts):
            os.environ[name] = val
    else:
        os.environ["CNRMA_DENSE_CHUNK"] = chunk
    ts = []
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); vol, cnt = rma.backproject_accum(feat, None, dims, 0.04, (0, 0, 0), stride, proj_scaled=proj); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    if ref is None:
        ref = (vol.clone(), cnt.clone())
    print(wl, "chunk", chunk, "ms", [round(t, 3) for t in ts], "same", torch.equal(vol, ref[0]) and torch.equal(cnt, ref[1]), flush=True)
    del vol, cnt
