"""layout pass + march: two launches vs the fused launch (cnrma_nchw_to_nhwc_march_f32); HIP-event times, equal outputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnrma_amd import rma, synth
dev = torch.device("cuda:0")
for wl in sys.argv[1:] or ["NS", "S"]:
    V, C, H, W, dims, stride = synth.SHAPES[wl]
    sc = synth.make_scene(wl, seed=0, device=dev, boxes=3)
    nchw = sc["features"][:, 0]
    out = torch.empty((V, H, W, C), dtype=torch.float32, device=dev)
    pinv = rma.projection_inverse(sc["projection"][:, 0], stride).to(dev)
    tsdf = sc["tsdf"][0, 0].to(dev)
    m = rma._March(out, pinv, tsdf, dims, 0.04, (0, 0, 0), 300, 0.05, "neus", 0)
    bufA, bufB = m.march_buffers(), m.march_buffers()

    def separate():
        rma.to_nhwc(nchw, out=out)
        return m.march(into=bufA)

    def fused():
        return m.march(layout_from=nchw, into=bufB)

    for name, fn in (("separate", separate), ("fused", fused), ("separate", separate), ("fused", fused)):
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print(wl, name, "ms", [round(t, 3) for t in ts[1:]], flush=True)
    ref = out.clone()
    separate(); torch.cuda.synchronize()
    same = torch.equal(ref, out) and all(torch.equal(x, y) for x, y in zip(bufA[:3], bufB[:3]))
    print(wl, "outputs equal:", same, flush=True)
