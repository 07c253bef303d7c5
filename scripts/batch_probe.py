"""probe: backbone time for 3 scenes one by one vs one batched sparse tensor (instnorm statistics are global in the
batched run: timing probe only)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, synth, rma
from cnrma_amd import sparse as S
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES["S"]
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
pts = []
for seed in range(3):
    sc = synth.make_scene("S", seed=seed)
    feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
    nhwc = rma.to_nhwc(feat)
    pinv = rma.projection_inverse(proj, stride).to(dev)
    p, _ = rma.aggregate_rows(nhwc, pinv, tsdf, dims, 0.04, [0.0, 0.0, 0.0], 300, 0.05, "neus", None)
    m = rma.sample_mask_device(torch.tensor([p.shape[0]], dtype=torch.int32, device=dev), p.shape[0], 500000, seed=seed)
    c, f = rma.select_rows(p, [0.0, 0.0, 0.0], m)
    pts.append((c, f))
torch.cuda.synchronize()

def run_single():
    for c, f in pts:
        x = S.sparse_collate([(c, f)], 0.01)
        lv = backbone(x)
        list(head(lv))
def run_batched():
    x = S.sparse_collate(pts, 0.01)
    lv = backbone(x)
    return lv
for fn in (run_single, run_batched, run_single, run_batched):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(fn.__name__, (time.perf_counter() - t0) / 5 * 1e3, "ms per 3 scenes")
