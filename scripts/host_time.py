"""host-side enqueue time per stage (no device sync inside) vs end-to-end time: are we host-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, rma, synth
from cnrma_amd import sparse as S
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES["S"]
sc = synth.make_scene("S", seed=0)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
for _ in range(5):
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf)
torch.cuda.synchronize()
import cProfile, pstats
N = 20
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf)
pr.disable()
torch.cuda.synchronize()
print("ms/scene", (time.perf_counter() - t0) / N * 1e3)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
