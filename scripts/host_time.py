"""host-side (Python) time per scene of the batched pipeline: cProfile top entries + wall time with the GPU idle-waits
separated (time spent inside .item()/.tolist() is waiting for the device, the rest is CPU work under the GIL)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, synth
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES["S"]
sc = synth.make_scene("S", seed=0)
scene = dict(features=sc["features"][:, 0].to(dev), projection=sc["projection"][:, 0], tsdf=sc["tsdf"][0, 0].to(dev))
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for _ in range(4):
    pipeline.forward_scenes(cfg, backbone, head, [scene] * B)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    pipeline.forward_scenes(cfg, backbone, head, [scene] * B)
torch.cuda.synchronize()
print("wall ms/scene (no profiler)", (time.perf_counter() - t0) / (N * B) * 1e3)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    pipeline.forward_scenes(cfg, backbone, head, [scene] * B)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
tot = sum(v[2] for v in st.stats.values())
wait = sum(v[2] for k, v in st.stats.items() if k[2] in ("<method 'item' of 'torch._C.TensorBase' objects>", "<method 'tolist' of 'torch._C.TensorBase' objects>"))
print("profiled total ms/scene", tot / (N * B) * 1e3, "of which waiting in item/tolist", wait / (N * B) * 1e3)
st.sort_stats("tottime").print_stats(32)
