"""where do the worker threads of the default bench configuration spend their time: waiting in device->host reads, or
enqueueing?  (monkey-patches _lib.read_ints with a timer; 3 threads x 2-scene passes like bench.py)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, synth, _lib
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES["S"]
sc = synth.make_scene("S", seed=0)
scene = dict(features=sc["features"][:, 0].to(dev), projection=sc["projection"][:, 0], tsdf=sc["tsdf"][0, 0].to(dev))
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = 2
pipeline.forward_scenes(cfg, backbone, head, [scene] * B)
wait = [0.0] * T
tls = threading.local()
orig = _lib.read_ints
def timed(t):
    t0 = time.perf_counter(); r = orig(t); wait[tls.i] += time.perf_counter() - t0; return r
for mod in ("cnrma_amd._lib", "cnrma_amd.sparse", "cnrma_amd.pipeline", "cnrma_amd.rma"):
    m = sys.modules[mod]
    if hasattr(m, "read_ints"): m.read_ints = timed
    if hasattr(m, "_lib"): m._lib.read_ints = timed
total = [0.0] * T
N = 12
streams = [torch.cuda.Stream() for _ in range(T)]
def worker(i):
    tls.i = i
    with torch.cuda.stream(streams[i]):
        for _ in range(3): pipeline.forward_scenes(cfg, backbone, head, [scene] * B)
        streams[i].synchronize(); wait[i] = 0.0
        t0 = time.perf_counter()
        for _ in range(N): pipeline.forward_scenes(cfg, backbone, head, [scene] * B)
        streams[i].synchronize()
        total[i] = time.perf_counter() - t0
ths = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
for i in range(T):
    print(f"thread {i}: {total[i]/N*1e3:.2f} ms per pass, waiting in reads {wait[i]/N*1e3:.2f} ms ({100*wait[i]/total[i]:.0f} %)")
print("scenes/s", T * N * B / max(total))
