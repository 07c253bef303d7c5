"""per-launch table of the sparse-conv calls of one scene (rows, Cin, Cout, K, ms, TF/s) -- diagnostics"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, synth
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES["S"]
sc = synth.make_scene("S", seed=0)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
B = int(os.environ.get("CNRMA_B", "1"))
scene = dict(features=feat, projection=proj, tsdf=tsdf)
run = (lambda: pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf)) if B == 1 else \
      (lambda: pipeline.forward_scenes(cfg, backbone, head, [scene] * B))
for _ in range(2):
    run()
prof = bench.KernelProfile(); prof.install()
run()
prof.uninstall(); torch.cuda.synchronize()
tot = 0
for name, a, e0, e1 in prof.records:
    ms = e0.elapsed_time(e1)
    if name in bench.CONV_ARGS:
        cin, K, cout, rows = (a[i] for i in bench.CONV_ARGS[name])
        fl = 2.0 * K * cin * cout * rows
        tot += ms
        print(f"conv rows={rows:7d} Cin={cin:4d} Cout={cout:4d} K={K:2d} {ms:8.3f} ms {fl/ms/1e9:7.1f} TF/s")
    else:
        print(f"{name} {ms:.3f} ms")
print("conv total", tot)

import collections, json
agg = collections.defaultdict(list)
for name, a, e0, e1 in prof.records:
    if name in bench.CONV_ARGS:
        cin, K, cout, rows = (a[i] for i in bench.CONV_ARGS[name])
        agg[(rows // 1000, cin, cout, K)].append(e0.elapsed_time(e1))
print("SUMMARY", json.dumps({f"{k[0]}k_{k[1]}_{k[2]}_K{k[3]}": round(sum(v) / len(v), 3) for k, v in sorted(agg.items())}))
