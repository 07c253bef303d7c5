"""stage times of pipeline.forward_scenes for a batch of 3 (HIP events)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, synth, rma
from cnrma_amd import sparse as S
dev = torch.device("cuda:0")
V, C, H, W, dims, stride = synth.SHAPES["S"]
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
sc = synth.make_scene("S", seed=0)
scene = dict(features=sc["features"][:, 0].to(dev), projection=sc["projection"][:, 0], tsdf=sc["tsdf"][0, 0].to(dev))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(2):
    pipeline.forward_scenes(cfg, backbone, head, [scene] * B)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
marks = [("start", ev())]
parts = []
for s_ in [scene] * B:
    feats = rma.to_nhwc(s_["features"])
    rma.backproject_accum(feats, s_["projection"], cfg.dims, cfg.voxel_size, cfg.origin, cfg.stride)
    pinv = rma.projection_inverse(s_["projection"], cfg.stride).to(dev)
    c, f, info = rma.aggregate_points(feats, pinv, s_["tsdf"], cfg.dims, cfg.voxel_size, cfg.origin, cfg.n_steps, cfg.thr,
                                      cfg.ray_marching_type, cfg.depth_points, max_points=cfg.max_points, sampler=cfg.sampler)
    parts.append((c, f))
marks.append(("front x%d" % B, ev()))
x = S.sparse_collate(parts, cfg.voxel_size_fcaf3d)
marks.append(("voxelize+collate", ev()))
levels = backbone(x)
marks.append(("backbone", ev()))
cen, box, cls, pts, scn = map(list, head(levels, fused=True))
marks.append(("head", ev()))
dets = head.get_bboxes_fused(cen, box, cls, pts, scn, B)
marks.append(("decode", ev()))
torch.cuda.synchronize()
for (n0, a), (n1, b) in zip(marks[:-1], marks[1:]):
    print(f"{n1:20s} {a.elapsed_time(b):8.3f} ms  ({a.elapsed_time(b)/B:6.3f} per scene)")
print("rows", [len(l) for l in levels], [len(c[0]) for c in cen])
