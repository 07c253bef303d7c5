"""per-layer table of the weight-gradient kernel in one training step at the ScanNet shape (bf16 autocast): rows, channels, chunks,
slab bytes, kernel time, time of the slab sum -- events around every call (the step is serialised: not a wall-time figure)"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import projects.mvsdetection  # noqa: F401
from projects.mvsdetection.registry import build_model
from cnrma_amd import synth, sparse as S

dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "S"
sc = synth.make_scene(shape, seed=0)
C = sc["features"].shape[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = runpy.run_path(os.path.join(root, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
m = dict(cfg["model"])
m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None)
m.update(save_path="/tmp/cnrma_train_probe", voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]),
         use_feature_transform=False, point_sampler="device", detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=34))
torch.manual_seed(0)
model = build_model(m)
model.detection_backbone.init_weights(); model.detection_head.init_weights()
model = model.to(dev).train()
dims = np.array(sc["dims"], dtype=np.float32) * 0.04
rng = np.random.RandomState(0)
boxes = torch.tensor([[rng.uniform(.2, .8) * dims[0], rng.uniform(.2, .8) * dims[1], rng.uniform(0, .3) * dims[2], .8, .6, .7, 0.]
                      for _ in range(12)], dtype=torch.float32, device=dev)
labels = torch.from_numpy(rng.randint(0, 18, size=12)).to(dev)
feats = sc["features"][:, 0].to(dev).requires_grad_(True)
data = dict(features=[feats], projection=[sc["projection"][:, 0].to(dev)], tsdf=sc["tsdf"].to(dev),
            offset=[torch.zeros(3, device=dev)], gt_bboxes_3d=[boxes], gt_labels_3d=[labels])
opt = torch.optim.SGD(model.parameters(), lr=1e-4)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model.train_step(dict(data), None)
    opt.zero_grad(); feats.grad = None
    out["loss"].backward()
    opt.step()


for _ in range(3):
    step()
rows = []
real_call = S.call


def timed_call(name, *a):
    if not name.startswith("cnrma_sparse_conv_wgrad") and name not in ("cnrma_sparse_conv_bf16", "cnrma_sparse_conv_go_bf16", "cnrma_sparse_kernel_map_transpose"):
        return real_call(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real_call(name, *a)
    e1.record(); e1.synchronize()
    if name == "cnrma_sparse_conv_wgrad_go_bf16":
        rows.append(("wgrad_go", a[1], a[4], 27, a[5], -(-a[5] // a[7]), e0.elapsed_time(e1) * 1e3))
    elif name.startswith("cnrma_sparse_conv_wgrad"):
        rows.append(("wgrad", a[1], a[5], a[3], a[6], a[8], e0.elapsed_time(e1) * 1e3))
    elif name == "cnrma_sparse_conv_go_bf16":
        rows.append(("conv_go", a[1], a[4], 27, a[6], 0, e0.elapsed_time(e1) * 1e3))
    elif name == "cnrma_sparse_conv_bf16":
        rows.append(("conv", a[1], a[5], a[3], a[11], 0, e0.elapsed_time(e1) * 1e3))
    else:
        rows.append(("transpose", 0, 0, a[3], a[1], 0, e0.elapsed_time(e1) * 1e3))
    return r


S.call = timed_call
step()
S.call = real_call
print(f"{'kind':10s} {'Cin':>4s} {'Cout':>4s} {'K':>3s} {'rows':>7s} {'chunks':>6s} {'slab MB':>8s} {'us':>8s}")
tot = {}
for kind, cin, cout, K, n, rpc, us in rows:
    chunks = -(-n // rpc) if rpc else 0
    mb = chunks * K * cin * cout * 4 / 2**20
    tot[kind] = tot.get(kind, 0) + us
    print(f"{kind:10s} {cin:4d} {cout:4d} {K:3d} {n:7d} {chunks:6d} {mb:8.1f} {us:8.1f}")
print({k: round(v) for k, v in tot.items()}, "us per step (serialised)")
