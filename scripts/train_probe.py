"""time of one training step of the detector at the ScanNet shape (features and TSDF given; detection losses only)"""
import os, sys, time, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import projects.mvsdetection  # noqa: F401
from projects.mvsdetection.registry import build_model
from cnrma_amd import synth
dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "S"
sc = synth.make_scene(shape, seed=0)
C = sc["features"].shape[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = runpy.run_path(os.path.join(root, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
m = dict(cfg["model"])
m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None)
m.update(save_path="/tmp/cnrma_train_probe", voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]),
         use_feature_transform=False, point_sampler=os.environ.get("CNRMA_SAMPLER", "device"), detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=34))
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
AUTOCAST = len(sys.argv) > 2 and sys.argv[2] == "bf16"
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=AUTOCAST):
        out = model.train_step(dict(data), None)
    opt.zero_grad(); feats.grad = None
    out["loss"].backward()
    opt.step()
    return float(out["loss"].detach())
from cnrma_amd import sparse as S_
AB = os.environ.get("CNRMA_PROBE_AB", "1") == "1"       # the BatchNorm A/B phases (off under the profiler: one configuration only)
for bn_hip in ((True, False, True, False) if AB else ()):
    S_.BN_TRAIN_HIP = bn_hip
    for _ in range(2): l = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): l = step()
    torch.cuda.synchronize()
    print(f"  BatchNorm through {'the HIP kernels' if bn_hip else 'torch'}: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per step", flush=True)
S_.BN_TRAIN_HIP = os.environ.get("CNRMA_BN_HIP", "1") == "1"
S_.FUSE_CONV_BN = os.environ.get("CNRMA_FUSE", "1") == "1"
for _ in range(2): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5 if AB else 28
for _ in range(n): l = step()
torch.cuda.synchronize()
if os.environ.get("CNRMA_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): step()
    torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(110)
print(f"{shape} ({'bf16 autocast' if AUTOCAST else 'fp32'}): {(time.perf_counter() - t0) / n * 1e3:.1f} ms per training step, loss {l:.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
