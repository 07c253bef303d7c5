"""Worker of tests/test_multirank_gpu.py::test_ddp_gradient_allreduce (launched with torch.distributed.run, 2 ranks on ONE
device over gloo -- RCCL refuses two ranks on a device; on a multi-GPU node the same code runs with backend nccl):
the registered detector wrapped in DistributedDataParallel, one train_step per rank on ITS OWN scene; the all-reduced
gradients must be identical on both ranks and equal the mean of the two single-process gradients
(reference: train.py:164-171 MMDistributedDataParallel, dist_train.sh:7-9)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(device):
    import projects.mvsdetection  # noqa: F401
    from cnrma_amd import synth
    from projects.mvsdetection.registry import build_model
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    m = dict(cfg["model"])
    dims = synth.SHAPES["tiny"][4]
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None, save_path=None, voxel_dim_test=list(dims),
             voxel_dim_train=list(dims), max_points=None, use_feature_transform=False,
             detection_backbone=dict(type="FCAF3DBackbone", in_channels=8, depth=14))
    torch.manual_seed(0)
    model = build_model(m)
    model.detection_backbone.init_weights()
    model.detection_head.init_weights()
    return model.to(device).train()


def scene_batch(seed, device):
    from cnrma_amd import synth
    sc = synth.make_scene("tiny", seed=seed)
    ext = np.array(sc["dims"], dtype=np.float32) * 0.04
    boxes = torch.tensor([[0.4 * ext[0], 0.5 * ext[1], 0.1 * ext[2], 0.5, 0.4, 0.5, 0.0]], device=device)
    return dict(features=[sc["features"][:, 0].to(device)], projection=[sc["projection"][:, 0].to(device)],
                tsdf=sc["tsdf"].to(device), offset=[torch.zeros(3, device=device)], gt_bboxes_3d=[boxes],
                gt_labels_3d=[torch.tensor([1 + seed % 3], device=device)])


def grads_of(model, batch):
    model.zero_grad(set_to_none=True)
    inner = model.module if hasattr(model, "module") else model
    losses = model(**inner.data_converter(dict(batch)))
    loss = sum(v for k, v in losses.items() if "loss" in k)
    loss.backward()
    m = model.module if hasattr(model, "module") else model
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # reference gradients: both scenes in one process
    ref_model = build(device)
    g = [grads_of(ref_model, scene_batch(s, device)) for s in range(world)]
    mean = {n: sum(gi[n] for gi in g) / world for n in g[0]}
    model = torch.nn.parallel.DistributedDataParallel(build(device))
    got = grads_of(model, scene_batch(rank, device))
    ok = set(got) == set(mean)
    worst = 0.0
    for n in mean:
        scale = float(mean[n].abs().max()) + 1e-12
        worst = max(worst, float((got[n] - mean[n]).abs().max()) / scale)
    # identical on every rank
    flat = torch.cat([got[n].flatten() for n in sorted(got)]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    same = all(torch.equal(both[0], b) for b in both)
    differ = float((g[0][sorted(mean)[0]] - g[1][sorted(mean)[0]]).abs().max()) > 0      # the two scenes pull differently
    ok = ok and len(got) > 50 and float(flat.abs().sum()) > 0 and differ
    print(f"rank {rank}: {len(got)} gradient tensors, keys {ok} worst_rel_err {worst:.2e} identical_across_ranks {same}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if (ok and worst < 2e-4 and same) else 1)


if __name__ == "__main__":
    main()
