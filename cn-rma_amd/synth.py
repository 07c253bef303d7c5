"""Deterministic synthetic scene generator.

Room TSDF (Atlas sign convention: negative in free space in front of the surface, positive behind),
V cameras on a circle looking outwards-ish at the walls, random-normal feature maps.
Pure torch/numpy on CPU; callers move the tensors to the device.
"""
import math

import numpy as np
import torch

SHAPES = {
    # name: (V, C, Hf, Wf, (X, Y, Z), stride)
    "P": (4, 64, 128, 128, (64, 64, 64), 1),          # BASELINE config 0 (plumbing)
    "S": (40, 32, 120, 160, (192, 192, 80), 4),        # ScanNet-train / ARKit-test shape
    "St": (50, 32, 120, 160, (256, 256, 96), 4),       # ScanNet-test shape
    "tiny": (3, 8, 30, 40, (48, 48, 20), 4),           # shrunken S for fixtures
    "NS": (40, 256, 480, 640, (192, 192, 192), 1),     # north-star synthetic
}


def room_tsdf(dims, voxel_size=0.04, boxes=0, seed=0):
    """tsdf[1,1,X,Y,Z] fp32 in [-1,1]: -dist_inside/(3*vs) clamped; + behind the walls."""
    X, Y, Z = dims
    ext = np.array([X, Y, Z], dtype=np.float64) * voxel_size
    lo = np.array([0.6, 0.6, 0.3])
    hi = ext - np.array([0.6, 0.6, 0.5])
    ax = [np.arange(n, dtype=np.float64) * voxel_size for n in (X, Y, Z)]
    px, py, pz = np.meshgrid(*ax, indexing="ij")
    d = np.minimum.reduce([px - lo[0], hi[0] - px, py - lo[1], hi[1] - py, pz - lo[2], hi[2] - pz])
    if boxes:
        rng = np.random.RandomState(seed + 1234)
        for _ in range(boxes):
            c = lo + rng.rand(3) * (hi - lo)
            c[2] = lo[2] + 0.3 * rng.rand()
            h = 0.15 + 0.35 * rng.rand(3)
            # signed distance to an axis-aligned box (negative inside the box)
            q = np.stack([np.abs(px - c[0]) - h[0], np.abs(py - c[1]) - h[1], np.abs(pz - c[2]) - h[2]])
            outside = np.sqrt((np.maximum(q, 0) ** 2).sum(0))
            inside = np.minimum(q.max(0), 0)
            d = np.minimum(d, outside + inside)
    tsdf = np.clip(-d / (3 * voxel_size), -1.0, 1.0).astype(np.float32)
    return torch.from_numpy(tsdf).view(1, 1, X, Y, Z)


def camera_projections(V, dims, voxel_size=0.04, img_hw=(480, 640), return_parts=False):
    """V x 3 x 4 fp32 projection matrices K @ [R|t] in full-resolution pixel units (return_parts: also the intrinsics
    K [3,3] and the camera -> world poses [V,4,4] they are made of)."""
    X, Y, Z = dims
    ext = np.array([X, Y, Z], dtype=np.float64) * voxel_size
    centre = ext / 2
    H, W = img_hw
    f = 577.0 * W / 1296.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    out, poses = [], []
    for i in range(V):
        a = 2 * math.pi * i / V
        eye = centre + np.array([1.2 * math.cos(a), 1.2 * math.sin(a), 0.2])
        tgt = centre + 3.0 * np.array([math.cos(a + 2.5), math.sin(a + 2.5), -0.1])
        fwd = tgt - eye
        fwd /= np.linalg.norm(fwd)
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])            # world -> camera (x right, y down, z forward)
        t = -R @ eye
        out.append(K @ np.concatenate([R, t[:, None]], axis=1))
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = R.T, eye                  # camera -> world
        poses.append(pose)
    proj = torch.from_numpy(np.stack(out).astype(np.float32))
    return (proj, K, np.stack(poses)) if return_parts else proj


def make_scene(shape="S", seed=0, boxes=0, V=None, device=None, channels_last=False):
    """Returns dict(features[V,1,C,H,W], projection[V,1,3,4], tsdf[1,1,X,Y,Z], dims, voxel_size, origin, stride).
    device: draw the feature maps directly on that device (the north-star shape holds 12.6 GB of them per scene; the
    values then come from the device generator's stream, not the CPU one -- geometry and TSDF are unchanged).
    channels_last (device draws only): the maps are channels-last IN MEMORY ([V,1,C,H,W] is a permuted view of a
    [V,1,H,W,C] block), the layout a 2D network run in torch.channels_last hands over."""
    Vd, C, Hf, Wf, dims, stride = SHAPES[shape] if isinstance(shape, str) else shape
    V = V or Vd
    if device is not None and torch.device(device).type != "cpu":
        g = torch.Generator(device=device).manual_seed(seed)
        if channels_last:
            feats = torch.randn(V, 1, Hf, Wf, C, generator=g, dtype=torch.float32, device=device).permute(0, 1, 4, 2, 3)
        else:
            feats = torch.randn(V, 1, C, Hf, Wf, generator=g, dtype=torch.float32, device=device)
    else:
        g = torch.Generator().manual_seed(seed)
        feats = torch.randn(V, 1, C, Hf, Wf, generator=g, dtype=torch.float32)
    proj = camera_projections(V, dims, img_hw=(Hf * stride, Wf * stride)).view(V, 1, 3, 4)
    tsdf = room_tsdf(dims, boxes=boxes, seed=seed)
    return dict(features=feats, projection=proj, tsdf=tsdf, dims=tuple(dims), voxel_size=0.04,
                origin=(0.0, 0.0, 0.0), stride=stride)


def write_scannet_like(root, n_scenes=2, V=6, dims=(48, 48, 24), img_hw=(120, 160), seed=0, boxes=2):
    """A tiny dataset in the directory layout AtlasScanNetDataset reads (datasets/scannet_dataset.py): random JPEG frames
    with the poses / intrinsics of camera_projections(), the room TSDF at 4 / 8 / 16 cm, an info pickle with two
    ground-truth boxes per scene.  World frame = the TSDF volume shifted by a per-scene origin, identity axis alignment.
    Returns the path of the info file."""
    import os
    import pickle
    from PIL import Image
    rng = np.random.RandomState(seed)
    infos = []
    for s in range(n_scenes):
        scene = f"scene{s:04d}_00"
        origin = np.array([0.2 * s - 0.4, 0.3, -0.1], dtype=np.float32)
        frames = os.path.join(root, "posed_images", scene)
        os.makedirs(frames, exist_ok=True)
        _, K, poses = camera_projections(V, dims, img_hw=img_hw, return_parts=True)
        K4 = np.eye(4)
        K4[:3, :3] = K
        np.savetxt(os.path.join(frames, "intrinsic.txt"), K4)
        for v in range(V):
            img = (rng.rand(img_hw[0], img_hw[1], 3) * 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(frames, f"{v:05d}.jpg"), quality=95)
            pose = poses[v].copy()
            pose[:3, 3] += origin
            np.savetxt(os.path.join(frames, f"{v:05d}.txt"), pose)
        vol_dir = os.path.join(root, "atlas_tsdf", scene)
        os.makedirs(vol_dir, exist_ok=True)
        full = room_tsdf(dims, boxes=boxes, seed=seed + s)[0, 0]
        for level, cm in enumerate((4, 8, 16)):
            vol = full if level == 0 else torch.nn.functional.avg_pool3d(full[None, None], 2 ** level)[0, 0]
            np.savez(os.path.join(vol_dir, f"tsdf_{cm:02d}.npz"), origin=origin, tsdf=vol.numpy())
        ext = np.array(dims, dtype=np.float32) * 0.04
        gt = np.array([[0.35 * ext[0], 0.4 * ext[1], 0.3 * ext[2], 0.5, 0.4, 0.5],
                       [0.65 * ext[0], 0.6 * ext[1], 0.35 * ext[2], 0.4, 0.6, 0.4]], dtype=np.float32)
        gt[:, :3] += origin
        infos.append(dict(scene=scene, total_image_ids=list(range(V)),
                          annos=dict(gt_num=2, gt_boxes_upright_depth=gt, **{"class": np.array([1, 3])},
                                     axis_align_matrix=np.eye(4))))
    ann = os.path.join(root, "scannet_infos_val.pkl")
    with open(ann, "wb") as f:
        pickle.dump(infos, f)
    return ann
