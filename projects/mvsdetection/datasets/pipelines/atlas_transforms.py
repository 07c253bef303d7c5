"""Loading-side transforms of the multi-view pipeline (registered names and data contract of the reference's
projects/mvsdetection/datasets/pipelines/atlas_transforms.py:12-227).  A sample is a dict: `imgs` (PIL images), per-view
`intrinsics` [3,3] / `extrinsics` [4,4] (camera -> world), `tsdf_dict` {tsdf_gt_004/008/016: TSDF}, `scene`,
`image_ids`, `ann_info`.  After the pipeline: imgs [V,3,H,W] fp32 (0..255), projection [V,3,4] (full-resolution pixel
units: the detector divides rows 0-1 by its 2D stride), offset [3] ..."""
import numpy as np
import torch
from PIL import Image, ImageOps

from ...core.data_container import DataContainer as DC
from ...registry import PIPELINES


@PIPELINES.register_module()
class AtlasToTensor(object):
    def __call__(self, data):
        data["imgs"] = torch.as_tensor(np.stack(data["imgs"]).transpose(0, 3, 1, 2), dtype=torch.float32)     # V,3,H,W
        data["intrinsics"] = torch.as_tensor(np.stack(data["intrinsics"]), dtype=torch.float32)
        data["extrinsics"] = torch.as_tensor(np.stack(data["extrinsics"]), dtype=torch.float32)
        ann = data.pop("ann_info", None)
        if ann is not None:
            data["gt_bboxes_3d"] = ann["gt_bboxes_3d"]
            data["gt_labels_3d"] = torch.as_tensor(ann["gt_labels_3d"]).long()
            data["axis_align_matrix"] = torch.as_tensor(ann["axis_align_matrix"], dtype=torch.float32)
        if "depths" in data:
            data["depths"] = torch.as_tensor(np.stack(data["depths"]), dtype=torch.float32)
        return data


@PIPELINES.register_module()
class AtlasCollectData(object):
    HOST_ONLY = ("tsdf_dict", "scene", "image_ids", "gt_bboxes_3d")

    def __call__(self, data):
        out = {}
        for key in ("imgs", "projection", "tsdf_dict", "scene", "image_ids", "depths", "offset", "gt_bboxes_3d",
                    "gt_labels_3d", "axis_align_matrix"):
            if key in data:
                if key in ("gt_labels_3d", "axis_align_matrix") and "gt_bboxes_3d" not in data:
                    continue
                out[key] = DC(data[key], cpu_only=key in self.HOST_ONLY)
        return out


def pad_scannet(img, intrinsics):
    """ScanNet colour frames are 1296 x 968; two rows of padding top and bottom make them 4:3"""
    if img.size == (1296, 968):
        img = ImageOps.expand(img, border=(0, 2))
        intrinsics[1, 2] += 2
    return img, intrinsics


@PIPELINES.register_module()
class AtlasResizeImage(object):
    """resize every view to `size` = (width, height); the intrinsics follow the resize"""

    def __init__(self, size):
        self.size = tuple(size)

    def __call__(self, data):
        for i, (im, K) in enumerate(zip(data["imgs"], data["intrinsics"])):
            im, K = pad_scannet(im, K)
            w, h = im.size
            K[0, :] /= w / self.size[0]
            K[1, :] /= h / self.size[1]
            data["imgs"][i] = np.array(im.resize(self.size, Image.BILINEAR), dtype=np.float32)
            data["intrinsics"][i] = K
        return data

    def __repr__(self):
        return f"{type(self).__name__}(size={self.size})"


@PIPELINES.register_module()
class AtlasIntrinsicsPoseToProjection(object):
    """projection = K @ (camera -> world)^-1 [:3]  per view"""

    def __call__(self, data):
        K, pose = data.pop("intrinsics"), data.pop("extrinsics")
        data["projection"] = torch.stack([k @ torch.inverse(p)[:3, :] for k, p in zip(K, pose)])
        return data


def transform_space(data, transform, voxel_dim, origin):
    """change of world frame: poses are mapped by transform^-1, every TSDF level is resampled into a volume of
    voxel_dim / (its voxel size / the finest voxel size) voxels at `origin`"""
    inv = torch.inverse(transform)
    for i in range(len(data["extrinsics"])):
        data["extrinsics"][i] = inv @ data["extrinsics"][i]
    sizes = sorted(int(key[8:]) for key in data["tsdf_dict"])
    for vs in sizes:
        key = "tsdf_gt_" + str(vs).zfill(3)
        dim = [int(d / (vs / sizes[0])) for d in voxel_dim]
        data["tsdf_dict"][key] = data["tsdf_dict"][key].transform(transform, dim, origin)
    return data


def _rotated_extent(tsdf, R):
    """bounding box of the TSDF volume's footprint after the in-plane rotation R (z unchanged)"""
    lo = tsdf.origin[0]
    hi = lo + torch.tensor(tsdf.tsdf_vol.shape) * tsdf.voxel_size
    xy = R @ torch.tensor([[lo[0], lo[0], hi[0], hi[0]], [lo[1], hi[1], lo[1], hi[1]]], dtype=torch.float32)
    return torch.stack((xy[0].min(), xy[1].min(), lo[2])), torch.stack((xy[0].max(), xy[1].max(), hi[2]))


@PIPELINES.register_module()
class AtlasRandomTransformSpaceRecon(object):
    """training augmentation of the reconstruction stage: random rotation about z and random placement of the
    voxel_dim crop inside the (padded) extent of the scene"""

    def __init__(self, voxel_dim, random_rotation=True, random_translation=True, paddingXY=1.5, paddingZ=.25,
                 origin=(0, 0, 0)):
        self.voxel_dim, self.origin = voxel_dim, list(origin)
        self.random_rotation, self.random_translation = random_rotation, random_translation
        self.pad_lo = torch.tensor([paddingXY, paddingXY, paddingZ])
        self.pad_hi = torch.tensor([paddingXY, paddingXY, 0.0])

    def __call__(self, data):
        tsdf = data["tsdf_dict"]["tsdf_gt_004"]
        r = float(torch.rand(1) * 2 * np.pi) if self.random_rotation else 0.0
        R = torch.tensor([[np.cos(r), -np.sin(r)], [np.sin(r), np.cos(r)]], dtype=torch.float32)
        lo, hi = _rotated_extent(tsdf, R)
        start = lo - self.pad_lo
        end = hi + self.pad_hi - torch.as_tensor(self.voxel_dim) * tsdf.voxel_size
        t = torch.rand(3) if self.random_translation else torch.full((3,), 0.5)
        t = t * start + (1 - t) * end
        T = torch.eye(4)
        T[:2, :2] = R
        T[:3, 3] = -t
        data["offset"] = -t
        return transform_space(data, torch.inverse(T), self.voxel_dim, self.origin)

    def __repr__(self):
        return type(self).__name__


@PIPELINES.register_module()
class AtlasTestTransformSpaceRecon(object):
    """test time: the volume starts at the scene's TSDF origin (snapped down by the half-metre the training crops use)"""

    def __init__(self, voxel_dim, origin):
        self.voxel_dim, self.origin = voxel_dim, origin

    def __call__(self, data):
        tsdf = data["tsdf_dict"]["tsdf_gt_004"]
        offset = tsdf.origin - (torch.tensor([.5, .5, .5]) // tsdf.voxel_size) * tsdf.voxel_size
        T = torch.eye(4)
        T[:3, 3] = offset
        data["offset"] = offset
        return transform_space(data, T, self.voxel_dim, self.origin)

    def __repr__(self):
        return type(self).__name__
