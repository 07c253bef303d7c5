"""Point-cloud side of the FCAF3D transforms used inside the detector
(reference: projects/mvsdetection/datasets/pipelines/fcaf3d_transforms.py:14-200, :283-296).
Host-side RNG stays numpy (global state) so that the drawn numbers are those of the reference."""
import numpy as np
import torch

from ...registry import PIPELINES


@torch.no_grad()
def sample_points(points, max_points=None):
    """Boolean keep-mask [N] on points.device: all rows when N <= max_points, else exactly max_points rows chosen by
    np.random.choice(N, max_points, replace=False) (reference :283-296; `np.bool` there is plain bool here)."""
    assert max_points is not None
    n = points.shape[0]
    mask = np.zeros(n, dtype=bool)
    if n > max_points:
        mask[np.random.choice(n, max_points, replace=False)] = True
    else:
        mask[:] = True
    return torch.from_numpy(mask).to(points.device)


def rotate_points(points, angle):
    """rotate xyz about +z by `angle` (reference :152-167: points @ R^T with R^T = [[c,-s,0],[s,c,0],[0,0,1]]^T)."""
    a = torch.as_tensor(angle)
    c, s = torch.cos(a), torch.sin(a)
    rot_t = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]).T.to(points.device)
    points[:, :3] = points[:, :3] @ rot_t
    return points


def flip_points(points, direction="horizontal"):
    col = 0 if direction == "horizontal" else 1
    points[:, col] = -points[:, col]
    return points


def translate_points(points, vec):
    points[:, :3] = points[:, :3] + torch.as_tensor(vec, dtype=points.dtype).to(points.device)
    return points


def scale_points(points, factor):
    points[:, :3] = points[:, :3] * factor
    return points


@PIPELINES.register_module()
class TransformFeaturesBBoxes(object):
    """Train-time flip / rotate / scale / translate of the aggregated points (and of the GT boxes when the box
    object implements flip/rotate/scale/translate, as mmdet3d's DepthInstance3DBoxes does).  Reference :14-146.
    Draw order of the random numbers follows the reference: flips (horizontal, vertical), rotation, scale,
    translation."""

    def __init__(self, rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05),
                 translation_std=(0, 0, 0), flip_ratio_horizontal=0.0, flip_ratio_vertical=0.0):
        if not isinstance(rot_range, (list, tuple, np.ndarray)):
            rot_range = [-rot_range, rot_range]
        if not isinstance(translation_std, (list, tuple, np.ndarray)):
            translation_std = [translation_std] * 3
        self.rot_range = rot_range
        self.scale_ratio_range = scale_ratio_range
        self.translation_std = translation_std
        self.flip_ratio_horizontal = flip_ratio_horizontal
        self.flip_ratio_vertical = flip_ratio_vertical

    def __call__(self, points, gt_bboxes):
        if np.random.rand() < self.flip_ratio_horizontal:
            points = flip_points(points, "horizontal")
            if gt_bboxes is not None:
                gt_bboxes.flip("horizontal")
        if np.random.rand() < self.flip_ratio_vertical:
            points = flip_points(points, "vertical")
            if gt_bboxes is not None:
                gt_bboxes.flip("vertical")
        angle = np.random.uniform(self.rot_range[0], self.rot_range[1])
        points = rotate_points(points, angle)
        if gt_bboxes is not None:
            gt_bboxes.rotate(angle)
        factor = np.random.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1])
        points = scale_points(points, factor)
        if gt_bboxes is not None:
            gt_bboxes.scale(factor)
        trans = np.random.normal(scale=np.array(self.translation_std, dtype=np.float32), size=3).T
        points = translate_points(points, trans)
        if gt_bboxes is not None:
            gt_bboxes.translate(trans)
        return points, gt_bboxes



@PIPELINES.register_module()
class AtlasTransformSpaceDetection(object):
    """World frame of the detection stage (reference :204-266): the scene is translated so that the voxel_dim volume at
    `origin` covers it -- centred on the scene's TSDF extent (mode 'middle') or anchored at the TSDF origin snapped
    down by half a metre (mode 'origin').  Training moves the ground-truth boxes along (offset = origin); testing keeps
    them and records `offset` = -translation, which the detector adds back to every aggregated point
    (ray_marching.py:364), so the saved boxes are in the original world frame."""

    def __init__(self, voxel_dim, origin=(0, 0, 0), test=False, mode="middle"):
        assert mode in ("middle", "origin")
        self.voxel_dim, self.origin, self.test, self.mode = voxel_dim, list(origin), test, mode

    def __call__(self, data):
        from .atlas_transforms import transform_space
        tsdf = data["tsdf_dict"]["tsdf_gt_004"]
        if self.mode == "middle":
            lo = tsdf.origin[0].float()
            hi = lo + torch.tensor(tsdf.tsdf_vol.shape) * tsdf.voxel_size
            end = hi - torch.as_tensor(self.voxel_dim) * tsdf.voxel_size
            t = -(lo * 0.5 + end * 0.5)
        else:
            t = (torch.tensor([.5, .5, .5]) // tsdf.voxel_size) * tsdf.voxel_size - tsdf.origin
            t = t.view(-1)
        if self.test:
            data["offset"] = -t
        else:
            data["offset"] = torch.tensor(self.origin, dtype=torch.float32)
            data["gt_bboxes_3d"].translate(t)
        T = torch.eye(4)
        T[:3, 3] = t
        return transform_space(data, torch.inverse(T), self.voxel_dim, self.origin)

    def __repr__(self):
        return type(self).__name__
