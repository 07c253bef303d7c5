"""Common part of the two posed-image datasets (reference: the shared halves of datasets/scannet_dataset.py:23-144 and
datasets/arkit_dataset.py:62-245, both subclasses of mmdet3d's Custom3DDataset, which is absent here): the info file
(`ann_file`: a pickled list of per-scene dicts), frame selection, the three TSDF levels of a scene, annotations ->
boxes, and the torch Dataset protocol.  With mmdet3d installed the boxes are DepthInstance3DBoxes; otherwise the
stand-in of core/boxes.py."""
import os
import pickle
import random
import warnings

import numpy as np
import torch
from torch.utils.data import Dataset

from ..core.boxes import GTBoxes
from .tsdf import TSDF
from ..registry import PIPELINES


class Compose:
    """the pipeline: a list of transform configs / callables applied in order"""

    def __init__(self, transforms):
        self.transforms = [PIPELINES.build(t) if isinstance(t, dict) else t for t in (transforms or [])]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


def make_boxes(array, with_yaw):
    """[K,6|7] gravity-centre boxes -> box object in the depth convention (bottom centre)"""
    try:
        from mmdet3d.core.bbox import DepthInstance3DBoxes
        return DepthInstance3DBoxes(array, box_dim=array.shape[-1], with_yaw=with_yaw, origin=(0.5, 0.5, 0.5))
    except Exception:
        return GTBoxes(torch.as_tensor(array), with_yaw=with_yaw, origin=(0.5, 0.5, 0.5))


class MultiViewDataset(Dataset):
    BOX_DIM, WITH_YAW = 6, False

    def __init__(self, data_root, ann_file, pipeline=None, classes=None, test_mode=False, num_frames=50, voxel_size=0.04,
                 select_type="random"):
        self.data_root, self.ann_file = data_root, ann_file
        self.CLASSES = classes
        self.test_mode = test_mode
        self.num_frames, self.voxel_size, self.select_type = num_frames, voxel_size, select_type
        with open(ann_file, "rb") as f:
            self.data_infos = sorted(pickle.load(f), key=lambda info: info["scene"])
        self.pipeline = Compose(pipeline)
        self.flag = np.zeros(len(self), dtype=np.uint8)         # mmdet's GroupSampler reads it

    def __len__(self):
        return len(self.data_infos)

    # ---- pieces of a sample ------------------------------------------------------------------------------------------
    def select_frames(self, ids):
        n = self.num_frames
        if n <= 0 or n > len(ids):
            picked = list(ids)
        elif self.select_type == "random":
            picked = random.sample(list(ids), n)
        elif self.select_type == "unit":
            step = (len(ids) - 1) // max(1, n - 1)
            picked = [ids[i * step] for i in range(n)]
        else:
            raise ValueError(f"select_type {self.select_type!r}")
        return sorted(picked)

    def read_scene_volumes(self, data_path, scene, voxel_size):
        """{tsdf_gt_004, _008, _016}: files tsdf_04.npz / _08 / _16 with `origin` and `tsdf`"""
        out = {}
        for level in range(3):
            vs = voxel_size * 2 ** level
            cm = int(round(vs * 100))
            z = np.load(os.path.join(data_path, scene, f"tsdf_{cm:02d}.npz"), allow_pickle=True)
            out[f"tsdf_gt_{cm:03d}"] = TSDF(vs, torch.as_tensor(z["origin"]).view(1, 3), torch.as_tensor(z["tsdf"]))
        return out

    def get_ann_info(self, index):
        info = self.data_infos[index]
        if "annos" not in info:
            return None
        ann = info["annos"]
        if "axis_align_matrix" in ann:
            align = ann["axis_align_matrix"].astype(np.float32)
        else:
            align = np.eye(4, dtype=np.float32)
            warnings.warn("no axis_align_matrix in the annotation: identity")
        if ann["gt_num"] != 0:
            boxes, labels = ann["gt_boxes_upright_depth"].astype(np.float32), ann["class"].astype(np.int64)
        else:
            boxes, labels = np.zeros((0, self.BOX_DIM), dtype=np.float32), np.zeros((0,), dtype=np.int64)
        return dict(gt_bboxes_3d=make_boxes(boxes, self.WITH_YAW), gt_labels_3d=labels, axis_align_matrix=align)

    def get_data_info(self, index):
        raise NotImplementedError

    # ---- Dataset protocol ----------------------------------------------------------------------------------------------
    def prepare_train_data(self, index):
        return self.pipeline(self.get_data_info(index))

    prepare_test_data = prepare_train_data

    def __getitem__(self, index):
        return self.prepare_test_data(index) if self.test_mode else self.prepare_train_data(index)

    def evaluate(self, outputs, voxel_size=0.04, save_path="./work_dir", logger=None):
        return {}           # detections are written to disk by the head and scored offline (post_process/)
