"""ScanNet posed-image scenes (registered name `AtlasScanNetDataset`; directory layout and sample dict of the reference's
datasets/scannet_dataset.py:23-144): <root>/posed_images/<scene>/{00000.jpg, 00000.txt (camera -> world), intrinsic.txt},
<root>/atlas_tsdf/<scene>/tsdf_{04,08,16}.npz; poses are pre-multiplied by the scene's axis-alignment matrix."""
import os

import numpy as np
from PIL import Image

from ..registry import DATASETS
from .multiview_dataset import MultiViewDataset


@DATASETS.register_module()
class AtlasScanNetDataset(MultiViewDataset):
    BOX_DIM, WITH_YAW = 6, False

    def get_data_info(self, index):
        info = self.data_infos[index]
        scene = info["scene"]
        image_ids = self.select_frames(info["total_image_ids"])
        ann = self.get_ann_info(index)
        frames = os.path.join(self.data_root, "posed_images", scene)
        K = np.loadtxt(os.path.join(frames, "intrinsic.txt"), delimiter=" ")[:3, :3].astype(np.float32)
        imgs, intrinsics, extrinsics = [], [], []
        for vid in image_ids:
            name = str(int(vid)).zfill(5)
            pose = np.loadtxt(os.path.join(frames, name + ".txt"))
            if ann is not None:
                pose = ann["axis_align_matrix"] @ pose
            if not np.isfinite(pose).all():
                raise ValueError(f"{scene} frame {name}: pose is not finite")
            imgs.append(Image.open(os.path.join(frames, name + ".jpg")))
            intrinsics.append(K.copy())
            extrinsics.append(pose)
        return dict(scene=scene, image_ids=image_ids, imgs=imgs, intrinsics=intrinsics, extrinsics=extrinsics,
                    tsdf_dict=self.read_scene_volumes(os.path.join(self.data_root, "atlas_tsdf"), scene, self.voxel_size),
                    ann_info=ann)
