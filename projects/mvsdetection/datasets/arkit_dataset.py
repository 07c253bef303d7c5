"""ARKitScenes low-resolution wide-camera scenes (registered name `AtlasARKitDataset`; layout and sample dict of the
reference's datasets/arkit_dataset.py:62-245): <root>/<split>/<scene>/<scene>_frames/{lowres_wide/<scene>_<ts>.png,
lowres_wide_intrinsics/<scene>_<ts>.pincam, lowres_wide.traj}; oriented (yaw) ground-truth boxes.  An info entry may
instead carry explicit `image_paths` / `intrinsics` / `extrinsics` per frame id."""
import os

import numpy as np
from PIL import Image

from ..registry import DATASETS
from .multiview_dataset import MultiViewDataset


def rodrigues(v):
    """axis-angle vector -> rotation matrix (cv2.Rodrigues without OpenCV)"""
    v = np.asarray(v, dtype=np.float64)
    theta = np.linalg.norm(v)
    if theta < 1e-12:
        return np.eye(3)
    k = v / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(theta) * K + (1 - np.cos(theta)) * (K @ K)


def traj_line_to_pose(line):
    """'timestamp rx ry rz tx ty tz' (world -> camera, axis-angle) -> (timestamp string, camera -> world 4x4)"""
    tok = line.split()
    assert len(tok) == 7
    world_to_cam = np.eye(4)
    world_to_cam[:3, :3] = rodrigues([float(t) for t in tok[1:4]])
    world_to_cam[:3, 3] = [float(t) for t in tok[4:7]]
    return tok[0], np.linalg.inv(world_to_cam)


def pincam_intrinsics(path):
    w, h, fx, fy, cx, cy = np.loadtxt(path)
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])


@DATASETS.register_module()
class AtlasARKitDataset(MultiViewDataset):
    BOX_DIM, WITH_YAW = 7, True

    def _frames_from_disk(self, info, image_ids):
        scene = info["scene"]
        root = os.path.join(self.data_root, info["split"], scene, scene + "_frames")
        with open(os.path.join(root, "lowres_wide.traj")) as f:
            poses = {f"{round(float(ts), 3):.3f}": pose for ts, pose in map(traj_line_to_pose, f.readlines())}
        imgs, intrinsics, extrinsics = [], [], []
        for vid in image_ids:
            cand = [str(vid), f"{float(vid) - 0.001:.3f}", f"{float(vid) + 0.001:.3f}"]       # timestamps jitter by 1 ms
            pin = next((p for p in (os.path.join(root, "lowres_wide_intrinsics", f"{scene}_{c}.pincam") for c in cand)
                        if os.path.exists(p)), None)
            if pin is None:
                raise FileNotFoundError(f"{scene}: no intrinsics for frame {vid}")
            pose = poses.get(str(vid))
            if pose is None:
                pose = next((p for ts, p in poses.items() if abs(float(vid) - float(ts)) < 0.005), None)
            if pose is None or not np.isfinite(pose).all():
                raise ValueError(f"{scene} frame {vid}: no finite pose")
            imgs.append(Image.open(os.path.join(root, "lowres_wide", f"{scene}_{vid}.png")))
            intrinsics.append(pincam_intrinsics(pin).astype(np.float32))
            extrinsics.append(pose.astype(np.float32))
        return imgs, intrinsics, extrinsics

    def get_data_info(self, index):
        info = self.data_infos[index]
        scene = info["scene"]
        image_ids = self.select_frames(info["total_image_ids"])
        if "image_paths" in info:
            imgs = [Image.open(os.path.join(self.data_root, info["image_paths"][v])) for v in image_ids]
            intrinsics = [info["intrinsics"][v].astype(np.float32) for v in image_ids]
            extrinsics = [info["extrinsics"][v].astype(np.float32) for v in image_ids]
        else:
            imgs, intrinsics, extrinsics = self._frames_from_disk(info, image_ids)
        return dict(split=info.get("split"), scene=scene, image_ids=image_ids, imgs=imgs, intrinsics=intrinsics,
                    extrinsics=extrinsics,
                    tsdf_dict=self.read_scene_volumes(os.path.join(self.data_root, "atlas_tsdf"), scene, self.voxel_size),
                    ann_info=self.get_ann_info(index))
