"""ARKitScenes CN-RMA config (hot path).  Reference: projects/configs/mvsdetection/ray_marching_arkit.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _hotpath_base import *  # noqa: F401,F403,E402
from _hotpath_base import make_model  # noqa: E402

plugin = True
plugin_dir = 'projects/mvsdetection'
class_names = ["cabinet", "refrigerator", "shelf", "stove", "bed", "sink", "washer", "toilet", "bathtub", "oven",
               "dishwasher", "fireplace", "stool", "chair", "table", "tv_monitor", "sofa"]
classes = len(class_names)
VOXEL_DIM_TRAIN = [192, 192, 80]
VOXEL_DIM_TEST = [192, 192, 80]
NUM_FRAMES_TRAIN = 40
NUM_FRAMES_TEST = 40
USE_BATCHNORM_TRAIN = True
USE_BATCHNORM_TEST = True
dist_params = dict(backend='nccl')      # RCCL on ROCm
work_dir = './work_dirs/ray_marching_arkit'
save_path = work_dir + '/results'
model = make_model(n_classes=classes, n_reg_outs=8, with_yaw=True, voxel_dim_train=VOXEL_DIM_TRAIN,
                   voxel_dim_test=VOXEL_DIM_TEST, use_batchnorm_test=USE_BATCHNORM_TEST, save_path=save_path)
