"""arkit, stage 1: the Atlas reconstruction network alone (reference: atlas_recon_arkit.py): posed images -> TSDF."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _builders as B  # noqa: E402

plugin = True
plugin_dir = 'projects/mvsdetection'
class_names = B.ARKIT_CLASSES
classes = len(class_names)
PIXEL_MEAN, PIXEL_STD, VOXEL_SIZE, N_SCALES = B.PIXEL_MEAN, B.PIXEL_STD, B.VOXEL_SIZE, B.N_SCALES
VOXEL_DIM_TRAIN, VOXEL_DIM_TEST = [160, 160, 64], [256, 256, 96]
NUM_FRAMES_TRAIN, NUM_FRAMES_TEST = 50, 500
RANDOM_ROTATION_3D, RANDOM_TRANSLATION_3D = True, True
PAD_XY_3D, PAD_Z_3D = 1.0, 0.25

work_dir = './work_dirs/atlas_recon_arkit'
R50_path = None                       # ImageNet ResNet-50 in Detectron2 layout (R-50.pth)
save_path = work_dir + '/results'
load_from = None
fp16 = dict(loss_scale=512.)
globals().update(B.schedule(total_epochs=80, lr_steps=[300], work_dir=work_dir, optimizer=dict(type='Adam', lr=5e-4),
                            max_norm=35, gamma=0.1))

train_pipeline, test_pipeline = B.recon_pipelines(VOXEL_DIM_TRAIN, VOXEL_DIM_TEST, PAD_XY_3D, PAD_Z_3D)
data = B.data_cfg('AtlasARKitDataset', './data/arkit', 'arkit', class_names, train_pipeline, test_pipeline, NUM_FRAMES_TRAIN,
                  NUM_FRAMES_TEST)
model = B.recon_model(VOXEL_DIM_TRAIN, VOXEL_DIM_TEST, save_path, R50_path)
