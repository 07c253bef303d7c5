"""ARKitScenes, stage 2 data dump (reference: arkit_middle.py): aggregated points of the TRAIN scenes for pre-training
FCAF3D with oriented boxes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _builders as B  # noqa: E402

plugin = True
plugin_dir = 'projects/mvsdetection'
class_names = B.ARKIT_CLASSES
classes = len(class_names)
PIXEL_MEAN, PIXEL_STD, VOXEL_SIZE, VOXEL_SIZE_FCAF3D, N_SCALES = B.PIXEL_MEAN, B.PIXEL_STD, B.VOXEL_SIZE, B.VOXEL_SIZE_FCAF3D, B.N_SCALES
VOXEL_DIM_TRAIN, VOXEL_DIM_TEST = [192, 192, 80], [192, 192, 80]
NUM_FRAMES_TRAIN, NUM_FRAMES_TEST = 40, 40
USE_BATCHNORM_TRAIN, USE_BATCHNORM_TEST = True, True
LOSS_WEIGHT_RECON, LOSS_WEIGHT_DETECTION = 0.5, 1.0
RAY_MARCHING_TYPE, NEUS_THRESHOLD, DEPTH_POINTS = 'neus', 0.05, None
MIDDLE_SAVE_PATH = './data/arkit/atlas_middle_data'
MIDDLE_VISUALIZE_PATH = None

work_dir = './work_dirs/arkit_middle'
R50_path = None
save_path = work_dir + '/results'
load_from = None                      # the stage-1 checkpoint (atlas_recon_arkit)
globals().update(B.schedule(total_epochs=360, lr_steps=[27, 36], work_dir=work_dir, checkpoint_interval=1))

train_pipeline, test_pipeline = B.detection_pipelines(VOXEL_DIM_TRAIN, VOXEL_DIM_TEST, test_mode='middle')
data = B.data_cfg('AtlasARKitDataset', './data/arkit', 'arkit', class_names, train_pipeline, test_pipeline,
                  NUM_FRAMES_TRAIN, NUM_FRAMES_TEST, test_split='train')
model = B.detection_model(n_classes=classes, n_reg_outs=8, with_yaw=True, voxel_dim_train=VOXEL_DIM_TRAIN,
                          voxel_dim_test=VOXEL_DIM_TEST, use_batchnorm_test=USE_BATCHNORM_TEST, save_path=save_path,
                          r50_path=R50_path, ray_marching_type=RAY_MARCHING_TYPE, neus_threshold=NEUS_THRESHOLD,
                          depth_points=DEPTH_POINTS, middle_save_path=MIDDLE_SAVE_PATH,
                          middle_visualize_path=MIDDLE_VISUALIZE_PATH)
