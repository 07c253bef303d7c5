"""Shared builder of the two hot-path configs.  Key names and values follow the reference's
projects/configs/mvsdetection/ray_marching_{scannet,arkit}.py (model section :114-210 and the module-level constants
:10-30); the 2D backbone / Atlas 3D entries are None because those networks are outside the hot-path scope (their
outputs -- features and TSDF -- are the detector's inputs), see INTEGRATION.md."""

PIXEL_MEAN = [103.53, 116.28, 123.675]
PIXEL_STD = [1.0, 1.0, 1.0]
VOXEL_SIZE = 0.04
VOXEL_SIZE_FCAF3D = 0.01
N_SCALES = 3
RAY_MARCHING_TYPE = 'neus'
NEUS_THRESHOLD = 0.05
DEPTH_POINTS = None


def make_model(n_classes, n_reg_outs, with_yaw, voxel_dim_train, voxel_dim_test, use_batchnorm_test, save_path):
    return dict(
        type='RayMarching',
        pixel_mean=PIXEL_MEAN,
        pixel_std=PIXEL_STD,
        voxel_size=VOXEL_SIZE,
        n_scales=N_SCALES,
        voxel_dim_train=voxel_dim_train,
        voxel_dim_test=voxel_dim_test,
        origin=[0, 0, 0],
        backbone2d_stride=4,
        loss_weight_detection=1.0,
        loss_weight_recon=0.5,
        voxel_size_fcaf3d=VOXEL_SIZE_FCAF3D,
        use_batchnorm_train=True,
        use_batchnorm_test=use_batchnorm_test,
        save_path=save_path,
        ray_marching_type=RAY_MARCHING_TYPE,
        neus_threshold=NEUS_THRESHOLD,
        depth_points=DEPTH_POINTS,
        backbone2d=None,        # reference: FPNDetectron (ResNet-50 FPN)      -- out of scope, features come in
        feature_2d=None,        # reference: AtlasFPNFeature (32 ch @ stride 4) -- out of scope
        backbone_3d=None,       # reference: AtlasBackbone3D                    -- out of scope, TSDF comes in
        tsdf_head=None,         # reference: AtlasTSDFHead                      -- out of scope
        detection_backbone=dict(type='FCAF3DBackbone', in_channels=32, depth=34),
        detection_head=dict(
            type='FCAF3DHead',
            in_channels=(64, 128, 256, 512),
            out_channels=128,
            pts_threshold=200000,
            n_classes=n_classes,
            n_reg_outs=n_reg_outs,
            voxel_size=VOXEL_SIZE_FCAF3D,
            assigner=dict(type='FCAF3DAssigner', limit=27, topk=18, n_scales=4),
            loss_bbox=dict(type='IoU3DLoss', loss_weight=1.0, with_yaw=with_yaw),
            train_cfg=dict(),
            test_cfg=dict(nms_pre=1000, iou_thr=.5, score_thr=.01)),
        max_points=500000,
        use_feature_transform=True,
        feature_transform=dict(
            flip_ratio_horizontal=0.5,
            flip_ratio_vertical=0.5,
            rot_range=[-0.087266, 0.087266],
            scale_ratio_range=[.9, 1.1],
            translation_std=[.1, .1, .1]))
