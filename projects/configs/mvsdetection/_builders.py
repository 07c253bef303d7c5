"""Shared builders of the six plugin configs.  Every key and value of the reference's
projects/configs/mvsdetection/*.py is produced here once (model section: ray_marching_scannet.py:114-210, data section
:56-112, schedule :32-54) so that the config files themselves only state what differs between datasets and stages.
`hot_path_only=True` drops the 2D network and the Atlas 3D network (their outputs -- feature maps and TSDF -- then come in
as inputs): the form bench.py and the GPU tests use."""

PIXEL_MEAN = [103.53, 116.28, 123.675]
PIXEL_STD = [1.0, 1.0, 1.0]
VOXEL_SIZE = 0.04
VOXEL_SIZE_FCAF3D = 0.01
N_SCALES = 3

SCANNET_CLASSES = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture', 'counter',
                   'desk', 'curtain', 'refrigerator', 'showercurtain', 'toilet', 'sink', 'bathtub', 'garbagebin']
ARKIT_CLASSES = ["cabinet", "refrigerator", "shelf", "stove", "bed", "sink", "washer", "toilet", "bathtub", "oven",
                 "dishwasher", "fireplace", "stool", "chair", "table", "tv_monitor", "sofa"]


def backbone2d_cfg(pretrained=None):
    """ResNet-50 FPN (Detectron2 layout), frozen through res2"""
    return dict(type='FPNDetectron',
                bottom_up_cfg=dict(input_channels=3, norm='BN', depth=50, out_features=["res2", "res3", "res4", "res5"],
                                   num_groups=1, width_per_group=64, stride_in_1x1=True, res5_dilation=1,
                                   res2_out_channels=256, stem_out_channels=64, freeze_at=2),
                in_features=["res2", "res3", "res4", "res5"], out_channels=256, norm='BN', fuse_type='sum',
                pretrained=pretrained)


def feature_2d_cfg():
    return dict(type='AtlasFPNFeature', feature_strides={'p2': 4, 'p3': 8, 'p4': 16, 'p5': 32, 'p6': 64},
                feature_channels={'p2': 256, 'p3': 256, 'p4': 256, 'p5': 256, 'p6': 256}, output_dim=32, output_stride=4,
                norm='BN')


def backbone_3d_cfg():
    return dict(type='AtlasBackbone3D', channels=[32, 64, 128, 256], layers_down=[1, 2, 3, 4], layers_up=[3, 2, 1], drop=0.0,
                zero_init_residual=True, cond_proj=False, norm='BN')


def tsdf_head_cfg():
    return dict(type='AtlasTSDFHead', input_channels=[32, 64, 128], n_scales=3, voxel_size=VOXEL_SIZE, label_smoothing=1.05,
                sparse_threshold=[0.99, 0.99, 0.99])


def recon_model(voxel_dim_train, voxel_dim_test, save_path, r50_path=None):
    """stage 1: type='Atlas' (2D network -> dense unprojection -> 3D U-Net -> TSDF)"""
    return dict(type='Atlas', pixel_mean=PIXEL_MEAN, pixel_std=PIXEL_STD, voxel_size=VOXEL_SIZE, n_scales=N_SCALES,
                voxel_dim_train=voxel_dim_train, voxel_dim_test=voxel_dim_test, origin=[0, 0, 0], backbone2d_stride=4,
                save_path=save_path, backbone2d=backbone2d_cfg(r50_path), feature_2d=feature_2d_cfg(),
                backbone_3d=backbone_3d_cfg(), tsdf_head=tsdf_head_cfg())


def detection_model(n_classes, n_reg_outs, with_yaw, voxel_dim_train, voxel_dim_test, use_batchnorm_test, save_path,
                    r50_path=None, ray_marching_type='neus', neus_threshold=0.05, depth_points=None,
                    loss_weight_recon=0.5, loss_weight_detection=1.0, middle_save_path=None, middle_visualize_path=None,
                    hot_path_only=False):
    """stages 2-3: type='RayMarching' (+ ray-marching aggregation -> FCAF3D)"""
    model = dict(
        type='RayMarching', pixel_mean=PIXEL_MEAN, pixel_std=PIXEL_STD, voxel_size=VOXEL_SIZE, n_scales=N_SCALES,
        voxel_dim_train=voxel_dim_train, voxel_dim_test=voxel_dim_test, origin=[0, 0, 0], backbone2d_stride=4,
        loss_weight_detection=loss_weight_detection, loss_weight_recon=loss_weight_recon,
        voxel_size_fcaf3d=VOXEL_SIZE_FCAF3D, use_batchnorm_train=True, use_batchnorm_test=use_batchnorm_test,
        save_path=save_path, ray_marching_type=ray_marching_type, neus_threshold=neus_threshold, depth_points=depth_points,
        backbone2d=None if hot_path_only else backbone2d_cfg(r50_path),
        feature_2d=None if hot_path_only else feature_2d_cfg(),
        backbone_3d=None if hot_path_only else backbone_3d_cfg(),
        tsdf_head=None if hot_path_only else tsdf_head_cfg(),
        detection_backbone=dict(type='FCAF3DBackbone', in_channels=32, depth=34),
        detection_head=dict(
            type='FCAF3DHead', in_channels=(64, 128, 256, 512), out_channels=128, pts_threshold=200000,
            n_classes=n_classes, n_reg_outs=n_reg_outs, voxel_size=VOXEL_SIZE_FCAF3D,
            assigner=dict(type='FCAF3DAssigner', limit=27, topk=18, n_scales=4),
            loss_bbox=dict(type='IoU3DLoss', loss_weight=1.0, with_yaw=with_yaw),
            train_cfg=dict(), test_cfg=dict(nms_pre=1000, iou_thr=.5, score_thr=.01)),
        max_points=500000, use_feature_transform=True,
        feature_transform=dict(flip_ratio_horizontal=0.5, flip_ratio_vertical=0.5, rot_range=[-0.087266, 0.087266],
                               scale_ratio_range=[.9, 1.1], translation_std=[.1, .1, .1]))
    if middle_save_path is not None:
        model.update(middle_save_path=middle_save_path, middle_visualize_path=middle_visualize_path)
    return model


def _pipeline(space_transform):
    return [dict(type='AtlasResizeImage', size=(640, 480)), dict(type='AtlasToTensor'), space_transform,
            dict(type='AtlasIntrinsicsPoseToProjection'), dict(type='AtlasCollectData')]


def recon_pipelines(voxel_dim_train, voxel_dim_test, pad_xy=1.5, pad_z=.25):
    train = _pipeline(dict(type='AtlasRandomTransformSpaceRecon', voxel_dim=voxel_dim_train, random_rotation=True,
                           random_translation=True, paddingXY=pad_xy, paddingZ=pad_z))
    test = _pipeline(dict(type='AtlasTestTransformSpaceRecon', voxel_dim=voxel_dim_test, origin=[0, 0, 0]))
    return train, test


def detection_pipelines(voxel_dim_train, voxel_dim_test, test_mode):
    train = _pipeline(dict(type='AtlasTransformSpaceDetection', voxel_dim=voxel_dim_train, origin=[0, 0, 0], test=False,
                           mode='middle'))
    test = _pipeline(dict(type='AtlasTransformSpaceDetection', voxel_dim=voxel_dim_test, origin=[0, 0, 0], test=True,
                          mode=test_mode))
    return train, test


def data_cfg(dataset, root, prefix, classes, train_pipeline, test_pipeline, frames_train, frames_test, test_split='val'):
    def split(name, pipeline, test, frames):
        return dict(type=dataset, data_root=root, ann_file=f'{root}/{prefix}_infos_{name}.pkl', classes=classes,
                    pipeline=pipeline, test_mode=test, num_frames=frames, voxel_size=VOXEL_SIZE, select_type='random')
    return dict(samples_per_gpu=1, workers_per_gpu=1, train_dataloader=dict(shuffle=True), test_dataloader=dict(shuffle=False),
                train=split('train', train_pipeline, False, frames_train), val=split('val', test_pipeline, True, frames_test),
                test=split(test_split, test_pipeline, True, frames_test))


def schedule(total_epochs, lr_steps, work_dir, checkpoint_interval=10, optimizer=None, max_norm=10, gamma=None):
    """optimiser / runner / hooks block shared by all stages (default: AdamW 1e-3, step schedule, gradient clipping)"""
    lr_config = dict(policy='step', warmup=None, step=list(lr_steps))
    if gamma is not None:
        lr_config['gamma'] = gamma
    return dict(
        optimizer=optimizer or dict(type='AdamW', lr=0.001, weight_decay=0.0001),
        optimizer_config=dict(grad_clip=dict(max_norm=max_norm, norm_type=2)),
        lr_config=lr_config,
        dist_params=dict(backend='nccl'),            # = RCCL on ROCm
        log_level='INFO', resume_from=None, workflow=[('train', 1)], total_epochs=total_epochs,
        evaluation=dict(interval=3000, voxel_size=VOXEL_SIZE, save_path=work_dir + '/results'),
        runner=dict(type='EpochBasedRunner', max_epochs=total_epochs),
        checkpoint_config=dict(interval=checkpoint_interval),
        log_config=dict(interval=10, hooks=[dict(type='TextLoggerHook')]))
