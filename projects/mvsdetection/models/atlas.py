"""Reconstruction-only detector (registered name `Atlas`; constructor keywords and contracts of the reference's
projects/mvsdetection/models/atlas.py:71-405, the first training stage of CN-RMA): posed images -> 2D features -> dense
unprojection + accumulate (the HIP kernel of the hot path) -> Atlas 3D U-Net -> coarse-to-fine
TSDF.  forward_test writes {save_path}/{scene}/{scene}.npz (+ .ply with scikit-image / trimesh) and returns [{}]."""
from ..registry import DETECTORS
from .multiview_base import MultiViewBase


@DETECTORS.register_module()
class Atlas(MultiViewBase):
    def __init__(self, pixel_mean, pixel_std, voxel_size, n_scales, voxel_dim_train, voxel_dim_test, origin,
                 backbone2d_stride, backbone2d, feature_2d, backbone_3d, tsdf_head, save_path, train_cfg=None,
                 test_cfg=None, pretrained=None):
        super().__init__(pixel_mean, pixel_std, voxel_size, n_scales, voxel_dim_train, voxel_dim_test, origin,
                         backbone2d_stride, backbone2d, feature_2d, backbone_3d, tsdf_head, save_path)
        assert self.backbone3d is not None and self.tsdf_head is not None, "Atlas is the 3D reconstruction network"
        self.initialize_volume()

    def inference1(self, projection, image=None, feature=None):
        """one view: 2D features of `image` (unless given) recorded for the accumulation (reference :120-153)"""
        if feature is None:
            feature = self.backbone2d(self.normalizer(image))
        self.aggregate_2d_features(projection, feature)

    def inference2(self, targets=None):
        """mean volume over the recorded views -> 3D network -> TSDF levels (+ losses against `targets`)"""
        self.clear_3d_features()
        return self.tsdf_head(self.backbone3d(self.volume), targets)

    def _run(self, inputs, test):
        self.voxel_dim = self.voxel_dim_test if test else self.voxel_dim_train
        self.initialize_volume()
        projections = inputs["projection"].transpose(0, 1)
        features = self._features(inputs, batched=not test)      # training shares the BatchNorm statistics over the views
        for projection, feature in zip(projections, features):
            self.inference1(projection, feature=feature)
        return self.inference2(inputs.get("tsdf_list") or None)

    def forward_train(self, inputs):
        return self._run(inputs, test=False)[1]

    def forward_test(self, inputs):
        outputs, losses = self._run(inputs, test=True)
        self.last_losses = losses
        if self.save_path is not None:
            self.save_reconstruction(outputs, inputs)
        return [{}]
