"""Coarse-to-fine TSDF regression head on the 3D U-Net features (wiring, not a hot-path kernel).

One 1x1x1 conv per scale (coarsest first), tsdf = tanh(.) * label_smoothing; from the second scale on, voxels that the
previous (x2 nearest-upsampled) scale puts further than its `sparse_threshold` from a surface are not regressed but
filled with +-0.999 by the previous sign.  Output keys `scene_tsdf_016 / _008 / _004` (voxel size in cm); the finest
one is the TSDF the ray marching samples (reference ray_marching.py:486).  Training: L1 in log space on the observed /
fully-outside voxels, restricted to the sparsified mask from the second scale on.  Parameter names as the reference's
(`tsdf_head.decoders.{0,1,2}.weight`; reference atlas_head.py:16-81).
"""
import torch
from torch import nn
from torch.nn import functional as F

from ..registry import HEADS


def log_transform(x, shift=1):
    """sign(x) * log(1 + |x| / shift): weights voxels near the surface more than those at the truncation distance"""
    return x.sign() * (1 + x.abs() / shift).log()


@HEADS.register_module()
class AtlasTSDFHead(nn.Module):
    def __init__(self, input_channels, n_scales, voxel_size, label_smoothing, sparse_threshold):
        super().__init__()
        self.fp16_enabled = False
        self.input_channels, self.n_scales, self.voxel_size = input_channels, n_scales, voxel_size
        self.label_smoothing, self.sparse_threshold = label_smoothing, sparse_threshold
        self.voxel_sizes = [voxel_size * 2 ** i for i in reversed(range(n_scales))]          # coarse -> fine
        self.keys = [str(int(v * 100)).zfill(3) for v in self.voxel_sizes]
        self.decoders = nn.ModuleList([nn.Conv3d(c, 1, 1, bias=False) for c in list(input_channels)[::-1]])

    def forward(self, xs, targets=None):
        output, masks = {}, []
        prev = None
        for i, (dec, x) in enumerate(zip(self.decoders, xs)):
            tsdf = torch.tanh(dec(x.float())) * self.label_smoothing
            if prev is not None:
                up = F.interpolate(prev, scale_factor=2).type_as(tsdf)
                near = up.abs() < self.sparse_threshold[i - 1]
                tsdf = torch.where(near, tsdf, up.sign() * 0.999)
                masks.append(near)
            output["scene_tsdf_" + self.keys[i]] = prev = tsdf
        losses = {}
        if targets is not None:
            for i, key in enumerate(self.keys):
                pred, trgt = output["scene_tsdf_" + key], targets["tsdf_gt_" + key]
                use = (trgt < 1) | (trgt == 1).all(-1, keepdim=True)            # observed, or a fully empty column
                if i > 0:
                    use = use & masks[i - 1]
                err = F.l1_loss(log_transform(pred, 1.0), log_transform(trgt, 1.0), reduction="none")
                losses["tsdf_loss_" + key] = err[use].mean() if (i == 0 or bool(use.any())) else 0 * err.sum()
        return output, losses
