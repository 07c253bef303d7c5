"""Pyramid -> one stride-4 feature map (registered name `AtlasFPNFeature`; keywords and keys of the reference's
projects/mvsdetection/models/backbone2d.py:28-67: levels p2..p5 each go through (3x3 conv - norm - ReLU [- 2x bilinear
upsampling]) as often as needed to reach `output_stride`, and are summed; modules `p2.0`, `p3.0`, `p4.0`, `p4.2`, ...).
Its output [N, output_dim, H/4, W/4] is the feature-map input of the hot path (cnrma_amd.rma)."""
import math

from torch import nn
from torch.nn import functional as F

from ..registry import BACKBONES
from .layers2d import Conv2d, make_norm, msra_fill


@BACKBONES.register_module()
class AtlasFPNFeature(nn.Module):
    LEVELS = ("p2", "p3", "p4", "p5")

    def __init__(self, feature_strides, feature_channels, output_dim=32, output_stride=4, norm="BN"):
        super().__init__()
        self.fp16_enabled = False
        for level in self.LEVELS:
            steps = max(1, int(math.log2(feature_strides[level]) - math.log2(output_stride)))
            ops = []
            for k in range(steps):
                conv = Conv2d(feature_channels[level] if k == 0 else output_dim, output_dim, kernel_size=3, stride=1, padding=1,
                              bias=not norm, norm=make_norm(norm, output_dim), activation=F.relu)
                msra_fill(conv)
                ops.append(conv)
                if feature_strides[level] != output_stride:
                    ops.append(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
            self.add_module(level, nn.Sequential(*ops))

    def forward(self, features):
        x = None
        for level in self.LEVELS:
            y = getattr(self, level)(features[level])
            x = y if x is None else x + y
        return x
