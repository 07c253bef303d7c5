"""Feature pyramid over the ResNet stages (registered name `FPNDetectron`; constructor keywords and state-dict keys of
the reference's projects/mvsdetection/models/fpn.py:49-200: `bottom_up.*`, `fpn_lateral{2..5}.*`, `fpn_output{2..5}.*`).
Top-down pathway with nearest-neighbour upsampling; an extra p6 by stride-2 subsampling of p5."""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ..registry import BACKBONES
from .layers2d import Conv2d, make_norm, xavier_fill
from .resnet import ResNetDetectron


@BACKBONES.register_module()
class FPNDetectron(nn.Module):
    def __init__(self, bottom_up_cfg, in_features, out_channels, norm="", fuse_type="sum", pretrained=None):
        super().__init__()
        assert fuse_type in ("sum", "avg")
        self.fp16_enabled = False
        self.bottom_up = ResNetDetectron(**bottom_up_cfg)
        shapes = self.bottom_up.output_shape()
        strides = [shapes[f][1] for f in in_features]
        assert all(b == 2 * a for a, b in zip(strides[:-1], strides[1:])), f"strides {strides} must double level to level"
        self.in_features = tuple(in_features)
        self._stages = [int(math.log2(s)) for s in strides]
        bias = not norm
        for f, stage in zip(in_features, self._stages):
            lateral = Conv2d(shapes[f][0], out_channels, kernel_size=1, bias=bias, norm=make_norm(norm, out_channels))
            output = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=bias,
                            norm=make_norm(norm, out_channels))
            xavier_fill(lateral)
            xavier_fill(output)
            self.add_module(f"fpn_lateral{stage}", lateral)
            self.add_module(f"fpn_output{stage}", output)
        self._fuse_type = fuse_type
        self._out_feature_strides = {f"p{s}": 2 ** s for s in self._stages}
        self._out_feature_strides[f"p{self._stages[-1] + 1}"] = 2 ** (self._stages[-1] + 1)
        self._out_features = list(self._out_feature_strides)
        self.size_divisibility = strides[-1]
        if pretrained is not None:
            self.load_state_dict(torch.load(pretrained, map_location="cpu"))

    def forward(self, x):
        """image batch [N,3,H,W] -> {"p2": ..., ..., "p6": ...} (finest first)"""
        feats = self.bottom_up(x)
        results = {}
        prev = None
        for f, stage in zip(reversed(self.in_features), reversed(self._stages)):
            lateral = getattr(self, f"fpn_lateral{stage}")(feats[f])
            if prev is not None:
                lateral = lateral + F.interpolate(prev, scale_factor=2.0, mode="nearest")
                if self._fuse_type == "avg":
                    lateral = lateral / 2
            prev = lateral
            results[f"p{stage}"] = getattr(self, f"fpn_output{stage}")(prev)
        top = self._stages[-1]
        results[f"p{top + 1}"] = F.max_pool2d(results[f"p{top}"], kernel_size=1, stride=2, padding=0)
        return {k: results[k] for k in self._out_features}
