"""Dense 3D U-Net between the two halves of the hot path (wiring, not a hot-path kernel).

`AtlasBackbone3D` refines the mean feature volume of the dense unprojection into multi-scale features for the TSDF
head whose finest output (`scene_tsdf_004`) drives the ray marching.  It is stock dense convolution work -- torch
Conv3d / BatchNorm3d, i.e. MIOpen on ROCm -- provided so that the `ray_marching_*` configs build with their
`backbone_3d` entry.  Sub-module and parameter names are the reference's (checkpoint keys `backbone3d.layers_down.1.0
.weight`, `...layers_up_res.0.1.bn2.running_var`, `...proj.0.conv.weight`; reference backbone3d.py:127-201), the
code is written from the architecture description:

  encoder level 0: `layers_down[0]` residual blocks at full resolution; level i > 0: 3x3x3 stride-2 conv + norm +
  ReLU, then residual blocks.  decoder step i: trilinear x2, 1x1x1 conv to the skip's width, projected skip
  (`proj[i]`: 1x1x1 conv [optionally kept only where the input volume was observed] + norm + ReLU), mean of the two,
  residual blocks.  Returns the decoder features coarse-to-fine.
"""
import torch
from torch import nn
from torch.nn import functional as F

from ..registry import BACKBONES

_NORMS = {"BN": nn.BatchNorm3d, "GN": lambda c: nn.GroupNorm(32, c), "nnSyncBN": nn.SyncBatchNorm}


def make_norm(kind, channels):
    """'BN' | 'GN' | 'nnSyncBN' | '' (none) | a callable channels -> module"""
    if callable(kind):
        return kind(channels)
    return _NORMS[kind](channels) if kind else None


def _conv(cin, cout, k, stride=1):
    return nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock3d(nn.Module):
    """conv3-norm-ReLU-conv3-norm + identity, ReLU (dropout slots kept for key compatibility; p = `drop`)"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, drop=0, norm="BN"):
        super().__init__()
        self.conv1, self.bn1, self.drop1 = _conv(inplanes, planes, 3, stride), make_norm(norm, planes), nn.Dropout(drop, True)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2, self.drop2 = _conv(planes, planes, 3), make_norm(norm, planes), nn.Dropout(drop, True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.drop1(self.bn1(self.conv1(x))))
        y = self.drop2(self.bn2(self.conv2(y)))
        y += x if self.downsample is None else self.downsample(x)
        return self.relu(y)


class ConditionalProjection(nn.Module):
    """skip connection: ReLU(norm(where(mask, conv1(enc), dec))) when `condition`, else ReLU(norm(conv1(enc)))"""

    def __init__(self, n, norm="BN", condition=True):
        super().__init__()
        self.conv, self.norm, self.relu = _conv(n, n, 1), make_norm(norm, n), nn.ReLU(True)
        self.condition = condition

    def forward(self, enc, dec, mask):
        s = self.conv(enc)
        if self.condition:
            s = torch.where(mask, s, dec)
        return self.relu(self.norm(s))


@BACKBONES.register_module()
class AtlasBackbone3D(nn.Module):
    def __init__(self, channels=(32, 64, 128), layers_down=(1, 2, 3), layers_up=(3, 3, 3), drop=0,
                 zero_init_residual=True, cond_proj=True, norm="BN"):
        super().__init__()
        self.cond_proj = cond_proj
        L = len(channels)

        def blocks(c, n):
            return [BasicBlock3d(c, c, drop=drop, norm=norm) for _ in range(n)]

        downs, projs = [], []
        for i, c in enumerate(channels):
            head = [] if i == 0 else [_conv(channels[i - 1], c, 3, 2), make_norm(norm, c), nn.Dropout(drop, True),
                                      nn.ReLU(inplace=True)]
            downs.append(nn.Sequential(*head, *blocks(c, layers_down[i])))
            if i < L - 1:
                projs.append(ConditionalProjection(c, norm, cond_proj))
        self.layers_down = nn.ModuleList(downs)
        self.proj = nn.ModuleList(projs[::-1])                      # decoder order: deepest skip first
        up = list(channels)[::-1]
        self.layers_up_conv = nn.ModuleList([_conv(up[j - 1], up[j], 1) for j in range(1, L)])
        self.layers_up_res = nn.ModuleList([nn.Sequential(*blocks(up[j], layers_up[j - 1])) for j in range(1, L)])
        if zero_init_residual:                                      # residual branches start as identities
            for m in self.modules():
                if isinstance(m, BasicBlock3d):
                    nn.init.constant_(m.bn2.weight, 0)

    def forward(self, x):
        observed = (x != 0).any(1, keepdim=True).float() if self.cond_proj else None
        skips = []
        for layer in self.layers_down:
            x = layer(x)
            skips.append(x)
        skips.reverse()
        steps = len(self.layers_up_conv)
        out = []
        for i in range(steps):
            x = self.layers_up_conv[i](F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False))
            mask = None
            if self.cond_proj:
                mask = F.interpolate(observed, scale_factor=1 / 2 ** (steps - i - 1)) != 0
            x = self.layers_up_res[i]((x + self.proj[i](skips[i + 1], x, mask)) / 2)
            out.append(x)
        return out
