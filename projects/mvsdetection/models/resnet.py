"""ResNet bottom-up network of the 2D feature extractor (registered name `ResNetDetectron`, constructor keywords and
state-dict keys of the reference's projects/mvsdetection/models/resnet.py:236-431, itself Detectron2's ResNet:
`stem.conv1.{weight,norm.*}`, `res2.0.{shortcut,conv1,conv2,conv3}.{weight,norm.*}` ...).  Plain torch (MIOpen)."""
from torch import nn
from torch.nn import functional as F

from ..registry import BACKBONES
from .layers2d import Conv2d, FrozenBatchNorm2d, make_norm, msra_fill

_BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


class _Block(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        FrozenBatchNorm2d.convert(self)
        return self


class BasicBlock(_Block):
    """two 3x3 convolutions (ResNet-18/34)"""

    def __init__(self, in_channels, out_channels, *, stride=1, norm="BN"):
        super().__init__(in_channels, out_channels, stride)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=make_norm(norm, out_channels))
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False,
                            norm=make_norm(norm, out_channels))
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                            norm=make_norm(norm, out_channels))
        for m in (self.conv1, self.conv2, self.shortcut):
            if m is not None:
                msra_fill(m)

    def forward(self, x):
        y = self.conv2(F.relu_(self.conv1(x)))
        return F.relu_(y + (x if self.shortcut is None else self.shortcut(x)))


class BottleneckBlock(_Block):
    """1x1 - 3x3 - 1x1 with a projection shortcut where the width changes (ResNet-50 and deeper)"""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1):
        super().__init__(in_channels, out_channels, stride)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=make_norm(norm, out_channels))
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=s1, bias=False,
                            norm=make_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=s3, padding=dilation, bias=False,
                            groups=num_groups, dilation=dilation, norm=make_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False, norm=make_norm(norm, out_channels))
        for m in (self.conv1, self.conv2, self.conv3, self.shortcut):
            if m is not None:
                msra_fill(m)

    def forward(self, x):
        y = self.conv3(F.relu_(self.conv2(F.relu_(self.conv1(x)))))
        return F.relu_(y + (x if self.shortcut is None else self.shortcut(x)))


class BasicStem(_Block):
    """7x7 stride-2 convolution, ReLU, 3x3 stride-2 max pool: stride 4"""

    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False,
                            norm=make_norm(norm, out_channels))
        msra_fill(self.conv1)

    def forward(self, x):
        return F.max_pool2d(F.relu_(self.conv1(x)), kernel_size=3, stride=2, padding=1)


@BACKBONES.register_module()
class ResNetDetectron(nn.Module):
    def __init__(self, input_channels=3, norm="BN", depth=50, out_features=("res2", "res3", "res4", "res5"), num_groups=1,
                 width_per_group=64, stride_in_1x1=True, res5_dilation=1, res2_out_channels=256, stem_out_channels=64,
                 freeze_at=0, num_classes=None):
        super().__init__()
        assert depth in _BLOCKS and res5_dilation in (1, 2) and num_classes is None
        self.fp16_enabled = False
        self.stem = BasicStem(input_channels, stem_out_channels, norm)
        self._strides, self._channels = {"stem": 4}, {"stem": stem_out_channels}
        self.stage_names = []
        in_ch, out_ch, mid = stem_out_channels, res2_out_channels, num_groups * width_per_group
        stride = 4
        for i, n_blocks in enumerate(_BLOCKS[depth]):
            dilation = res5_dilation if i == 3 else 1
            first = 1 if (i == 0 or (i == 3 and dilation == 2)) else 2
            blocks = []
            for b in range(n_blocks):
                s = first if b == 0 else 1
                if depth < 50:
                    blocks.append(BasicBlock(in_ch, out_ch, stride=s, norm=norm))
                else:
                    blocks.append(BottleneckBlock(in_ch, out_ch, bottleneck_channels=mid, stride=s, num_groups=num_groups,
                                                  norm=norm, stride_in_1x1=stride_in_1x1, dilation=dilation))
                in_ch = out_ch
            name = f"res{i + 2}"
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
            stride *= first
            self._strides[name], self._channels[name] = stride, out_ch
            out_ch, mid = out_ch * 2, mid * 2
        self._out_features = list(out_features) if out_features is not None else [self.stage_names[-1]]
        last = max(self.stage_names.index(f) for f in self._out_features if f != "stem") if self._out_features != ["stem"] else -1
        for name in self.stage_names[last + 1:]:                 # stages nobody reads are not built into the module
            delattr(self, name)
        self.stage_names = self.stage_names[:last + 1]
        self.freeze(freeze_at)

    def output_shape(self):
        """name -> (channels, stride) of every returned feature map"""
        return {n: (self._channels[n], self._strides[n]) for n in self._out_features}

    def freeze(self, freeze_at=0):
        """freeze the stem (freeze_at >= 1) and the first freeze_at - 1 residual stages: no gradients, BatchNorm fixed"""
        if freeze_at >= 1:
            self.stem.freeze()
        for idx, name in enumerate(self.stage_names, start=2):
            if freeze_at >= idx:
                for block in getattr(self, name):
                    block.freeze()
        return self

    def forward(self, x):
        assert x.dim() == 4, f"ResNet takes [N,C,H,W], got {tuple(x.shape)}"
        out = {}
        x = self.stem(x)
        if "stem" in self._out_features:
            out["stem"] = x
        for name in self.stage_names:
            x = getattr(self, name)(x)
            if name in self._out_features:
                out[name] = x
        return out
