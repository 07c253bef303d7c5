"""Building blocks of the 2D feature extractor (first half).

The reference ships a trimmed copy of Detectron2's layers (projects/mvsdetection/models/detectron_base.py); these are
the few pieces its ResNet-50 FPN needs, written for this repo on plain torch (MIOpen executes the convolutions on ROCm;
the 2D network is outside the measured hot path -- its OUTPUT, the feature maps, is the hot path's input).
Parameter names follow Detectron2 so that the reference's checkpoints (`fpn.*`, `feature_2d.*`) load unchanged:
a convolution owns `weight` (+ `bias`) and its normalisation lives in the child module `norm`.
"""
import torch
from torch import nn
from torch.nn import functional as F


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine terms: y = x * scale + shift (four BUFFERS named like
    BatchNorm's tensors, so either kind of state dict loads into the other)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)       # present in BatchNorm2d checkpoints only
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @classmethod
    def convert(cls, module):
        """replace every BatchNorm2d below `module` by a frozen copy (ResNetDetectron.freeze, reference resnet.py:408-431)"""
        out = module
        if isinstance(module, nn.modules.batchnorm._BatchNorm):
            out = cls(module.num_features, module.eps)
            if module.affine:
                out.weight.data.copy_(module.weight.data)
                out.bias.data.copy_(module.bias.data)
            out.running_mean.data.copy_(module.running_mean.data)
            out.running_var.data.copy_(module.running_var.data)
        else:
            for name, child in module.named_children():
                new = cls.convert(child)
                if new is not child:
                    setattr(out, name, new)
        return out


def make_norm(norm, channels):
    """'' / None -> no normalisation; 'BN' -> BatchNorm2d; 'FrozenBN'; 'GN' (32 groups)"""
    if not norm:
        return None
    if callable(norm):
        return norm(channels)
    return {"BN": nn.BatchNorm2d, "FrozenBN": FrozenBatchNorm2d, "GN": lambda c: nn.GroupNorm(32, c)}[norm](channels)


class Conv2d(nn.Conv2d):
    """nn.Conv2d followed by an optional normalisation (child `norm`) and activation"""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def msra_fill(m):
    """Caffe2 "MSRAFill": kaiming normal on the fan-out, zero bias"""
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


def xavier_fill(m):
    """Caffe2 "XavierFill": kaiming uniform with a = 1, zero bias"""
    nn.init.kaiming_uniform_(m.weight, a=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)
