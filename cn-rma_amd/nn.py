"""torch.nn modules with the parameter names of the MinkowskiEngine layers the reference builds, executing on the
HIP sparse engine (sparse.py).  State-dict keys match ME's so that the reference's checkpoints load unchanged:
convolutions own `kernel` ([K,Cin,Cout], or [Cin,Cout] when K == 1) and optionally `bias` ([1,Cout]);
MinkowskiBatchNorm owns `bn.*` (an nn.BatchNorm1d); MinkowskiInstanceNorm owns `weight`/`bias` ([1,C]).

Inference (eval mode) fuses conv + BatchNorm + activation (+ residual) into one kernel launch; FusedSequential
does that pattern matching for the reference's nn.Sequential(conv, BN, ELU) blocks without changing key names.
Reference: projects/mvsdetection/models/fcaf3d_backbone.py:14-107, fcaf3d_head.py:61-98 and
MinkowskiEngine/modules/resnet_block.py (BasicBlock; third-party).
"""
import math

import torch
from torch import nn

from . import sparse as S


class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        kv = kernel_size ** 3
        shape = (kv, in_channels, out_channels) if kv > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None
        # ME's default init: uniform(-stdv, stdv), stdv = 1/sqrt(in_channels * kernel_volume)
        stdv = 1.0 / math.sqrt(in_channels * kv)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def train(self, mode=True):
        S.invalidate_weight_cache(self)      # writes through param.data (EMA hooks) bump no version counter
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        S.invalidate_weight_cache(self)
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, scale=None, shift=None, residual=None, act=None):
        if torch.is_grad_enabled() and (x.F.requires_grad or self.kernel.requires_grad) and scale is None and \
                residual is None and act is None:
            # training: differentiable convolution (dgrad / wgrad kernels), bias added by torch
            y = S.conv_autograd(x, self.kernel, self.kernel_size, self.stride)
            if shift is None and self.bias is not None:
                y = S.SparseTensor(y.F + self.bias.view(1, -1), y.cs)
            elif shift is not None:
                y = S.SparseTensor(y.F + shift.view(1, -1), y.cs)
            return y
        if shift is None and self.bias is not None:
            shift = self.bias.view(-1).contiguous()
        if torch.is_grad_enabled() and (x.F.requires_grad or self.kernel.requires_grad):
            # fused epilogue requested while gradients are wanted (eval-mode fine-tuning with frozen BatchNorm, input-gradient
            # analysis): the fused kernel has no backward -- the same result through the differentiable convolution + torch
            y = S.conv_autograd(x, self.kernel, self.kernel_size, self.stride).F
            if scale is not None:
                y = y * scale.view(1, -1)
            if shift is not None:
                y = y + shift.view(1, -1)
            if residual is not None:
                y = y + (residual.F if isinstance(residual, S.SparseTensor) else residual)
            y = torch.relu(y) if act == "relu" else (nn.functional.elu(y) if act == "elu" else y)
            out_cs = x.cs if self.stride == 1 else x.cs.strided(self.stride)
            return S.SparseTensor(y, out_cs)
        return S.conv(x, self.kernel, self.kernel_size, self.stride, scale, shift, residual, act)


class MinkowskiGenerativeConvolutionTranspose(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and kernel_size == 2 and stride == 2 and not bias
        self.kernel = nn.Parameter(torch.empty(8, in_channels, out_channels))
        stdv = 1.0 / math.sqrt(in_channels * 8)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def forward(self, x, scale=None, shift=None, act=None):
        return S.conv_transpose_generative(x, self.kernel, scale, shift, act)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)
        self._folded = {}

    def train(self, mode=True):
        self._folded = {}
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._folded = {}
        return super()._load_from_state_dict(*args, **kwargs)

    def folded(self, bias=None):
        """(scale, shift) of the eval-mode affine map, cached until train()/load_state_dict()/device change."""
        key = (None if bias is None else bias.data_ptr(), self.bn.weight.data_ptr(), self.bn.weight._version,
               self.bn.running_var._version)
        if key not in self._folded:
            with torch.no_grad():
                self._folded = {key: S.fold_bn(self.bn, bias)}
        return self._folded[key]

    def batch_stats(self):
        """True when this layer normalises with batch statistics and updates the running ones.  Frozen BatchNorm inside a
        train-mode network (mmcv's norm_eval walks modules() and calls .eval() on the nn.BatchNorm1d, i.e. on self.bn, or on
        this wrapper) must keep its running statistics: every fused training path asks here first."""
        return self.training and self.bn.training

    def forward(self, x, relu=False, residual=None):
        """relu / residual (training mode): [relu]( bn(x) [+ residual.F] ) in one pass (S.batch_norm_train)"""
        if self.batch_stats():
            return S.SparseTensor(S.batch_norm_train(x.F, self.bn, relu, None if residual is None else residual.F), x.cs)
        if torch.is_grad_enabled() and (self.bn.weight.requires_grad or self.bn.bias.requires_grad):
            # frozen statistics, trainable affine map (fine-tuning): torch's eval-mode batch_norm keeps the gradients of
            # weight / bias, which the cached folded() constants would cut
            y = nn.functional.batch_norm(x.F, self.bn.running_mean, self.bn.running_var, self.bn.weight, self.bn.bias, False, 0.0,
                                         self.bn.eps)
        else:
            scale, shift = self.folded()
            y = x.F * scale + shift
        if residual is not None:
            y = y + residual.F
        y = nn.functional.elu(y) if relu == "elu" else (torch.relu(y) if relu else y)
        return S.SparseTensor(y, x.cs)


class MinkowskiInstanceNorm(nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))
        self.eps = 1e-8

    def forward(self, x, relu=False):
        return S.instance_norm(x, self.weight, self.bias, self.eps, relu)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return S.SparseTensor(torch.relu(x.F), x.cs, None, x.amax)


class MinkowskiELU(nn.Module):
    def forward(self, x):
        return S.SparseTensor(nn.functional.elu(x.F), x.cs)


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size=2, stride=2, dimension=3):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x):
        return S.max_pool(x, self.kernel_size, self.stride)


class MinkowskiPruning(nn.Module):
    def forward(self, x, mask):
        return S.prune(x, mask)


def _act_name(m):
    if isinstance(m, MinkowskiReLU):
        return "relu"
    if isinstance(m, MinkowskiELU):
        return "elu"
    return None


class FusedSequential(nn.Sequential):
    """nn.Sequential whose eval-mode forward fuses [conv | generative-transpose] -> BatchNorm -> ReLU/ELU and
    conv -> InstanceNorm(+ReLU) runs; parameter names are those of a plain nn.Sequential."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            nxt2 = mods[i + 2] if i + 2 < len(mods) else None
            if (not self.training) and isinstance(m, (MinkowskiConvolution, MinkowskiGenerativeConvolutionTranspose)) \
                    and isinstance(nxt, MinkowskiBatchNorm):
                scale, shift = nxt.folded(getattr(m, "bias", None))
                act = _act_name(nxt2)
                x = m(x, scale=scale, shift=shift, act=act)
                i += 3 if act else 2
            elif (not self.training) and isinstance(m, MinkowskiInstanceNorm) and isinstance(nxt, MinkowskiReLU) and \
                    isinstance(nxt2, MinkowskiMaxPooling) and x.cs.n_batch <= 1 and x.F.shape[1] % 4 == 0 and \
                    not (torch.is_grad_enabled() and x.F.requires_grad):
                # the stem: the normalised tensor is never written -- the pooling normalises its candidates on the fly
                x = S.instance_norm_max_pool(x, m.weight, m.bias, m.eps, relu=True, kernel_size=nxt2.kernel_size, stride=nxt2.stride)
                i += 3
            elif self.training and isinstance(m, MinkowskiConvolution) and isinstance(nxt, MinkowskiBatchNorm) and \
                    nxt.batch_stats() and torch.is_grad_enabled():
                act = _act_name(nxt2)                          # conv -> BatchNorm -> [ReLU / ELU]: one autograd node
                x = _conv_bn_act(m, nxt, x, act)
                i += 3 if act else 2
            elif self.training and isinstance(m, MinkowskiBatchNorm) and m.batch_stats() and _act_name(nxt) is not None:
                x = m(x, relu=_act_name(nxt))                  # BatchNorm + ReLU / ELU in one pass (S.batch_norm_train)
                i += 2
            elif isinstance(m, MinkowskiInstanceNorm) and isinstance(nxt, MinkowskiReLU):
                x = m(x, relu=True)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


def _conv_bn_act(conv, norm, x, act=None, residual=None):
    """training mode: conv -> BatchNorm -> [+ residual] -> act as one autograd node (S.conv_bn_act_train) for bias-free
    convolutions whose BatchNorm runs on batch statistics; the module-by-module composition otherwise (a frozen BatchNorm --
    norm.eval() / norm.bn.eval() inside a train-mode block -- normalises with its running statistics and leaves them alone)"""
    if conv.bias is None and isinstance(conv, MinkowskiConvolution) and norm.batch_stats():
        return S.conv_bn_act_train(x, conv.kernel, norm.bn, conv.kernel_size, conv.stride, act, residual)
    return norm(conv(x), relu=act if act else False, residual=residual)


class BasicBlock(nn.Module):
    """ME.modules.resnet_block.BasicBlock: conv1(k3, stride) - norm1 - ReLU - conv2(k3) - norm2 - (+ shortcut) - ReLU.
    Three launches in eval mode (two when there is no downsample branch)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=3):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        if self.training:                                  # BatchNorm fused with the ReLU / the shortcut add + ReLU behind it
            out = _conv_bn_act(self.conv1, self.norm1, x, "relu")
            res = self.downsample(x) if self.downsample is not None else x
            return _conv_bn_act(self.conv2, self.norm2, out, "relu", res)
        s1, b1 = self.norm1.folded()
        out = self.conv1(x, scale=s1, shift=b1, act="relu")
        res = self.downsample(x) if self.downsample is not None else x
        s2, b2 = self.norm2.folded()
        return self.conv2(out, scale=s2, shift=b2, residual=res, act="relu")


class Bottleneck(nn.Module):
    """ME.modules.resnet_block.Bottleneck (MinkowskiEngine v0.5.4, third party, not under /root/reference; the reference
    selects it for depth 50 / 101, fcaf3d_backbone.py:122-127): conv1(k1) - norm1 - ReLU - conv2(k3, stride) - norm2 - ReLU -
    conv3(k1, planes -> 4 * planes) - norm3 - (+ shortcut) - ReLU; attribute names conv1..3, norm1..3, downsample.
    Eval mode: the three BatchNorms fold into the convolution epilogues -- three launches (four with the shortcut branch)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=3):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        if self.training:
            out = _conv_bn_act(self.conv1, self.norm1, x, "relu")
            out = _conv_bn_act(self.conv2, self.norm2, out, "relu")
            res = self.downsample(x) if self.downsample is not None else x
            return _conv_bn_act(self.conv3, self.norm3, out, "relu", res)
        s1, b1 = self.norm1.folded()
        out = self.conv1(x, scale=s1, shift=b1, act="relu")
        s2, b2 = self.norm2.folded()
        out = self.conv2(out, scale=s2, shift=b2, act="relu")
        res = self.downsample(x) if self.downsample is not None else x
        s3, b3 = self.norm3.folded()
        return self.conv3(out, scale=s3, shift=b3, residual=res, act="relu")

