"""FCAF3D sparse backbone (MinkResNet) on the HIP sparse engine.

Same registered name, constructor and output contract as the reference's
projects/mvsdetection/models/fcaf3d_backbone.py:14-130 (ResNetBase / FCAF3DBackbone); the layers come from
cnrma_amd.nn (ME-compatible parameter names: conv1.0.kernel, layer1.0.conv1.kernel, layer1.0.norm1.bn.weight,
layer1.0.downsample.0.kernel, ...), so detection_backbone.* checkpoint keys load unchanged."""
from torch import nn

from cnrma_amd import nn as snn

from ..registry import BACKBONES

# depth -> (block, blocks per stage)   (reference :112-127; the Bottleneck depths widen the outputs to 256 / 512 / 1024 / 2048)
_DEPTHS = {14: (snn.BasicBlock, (1, 1, 1, 1)), 18: (snn.BasicBlock, (2, 2, 2, 2)), 34: (snn.BasicBlock, (3, 4, 6, 3)),
           50: (snn.Bottleneck, (4, 3, 6, 3)), 101: (snn.Bottleneck, (3, 4, 23, 3))}


@BACKBONES.register_module()
class FCAF3DBackbone(nn.Module):
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)

    def __init__(self, in_channels, depth, n_outs=4):
        super().__init__()
        if depth not in _DEPTHS:
            raise ValueError(f"invalid depth={depth}")
        self.block, layers = _DEPTHS[depth]
        self.fp16_enabled = False
        self.n_outs = n_outs
        self.inplanes = self.INIT_DIM
        # stem: conv k3 s2 -> InstanceNorm -> ReLU -> MaxPool k2 s2   (reference :25-32)
        self.conv1 = snn.FusedSequential(
            snn.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=3, stride=2, dimension=3),
            snn.MinkowskiInstanceNorm(self.inplanes),
            snn.MinkowskiReLU(inplace=True),
            snn.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=3))
        for i in range(n_outs):
            setattr(self, f"layer{i + 1}", self._make_layer(self.PLANES[i], layers[i], stride=2))

    def _make_layer(self, planes, blocks, stride):
        """first block strided with a 1x1 strided conv + BN shortcut, then `blocks-1` plain blocks (reference :59-87)"""
        block, out_planes = self.block, planes * self.block.expansion
        downsample = None
        if stride != 1 or self.inplanes != out_planes:
            downsample = snn.FusedSequential(
                snn.MinkowskiConvolution(self.inplanes, out_planes, kernel_size=1, stride=stride, dimension=3),
                snn.MinkowskiBatchNorm(out_planes))
        layers = [block(self.inplanes, planes, stride=stride, downsample=downsample)]
        self.inplanes = out_planes
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def init_weights(self):
        """kaiming-normal (fan_out, relu) on conv kernels, BN affine = (1, 0)  (reference :50-57)"""
        for m in self.modules():
            if isinstance(m, snn.MinkowskiConvolution):
                fan_out = m.out_channels * m.kernel_size ** 3
                nn.init.normal_(m.kernel, std=(2.0 / fan_out) ** 0.5)
            if isinstance(m, snn.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def forward(self, x):
        outs = []
        # the coordinate sets of all stride-2 steps (stem conv, max pool, one per layer) depend on the input sites only:
        # build the whole chain now with a single device->host read of the row counts
        x.cs.prefetch_strided(2 + self.n_outs)
        x = self.conv1(x)
        for i in range(self.n_outs):
            x = getattr(self, f"layer{i + 1}")(x)
            outs.append(x)
        return outs
