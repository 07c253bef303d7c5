"""ORACLE (test infrastructure only) -- the sparse half once more, in fp32 on torch's CPU
backend, for the `cpu_baseline` leg of bench.py: the same operator semantics as oracle/sparse_oracle.py (which stays the
parity checker: float64, numpy) but multi-threaded (torch index / GEMM kernels use every host core) and with the
neighbour look-ups of a coordinate set shared by the convolutions that run on it -- what an optimised CPU
implementation of the reference's MinkowskiEngine path would also do.  PARITY UNPINNED like sparse_oracle.py (ME is not
available); pinned to it instead: tests/test_sparse_oracle_cpu.py::test_fp32_torch_port_matches_the_fp64_oracle.

Structure: fcaf3d_backbone.py:14-107, fcaf3d_head.py:61-139, :238-349 of the reference.
Only tests/ and bench.py's cpu_baseline leg may import this module.
"""
import itertools

import torch


def _key(c):
    c = c.to(torch.int64)
    return ((c[:, 0] + 8) << 54) | ((c[:, 1] + (1 << 17)) << 36) | ((c[:, 2] + (1 << 17)) << 18) | (c[:, 3] + (1 << 17))


def kernel_offsets(kernel_size, tensor_stride):
    k = kernel_size
    rng = [(i - k // 2) if k % 2 == 1 else i for i in range(k)]
    return torch.tensor([(ix, iy, iz) for iz, iy, ix in itertools.product(rng, rng, rng)], dtype=torch.int64) * tensor_stride


class CoordSet:
    """coordinates int64 [N,4] + sorted keys; neighbour tables cached per (kernel, stride, output set)"""

    def __init__(self, coords, stride):
        self.C = coords.to(torch.int64)
        self.stride = stride
        k = _key(self.C)
        self.sorted, self.order = torch.sort(k)
        self._nbr = {}
        self._strided = {}

    def __len__(self):
        return self.C.shape[0]

    def lookup(self, q):
        k = _key(q)
        pos = torch.searchsorted(self.sorted, k).clamp(max=len(self.sorted) - 1)
        hit = self.sorted[pos] == k
        return torch.where(hit, self.order[pos], torch.full_like(pos, -1))

    def strided(self, factor=2):
        """unique(floor(p / s') * s') in first-occurrence order"""
        ns = self.stride * factor
        if ns not in self._strided:
            c = self.C.clone()
            c[:, 1:] = torch.div(c[:, 1:], ns, rounding_mode="floor") * ns
            k = _key(c)
            uniq, inv = torch.unique(k, return_inverse=True)
            first = torch.full((len(uniq),), len(k), dtype=torch.int64).scatter_reduce(0, inv, torch.arange(len(k)), "amin")
            self._strided[ns] = CoordSet(c[torch.sort(first)[0]], ns)
        return self._strided[ns]

    def neighbours(self, out_set, kernel_size, offset_stride):
        """list over kernel offsets of (out_rows, in_rows) index pairs"""
        key = (kernel_size, offset_stride, id(out_set))
        if key not in self._nbr:
            pairs = []
            for off in kernel_offsets(kernel_size, offset_stride):
                q = out_set.C.clone()
                q[:, 1:] += off
                idx = self.lookup(q)
                o = torch.nonzero(idx >= 0).view(-1)
                pairs.append((o, idx[o]))
            self._nbr[key] = (pairs, out_set)
        return self._nbr[key][0]


def conv(cs, F, weight, kernel_size, stride=1, out_set=None):
    W = weight.detach().float()
    if W.dim() == 2:
        W = W.unsqueeze(0)
    if out_set is None:
        out_set = cs if stride == 1 else cs.strided(stride)
    out = torch.zeros((len(out_set), W.shape[2]), dtype=torch.float32)
    if kernel_size == 1 and stride == 1:
        return out_set, F @ W[0]
    for k, (o, i) in enumerate(cs.neighbours(out_set, kernel_size, cs.stride)):
        if len(o):
            out.index_add_(0, o, F.index_select(0, i) @ W[k])
    return out_set, out


def conv_transpose_generative(cs, F, weight):
    half = cs.stride // 2
    W = weight.detach().float()
    oc, of = [], []
    for k, off in enumerate(kernel_offsets(2, half)):
        c = cs.C.clone()
        c[:, 1:] += off
        oc.append(c)
        of.append(F @ W[k])
    return CoordSet(torch.cat(oc), half), torch.cat(of)


def max_pool(cs, F):
    out_set = cs.strided(2)
    out = torch.full((len(out_set), F.shape[1]), float("-inf"))
    for o, i in cs.neighbours(out_set, 2, cs.stride):
        if len(o):
            out[o] = torch.maximum(out[o], F.index_select(0, i))
    return out_set, out


def _bn(F, bn):
    s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    return F * s + (bn.bias.detach() - bn.running_mean * s)


def instance_norm(F, weight, bias, eps=1e-8):
    mean = F.mean(0, keepdim=True)
    var = ((F - mean) ** 2).mean(0, keepdim=True)
    return (F - mean) / torch.sqrt(var + eps) * weight.detach().view(1, -1) + bias.detach().view(1, -1)


def union_add(ca, fa, cb, fb):
    idx = ca.lookup(cb.C)
    m = idx >= 0
    f = fa.clone()
    f.index_add_(0, idx[m], fb[m])
    return CoordSet(torch.cat((ca.C, cb.C[~m])), ca.stride), torch.cat((f, fb[~m]))


def interpolate(score_set, score, query):
    s = score_set.stride
    base = query.clone()
    base[:, 1:] = torch.div(query[:, 1:], s, rounding_mode="floor") * s
    out = torch.zeros(len(query))
    val = score.view(-1)
    for off in kernel_offsets(2, s):
        c = base.clone()
        c[:, 1:] += off
        w = torch.prod(1.0 - (query[:, 1:] - c[:, 1:]).abs().float() / s, dim=1)
        idx = score_set.lookup(c)
        m = idx >= 0
        out[m] += w[m] * val[idx[m]]
    return out


def _basic_block(cs, F, blk):
    stride = blk.conv1.stride
    oc, o = conv(cs, F, blk.conv1.kernel, 3, stride)
    o = torch.relu(_bn(o, blk.norm1.bn))
    _, o = conv(oc, o, blk.conv2.kernel, 3, 1)
    o = _bn(o, blk.norm2.bn)
    if blk.downsample is not None:
        _, r = conv(cs, F, blk.downsample[0].kernel, 1, stride, out_set=oc)
        r = _bn(r, blk.downsample[1].bn)
    else:
        r = F
    return oc, torch.relu(o + r)


@torch.no_grad()
def backbone_forward(backbone, coords, feats):
    cs, F = CoordSet(torch.as_tensor(coords), 1), torch.as_tensor(feats).float()
    stem = backbone.conv1
    cs, F = conv(cs, F, stem[0].kernel, 3, 2)
    F = torch.relu(instance_norm(F, stem[1].weight, stem[1].bias))
    cs, F = max_pool(cs, F)
    outs = []
    for i in range(backbone.n_outs):
        for blk in getattr(backbone, f"layer{i + 1}"):
            cs, F = _basic_block(cs, F, blk)
        outs.append((cs, F))
    return outs


def _conv_bn_elu(cs, F, conv_m, bn_m):
    _, o = conv(cs, F, conv_m.kernel, 3, 1)
    return torch.nn.functional.elu(_bn(o, bn_m.bn))


@torch.no_grad()
def head_forward(head, levels):
    results = [None] * len(levels)
    x = scores = None
    for i in range(len(levels) - 1, -1, -1):
        ci, fi = levels[i]
        if i == len(levels) - 1:
            cs, F = ci, fi
        else:
            up = getattr(head, f"up_block_{i + 1}")
            uc, uf = conv_transpose_generative(x[0], x[1], up[0].kernel)
            uf = torch.nn.functional.elu(_bn(uf, up[1].bn))
            uf = _conv_bn_elu(uc, uf, up[3], up[4])
            cs, F = union_add(ci, fi, uc, uf)
            if 0 <= head.pts_threshold < len(cs):
                interp = interpolate(scores[0], scores[1], cs.C)
                keep = torch.sort(torch.topk(interp, head.pts_threshold)[1])[0]
                cs, F = CoordSet(cs.C[keep], cs.stride), F[keep]
        x = (cs, F)
        ob = getattr(head, f"out_block_{i}")
        o = _conv_bn_elu(cs, F, ob[0], ob[1])
        ctr = o @ head.centerness_conv.kernel.detach()
        cls = o @ head.cls_conv.kernel.detach() + head.cls_conv.bias.detach()
        reg = o @ head.reg_conv.kernel.detach()
        bbox = torch.cat((torch.exp(reg[:, :6] * float(head.scales[i].scale.detach())), reg[:, 6:]), dim=1)
        scores = (cs, cls.max(dim=1, keepdim=True)[0])
        results[i] = dict(coords=cs.C, centerness=ctr, bbox_pred=bbox, cls_score=cls, points=cs.C[:, 1:].float() * head.voxel_size)
    return results


@torch.no_grad()
def get_bboxes(head, results):
    from .sparse_oracle import decode_boxes
    boxes, scores = [], []
    for r in results:
        s = torch.sigmoid(r["cls_score"]) * torch.sigmoid(r["centerness"])
        bp, pt = r["bbox_pred"], r["points"]
        k = head.test_cfg.nms_pre
        if len(s) > k > 0:
            ids = s.max(dim=1)[0].topk(k)[1]
            bp, s, pt = bp[ids], s[ids], pt[ids]
        boxes.append(decode_boxes(pt, bp, head.yaw_parametrization))
        scores.append(s)
    return torch.cat(boxes), torch.cat(scores)
