"""ORACLE (test infrastructure only) -- CPU restatement of the sparse half.

PARITY UNPINNED for a9-a11: MinkowskiEngine v0.5.4 (pinned by the reference's README.md:18 / doc/install.md:49-58)
is not vendored under /root/reference and cannot be built offline, and the reference has no tests.  The operator
semantics restated here are those of SURVEY.md Appendix A (upstream ME behaviour); what pins this file instead is
(i) the dense equivalence checked in tests/test_sparse_oracle_cpu.py -- a sparse convolution must equal
torch.nn.functional.conv3d on the densified tensor sampled at the active output sites -- and (ii) known-answer
cases.  The box decoder (a12) IS pinned: tests/golden/decode.npz comes from the reference's fcaf3d_head.py.

Structure follows the reference's files: fcaf3d_backbone.py (ResNetBase :14-107), fcaf3d_head.py (:61-139,
:238-349) and ray_marching.py:322-336.  Plain numpy / torch-CPU, float64 accumulation in the convolutions.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import itertools

import numpy as np
import torch


def _key(c):
    c = np.asarray(c, dtype=np.int64)
    return ((c[:, 0] + 8) << 54) | ((c[:, 1] + (1 << 17)) << 36) | ((c[:, 2] + (1 << 17)) << 18) | (c[:, 3] + (1 << 17))


class Lookup:
    """coordinate -> row index (or -1) through a sorted key array."""

    def __init__(self, coords):
        k = _key(coords)
        self.order = np.argsort(k, kind="stable")
        self.sorted = k[self.order]
        assert len(np.unique(k)) == len(k), "coordinates must be unique"

    def __call__(self, coords):
        k = _key(coords)
        pos = np.searchsorted(self.sorted, k)
        pos = np.clip(pos, 0, len(self.sorted) - 1) if len(self.sorted) else np.zeros_like(pos)
        hit = (self.sorted[pos] == k) if len(self.sorted) else np.zeros(len(k), dtype=bool)
        return np.where(hit, self.order[pos] if len(self.sorted) else -1, -1)


def kernel_offsets(kernel_size, tensor_stride):
    """ME region order: x fastest; odd kernels centred, even kernels 0..k-1."""
    k = kernel_size
    rng = [(i - k // 2) if k % 2 == 1 else i for i in range(k)]
    return np.array([(ix, iy, iz) for iz, iy, ix in itertools.product(rng, rng, rng)], dtype=np.int64) * tensor_stride


def unique_first(coords):
    """rows of the first occurrence of every distinct coordinate, in first-occurrence order."""
    k = _key(coords)
    _, first = np.unique(k, return_index=True)
    return np.sort(first)


def stride_coords(coords, new_stride):
    c = np.asarray(coords, dtype=np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], new_stride) * new_stride
    return c[unique_first(c)]


def conv(coords, feats, weight, kernel_size, stride, tensor_stride, out_coords=None):
    """MinkowskiConvolution: out[p] = sum_k in[p + off_k] @ W[k] over existing inputs.
    stride 2: output sites floor(p/(2s))*2s, offsets (-1,0,1)*s around the output site."""
    coords = np.asarray(coords, dtype=np.int64)
    W = np.asarray(weight, dtype=np.float64)
    if W.ndim == 2:
        W = W[None]
    if out_coords is None:
        out_coords = coords if stride == 1 else stride_coords(coords, tensor_stride * stride)
    look = Lookup(coords)
    F = np.asarray(feats, dtype=np.float64)
    out = np.zeros((len(out_coords), W.shape[2]))
    for k, off in enumerate(kernel_offsets(kernel_size, tensor_stride)):
        q = out_coords.copy()
        q[:, 1:] += off
        idx = look(q)
        m = idx >= 0
        if m.any():
            out[m] += F[idx[m]] @ W[k]
    return out_coords, out


def conv_backward(coords, feats, weight, grad_out, kernel_size, stride, tensor_stride, out_coords=None):
    """gradients of conv() w.r.t. feats and weight (fp64): grad_in[i] += G[o] @ W[k]^T, gradW[k] += F[i]^T @ G[o] for
    every pair (i, o) of offset k"""
    coords = np.asarray(coords, dtype=np.int64)
    W = np.asarray(weight, dtype=np.float64)
    W3 = W[None] if W.ndim == 2 else W
    if out_coords is None:
        out_coords = coords if stride == 1 else stride_coords(coords, tensor_stride * stride)
    look = Lookup(coords)
    F = np.asarray(feats, dtype=np.float64)
    G = np.asarray(grad_out, dtype=np.float64)
    gF, gW = np.zeros_like(F), np.zeros_like(W3)
    for k, off in enumerate(kernel_offsets(kernel_size, tensor_stride)):
        q = out_coords.copy()
        q[:, 1:] += off
        idx = look(q)
        m = idx >= 0
        if m.any():
            np.add.at(gF, idx[m], G[m] @ W3[k].T)
            gW[k] += F[idx[m]].T @ G[m]
    return gF, gW.reshape(W.shape)


def conv_transpose_generative(coords, feats, weight, tensor_stride):
    """k=2 s=2 generative transpose: out[p + off_k*(s/2)] = in[p] @ W[k]; rows ordered k-major (k*N + i)."""
    coords = np.asarray(coords, dtype=np.int64)
    half = tensor_stride // 2
    F = np.asarray(feats, dtype=np.float64)
    W = np.asarray(weight, dtype=np.float64)
    oc, of = [], []
    for k, off in enumerate(kernel_offsets(2, half)):
        c = coords.copy()
        c[:, 1:] += off
        oc.append(c)
        of.append(F @ W[k])
    return np.concatenate(oc), np.concatenate(of)


def max_pool(coords, feats, tensor_stride):
    coords = np.asarray(coords, dtype=np.int64)
    out_coords = stride_coords(coords, tensor_stride * 2)
    look = Lookup(coords)
    F = np.asarray(feats, dtype=np.float64)
    out = np.full((len(out_coords), F.shape[1]), -np.inf)
    for off in kernel_offsets(2, tensor_stride):
        q = out_coords.copy()
        q[:, 1:] += off
        idx = look(q)
        m = idx >= 0
        out[m] = np.maximum(out[m], F[idx[m]])
    return out_coords, out


def instance_norm(feats, weight, bias, eps=1e-8):
    F = np.asarray(feats, dtype=np.float64)
    mean = F.mean(0, keepdims=True)
    var = ((F - mean) ** 2).mean(0, keepdims=True)
    return (F - mean) / np.sqrt(var + eps) * np.asarray(weight, dtype=np.float64).reshape(1, -1) + \
        np.asarray(bias, dtype=np.float64).reshape(1, -1)


def batch_norm_eval(feats, bn):
    F = np.asarray(feats, dtype=np.float64)
    g, b = bn.weight.detach().double().numpy(), bn.bias.detach().double().numpy()
    m, v = bn.running_mean.double().numpy(), bn.running_var.double().numpy()
    return (F - m) / np.sqrt(v + bn.eps) * g + b


def relu(x):
    return np.maximum(x, 0)


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def union_add(ca, fa, cb, fb):
    """A rows first (in order), then the B-only rows (in order); features summed where both exist."""
    ca, cb = np.asarray(ca, dtype=np.int64), np.asarray(cb, dtype=np.int64)
    look = Lookup(ca)
    idx = look(cb)
    out_f = [np.asarray(fa, dtype=np.float64).copy()]
    m = idx >= 0
    out_f[0][idx[m]] += np.asarray(fb, dtype=np.float64)[m]
    out_f.append(np.asarray(fb, dtype=np.float64)[~m])
    return np.concatenate((ca, cb[~m])), np.concatenate(out_f)


def interpolate(score_coords, score, score_stride, query_coords):
    """features_at_coordinates: 8 corners floor(q/s)*s + {0,s}^3, weight prod(1-|q-c|/s), missing corners = 0."""
    sc = np.asarray(score_coords, dtype=np.int64)
    q = np.asarray(query_coords, dtype=np.int64)
    look = Lookup(sc)
    s = score_stride
    base = q.copy()
    base[:, 1:] = np.floor_divide(q[:, 1:], s) * s
    val = np.asarray(score, dtype=np.float64).reshape(-1)
    out = np.zeros(len(q))
    for off in kernel_offsets(2, s):
        c = base.copy()
        c[:, 1:] += off
        w = np.prod(1.0 - np.abs(q[:, 1:] - c[:, 1:]) / s, axis=1)
        idx = look(c)
        m = idx >= 0
        out[m] += w[m] * val[idx[m]]
    return out.reshape(-1, 1)


# ------------------------------------------------------------------------------------------------------------
# a12  _bbox_pred_to_bbox  (fcaf3d_head.py:300-349) and score product (:249)
# ------------------------------------------------------------------------------------------------------------
def decode_boxes(points, bbox_pred, yaw_parametrization="fcaf3d"):
    p, b = torch.as_tensor(points), torch.as_tensor(bbox_pred)
    if b.shape[0] == 0:
        return b
    xc = p[:, 0] + (b[:, 1] - b[:, 0]) / 2
    yc = p[:, 1] + (b[:, 3] - b[:, 2]) / 2
    zc = p[:, 2] + (b[:, 5] - b[:, 4]) / 2
    base = torch.stack([xc, yc, zc, b[:, 0] + b[:, 1], b[:, 2] + b[:, 3], b[:, 4] + b[:, 5]], -1)
    if b.shape[1] == 6:
        return base
    if yaw_parametrization == "naive":
        return torch.cat((base, b[:, 6:7]), -1)
    if yaw_parametrization == "sin-cos":
        norm = torch.pow(torch.pow(b[:, 6:7], 2) + torch.pow(b[:, 7:8], 2), 0.5)
        return torch.cat((base, torch.atan2(b[:, 6:7] / norm, b[:, 7:8] / norm)), -1)
    scale = b[:, 0] + b[:, 1] + b[:, 2] + b[:, 3]
    q = torch.exp(torch.sqrt(torch.pow(b[:, 6], 2) + torch.pow(b[:, 7], 2)))
    alpha = 0.5 * torch.atan2(b[:, 6], b[:, 7])
    return torch.stack((xc, yc, zc, scale / (1 + q), scale / (1 + q) * q, b[:, 5] + b[:, 4], alpha), dim=-1)


# ------------------------------------------------------------------------------------------------------------
# a10-a11  whole FCAF3D forward with the weights of the product modules (structure: fcaf3d_backbone.py:89-107,
#          fcaf3d_head.py:107-139, :275-298).  Returns per level (coords, centerness, bbox_pred, cls_score).
# ------------------------------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().double().numpy()


def _basic_block(c, f, ts, blk):
    stride = blk.conv1.stride
    oc, o = conv(c, f, _np(blk.conv1.kernel), 3, stride, ts)
    ots = ts * stride
    o = relu(batch_norm_eval(o, blk.norm1.bn))
    _, o = conv(oc, o, _np(blk.conv2.kernel), 3, 1, ots)
    o = batch_norm_eval(o, blk.norm2.bn)
    if blk.downsample is not None:
        _, r = conv(c, f, _np(blk.downsample[0].kernel), 1, stride, ts, out_coords=oc)
        r = batch_norm_eval(r, blk.downsample[1].bn)
    else:
        r = np.asarray(f, dtype=np.float64)
    return oc, relu(o + r), ots


def _bottleneck(c, f, ts, blk):
    """ME.modules.resnet_block.Bottleneck (MinkowskiEngine v0.5.4; selected by fcaf3d_backbone.py:122-127 for depth 50 / 101):
    conv1 k1 - BN - ReLU - conv2 k3 (stride) - BN - ReLU - conv3 k1 (x4 channels) - BN - (+ shortcut) - ReLU"""
    stride = blk.conv2.stride
    _, o = conv(c, f, _np(blk.conv1.kernel), 1, 1, ts)
    o = relu(batch_norm_eval(o, blk.norm1.bn))
    oc, o = conv(c, o, _np(blk.conv2.kernel), 3, stride, ts)
    ots = ts * stride
    o = relu(batch_norm_eval(o, blk.norm2.bn))
    _, o = conv(oc, o, _np(blk.conv3.kernel), 1, 1, ots)
    o = batch_norm_eval(o, blk.norm3.bn)
    if blk.downsample is not None:
        _, r = conv(c, f, _np(blk.downsample[0].kernel), 1, stride, ts, out_coords=oc)
        r = batch_norm_eval(r, blk.downsample[1].bn)
    else:
        r = np.asarray(f, dtype=np.float64)
    return oc, relu(o + r), ots


def backbone_forward(backbone, coords, feats):
    c, f, ts = np.asarray(coords, dtype=np.int64), np.asarray(feats, dtype=np.float64), 1
    stem = backbone.conv1
    c, f = conv(c, f, _np(stem[0].kernel), 3, 2, ts)
    ts = 2
    f = relu(instance_norm(f, _np(stem[1].weight), _np(stem[1].bias)))
    c, f = max_pool(c, f, ts)
    ts = 4
    outs = []
    for i in range(backbone.n_outs):
        for blk in getattr(backbone, f"layer{i + 1}"):
            c, f, ts = (_bottleneck if hasattr(blk, "conv3") else _basic_block)(c, f, ts, blk)
        outs.append((c, f, ts))
    return outs


def _seq_conv_bn_elu(c, f, ts, conv_m, bn_m):
    _, o = conv(c, f, _np(conv_m.kernel), 3, 1, ts)
    return elu(batch_norm_eval(o, bn_m.bn))


def head_forward(head, levels):
    """levels: list of (coords, feats, tensor_stride) from the backbone. pts_threshold pruning applied like the
    reference when a level exceeds it (top-k by interpolated score; ties arbitrary)."""
    results = [None] * len(levels)
    x = None
    scores = None
    for i in range(len(levels) - 1, -1, -1):
        ci, fi, ts = levels[i]
        if i == len(levels) - 1:
            c, f = ci, np.asarray(fi, dtype=np.float64)
        else:
            up = getattr(head, f"up_block_{i + 1}")
            xc, xf, xts = x
            uc, uf = conv_transpose_generative(xc, xf, _np(up[0].kernel), xts)
            uf = elu(batch_norm_eval(uf, up[1].bn))
            uf = _seq_conv_bn_elu(uc, uf, ts, up[3], up[4])
            c, f = union_add(ci, fi, uc, uf)
            if 0 <= head.pts_threshold < len(c):
                sc, sv, sts = scores
                interp = interpolate(sc, sv, sts, c).reshape(-1)
                keep = np.sort(np.argsort(-interp, kind="stable")[:head.pts_threshold])
                c, f = c[keep], f[keep]
        x = (c, f, ts)
        ob = getattr(head, f"out_block_{i}")
        o = _seq_conv_bn_elu(c, f, ts, ob[0], ob[1])
        ctr = o @ _np(head.centerness_conv.kernel)
        cls = o @ _np(head.cls_conv.kernel) + _np(head.cls_conv.bias)
        reg = o @ _np(head.reg_conv.kernel)
        dist = np.exp(reg[:, :6] * float(head.scales[i].scale.detach()))
        bbox = np.concatenate((dist, reg[:, 6:]), axis=1)
        scores = (c, cls.max(axis=1, keepdims=True), ts)
        results[i] = dict(coords=c, centerness=ctr, bbox_pred=bbox, cls_score=cls, points=c[:, 1:] * head.voxel_size)
    return results


def get_bboxes(head, results):
    """_get_bboxes_single (fcaf3d_head.py:238-271) without the file dump."""
    boxes, scores = [], []
    for r in results:
        s = torch.sigmoid(torch.from_numpy(r["cls_score"])) * torch.sigmoid(torch.from_numpy(r["centerness"]))
        mx = s.max(dim=1)[0]
        bp, pt = torch.from_numpy(r["bbox_pred"]), torch.from_numpy(r["points"])
        k = head.test_cfg.nms_pre
        if len(s) > k > 0:
            ids = mx.topk(k)[1]
            bp, s, pt = bp[ids], s[ids], pt[ids]
        boxes.append(decode_boxes(pt, bp, head.yaw_parametrization))
        scores.append(s)
    return torch.cat(boxes), torch.cat(scores)
