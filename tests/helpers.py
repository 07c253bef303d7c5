"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ["tiny", "tiny_boxes_origin", "mini_p", "edge_empty_view"]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"rma_{name}.npz"))
    g = {k: z[k] for k in z.files}
    g["dims"] = tuple(int(x) for x in g["dims"])
    g["voxel_size"] = float(g["voxel_size"])
    g["stride"] = int(g["stride"])
    g["thr"] = float(g["thr"])
    g["n_steps"] = int(g["n_steps"])
    g["origin"] = tuple(float(x) for x in g["origin"])
    return g


def t(a, device=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.to(device) if device is not None else x


def bits_equal(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype.kind == "f":
        return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)) | ((a == 0) & (b == 0))
    return a == b


def count_mismatch(a, b):
    return int(np.count_nonzero(~bits_equal(a, b)))


def fill_state_deterministic(module):
    """every floating tensor of the state dict <- a function of its KEY only (golden tests of networks too large to keep
    their weights as fixtures: the reference-side generator and the test fill both models identically)"""
    import zlib
    with torch.no_grad():
        for k, v in sorted(module.state_dict().items()):
            if not v.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
            r = torch.randn(v.shape, generator=g)
            if k.endswith("running_var"):
                r = r.abs() * 0.5 + 0.5
            elif k.endswith("norm.weight"):
                r = 1.0 + 0.1 * r
            elif k.endswith(("norm.bias", "running_mean", ".bias")):
                r = 0.1 * r
            else:
                fan_in = max(1, v[0].numel()) if v.dim() > 1 else 1
                r = r * (2.0 / fan_in) ** 0.5
            v.copy_(r)
    return module


def elementwise_error(got, exp):
    """worst over the elements of min(|got - exp|, |got - exp| / |exp|): "within tol absolutely OR relatively", the north
    star's criterion for fp32 outputs (parity tolerances)"""
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    err = np.abs(got - exp)
    worst = np.minimum(err, err / np.maximum(np.abs(exp), 1e-300))
    return float(worst.max()) if worst.size else 0.0
