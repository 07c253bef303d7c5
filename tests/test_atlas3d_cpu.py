"""Atlas 3D U-Net + TSDF head against outputs of the reference's own modules
(tests/golden/atlas3d.npz, made by tests/golden/make_golden.py from /root/reference): same parameter names, same
forward.  Stock torch ops on both sides, so the comparison is tight (conv algorithms may differ per host CPU)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(tag):
    z = np.load(os.path.join(HERE, "golden", "atlas3d.npz"))
    pick = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    return z, pick(f"{tag}_net."), pick(f"{tag}_head.")


def _build(cond):
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_backbone, build_head
    net = build_backbone(dict(type="AtlasBackbone3D", channels=[2, 4, 8, 16], layers_down=[1, 2, 1, 1], layers_up=[1, 2, 1],
                              drop=0.0, zero_init_residual=False, cond_proj=cond, norm="BN"))
    head = build_head(dict(type="AtlasTSDFHead", input_channels=[2, 4, 8], n_scales=3, voxel_size=0.04,
                           label_smoothing=1.05, sparse_threshold=[0.99, 0.99]))
    return net.eval(), head.eval()


@pytest.mark.parametrize("tag,cond", [("plain", False), ("cond", True)])
def test_atlas3d_matches_reference_outputs(tag, cond):
    z, net_sd, head_sd = _load(tag)
    net, head = _build(cond)
    assert set(net.state_dict()) == set(net_sd) and set(head.state_dict()) == set(head_sd)     # checkpoint-compatible keys
    net.load_state_dict(net_sd, strict=True)
    head.load_state_dict(head_sd, strict=True)
    with torch.no_grad():
        feats = net(torch.from_numpy(z[f"{tag}_x"]))
        tsdf, losses = head(feats)
    assert losses == {}
    for i, f in enumerate(feats):
        np.testing.assert_allclose(f.numpy(), z[f"{tag}_feat{i}"], rtol=1e-5, atol=1e-6)
    assert list(tsdf) == ["scene_tsdf_016", "scene_tsdf_008", "scene_tsdf_004"]
    for k, v in tsdf.items():
        np.testing.assert_allclose(v.numpy(), z[f"{tag}_{k}"], rtol=1e-5, atol=1e-6)
        assert float(v.abs().max()) <= 1.05


def test_tsdf_head_loss_terms():
    """log-space L1 on observed / fully-empty voxels, sparsified from the second scale on (reference atlas_head.py:51-81)"""
    z, net_sd, head_sd = _load("plain")
    net, head = _build(False)
    net.load_state_dict(net_sd); head.load_state_dict(head_sd)
    torch.manual_seed(0)
    with torch.no_grad():
        feats = net(torch.from_numpy(z["plain_x"]))
        targets = {f"tsdf_gt_{k}": (torch.rand(1, 1, *s) * 2 - 1).clamp(-1, 1) for k, s in
                   (("016", (4, 4, 2)), ("008", (8, 8, 4)), ("004", (16, 16, 8)))}
        targets["tsdf_gt_004"][..., :2, :] = 1.0
        out, losses = head(feats, targets)
    assert set(losses) == {"tsdf_loss_016", "tsdf_loss_008", "tsdf_loss_004"}
    from projects.mvsdetection.models.atlas_head import log_transform
    p, t = out["scene_tsdf_016"], targets["tsdf_gt_016"]
    use = (t < 1) | (t == 1).all(-1, keepdim=True)
    exp = (log_transform(p) - log_transform(t)).abs()[use].mean()
    assert torch.allclose(losses["tsdf_loss_016"], exp)
    assert all(torch.isfinite(torch.as_tensor(v)) for v in losses.values())
