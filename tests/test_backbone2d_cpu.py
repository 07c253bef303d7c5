"""2D feature extractor: ResNet-50 FPN + AtlasFPNFeature against outputs of the reference's own
modules (tests/golden/backbone2d.npz, made by tests/golden/make_golden.py --backbone2d from /root/reference).  The
weights of both sides are a function of their state-dict KEY (helpers.fill_state_deterministic), so identical key sets
are part of what is tested (checkpoint compatibility) and the fixture holds input + outputs only."""
import os

import numpy as np
import torch

from helpers import fill_state_deterministic

HERE = os.path.dirname(os.path.abspath(__file__))
CFG_FPN = dict(type="FPNDetectron",
               bottom_up_cfg=dict(input_channels=3, norm="BN", depth=50, out_features=["res2", "res3", "res4", "res5"],
                                  num_groups=1, width_per_group=64, stride_in_1x1=True, res5_dilation=1,
                                  res2_out_channels=256, stem_out_channels=64, freeze_at=2),
               in_features=["res2", "res3", "res4", "res5"], out_channels=256, norm="BN", fuse_type="sum")
CFG_HEAD = dict(type="AtlasFPNFeature", feature_strides={"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64},
                feature_channels={"p2": 256, "p3": 256, "p4": 256, "p5": 256, "p6": 256}, output_dim=32, output_stride=4,
                norm="BN")


def _build():
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_backbone
    return build_backbone(dict(CFG_FPN)).eval(), build_backbone(dict(CFG_HEAD)).eval()


def test_backbone2d_matches_reference_outputs():
    z = np.load(os.path.join(HERE, "golden", "backbone2d.npz"))
    fpn, head = _build()
    # frozen stages keep their statistics as buffers: BatchNorm's step counter is the only key that may be missing
    mine = set(fpn.state_dict())
    ref = set(z["fpn_keys"].tolist())
    assert mine <= ref and all(k.endswith("num_batches_tracked") for k in ref - mine)
    assert set(head.state_dict()) == set(z["head_keys"].tolist())
    fill_state_deterministic(fpn)
    fill_state_deterministic(head)
    with torch.no_grad():
        pyr = fpn(torch.from_numpy(z["x"]))
        y = head(pyr)
    assert list(pyr) == ["p2", "p3", "p4", "p5", "p6"]
    # 50 layers deep with key-derived (not trained) weights: activations reach 1e4, so the tolerance is relative to each
    # map's scale (frozen BatchNorm folds its statistics with rsqrt here, the reference divides by a square root)
    for got, ref in ((pyr["p2"].numpy(), z["p2"]), (pyr["p6"].numpy(), z["p6"]), (y.numpy(), z["y"])):
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max()


def test_frozen_stages_have_no_trainable_parameters_and_load_batchnorm_checkpoints():
    fpn, _ = _build()
    frozen = [n for n, p in fpn.named_parameters() if not p.requires_grad]
    assert frozen and all(n.startswith(("bottom_up.stem", "bottom_up.res2")) for n in frozen)
    # a checkpoint written by a BatchNorm2d model (with num_batches_tracked everywhere) loads strictly
    sd = dict(fpn.state_dict())
    for k in list(sd):
        if k.endswith("running_var") and k[:-11] + "num_batches_tracked" not in sd:
            sd[k[:-11] + "num_batches_tracked"] = torch.tensor(0)
    fpn.load_state_dict(sd, strict=True)


def test_channels_last_2d_stack_gives_the_same_maps_in_the_layout_the_hot_path_reads():
    """the plugin runs the 2D stack in torch.channels_last (MultiViewBase.channels_last_2d): same values as the reference's
    NCHW run (golden fixture, same tolerance), and the output's MEMORY is channels-last -- the hand-off layout of the
    aggregation kernels, consumed without a layout pass"""
    z = np.load(os.path.join(HERE, "golden", "backbone2d.npz"))
    fpn, head = _build()
    fill_state_deterministic(fpn)
    fill_state_deterministic(head)
    fpn.to(memory_format=torch.channels_last)
    head.to(memory_format=torch.channels_last)
    x = torch.from_numpy(z["x"]).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = head(fpn(x))
    assert y.shape == z["y"].shape
    assert np.abs(y.numpy() - z["y"]).max() <= 2e-4 * np.abs(z["y"]).max()
    assert y.permute(0, 2, 3, 1).is_contiguous()                      # one contiguous channel vector per pixel
    # the detector base class does exactly this
    from projects.mvsdetection.models.multiview_base import MultiViewBase
    assert MultiViewBase.channels_last_2d
