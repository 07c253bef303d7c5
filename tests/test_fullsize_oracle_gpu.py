"""Parity AT THE BENCHMARKED SIZE: the graph replay of the sparse half (pipeline.StaticNet: voxelise
-> MinkResNet34 -> neck / head -> decode) on one full ScanNet-shape scene -- 500 000 aggregated points, ~490 k voxels, head
level 0 pruned to 200 000 rows -- and the north-star stem (C = 256 pair-list convolution, ~474 k output rows) against
oracle/sparse_torch.py (fp32 on the host cores, pinned to the fp64 oracle by tests/test_sparse_oracle_cpu.py): every level's
coordinate set bit-exact, features / head outputs / decoded boxes ELEMENT-WISE within 1e-4 (absolute or relative), in the
default f16x3 arithmetic and in exact fp32.  The test also proves WHICH kernel variants ran (sparse.conv_plan: the launcher's
choice is a pure function of the capacities): the 128x64 tile (>= 200 k rows), the 128x128 tile (16 k-40 k rows), 64x128,
64x64, the un-split path and the split-over-offsets path are all covered here -- the small-size oracle tests never reach the
first two.  Reference: fcaf3d_backbone.py:89-107, fcaf3d_head.py:107-139, :275-349."""
import copy

import numpy as np
import pytest
import torch

from oracle import rma_oracle as RO
from oracle import sparse_oracle as SO
from oracle import sparse_torch as ST

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _within(got, exp, tol=TOL):
    """element-wise: |got - exp| <= tol OR |got - exp| <= tol |exp|; returns the worst min(abs, rel) error"""
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    err = np.abs(got - exp)
    worst = np.minimum(err, err / np.maximum(np.abs(exp), 1e-300))
    return float(worst.max()) if worst.size else 0.0


@pytest.fixture(scope="module")
def s_scene(device):
    """the 500 000 points the ScanNet-shape scene hands to the detector + the oracle's levels / head outputs / boxes"""
    import bench
    from cnrma_amd import rma, synth
    sc = synth.make_scene("S", seed=1, boxes=4)
    nhwc = rma.to_nhwc(sc["features"][:, 0].to(device))
    pinv = rma.projection_inverse(sc["projection"][:, 0], sc["stride"]).to(device)
    P, F, info = rma.aggregate_points(nhwc, pinv, sc["tsdf"][0, 0].to(device), sc["dims"], 0.04, sc["origin"],
                                      max_points=500000, sampler="device", seed=5)
    assert P.shape[0] == 500000 and info["M"] > 3_000_000
    backbone, head = bench.build_model(32, device)            # the benchmark's model: pts_threshold 200 000, nms_pre 1000
    b_cpu, h_cpu = copy.deepcopy(backbone).cpu(), copy.deepcopy(head).cpu()
    torch.set_num_threads(min(64, torch.get_num_threads() or 64))
    Cq, Fq, _ = RO.voxelize(P.cpu(), F.cpu(), 0.01)
    levels = ST.backbone_forward(b_cpu, Cq.numpy(), Fq.numpy())
    results = ST.head_forward(h_cpu, levels)
    boxes, scores = ST.get_bboxes(h_cpu, results)
    return dict(P=P, F=F, backbone=backbone, head=head, n_vox=len(Cq), levels=levels, results=results, boxes=boxes, scores=scores)


def _record_plans(S):
    """wrap sparse.conv / conv_transpose_generative: the launcher's plan of every convolution that runs"""
    plans = []
    orig_conv, orig_tr = S.conv, S.conv_transpose_generative

    def conv(x, weight, kernel_size=3, stride=1, *a, **k):
        y = orig_conv(x, weight, kernel_size, stride, *a, **k)
        Cin, Cout = x.F.shape[1], y.F.shape[1]
        if Cin % 32 == 0:
            pair = (S.PAIR_CONV and stride == 2 and kernel_size == 3 and Cin >= S.PAIR_CONV_MIN_CIN and S.CONV_PRECISION == "f16x3")
            plans.append(dict(S.conv_plan(y.cs.n, Cin, Cout, kernel_size ** 3), rows=y.cs.n, Cin=Cin, Cout=Cout, pair_list=pair))
        return y

    def convtr(x, weight, *a, **k):
        y = orig_tr(x, weight, *a, **k)
        plans.append(dict(S.conv_plan(x.cs.n, x.F.shape[1], y.F.shape[1], 1, slices=8), rows=x.cs.n, Cin=x.F.shape[1], Cout=y.F.shape[1],
                          pair_list=False))
        return y
    import sys
    S.conv, S.conv_transpose_generative = conv, convtr
    sys.modules["cnrma_amd.nn"].S.conv = conv

    def undo():
        S.conv, S.conv_transpose_generative = orig_conv, orig_tr
        sys.modules["cnrma_amd.nn"].S.conv = orig_conv
    return plans, undo


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_graph_replay_of_the_full_scannet_scene_vs_oracle(device, s_scene, precision):
    from cnrma_amd import pipeline
    from cnrma_amd import sparse as S
    d = s_scene
    prev = S.CONV_PRECISION
    S.CONV_PRECISION = precision
    plans, undo = _record_plans(S)
    try:
        net = pipeline.StaticNet(d["backbone"], d["head"], 0.01, device)
        net.build(d["P"], d["F"])
        assert net.graph is not None
    finally:
        undo()
    try:
        net.run(d["P"].flip(0).contiguous(), d["F"] * 0.5)                 # other inputs through the graph first
        out = net.run(d["P"], d["F"])
        b, s, info = pipeline.StaticScene.detections(out)
    finally:
        S.CONV_PRECISION = prev
    # ---- which kernel variants this covered (static trace = the graph's launches; capacities, not live rows, decide)
    if precision == "f16x3":
        shapes = {(p["shape"], p["splits"] > 1) for p in plans}
        big = [p for p in plans if p["rows"] >= 200000 and p["Cin"] > 32 and p["Cout"] == 64]
        mid = [p for p in plans if 16384 <= p["rows"] < 40000 and p["Cout"] >= 128 and p["Cin"] >= 64 and p["tile"] != (128, 32)]
        assert big and all(p["shape"] == "128x64" and p["splits"] == 1 for p in big), big           # T128x64, no split
        assert mid and any(p["shape"] == "128x128" for p in mid), mid                                # T128x128
        assert {("64x128", True), ("64x128", False), ("64x64", False)} <= shapes, shapes
        assert any(p["splits"] >= 9 for p in plans)                                                  # deep split of the short layers
    # ---- coordinate sets of every backbone level: bit-exact; features element-wise
    assert info["M_unique"] == d["n_vox"]
    worst = {}
    for li, (o, n, (cs, f)) in enumerate(zip(out["levels"], info["level_rows"], d["levels"])):
        c = cs.C.numpy()
        assert n == len(c) and o.cs.stride == cs.stride
        got_c = o.cs.C[:n].cpu().numpy().astype(np.int64)
        k1, k2 = np.argsort(SO._key(got_c), kind="stable"), np.argsort(SO._key(c), kind="stable")
        assert (got_c[k1] == c[k2]).all()
        worst[f"level{li}"] = _within(o.F[:n].cpu().numpy()[k1], f.numpy()[k2])
    # ---- head: rows of a pruned level may differ at ties of the pruning score; compare through the coordinate key
    hd = out["head"]
    for i in range(4):
        e, n = d["results"][i], info["head_rows"][i]
        got_pts = hd["points"][i][:n].cpu().numpy()
        ck = SO._key(np.concatenate((np.zeros((n, 1)), np.round(got_pts / 0.01)), axis=1).astype(np.int64))
        ek = SO._key(e["coords"].numpy())
        assert len(ck) == len(ek), (i, len(ck), len(ek))
        common = np.intersect1d(ck, ek)
        assert len(common) >= 0.999 * len(ek), (i, len(common), len(ek))
        gi = np.argsort(ck)[np.searchsorted(np.sort(ck), common)]
        ei = np.argsort(ek)[np.searchsorted(np.sort(ek), common)]
        for key in ("centerness", "bbox_pred", "cls_score"):
            g_, e_ = hd[key][i][:n].cpu().numpy()[gi].astype(np.float64), e[key].numpy()[ei].astype(np.float64)
            if key == "bbox_pred":                   # exp(scale * reg): the regression itself is the exponent
                g_ = np.concatenate((np.log(g_[:, :6]), g_[:, 6:]), axis=1)
                e_ = np.concatenate((np.log(e_[:, :6]), e_[:, 6:]), axis=1)
            worst[f"head{i}.{key}"] = _within(g_, e_)
    # ---- decoded boxes + scores (nms_pre = 1000 per level): rows matched through (level, box centre); membership may differ at
    # score ties of the top-k cut
    bx, sx = b.cpu().numpy(), s.cpu().numpy()
    eb, es = d["boxes"].numpy(), d["scores"].numpy()
    assert bx.shape == eb.shape and sx.shape == es.shape
    kg = {tuple(np.round(r[:6] * 1e4).astype(np.int64)): j for j, r in enumerate(bx)}
    pairs = [(kg[k], j) for j, r in enumerate(eb) for k in [tuple(np.round(r[:6] * 1e4).astype(np.int64))] if k in kg]
    assert len(pairs) >= 0.98 * len(eb), (len(pairs), len(eb))
    gi, ei = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    worst["boxes"] = _within(bx[gi], eb[ei])
    worst["scores"] = _within(sx[gi], es[ei])
    print(f"\n[{precision}] worst element-wise min(abs, rel) error vs oracle/sparse_torch.py:", {k: f"{v:.2e}" for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if v > TOL}
    assert not bad, bad


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_north_star_stem_vs_oracle(device, s_scene, precision):
    """the NS stem: 500 000 points x 256 channels -> stride-2 3x3x3 convolution to 64 channels (pair-list kernel in f16x3:
    the kernel map is 5 % full), instance norm + ReLU, max pool -- every output row element-wise against the oracle"""
    from cnrma_amd import sparse as S
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    d = s_scene
    g = torch.Generator().manual_seed(3)
    F256 = torch.randn(d["P"].shape[0], 256, generator=g)
    torch.manual_seed(1)
    bb = FCAF3DBackbone(256, 34).eval()
    bb.init_weights()
    Cq, Fq, _ = RO.voxelize(d["P"].cpu(), F256, 0.01)
    cs = ST.CoordSet(torch.as_tensor(Cq.numpy()), 1)
    oc, of = ST.conv(cs, Fq, bb.conv1[0].kernel, 3, 2)
    onorm = torch.relu(ST.instance_norm(of, bb.conv1[1].weight, bb.conv1[1].bias))
    pc, pf = ST.max_pool(oc, onorm)
    bb.to(device)
    prev = S.CONV_PRECISION
    S.CONV_PRECISION = precision
    plans, undo = _record_plans(S)
    try:
        with torch.no_grad():
            x, _ = S.voxelize(d["P"], F256.to(device), 0.01)
            y = S.conv(x, bb.conv1[0].kernel, 3, 2)
            z = bb.conv1(x)
    finally:
        undo()
        S.CONV_PRECISION = prev
    assert y.cs.n == len(oc) > 400000
    if precision == "f16x3":
        assert plans[0]["pair_list"] and plans[0]["Cin"] == 256
    for got, (ecs, ef), name in ((y, (oc, of), "stem conv"), (z, (pc, pf), "stem conv + norm + pool")):
        c = got.cs.C.cpu().numpy().astype(np.int64)
        k1, k2 = np.argsort(SO._key(c), kind="stable"), np.argsort(SO._key(ecs.C.numpy()), kind="stable")
        assert (c[k1] == ecs.C.numpy()[k2]).all()
        w = _within(got.F.cpu().numpy()[k1], ef.numpy()[k2])
        print(f"\n[{precision}] {name}: worst element-wise error {w:.2e} over {c.shape[0]} rows")
        assert w <= TOL, (name, w)
