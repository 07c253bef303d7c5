"""GPU property tests at BASELINE.json's full size (ScanNet shape, 40 views, 500 k points): things the oracle cannot
reach in seconds are checked through size-independent properties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(device):
    from cnrma_amd import rma, synth
    sc = synth.make_scene("S", seed=1, boxes=4)
    feat = sc["features"][:, 0].to(device)
    proj = sc["projection"][:, 0]
    return dict(sc=sc, nhwc=rma.to_nhwc(feat), proj=proj, pinv=rma.projection_inverse(proj, sc["stride"]).to(device),
                tsdf=sc["tsdf"][0, 0].to(device))


def test_single_march_equals_two_pass_at_full_size(scene, device):
    """kept-record emission == re-marching emission, bit for bit, over all 768 k rays"""
    from cnrma_amd import rma
    s = scene
    a, pa = rma.rma_view_rows(s["nhwc"], s["pinv"], s["tsdf"], s["sc"]["dims"], 0.04, s["sc"]["origin"], single_march=True)
    b, pb = rma.rma_view_rows(s["nhwc"], s["pinv"], s["tsdf"], s["sc"]["dims"], 0.04, s["sc"]["origin"], single_march=False)
    assert torch.equal(pa, pb) and torch.equal(a, b)
    assert a.shape[0] > 3_000_000
    w = a[:, 3]
    assert float(w.min()) >= 0.05 - 1e-9 and float(w.max()) <= 1.0 + 1e-6           # kept weights in [thr, 1]


def test_ray_weights_sum_to_at_most_one_and_points_lie_in_grid(scene, device):
    from cnrma_amd import rma
    s = scene
    rows, per_view, samples = rma.rma_view_rows(s["nhwc"], s["pinv"], s["tsdf"], s["sc"]["dims"], 0.04, s["sc"]["origin"],
                                                with_samples=True)
    ray = samples[:, 0].long()
    wsum = torch.zeros(int(ray.max()) + 1, device=device).index_add_(0, ray, rows[:, 3])
    assert float(wsum.max()) <= 1.0 + 1e-5
    # order: (ray, step) strictly increasing lexicographically
    key = ray * 512 + samples[:, 1].long()
    assert bool((key[1:] > key[:-1]).all())
    X, Y, Z = s["sc"]["dims"]
    ext = torch.tensor([X, Y, Z], device=device) * 0.04
    assert bool((rows[:, :3] > -0.03).all()) and bool((rows[:, :3] < ext + 0.03).all())
    # features are the pixel's channel vector: every row of a ray carries the same vector
    same = ray[1:] == ray[:-1]
    assert bool((rows[1:, 4:][same] == rows[:-1, 4:][same]).all())


def test_fused_selection_is_a_subset_in_order_and_mean_scaled(scene, device):
    from cnrma_amd import rma
    s = scene
    full, info = rma.aggregate_rows(s["nhwc"], s["pinv"], s["tsdf"], s["sc"]["dims"], 0.04, s["sc"]["origin"])
    M = info["M"]
    mask = rma.sample_mask_device(torch.tensor([M], dtype=torch.int32, device=device), M, 500_000, seed=5)
    c, f, info2 = rma.aggregate_points(s["nhwc"], s["pinv"], s["tsdf"], s["sc"]["dims"], 0.04, s["sc"]["origin"],
                                       offset=(1.0, 2.0, 3.0), mask=mask)
    assert c.shape[0] == 500_000 and info2["M"] == M
    idx = torch.nonzero(mask).squeeze(1)
    off = torch.tensor([1.0, 2.0, 3.0], device=device)
    assert torch.equal(c, full[idx, :3] + off) and torch.equal(f, full[idx, 3:])
    # mean of the (unscaled) weights is what the features were divided by: sum_i w_i/mean = M
    raw, _ = rma.rma_view_rows(s["nhwc"], s["pinv"], s["tsdf"], s["sc"]["dims"], 0.04, s["sc"]["origin"])
    scale = raw[:, 3].double() / raw[:, 3].double().mean()
    np.testing.assert_allclose(float(scale.sum()), M, rtol=1e-9)
    nz = raw[:, 4].abs() > 1e-3
    np.testing.assert_allclose((full[:, 3][nz] / raw[:, 4][nz]).cpu().numpy(), scale[nz].float().cpu().numpy(), rtol=2e-6)


def test_sparse_levels_are_unique_lattices_and_forward_is_deterministic(scene, device):
    import bench
    from cnrma_amd import pipeline
    s = scene
    backbone, head = bench.build_model(32, device)
    cfg = pipeline.SceneConfig(s["sc"]["dims"], stride=4, max_points=500_000, sampler="device")
    feat = s["sc"]["features"][:, 0].to(device)
    mask = None
    outs = []
    for _ in range(2):
        torch.manual_seed(0)
        from cnrma_amd import rma
        rma._SAMPLE_CALLS[0] = 0
        outs.append(pipeline.forward_scene(cfg, backbone, head, feat, s["proj"], s["tsdf"]))
    a, b = outs
    assert torch.equal(a["bboxes"], b["bboxes"]) and torch.equal(a["scores"], b["scores"])     # run-to-run deterministic
    assert a["M_selected"] == 500_000 and a["M_unique"] <= 500_000
    assert a["level_rows"] == b["level_rows"] and a["head_rows"][0] == 200_000
    assert torch.isfinite(a["bboxes"]).all() and torch.isfinite(a["scores"]).all()
    assert float(a["scores"].min()) >= 0.0 and float(a["scores"].max()) <= 1.0
    assert a["bboxes"].shape[0] == a["scores"].shape[0] <= 4000
    assert bool((a["bboxes"][:, 3:6] > 0).all())                                                 # sizes = sums of exp()


def test_conv_is_linear_at_full_size(scene, device):
    """conv(a*x + b*y) == a*conv(x) + b*conv(y) on a 200 k-row tensor (no epilogue): catches indexing / race bugs"""
    from cnrma_amd import sparse as S
    g = torch.Generator().manual_seed(0)
    xyz = torch.randint(0, 90, (260_000, 3), generator=g, dtype=torch.int32) * 2
    c = torch.unique(torch.cat((torch.zeros(len(xyz), 1, dtype=torch.int32), xyz), dim=1), dim=0).to(device)
    cs = S.CoordSet(c.contiguous(), 2)
    x = torch.randn(c.shape[0], 64, generator=g).to(device)
    y = torch.randn(c.shape[0], 64, generator=g).to(device)
    W = (torch.randn(27, 64, 128, generator=g) / 40).to(device)
    fx = S.conv(S.SparseTensor(x, cs), W, 3, 1).F
    fy = S.conv(S.SparseTensor(y, cs), W, 3, 1).F
    fz = S.conv(S.SparseTensor(2.0 * x - 0.5 * y, cs), W, 3, 1).F
    assert torch.allclose(fz, 2.0 * fx - 0.5 * fy, rtol=1e-4, atol=1e-4)
    assert torch.equal(S.conv(S.SparseTensor(x, cs), W, 3, 1).F, fx)                              # deterministic


def test_concurrent_scenes_on_separate_streams_match_sequential(scene, device):
    """bench.py keeps several scenes in flight (one host thread + HIP stream each): the results must be those of the
    sequential run (per-stream workspaces, shared read-only caches)."""
    import threading
    import bench
    from cnrma_amd import pipeline, rma
    s = scene
    backbone, head = bench.build_model(32, device)
    cfg = pipeline.SceneConfig(s["sc"]["dims"], stride=4, max_points=500_000, sampler="device")
    feat = s["sc"]["features"][:, 0].to(device)
    full, info = rma.aggregate_rows(s["nhwc"], s["pinv"], s["tsdf"], s["sc"]["dims"], 0.04, s["sc"]["origin"])
    mask = rma.sample_mask_device(torch.tensor([info["M"]], dtype=torch.int32, device=device), info["M"], 500_000, seed=9)
    ref = pipeline.forward_scene(cfg, backbone, head, feat, s["proj"], s["tsdf"], mask=mask)
    torch.cuda.synchronize()
    n_threads, per_thread = 3, 2
    outs, errs = [[] for _ in range(n_threads)], []
    streams = [torch.cuda.Stream(device=device) for _ in range(n_threads)]

    def worker(w):
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(streams[w]):
                for _ in range(per_thread):
                    o = pipeline.forward_scene(cfg, backbone, head, feat, s["proj"], s["tsdf"], mask=mask)
                    outs[w].append((o["bboxes"].clone(), o["scores"].clone()))
                streams[w].synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not errs, errs
    for w in range(n_threads):
        assert len(outs[w]) == per_thread
        for b, sc in outs[w]:
            assert torch.equal(b, ref["bboxes"]) and torch.equal(sc, ref["scores"])


def test_batched_scenes_equal_single_scenes(device):
    """pipeline.forward_scenes (several scenes through one sparse network pass: collated voxels, per-scene instance norm,
    per-scene pruning and decode) gives each scene the detections of its own forward_scene pass"""
    import bench
    from cnrma_amd import pipeline, synth
    shape = "St"
    V, C, H, W, dims, stride = synth.SHAPES[shape]
    torch.manual_seed(0)
    backbone, head = bench.build_model(C, device)
    head.pts_threshold = 3000                     # make the per-scene pruning bite on the finest level
    cfg = pipeline.SceneConfig(dims, stride=stride, max_points=None)      # no random point selection: deterministic
    scenes = []
    for seed in (0, 1, 2):
        sc = synth.make_scene(shape, seed=seed)
        scenes.append(dict(features=sc["features"][:, 0].to(device), projection=sc["projection"][:, 0],
                           tsdf=sc["tsdf"][0, 0].to(device), offset=(0.1 * seed, 0.0, -0.05 * seed)))
    singles = [pipeline.forward_scene(cfg, backbone, head, s_["features"], s_["projection"], s_["tsdf"], offset=s_["offset"])
               for s_ in scenes]
    for s_, a in zip(scenes, singles):             # a batch of one is the single-scene path
        b = pipeline.forward_scenes(cfg, backbone, head, [s_])[0]
        assert torch.equal(a["bboxes"], b["bboxes"]) and torch.equal(a["scores"], b["scores"])
    batched = pipeline.forward_scenes(cfg, backbone, head, scenes)
    assert len(batched) == 3
    for a, b in zip(singles, batched):
        assert a["M"] == b["M"] and a["M_unique"] == b["M_unique"]
        assert a["bboxes"].shape == b["bboxes"].shape and a["bboxes"].shape[0] > 100
        ka = a["scores"].max(dim=1)[0].argsort(descending=True, stable=True)
        kb = b["scores"].max(dim=1)[0].argsort(descending=True, stable=True)
        # the same rows were selected and they rank the same, up to swaps between near-equal scores (the split over
        # kernel offsets of a layer depends on its row count, so sums are rounded in a different order)
        ra, rb = a["scores"][ka].cpu().numpy(), b["scores"][kb].cpu().numpy()
        sa, sb = a["bboxes"][ka].cpu().numpy(), b["bboxes"][kb].cpu().numpy()
        same = np.isclose(ra, rb, rtol=2e-4, atol=2e-6).all(axis=1) & np.isclose(sa, sb, rtol=1e-3, atol=1e-4).all(axis=1)
        assert same.mean() > 0.995, same.mean()
        np.testing.assert_allclose(np.sort(ra.max(axis=1)), np.sort(rb.max(axis=1)), rtol=2e-4, atol=2e-6)
