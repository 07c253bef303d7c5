"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors (generated from the reference's own
Python by tests/golden/make_golden.py), the host logic, and the C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import SCENES, bits_equal, count_mismatch, load_golden, t
from oracle import rma_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", SCENES)
def test_oracle_dense_matches_golden(name):
    g = load_golden(name)
    vol, cnt = O.backproject_accum(g["dims"], g["voxel_size"], g["origin"], t(g["projection"]), t(g["features"]),
                                   g["stride"])
    assert count_mismatch(vol, g["dense_volume"]) == 0
    assert (cnt.numpy() == g["dense_count"]).all()
    _, valid, px, py = O.backproject_view(g["dims"], g["voxel_size"], g["origin"],
                                          O.scale_projection(t(g["projection"][0]), g["stride"]), t(g["features"][0]))
    assert (px.numpy() == g["view0_px"]).all() and (py.numpy() == g["view0_py"]).all()
    assert (valid.numpy() == g["view0_valid"]).all()


@pytest.mark.parametrize("name", SCENES)
def test_oracle_ray_params_match_golden(name):
    g = load_golden(name)
    H, W = g["features"].shape[-2:]
    for v in range(g["features"].shape[0]):
        o, d = O.ray_params(O.scale_projection(t(g["projection"][v]), g["stride"]), H, W)
        assert count_mismatch(o, g["ray_o"][v]) == 0
        assert count_mismatch(d, g["ray_d"][v]) == 0


@pytest.mark.parametrize("name", SCENES)
def test_oracle_neus_rows_and_points_match_golden(name):
    g = load_golden(name)
    tsdf = t(g["tsdf"])
    rows, counts = [], []
    for v in range(g["features"].shape[0]):
        r = O.rma_neus_view(O.scale_projection(t(g["projection"][v]), g["stride"]), t(g["features"][v]), tsdf,
                            g["dims"], g["voxel_size"], g["origin"], g["n_steps"], g["thr"])
        counts.append(0 if r is None else r.shape[0])
        if r is not None:
            rows.append(r)
    assert counts == list(g["neus_counts"])
    assert count_mismatch(torch.cat(rows), g["neus_rows"]) == 0
    pts = O.aggregate_rma(t(g["projection"]), t(g["features"]), tsdf, g["dims"], g["voxel_size"], g["origin"],
                          g["stride"], 300, g["thr"])
    assert count_mismatch(pts, g["points"]) == 0


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_depth_rows_match_golden(name, k):
    g = load_golden(name)
    r = O.rma_depth_view(O.scale_projection(t(g["projection"][0]), g["stride"]), t(g["features"][0]), t(g["tsdf"]),
                         g["dims"], g["voxel_size"], g["origin"], g["n_steps"], k)
    exp = g[f"depth_rows_k{k}"]
    if exp.shape[0] == 0:
        assert r is None
    else:
        assert count_mismatch(r, exp) == 0


@pytest.mark.parametrize("name", SCENES)
def test_oracle_select_and_voxelize_match_golden(name):
    g = load_golden(name)
    M = g["points"].shape[0]
    mask = np.unpackbits(g["sel_mask"])[:M].astype(bool)
    # the mask is what numpy's global RNG gives under the recorded seed (sample_points semantics)
    np.random.seed(7)
    assert (O.sample_mask_numpy(M, int(g["sel_max_points"])) == mask).all()
    c, f = O.select_rows(t(g["points"]), g["sel_offset"], mask)
    assert count_mismatch(c, g["sel_coords"]) == 0 and count_mismatch(f, g["sel_feats"]) == 0
    Cq, Fq, src = O.voxelize(c, f, 0.01)
    assert (Cq.numpy() == g["vox_coords"]).all() and (src.numpy() == g["vox_src"]).all()
    # first-wins: every source index is the smallest index of its voxel
    q = torch.floor(c / 0.01).to(torch.int32).numpy()
    seen = {}
    for i, row in enumerate(map(tuple, q)):
        seen.setdefault(row, i)
    assert sorted(seen.values()) == list(src.numpy())


def test_all_views_empty_raises_like_reference():
    g = load_golden("edge_empty_view")
    with pytest.raises(TypeError):
        O.aggregate_rma(t(g["projection"][:1]), t(g["features"][:1]), t(g["tsdf"]), g["dims"], g["voxel_size"],
                        g["origin"], g["stride"])


def test_host_projection_inverse_matches_golden():
    from cnrma_amd import rma
    for name in SCENES:
        g = load_golden(name)
        pinv = rma.projection_inverse(t(g["projection"]), g["stride"])
        # same call as the oracle's, so identical on any one host; vs the golden (generated on the build container's
        # CPU) LAPACK may differ in the last bits on another CPU model
        for v in range(pinv.shape[0]):
            assert torch.equal(pinv[v], O.projection_inverse(O.scale_projection(t(g["projection"][v]), g["stride"])))
        np.testing.assert_allclose(pinv.numpy(), g["proj_inv"], rtol=1e-4, atol=1e-5)


def test_synth_scene_is_deterministic():
    from cnrma_amd import synth
    a = synth.make_scene("tiny", seed=3, boxes=2)
    b = synth.make_scene("tiny", seed=3, boxes=2)
    for k in ("features", "projection", "tsdf"):
        assert torch.equal(a[k], b[k])
    g = load_golden("tiny")
    c = synth.make_scene("tiny", seed=0)
    assert count_mismatch(c["features"][:, 0], g["features"]) == 0
    assert count_mismatch(c["projection"][:, 0], g["projection"]) == 0
    assert count_mismatch(c["tsdf"][0, 0], g["tsdf"]) == 0


# ---------------------------------------------------------------------------------------------------------------
# C-ABI surface: the library loads, exports every symbol of include/cnrma.h, and the ctypes table agrees in arity
# ---------------------------------------------------------------------------------------------------------------
def _header_functions(experiments=False):
    """prototypes of include/cnrma.h: the product surface, or (experiments=True) only those under #ifdef CNRMA_EXPERIMENTS"""
    src = open(os.path.join(ROOT, "include", "cnrma.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    exp = "\n".join(re.findall(r"#ifdef CNRMA_EXPERIMENTS\n(.*?)#endif", src, flags=re.S))
    src = exp if experiments else re.sub(r"#ifdef CNRMA_EXPERIMENTS\n.*?#endif", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(cnrma_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        out[m.group(1)] = n
    return out


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("cnrma_")}


def test_library_exports_every_declared_symbol():
    from cnrma_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(_lib.EXP_LIB_PATH):   # fresh checkout: hipcc cross-compiles without a GPU
        import subprocess
        subprocess.run(["make", "-C", os.path.dirname(_lib.LIB_PATH), "-j8"], check=True)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _header_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/cnrma.h but not exported"
    assert lib.cnrma_abi_version() == _lib.ABI_VERSION
    # the product library exports exactly the product surface: no cnrma_debug_* hook, nothing undeclared
    product = _exported(_lib.LIB_PATH)
    assert product == set(decl), product ^ set(decl)
    assert not [n for n in product if n.startswith("cnrma_debug")]
    # the experiments library = the same surface + the prototypes under #ifdef CNRMA_EXPERIMENTS
    extra = _header_functions(experiments=True)
    assert extra and all(n.startswith("cnrma_debug_") for n in extra)
    assert _exported(_lib.EXP_LIB_PATH) == set(decl) | set(extra)
    assert ctypes.CDLL(_lib.EXP_LIB_PATH).cnrma_abi_version() == _lib.ABI_VERSION


def test_ctypes_table_matches_header():
    from cnrma_amd import _lib
    for decl, table in ((_header_functions(), _lib.SIGNATURES), (_header_functions(experiments=True), _lib.EXPERIMENT_SIGNATURES)):
        assert set(decl) == set(table), set(decl) ^ set(table)
        for name, n in decl.items():
            assert len(table[name][1]) == n, (name, n, len(table[name][1]))
    assert not set(_lib.SIGNATURES) & set(_lib.EXPERIMENT_SIGNATURES)


def test_experiments_library_is_opt_in_and_reference_counted():
    """product code binds libcnrma_hip.so only; `_lib.experiments(True, who)` routes the process's calls through
    libcnrma_hip_exp.so until every `who` has let go (sparse.conv_tuning / rma.dense_tuning / tests are such users)"""
    from cnrma_amd import _lib
    assert not _lib.experiments_active()
    prod = _lib.load()
    assert not hasattr(prod, "cnrma_debug_conv_tuning") and not hasattr(prod, "cnrma_debug_dense_tuning")
    try:
        _lib.experiments(True, "a")
        _lib.experiments(True, "b")
        exp = _lib.load()
        assert exp is not prod and hasattr(exp, "cnrma_debug_conv_tuning") and hasattr(exp, "cnrma_sparse_conv_go_f16x3")
        assert _lib.load(experiments=False) is prod
        _lib.experiments(False, "a")
        assert _lib.experiments_active() and _lib.load() is exp          # "b" still holds it
        _lib.experiments(False, "b")
        assert not _lib.experiments_active() and _lib.load() is prod
    finally:
        _lib.experiments(False, "a")
        _lib.experiments(False, "b")


def test_product_path_fails_loudly_without_gpu():
    from cnrma_amd import _lib, rma
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.CnrmaError):
        rma.to_nhwc(torch.zeros(1, 4, 2, 2))
