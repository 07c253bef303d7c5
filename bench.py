"""bench.py -- scenes/s of the CN-RMA hot path (dense unprojection -> RMA -> voxelise -> FCAF3D -> decode), forward.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NS|S|St|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload = the configuration BASELINE.json's metric is quoted on: NS = 40 views x 256 channels x 480x640 feature planes
into a 192^3 grid (SURVEY.md 8d "Headline (NS)"); the ScanNet-config shape S (40 x 32ch x 120x160 -> 192x192x80) is
measured in the same run and printed as the labelled block "S" of the same JSON line.

A *scene* goes through the whole path as ONE replayed HIP graph (pipeline.StaticScene: ~350 launches, no device->host
read); inputs (feature maps as the 2D backbone writes them, NCHW fp32, and the TSDF) are resident in HBM; `--slots`
graphs on their own streams are in flight so that the latency-bound small kernels of one scene overlap the large kernels
of another.  A *step* = one wave of `scenes_per_step` scenes per GPU (chosen from the warm-up so that the K timed steps
take >= --window-s seconds; printed in `config`).  >= 8 DISTINCT scenes (own feature tensors, own furniture in the TSDF)
are rotated.  The K steps are timed three times (each window bracketed by barrier + synchronize, MAX over ranks); `value`
is the median window.  Scenes are independent: N GPUs process N x scenes_per_step scenes per step (weak scaling), the
only exchange is ONE all-gather of the padded detections per step over RCCL.
Rank 0 prints ONE JSON line carrying `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0   # measured float4 copy (same guide)
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_F16_PEAK_TFLOPS = 2500.0
# HBM-side traffic of the dominant kernel per launch: READ from the per-kernel sums of the PMC passes committed under
# profiles/ (scripts/profile_round.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs folded by scripts/pmc_sum.py;
# counter unit KB; FETCH_SIZE x 2 -- gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md "HBM")
PMC_TAG = "r06"
PMC_KERNELS = {"dense": ("backproject_accum",), "conv": ("sparse_conv_",)}


def pmc_traffic(workload, family, tag=None, profiles=None):
    """(bytes per launch, source) of a kernel family from profiles/<tag>_<workload>_pmc_{FETCH_SIZE,WRITE_SIZE}.csv:
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the family's instantiations / their dispatches.  None when the
    passes are not there (roofline.traffic is then null, never a stale constant)."""
    import csv
    tag = tag or PMC_TAG
    profiles = profiles or os.path.join(ROOT, "profiles")
    total, launches, files = 0.0, None, []
    for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = os.path.join(profiles, f"{tag}_{workload.lower()}_pmc_{counter}.csv")
        if not os.path.exists(path):
            return None
        kb, n = 0.0, 0
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in PMC_KERNELS[family]):
                    kb += float(r["Sum"])
                    n += int(r["Dispatches"])
        if n == 0:
            return None
        total += mult * kb * 1024.0 / n
        launches = n
        files.append(os.path.relpath(path, ROOT))
    return total, f"{files[0]} (x2) + {files[1]}: per-kernel sums over {launches} dispatches of {'/'.join(PMC_KERNELS[family])}*"


SQ_FAMILIES = {"gather_once_conv": ("sparse_conv_go2_kernel", "sparse_conv_go_kernel"), "stage_conv": ("sparse_conv_bf16x6_kernel",),
               "dense": ("backproject_accum",), "march": ("neus_march_kernel", "layout_march_kernel")}


def pmc_sq(workload, tag=None, profiles=None):
    """per kernel family, from the SQ counter pass committed under profiles/ (scripts/profile_sq.sh, one slot): MFMA-busy fraction
    = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), the split of the wave cycles (parked in s_waitcnt / issue-stalled /
    issuing) and LDS bank-conflict cycles per LDS-active cycle.  None when the pass is not there."""
    import csv
    path = os.path.join(profiles or os.path.join(ROOT, "profiles"), f"{tag or PMC_TAG}_{workload.lower()}_pmc_SQ.csv")
    if not os.path.exists(path):
        return None
    acc = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            for fam, pats in SQ_FAMILIES.items():
                if any(p_ in r["Kernel_Name"] for p_ in pats):
                    d = acc.setdefault(fam, {})
                    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Sum"])
    out = {"source": os.path.relpath(path, ROOT)}
    for fam, d in acc.items():
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        if not wc or not d.get("SQ_BUSY_CU_CYCLES"):
            continue
        out[fam] = dict(mfma_busy=d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * d["SQ_BUSY_CU_CYCLES"]),
                        wave_cycles_waitcnt=d.get("SQ_WAIT_ANY", 0.0) / wc, wave_cycles_issue_stall=d.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                        wave_cycles_issuing=d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                        lds_conflict_per_active=d.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, d.get("SQ_LDS_IDX_ACTIVE", 0.0)))
    return out


def mfma_count(workload, precision="f16x3", tag=None, profiles=None):
    """executed-over-algorithmic MFMA instructions of the gather-once convolutions of a scene, from the counter pass committed
    under profiles/ (scripts/profile_mfma.sh -> <tag>_mfma_count_<workload>_<precision>.log, last line).  None when not there."""
    import re
    path = os.path.join(profiles or os.path.join(ROOT, "profiles"), f"{tag or PMC_TAG}_mfma_count_{workload}_{precision}.log")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        last = [l for l in f.read().splitlines() if l.startswith("sum:")]
    m = last and re.search(r"executed / algorithmic ([\d.]+); executed / tile-mask ([\d.]+); busy / \(executed x (\d+)\) ([\d.]+); "
                           r"busy / \(4 x CU-busy\) ([\d.]+)", last[-1])
    if not m:
        return None
    return dict(executed_over_algorithmic=float(m.group(1)), executed_over_tile_mask_prediction=float(m.group(2)),
                busy_cycles_over_executed_x_cycles=float(m.group(4)), mfma_busy_in_this_pass=float(m.group(5)),
                source=os.path.relpath(path, ROOT))


def pmc_provenance(tag=None, profiles=None):
    """where the counter-derived numbers of the line come from, and whether they describe THIS build: the PMC
    passes are separate rocprofv3 runs committed under profiles/, not part of a bench run; scripts/profile_round.sh /
    profile_sq.sh record the sha256 of the library they profiled in profiles/<tag>_pmc_meta.json.  Returns
    dict(src="profiles/<tag>@<sha12>", matches_build=bool | None)."""
    import hashlib
    tag = tag or PMC_TAG
    path = os.path.join(profiles or os.path.join(ROOT, "profiles"), f"{tag}_pmc_meta.json")
    if not os.path.exists(path):
        return dict(src=f"profiles/{tag}@unrecorded", matches_build=None)
    with open(path) as f:
        meta = json.load(f)
    lib = os.path.join(ROOT, "cn-rma_amd", "csrc", "libcnrma_hip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
    return dict(src=f"profiles/{tag}@{str(meta.get('so_sha256', ''))[:12]}", matches_build=(sha == meta.get("so_sha256")) if sha else None)


PASS_DEFAULT = {}             # workload -> scenes per sparse-network pass (measured: see DESIGN.md "Scenes per pass")

_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries the ONE JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="NS", help="NS = north-star shape (default); S = ScanNet config; St; tiny")
    ap.add_argument("--slots", type=int, default=3, help="scene graphs in flight per GPU (each on its own HIP stream)")
    ap.add_argument("--scenes", type=int, default=8, help="distinct scenes rotated per GPU")
    ap.add_argument("--scenes-per-pass", type=int, default=0,
                    help="scenes collated into one sparse-network pass per graph (pipeline.StaticBatch; 1 = one scene per graph, "
                         "0 = the workload's default)")
    ap.add_argument("--scenes-per-step", type=int, default=0, help="0 = chosen from the warm-up (see --window-s)")
    ap.add_argument("--window-s", type=float, default=2.0, help="minimum length of one timed window of K steps")
    ap.add_argument("--windows", type=int, default=3)
    ap.add_argument("--eager", action="store_true", help="no graphs: the eager path (device->host reads per scene)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the S block, the f32-convolution block and the plugin block")
    ap.add_argument("--through-plugin", action="store_true",
                    help="drive the registered RayMarching detector -- model(return_loss=False, **data), {scene}_bbox_raw.npz "
                         "written per scene -- instead of pipeline.StaticScene directly")
    ap.add_argument("--plugin-static", type=int, default=1, help="--through-plugin only: 0 = force the detector's eager path "
                    "(what the reference's default numpy point sampler takes)")
    ap.add_argument("--dense-branch", type=int, default=0, help="A/B aid: 1 = the dense kernel as a parallel branch of the scene graph")
    ap.add_argument("--pace", type=int, default=1, help="1 = the host waits for a slot's previous scene before refilling it "
                    "(what a caller that consumes the detections does anyway; +1 %% over enqueueing blindly), 0 = enqueue as fast as possible")
    ap.add_argument("--feature-layout", default="channels_last", choices=["channels_last", "nchw"],
                    help="memory layout of the resident feature maps [V,C,H,W]: channels_last (default) = what the plugin's 2D stack "
                         "hands over (MultiViewBase.channels_last_2d: the 2D network runs in torch.channels_last), read in place; "
                         "nchw = the reference's layout, converted by the layout pass inside the timed path")
    ap.add_argument("--detail", default="", help="where the full result (kernel tables, per-layer convolution table, stage times, notes) "
                    "goes; default: bench_detail.json beside bench.py (and a copy under gpurun_out/ when that directory exists)")
    ap.add_argument("--dump-exchange", default="", help="test aid: directory that receives, after the last timed step of the main "
                    "workload, every rank's own padded detections (rank<r>_of<N>.npz) and rank 0's gathered block (gathered.npz)")
    ap.add_argument("--as-rank", type=int, default=-1, help="test aid (single process): generate the scenes rank R of a multi-rank "
                    "run would generate")
    ap.add_argument("--dense-tuning", default="", help="A/B aid: schedule switches of the dense kernel for this run, e.g. "
                    "'variant=0' = the round-2 kernel (cnrma_debug_dense_tuning; never set by the driver)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (torch.distributed.run)
    BEFORE anything in this process touches the GPU, and exit with their code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_model(C, device, n_classes=18, n_reg=6):
    import torch
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    from projects.mvsdetection.models.fcaf3d_head import FCAF3DHead
    torch.manual_seed(0)
    backbone = FCAF3DBackbone(C, 34)
    head = FCAF3DHead(n_classes, (64, 128, 256, 512), 128, n_reg, 0.01, 200000, None,
                      test_cfg=dict(nms_pre=1000, iou_thr=.5, score_thr=.01))
    backbone.init_weights()
    head.init_weights()
    return backbone.to(device).eval(), head.to(device).eval()


# ------------------------------------------------------------------------------------------------------------------
# per-kernel profile of ONE eager scene (outside the timed region): HIP events around every C-ABI call on the launch
# stream + the algorithmic work of every convolution (rows, kernel-map pairs)
# ------------------------------------------------------------------------------------------------------------------
CONV_CALLS = ("cnrma_sparse_conv_f32", "cnrma_sparse_conv_go_f32", "cnrma_sparse_conv_pairs_f32", "cnrma_sparse_conv_bf16x6", "cnrma_sparse_conv_f16x3", "cnrma_sparse_conv_pairs_f16x3",
              "cnrma_sparse_conv_go_f16x3", "cnrma_sparse_convtr_gen_f32",
              "cnrma_sparse_convtr_gen_bf16x6", "cnrma_sparse_convtr_gen_f16x3")


class KernelProfile:
    TIMED = ("cnrma_backproject_accum_f32", "cnrma_rma_neus_march_f32", "cnrma_rma_neus_emit_rows_f32",
             "cnrma_sparse_kernel_map_symmetric", "cnrma_sparse_kernel_map_strided", "cnrma_sparse_kernel_map",
             "cnrma_nchw_to_nhwc_f32", "cnrma_sparse_maxpool_f32", "cnrma_voxelize_f32", "cnrma_sample_mask",
             "cnrma_sparse_stride_coords", "cnrma_sparse_union_add_f32", "cnrma_sparse_instnorm_f32",
             "cnrma_sparse_tile_union_build") + CONV_CALLS

    def __init__(self):
        self.records, self.layers = [], []

    def install(self):
        import torch
        from cnrma_amd import _lib
        from cnrma_amd import sparse as S
        self._orig_call, self._orig_conv, self._orig_convtr = _lib.call, S.conv, S.conv_transpose_generative
        prof = self

        def timed_call(name, *args):
            if name not in prof.TIMED:
                return prof._orig_call(name, *args)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = prof._orig_call(name, *args)
            b.record()
            prof.records.append((name, a, b))
            return rc

        def conv(x, weight, kernel_size=3, stride=1, *a, **k):
            y = prof._orig_conv(x, weight, kernel_size, stride, *a, **k)
            K = kernel_size ** 3
            if K == 1 and stride == 1:
                pairs = y.cs.n
            else:
                pairs = int((x.cs.neighbours(y.cs, kernel_size, x.cs.stride) >= 0).sum())
            prof.layers.append(dict(K=K, Cin=x.F.shape[1], Cout=y.F.shape[1], n_in=x.cs.n, n_out=y.cs.n, pairs=pairs, taps=K))
            return y

        def convtr(x, weight, *a, **k):
            y = prof._orig_convtr(x, weight, *a, **k)
            prof.layers.append(dict(K=8, Cin=x.F.shape[1], Cout=y.F.shape[1], n_in=x.cs.n, n_out=y.cs.n, pairs=y.cs.n, taps=1))
            return y
        for mod in ("cnrma_amd._lib", "cnrma_amd.rma", "cnrma_amd.sparse"):
            sys.modules[mod].call = timed_call
        S.conv, S.conv_transpose_generative = conv, convtr
        sys.modules["cnrma_amd.nn"].S.conv = conv

    def uninstall(self):
        from cnrma_amd import sparse as S
        for mod in ("cnrma_amd._lib", "cnrma_amd.rma", "cnrma_amd.sparse"):
            sys.modules[mod].call = self._orig_call
        S.conv, S.conv_transpose_generative = self._orig_conv, self._orig_convtr

    def summary(self, reps):
        import torch
        torch.cuda.synchronize()
        agg = {}
        conv_ms = []
        for name, a, b in self.records:
            ms = a.elapsed_time(b)
            d = agg.setdefault(name, dict(ms=0.0, n=0))
            d["ms"] += ms
            d["n"] += 1
            if name in CONV_CALLS:
                conv_ms.append(ms)
        kern = {k: dict(ms_per_scene=v["ms"] / reps, launches_per_scene=v["n"] / reps) for k, v in agg.items()}
        assert len(conv_ms) == len(self.layers), (len(conv_ms), len(self.layers))
        n = len(self.layers) // reps
        layers = []
        for i in range(n):                       # same layer over the repetitions
            L = dict(self.layers[i])
            L["ms"] = sum(conv_ms[i + r * n] for r in range(reps)) / reps
            layers.append(L)
        return kern, layers


def algorithmic_bytes(V, C, H, W, dims, Ms, Mu, layers):
    """compulsory traffic of one scene, SURVEY.md 8(d) (every input read once, every output written once, fp32)"""
    G = dims[0] * dims[1] * dims[2]
    b_dense = 4 * V * C * H * W + 4 * C * G + 4 * G + 48 * V
    b_rma = 4 * G + 4 * Ms * C + 4 * Ms * (3 + C)          # selection fused into the emission: only kept rows move
    b_vox = 4 * Ms * (3 + C) + 16 * Mu + 4 * Mu * C
    b_conv = sum(4 * L["n_in"] * L["Cin"] + 4 * L["n_out"] * L["Cout"] + 4 * L["K"] * L["Cin"] * L["Cout"] + 8 * L["pairs"]
                 for L in layers)
    return dict(dense=b_dense, rma=b_rma, voxelize=b_vox, conv=b_conv, scene=b_dense + b_rma + b_vox + b_conv)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a CPU restatement of the reference's algorithm) on this host's cores
# ------------------------------------------------------------------------------------------------------------------
def _timeit(fn, warm=1, reps=3, budget_s=20.0):
    """1 warm-up + up to `reps` repetitions within a time budget; returns (best seconds, repetitions done)"""
    t0 = time.perf_counter()
    for _ in range(warm):
        fn()
    first = time.perf_counter() - t0
    times = []
    while len(times) < reps and (not times or (time.perf_counter() - t0) + min(times) < budget_s):
        t1 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t1)
    return (min(times) if times else first), len(times), first


def cpu_baseline(shape_name, Ms_full, C):
    """The oracle timed on this host: dense unprojection and RMA on a BOUNDED sample (the work is uniform per view, per
    channel and per image row, so the sample scales linearly; what was sampled is said in `sample`), the sparse network on
    the FULL selected point set.  All host cores, 1 warm-up + up to 3 repetitions per leg inside a time budget."""
    import torch
    from cnrma_amd import synth
    from oracle import rma_oracle as O
    from oracle import sparse_torch as ST
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    from projects.mvsdetection.models.fcaf3d_head import FCAF3DHead
    # torch's intra-op pool: all cores up to 64 -- on the 256-core host of the MI355X box 16 and 64 threads run the oracle's
    # legs equally fast, 256 threads run them 60x SLOWER (oversubscribed barriers; scripts/cpu_threads_probe.py)
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    V_full, C_full, H, W, dims, stride = synth.SHAPES[shape_name]
    big = H * W * 300 > 5e7                  # NS: 92 M samples per view in the oracle's [rays, steps] arrays
    n_views = 1 if big else 2
    C_s = min(C_full, 32)                    # channels of the sample (the gather / store work is linear in C)
    rows = H // 4 if big else H              # image rows of the RMA sample (one ray per pixel: linear in rows)
    shape = (n_views, C_s, H, W, dims, stride)
    sc = synth.make_scene(shape, seed=0)
    proj, feat, tsdf = sc["projection"][:, 0], sc["features"][:, 0], sc["tsdf"][0, 0]
    log(f"  dense leg: {n_views} view(s) x {C_s} channels")
    t_dense, r_dense, _ = _timeit(lambda: [O.backproject_view(sc["dims"], 0.04, sc["origin"], O.scale_projection(proj[v], stride),
                                                             feat[v]) for v in range(n_views)], reps=3, budget_s=12.0)
    keep = {}
    feat_r = feat[:, :, :rows].contiguous()  # the top `rows` image rows: ray (u, v) only depends on its own pixel

    def rma_leg():
        keep["pts"] = O.aggregate_rma(proj, feat_r, tsdf, sc["dims"], 0.04, sc["origin"], stride)
    log(f"  RMA leg: {n_views} view(s) x {rows} of {H} rows")
    t_rma, r_rma, _ = _timeit(rma_leg, reps=3, budget_s=20.0)
    pts = keep["pts"]
    torch.manual_seed(0)
    backbone = FCAF3DBackbone(C, 34).eval()
    head = FCAF3DHead(18, (64, 128, 256, 512), 128, 6, 0.01, 200000, None, test_cfg=dict(nms_pre=1000)).eval()
    backbone.init_weights()
    head.init_weights()
    # a full-size point set for the sparse leg: the sampled rows (randomly thinned, or repeated with a one-voxel offset
    # per copy) with full-width random features -- its cost depends on the voxel structure, not on the feature values
    if pts.shape[0] >= Ms_full:
        pick = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(0))[:Ms_full].sort()[0]
        xyz = pts[pick, :3]
    else:
        reps_needed = -(-Ms_full // pts.shape[0])
        xyz = torch.cat([pts[:, :3] + torch.tensor([0.0, 0.0, 0.01 * r]) for r in range(reps_needed)])[:Ms_full]
    f = torch.randn(xyz.shape[0], C, generator=torch.Generator().manual_seed(1))

    def sparse_leg():
        Cq, Fq, _ = O.voxelize(xyz, f, 0.01)
        res = ST.head_forward(head, ST.backbone_forward(backbone, Cq.numpy(), Fq.numpy()))
        ST.get_bboxes(head, res)
    log(f"  sparse leg: {xyz.shape[0]} points x {C} channels")
    t_sparse, r_sparse, _ = _timeit(sparse_leg, reps=3, budget_s=25.0)
    dense_scene = t_dense / n_views * V_full * (C_full / C_s)
    rma_scene = t_rma / n_views * V_full * (H / rows)
    est = dense_scene + rma_scene + t_sparse
    fx_dense, fx_rma = V_full / n_views * (C_full / C_s), V_full / n_views * (H / rows)
    return dict(value=1.0 / est, unit="scenes/s", cores=cores, kind="port",
                stage_s=dict(dense=dense_scene, rma=rma_scene, sparse=t_sparse),
                sample_short=f"oracle/ on {cores} threads: dense {n_views}/{V_full} views x {C_s}/{C_full} ch (x{fx_dense:g}), RMA {n_views}/"
                             f"{V_full} views x {rows}/{H} rows (x{fx_rma:g}), sparse net full {Ms_full}-point set (x1)",
                extrapolated=dict(dense=f"x{fx_dense:g} ({n_views} of {V_full} views x {C_s} of {C_full} channels, linear)",
                                  rma=f"x{fx_rma:g} ({n_views} of {V_full} views x {rows} of {H} image rows, linear)",
                                  sparse="x1 (full point set)"),
                sample=f"oracle/ (torch-CPU restatement of the reference, {cores} threads of {os.cpu_count()} host cores -- more "
                       f"threads ran slower --, best of up to 3 reps after 1 warm-up): dense unprojection on {n_views} of {V_full} views x {C_s} of {C_full} channels "
                       f"({t_dense:.2f} s, {r_dense} reps); RMA on {n_views} of {V_full} views x {rows} of {H} image rows "
                       f"({t_rma:.2f} s, {r_rma} reps); both scaled linearly to the scene ({dense_scene:.1f} s + {rma_scene:.1f} s); "
                       f"voxelise + FCAF3D + decode on a FULL {Ms_full}-point set ({t_sparse:.2f} s, {r_sparse} reps, fp32 port "
                       f"oracle/sparse_torch.py)")


# ------------------------------------------------------------------------------------------------------------------
class Workload:
    """>= n_scenes distinct synthetic scenes of one shape resident in HBM + `slots` captured scene graphs"""

    def __init__(self, name, device, rank, world, args, n_classes=18, n_reg=6, plugin=False, layout=None,
                 config="ray_marching_scannet.py"):
        import torch
        self.plugin = plugin
        self.config_file = config
        self.layout = layout or args.feature_layout
        from cnrma_amd import pipeline, synth
        self.torch, self.pipeline = torch, pipeline
        self.name, self.device, self.rank, self.world, self.args = name, device, rank, world, args
        self.V, self.C, self.H, self.W, self.dims, self.stride = synth.SHAPES[name]
        self.scenes = []
        self.seed_rank = rank if getattr(args, "as_rank", -1) < 0 else args.as_rank
        for i in range(args.scenes):
            sc = synth.make_scene(name, seed=1000 * self.seed_rank + i, boxes=2 + i % 4, device=device, channels_last=self.layout == "channels_last")
            f = sc["features"][:, 0]                       # [V,C,H,W]; channels_last: a permuted view of a [V,H,W,C] block
            self.scenes.append(dict(features=f if self.layout == "channels_last" else f.contiguous(), projection=sc["projection"][:, 0],
                                    tsdf=sc["tsdf"][0, 0].to(device)))
        self.input_bytes = sum(s["features"].numel() * 4 + s["tsdf"].numel() * 4 for s in self.scenes)
        if plugin:
            self._build_plugin_model(n_classes, n_reg)
            for i, sc_ in enumerate(self.scenes):       # what the DataContainer scatter hands the detector: device tensors
                sc_.update(projection_dev=sc_["projection"].to(device), name=f"scene{rank:02d}{i:02d}_00",
                           offset=torch.tensor([0.04 * i, -0.04 * i, 0.0], device=device))
        else:
            self.backbone, self.head = build_model(self.C, device, n_classes, n_reg)
        self.cfg = pipeline.SceneConfig(self.dims, stride=self.stride, max_points=500000, sampler="device")
        self.det_w = (6 if n_reg == 6 else 7) + n_classes
        self.slots = []
        self._k = 0

    def _build_plugin_model(self, n_classes, n_reg):
        """the registered detector built from the ScanNet config's model section (hot-path form: feature maps and TSDF
        come in as inputs), at this workload's shape; device point sampler (the graph path's precondition)"""
        import runpy
        import tempfile
        import torch
        import projects.mvsdetection  # noqa: F401
        from projects.mvsdetection.registry import build_model as build_detector
        cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", self.config_file))
        m = dict(cfg["model"])
        self.save_dir = tempfile.mkdtemp(prefix="cnrma_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        # the config's model section as shipped (max_points, point sampler, graph path: its defaults); only what defines the
        # measured hot path is replaced: the 2D / Atlas 3D networks are not built (their OUTPUTS are the resident inputs, SURVEY
        # 8d), the grid and the stride are the workload's, the result files go to tmpfs
        m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None, save_path=self.save_dir,
                 voxel_dim_test=list(self.dims), voxel_dim_train=list(self.dims), backbone2d_stride=self.stride)
        m["detection_backbone"] = dict(type="FCAF3DBackbone", in_channels=self.C, depth=34)
        torch.manual_seed(0)
        model = build_detector(m)
        model.detection_backbone.init_weights()
        model.detection_head.init_weights()
        self.model = model.to(self.device).eval()
        # runtime attributes, not constructor keywords: slots in flight as the rest of the bench, every resident scene calibrates
        self.model.static_slots, self.model.static_calibration = self.args.slots, self.args.scenes
        if not self.args.plugin_static:
            self.model.static_test = False
        self.backbone, self.head = self.model.detection_backbone, self.model.detection_head

    def _plugin_scene(self, s):
        return self.model(return_loss=False, features=[s["features"]], projection=[s["projection_dev"]], tsdf=s["tsdf"][None, None],
                          offset=[s["offset"]], scene=[s["name"]])

    def build(self):
        torch, pipeline = self.torch, self.pipeline
        if self.plugin:
            with torch.no_grad():
                for s in self.scenes:                            # the calibration scenes (eager); the last one builds the slots
                    self._plugin_scene(s)
            if not self.args.plugin_static:
                self.slots = []
                return
            ctx = next(iter(self.model._static.values()))
            assert ctx["built"]
            self.slots = ctx["slots"]
            return
        if self.args.eager:
            return
        first = pipeline.StaticScene(self.cfg, self.backbone, self.head, self.device)
        for s in self.scenes:                                    # capacities cover every scene of the rotation
            first.calibrate(s["features"], s["projection"], s["tsdf"])
        s0 = self.scenes[0]
        self.B = self.args.scenes_per_pass or PASS_DEFAULT.get(self.name, 1)
        if self.B > 1:
            group = [(s_["features"], s_["projection"], s_["tsdf"]) for s_ in self.scenes[:self.B]]
            self.slots = []
            for _ in range(self.args.slots):
                sb = pipeline.StaticBatch(self.cfg, self.backbone, self.head, self.device, self.B)
                sb.build(group, first.plan)
                self.slots.append(sb)
            del first
            self.bad = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in self.slots]
            self.det = None
            return
        first.build(s0["features"], s0["projection"], s0["tsdf"])
        self.slots = [first]
        for _ in range(1, self.args.slots):
            st = pipeline.StaticScene(self.cfg, self.backbone, self.head, self.device)
            st.build(s0["features"], s0["projection"], s0["tsdf"], plan=first.plan)
            self.slots.append(st)
        self.bad = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in self.slots]
        self.det = None

    def alloc_step_buffers(self, sps):
        torch = self.torch
        self.sps = sps
        if self.args.eager or self.plugin:
            return
        torch.cuda.synchronize()
        o0 = self.slots[0].out
        n_det, n_lvl = o0["bboxes"].shape[-2], o0["valid"].shape[-1]                                 # padded rows, levels
        self.det = torch.zeros((sps, n_det, self.det_w), dtype=torch.float32, device=self.device)
        self.det_valid = torch.zeros((sps, n_lvl), dtype=torch.int32, device=self.device)
        if self.world > 1:
            self.det_all = torch.empty((self.world,) + tuple(self.det.shape), dtype=torch.float32, device=self.device)
            self.valid_all = torch.empty((self.world, sps, n_lvl), dtype=torch.int32, device=self.device)

    def step(self):
        """one wave of sps scenes on this GPU (+ one all-gather of the detections when there are several ranks)"""
        torch = self.torch
        if self.plugin:
            with torch.no_grad():
                for j in range(self.sps):
                    self._plugin_scene(self.scenes[self._k % len(self.scenes)])
                    self._k += 1
            return
        if self.args.eager:
            for j in range(self.sps):
                s = self.scenes[self._k % len(self.scenes)]
                self._k += 1
                self.last = self.pipeline.forward_scene(self.cfg, self.backbone, self.head, s["features"], s["projection"], s["tsdf"])
            return
        main = torch.cuda.current_stream()
        if getattr(self, "B", 1) > 1:
            B = self.B
            for j in range(0, self.sps, B):
                group = []
                for _ in range(B):
                    s = self.scenes[self._k % len(self.scenes)]
                    self._k += 1
                    group.append((s["features"], s["projection"], s["tsdf"]))
                i = (self._k // B) % len(self.slots)
                sb = self.slots[i]
                out = sb.run(group)
                with torch.cuda.stream(sb.stream):
                    nb = out["bboxes"].shape[2]
                    self.det[j:j + B, :, :nb].copy_(out["bboxes"], non_blocking=True)
                    self.det[j:j + B, :, nb:].copy_(out["scores"], non_blocking=True)
                    self.det_valid[j:j + B].copy_(out["valid"], non_blocking=True)
                    self.bad[i] += out["status"]
                self.last_out, self.last_slot = out, sb
        for j in range(self.sps if getattr(self, "B", 1) <= 1 else 0):
            s = self.scenes[self._k % len(self.scenes)]
            i = self._k % len(self.slots)
            self._k += 1
            st = self.slots[i]
            if self.args.pace and getattr(st, "done", None) is not None:
                st.done.synchronize()                           # host paces itself: a slot is refilled when its scene has left it
            out = st.run(s["features"], s["projection"], s["tsdf"])
            with torch.cuda.stream(st.stream):                  # results leave the slot's static buffers in stream order
                nb = out["bboxes"].shape[1]
                self.det[j, :, :nb].copy_(out["bboxes"], non_blocking=True)
                self.det[j, :, nb:].copy_(out["scores"], non_blocking=True)
                self.det_valid[j].copy_(out["valid"], non_blocking=True)
                self.bad[i] += out["status"]
            self.last_out = out
        if self.world > 1:
            import torch.distributed as dist
            for st in self.slots:
                main.wait_stream(st.stream)
            if dist.get_backend() != "nccl":
                main.synchronize()                              # test hook (gloo): the exchange goes through the host
            self.pipeline.gather_padded_detections(self.det, self.det_valid, self.det_all, self.valid_all)
            for st in self.slots:                               # the next wave overwrites det: after the collective
                st.stream.wait_stream(main)

    def drain(self):
        """end of a timed window: every result of the window is where its consumer reads it (plugin: the files)"""
        if self.plugin:
            self.model.flush()

    def violations(self):
        if self.plugin:
            return int(getattr(self.model, "static_fallbacks", 0))
        if self.args.eager:
            return 0
        self.torch.cuda.synchronize()
        return int(sum(int(b.item()) for b in self.bad))

    def sizes(self):
        if self.plugin and not self.slots:
            return dict(M_rows=None, M_selected=None, M_unique=None, level_rows=None, head_rows=None)
        if self.plugin:
            st = self.slots[0]
            st.run(self.scenes[0]["features"], self.scenes[0]["projection"], self.scenes[0]["tsdf"])
            _, _, info = self.pipeline.StaticScene.detections(st.out)
            return dict(M_rows=info["M"], M_selected=info["M_selected"], M_unique=info["M_unique"], level_rows=info["level_rows"],
                        head_rows=info["head_rows"])
        if self.args.eager:
            o = self.last
            return dict(M_rows=o["M"], M_selected=o["M_selected"], M_unique=o["M_unique"], level_rows=o["level_rows"],
                        head_rows=o["head_rows"])
        if getattr(self, "B", 1) > 1:
            info = self.last_slot.detections(self.last_out)[0][2]
        else:
            _, _, info = self.pipeline.StaticScene.detections(self.last_out)
        return dict(M_rows=info["M"], M_selected=info["M_selected"], M_unique=info["M_unique"], level_rows=info["level_rows"],
                    head_rows=info["head_rows"])


def time_windows(wl, args, world, barrier):
    """warm-up, choice of scenes_per_step, then `windows` windows of exactly `steps` steps; returns per-window seconds"""
    import torch
    q = max(1, len(wl.slots)) * getattr(wl, "B", 1)                     # scenes of one round over the slots
    sps = args.scenes_per_step or max(2, q)
    sps = -(-sps // getattr(wl, "B", 1)) * getattr(wl, "B", 1)
    wl.alloc_step_buffers(sps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(max(1, args.warmup)):
        wl.step()
    wl.drain()
    barrier()
    warm = (time.perf_counter() - t0) / max(1, args.warmup) / sps        # seconds per scene, warm-up estimate
    if not args.scenes_per_step:
        want = args.window_s / max(1, args.steps) / max(warm, 1e-6)
        sps = int(min(256, max(q, -(-int(want + 0.999) // q) * q)))
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([sps], dtype=torch.int64, device=wl.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sps = int(t.item())
        wl.alloc_step_buffers(sps)
        for _ in range(2):                                               # warm passes at the final wave size
            wl.step()
    barrier()
    secs, per_rank = [], []
    for _ in range(max(1, args.windows)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        wl.drain()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            mine = torch.tensor([dt], dtype=torch.float64, device=wl.device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)                                 # each rank's own clock around the same K steps
            per_rank.append([float(x.item()) for x in every])
            dt = max(per_rank[-1])                                       # MAX over ranks
        secs.append(dt)
    wl.per_rank_window_s = per_rank
    return sps, secs


def measure(name, device, rank, world, args, barrier, precision=None, plugin=False, layout=None, config=None):
    """build + time one workload; returns the result block"""
    import torch
    from cnrma_amd import sparse as S
    prev = S.CONV_PRECISION
    if precision:
        S.CONV_PRECISION = precision
    try:
        log(f"workload {name}{' (' + precision + ' conv)' if precision else ''}: generating {args.scenes} scenes")
        wl = Workload(name, device, rank, world, args, plugin=plugin, layout=layout, **({"config": config} if config else {}))
        log("calibrating + capturing the scene graphs")
        wl.build()
        log("timing")
        c0 = os.times()
        sps, secs = time_windows(wl, args, world, barrier)
        c1 = os.times()
        if getattr(args, "dump_exchange", "") and not plugin and precision is None and layout is None and getattr(wl, "det", None) is not None:
            import numpy as np
            torch.cuda.synchronize()
            os.makedirs(args.dump_exchange, exist_ok=True)
            np.savez(os.path.join(args.dump_exchange, f"rank{wl.seed_rank}_of{world}.npz"), det=wl.det.cpu().numpy(),
                     valid=wl.det_valid.cpu().numpy(), sizes=np.asarray(wl.slots[0].out["sizes"], dtype=np.int64))
            if world > 1 and rank == 0:
                np.savez(os.path.join(args.dump_exchange, "gathered.npz"), det_all=wl.det_all.cpu().numpy(),
                         valid_all=wl.valid_all.cpu().numpy())
        bad = wl.violations()
        med = sorted(secs)[len(secs) // 2]
        block = dict(value=world * args.steps * sps / med, ms_per_step=med / args.steps * 1e3, ms_per_scene=med / args.steps / sps * 1e3,
                     scenes_per_step=sps, windows_scenes_per_s=[world * args.steps * sps / s for s in secs],
                     window_s=med, plan_violations=bad, distinct_scenes=len(wl.scenes), input_GB=wl.input_bytes / 1e9,
                     host_cpu_cores_busy=round(((c1.user - c0.user) + (c1.system - c0.system)) / max(sum(secs), 1e-9), 2))
        block.update(wl.sizes())
        if world > 1:
            block["per_rank_window_s"] = wl.per_rank_window_s            # [window][rank]: what every rank's clock saw
        n_nodes = getattr(wl.slots[0], "n_nodes", None) if wl.slots else None
        block["graph_nodes_per_scene"] = n_nodes
        log(f"{name}: {block['value']:.1f} scenes/s, {block['ms_per_scene']:.2f} ms/scene, windows "
            f"{[round(x, 1) for x in block['windows_scenes_per_s']]}, {sps} scenes/step, violations {bad}")
        return wl, block
    finally:
        S.CONV_PRECISION = prev


def profile_block(wl, block, name):
    """per-kernel times (HIP events, eager pass outside the timed region), conv layer table, roofline objects"""
    import torch
    from cnrma_amd import pipeline
    prof = KernelProfile()
    s = wl.scenes[0]
    reps = 2
    pipeline.forward_scene(wl.cfg, wl.backbone, wl.head, s["features"], s["projection"], s["tsdf"])     # warm
    prof.install()
    try:
        for _ in range(reps):
            pipeline.forward_scene(wl.cfg, wl.backbone, wl.head, s["features"], s["projection"], s["tsdf"])
    finally:
        prof.uninstall()
    kern, layers = prof.summary(reps)
    stage = None
    for _ in range(3):                  # per-stage minimum over three eager passes (allocator growth lands in single passes)
        st_ = pipeline.forward_scene(wl.cfg, wl.backbone, wl.head, s["features"], s["projection"], s["tsdf"], timing=True)["stage_ms"]
        stage = st_ if stage is None else {k: min(stage[k], st_[k]) for k in stage}
    B = algorithmic_bytes(wl.V, wl.C, wl.H, wl.W, wl.dims, block["M_selected"], block["M_unique"], layers)
    dense_ms = kern["cnrma_backproject_accum_f32"]["ms_per_scene"]
    kern["cnrma_backproject_accum_f32"].update(algorithmic_GB=B["dense"] / 1e9, GBps=B["dense"] / 1e6 / dense_ms,
                                               frac_hbm=B["dense"] / 1e6 / dense_ms / HBM_PEAK_GBS)
    conv_ms = sum(L["ms"] for L in layers)
    F_alg = sum(2.0 * L["pairs"] * L["Cin"] * L["Cout"] for L in layers)
    F_dense = sum(2.0 * L["taps"] * L["n_out"] * L["Cin"] * L["Cout"] for L in layers)     # every tap of every output row
    conv = dict(ms_per_scene=conv_ms, launches=len(layers), pairs=int(sum(L["pairs"] for L in layers)),
                algorithmic_TFLOP=F_alg / 1e12, algorithmic_TFLOPps=F_alg / 1e9 / conv_ms,
                occupancy_pairs_over_K_rows=F_alg / F_dense, algorithmic_GB=B["conv"] / 1e9,
                frac_f16_mfma_peak_x3=3.0 * F_alg / 1e9 / conv_ms / MFMA_F16_PEAK_TFLOPS,
                frac_f32_mfma_peak=F_alg / 1e9 / conv_ms / MFMA_F32_PEAK_TFLOPS,
                note="algorithmic = 2*pairs*Cin*Cout per layer (SURVEY 8d); the default f16x3 path spends 3 fp16 MFMA products "
                     "per fp32 product, so the fp16-MFMA fraction is quoted on 3x the algorithmic flops")
    block["kernels"] = kern
    block["stage_ms"] = stage
    block["conv"] = conv
    block["pmc_sq"] = pmc_sq(name)
    block["pmc_provenance"] = pmc_provenance()
    if block["pmc_sq"] and "gather_once_conv" in block["pmc_sq"]:
        # a counter pass of ANOTHER build (or of the exact-fp32 kernels) says nothing about these launches: null, not a stale number
        from cnrma_amd import sparse as _S
        f16x3 = _S.CONV_PRECISION == "f16x3"
        ok = block["pmc_provenance"]["matches_build"] is not False and f16x3
        conv["mfma_busy_gather_once"] = block["pmc_sq"]["gather_once_conv"]["mfma_busy"] if ok else None
        conv["mfma_busy_src"] = block["pmc_provenance"]["src"] + ("" if ok else " (another build: not reported)")
    conv["mfma_count"] = mfma_count(name) if name in ("S", "NS") else None     # executed vs algorithmic MFMAs (counter pass)
    block["conv_layers"] = [dict(K=L["K"], Cin=L["Cin"], Cout=L["Cout"], rows=L["n_out"], pairs=L["pairs"], ms=round(L["ms"], 4))
                            for L in layers]
    ms_scene = block["ms_per_scene"]
    block["algorithmic_GB_per_scene"] = {k: v / 1e9 for k, v in B.items()}
    block["whole_path_hbm"] = dict(GBps=B["scene"] / 1e6 / ms_scene, frac_of_8TBps=B["scene"] / 1e6 / ms_scene / HBM_PEAK_GBS,
                                   frac_of_6p3TBps=B["scene"] / 1e6 / ms_scene / HBM_ACHIEVABLE_GBS)
    # ---- roofline of the dominant kernel
    times = {"dense": dense_ms, "conv": conv_ms, "march": kern.get("cnrma_rma_neus_march_f32", {}).get("ms_per_scene", 0.0),
             "nhwc": kern.get("cnrma_nchw_to_nhwc_f32", {}).get("ms_per_scene", 0.0)}
    dominant = max(times, key=times.get)
    if dominant in ("dense", "nhwc", "march"):
        # the march is cache-resident VALU work and the layout pass is not algorithmic traffic: the HBM-bound kernel the
        # metric is about is the dense unprojection; report it (and say which kernel is actually the longest)
        tr = pmc_traffic(name, "dense")
        roof = dict(kernel="backproject_accum kernel (cnrma_backproject_accum_f32)", bound="hbm", achieved=B["dense"] / 1e6 / dense_ms,
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=B["dense"] / 1e6 / dense_ms / HBM_PEAK_GBS,
                    traffic=tr[0] if tr else None, traffic_source=tr[1] if tr else None, launch_ms=dense_ms,
                    algorithmic_bytes_per_launch=B["dense"], longest_kernel=dominant,
                    note="algorithmic bytes = 4*V*C*H*W + 4*C*G + 4*G + 48*V (SURVEY 8d) / HIP-event launch time on the launch "
                         "stream (eager pass outside the timed region)")
    else:
        tr = pmc_traffic(name, "conv")
        roof = dict(kernel="sparse_conv_go2_kernel + sparse_conv_bf16x6_kernel<.., MODE=1> (all f16x3 conv launches of a scene)",
                    bound="mfma", achieved=F_alg / 1e9 / conv_ms, peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=F_alg / 1e9 / conv_ms / MFMA_F16_PEAK_TFLOPS, traffic=tr[0] if tr else None,
                    traffic_source=tr[1] if tr else None, launch_ms=conv_ms / max(1, len(layers)),
                    frac_counting_the_3_fp16_products=3.0 * F_alg / 1e9 / conv_ms / MFMA_F16_PEAK_TFLOPS,
                    note="achieved = algorithmic flops (2*pairs*Cin*Cout per layer, SURVEY 8d) / summed launch time, priced "
                         "against the dense fp16 MFMA peak; the f16x3 arithmetic issues 3 fp16 products per fp32 product, "
                         "so the MFMA pipe itself runs at 3x this fraction")
    return roof


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] on one GPU: forward + backward + optimiser step of the detector at the ScanNet shape, bf16 autocast
# ------------------------------------------------------------------------------------------------------------------
def train_block(device, steps=8, warm=3):
    """RayMarching.train_step built from projects/configs/mvsdetection/ray_marching_scannet.py's model section (2D / Atlas networks
    not built: feature maps and TSDF are the inputs, gradients flow back into the feature maps), synthetic scene + 12 boxes, SGD;
    the eager path (the static trace covers inference only).  Returns ms per step and scenes/s (one scene per step)."""
    import runpy
    import numpy as np
    import torch
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model as build_registered
    from cnrma_amd import synth
    import gc
    gc.collect()                                 # graphs / plans of the inference blocks measured before (reference cycles)
    torch.cuda.empty_cache()
    resident = torch.cuda.memory_allocated()     # what other blocks of this process still hold: not this block's footprint
    sc = synth.make_scene("S", seed=0)
    C = sc["features"].shape[2]
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None, save_path="/tmp/cnrma_bench_train",
             voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]), use_feature_transform=False,
             detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=34))
    torch.manual_seed(0)
    model = build_registered(m)
    model.detection_backbone.init_weights()
    model.detection_head.init_weights()
    model = model.to(device).train()
    ext = np.array(sc["dims"], dtype=np.float32) * 0.04
    rng = np.random.RandomState(0)
    boxes = torch.tensor([[rng.uniform(.2, .8) * ext[0], rng.uniform(.2, .8) * ext[1], rng.uniform(0, .3) * ext[2], .8, .6, .7, 0.]
                          for _ in range(12)], dtype=torch.float32, device=device)
    labels = torch.from_numpy(rng.randint(0, 18, size=12)).to(device)
    feats = sc["features"][:, 0].to(device).requires_grad_(True)
    data = dict(features=[feats], projection=[sc["projection"][:, 0].to(device)], tsdf=sc["tsdf"].to(device),
                offset=[torch.zeros(3, device=device)], gt_bboxes_3d=[boxes], gt_labels_3d=[labels])
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model.train_step(dict(data), None)
        opt.zero_grad()
        feats.grad = None
        out["loss"].backward()
        opt.step()
        return out["loss"].detach()
    for _ in range(warm):
        loss = step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()        # the block's own peak, not that of the inference workloads measured before it
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return dict(value=1e3 / ms, unit="scenes/s", ms_per_step=ms, steps=steps, loss=float(loss), dtype="bf16 autocast (fp32 master weights)",
                peak_memory_GiB=(torch.cuda.max_memory_allocated() - resident) / 2 ** 30,      # model + scene + the step's own peak
                resident_before_GiB=resident / 2 ** 30,
                note="BASELINE configs[4] on ONE GPU: forward + backward (aggregation, sparse convolutions dgrad + wgrad, losses) + SGD "
                     "step per scene at the ScanNet shape, eager path (~2 k launches per step)")


# ------------------------------------------------------------------------------------------------------------------
# output: ONE stdout line of <= 4 KB (what the driver parses) + everything else in bench_detail.json beside bench.py
# ------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms")
DETAIL_NAME = "bench_detail.json"


def _r(x, nd=4):
    """floats to `nd` significant digits (the line is for reading and parsing, the detail file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    return x


def _roof(r):
    if not r:
        return None
    out = {k: r.get(k) for k in ROOF_KEYS}
    out["kernel"] = str(out["kernel"])[:96]
    return out


def _cpu(c, sample=True):
    if not c:
        return None
    out = dict(value=c["value"], unit=c["unit"], cores=c["cores"], kind=c["kind"], stage_s=c.get("stage_s"))
    if sample:
        out["sample"] = str(c.get("sample_short") or c.get("sample", ""))[:200]
    return out


def compact_line(result):
    """The driver-facing line: the contract's keys + `roofline` + `cpu_baseline` (+ the S block and the three labelled
    variants as numbers only).  Kernel tables, per-layer convolution tables, stage times, notes and the long `sample` /
    `traffic_source` strings live in bench_detail.json.  Pure function of `result` (tests/test_bench_line_cpu.py)."""
    cfg = result.get("config", {})
    line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg[k] for k in ("workload", "scenes_per_step", "M_rows", "M_selected", "M_unique", "level_rows", "head_rows")
                      if k in cfg}
    for k in ("ms_per_scene", "plan_violations", "value_f32_conv", "feature_layout", "graph_nodes_per_scene"):
        if k in result:
            line[k] = result[k]
    if result.get("conv"):
        line["conv_ms_per_scene"] = result["conv"].get("ms_per_scene")
        line["conv_mfma_busy"] = result["conv"].get("mfma_busy_gather_once")
        line["pmc_src"] = result["conv"].get("mfma_busy_src")            # roofline.traffic and conv_mfma_busy: committed counter passes
        line["conv_mfma_executed_over_algorithmic"] = (result["conv"].get("mfma_count") or {}).get("executed_over_algorithmic")
    if result.get("whole_path_hbm"):
        line["whole_path_hbm_frac"] = result["whole_path_hbm"].get("frac_of_8TBps")
    line["roofline"] = _roof(result.get("roofline"))
    line["cpu_baseline"] = _cpu(result.get("cpu_baseline"))
    if result.get("S"):
        s = result["S"]
        line["S"] = dict(value=s["value"], ms_per_step=s["ms_per_step"], ms_per_scene=s.get("ms_per_scene"),
                         graph_nodes_per_scene=s.get("graph_nodes_per_scene"),
                         conv_ms_per_scene=(s.get("conv") or {}).get("ms_per_scene"),
                         conv_mfma_busy=(s.get("conv") or {}).get("mfma_busy_gather_once"),
                         conv_mfma_executed_over_algorithmic=((s.get("conv") or {}).get("mfma_count") or {}).get("executed_over_algorithmic"),
                         roofline=_roof(s.get("roofline")), cpu_baseline=_cpu(s.get("cpu_baseline"), sample=False))
    for k in ("through_plugin", "nchw_input", "f32_conv", "St", "A", "train_S"):
        if result.get(k):
            line[k] = {kk: result[k][kk] for kk in ("value", "ms_per_step", "graph_nodes_per_scene", "n_reg_outs") if kk in result[k]}
    if result.get("dist"):
        d = result["dist"]
        line["dist"] = dict(world_size=d.get("world_size"), backend=d.get("backend"))
    line["detail"] = DETAIL_NAME
    line = _r(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:                                          # never print a line the driver cannot parse
        for k in ("train_S", "St", "A", "f32_conv", "nchw_input", "through_plugin", "dist"):
            line.pop(k, None)
        line["config"] = {k: v for k, v in line["config"].items() if k in ("workload", "scenes_per_step")}
        line["config"]["workload"] = line["config"].get("workload", "")[:200]
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(result, detail_path=""):
    """detail file + stderr first, then the ONE stdout line, last thing this process prints"""
    text = compact_line(result)
    detail = json.dumps(result, indent=1, default=str)
    targets = [detail_path] if detail_path else [os.path.join(ROOT, DETAIL_NAME), os.path.join(ROOT, "gpurun_out", DETAIL_NAME)]
    for path in targets:
        try:
            if os.path.isdir(os.path.dirname(os.path.abspath(path))):
                with open(path, "w") as f:
                    f.write(detail)
        except OSError as e:                                             # a read-only tree must not cost the line
            log(f"could not write {path}: {e}")
    log(f"detail: {DETAIL_NAME} ({len(detail)} bytes); line: {len(text)} bytes")
    sys.stderr.flush()
    sys.stdout.write(text + "\n")
    sys.stdout.flush()


def _released(tag):
    """after a block: collect reference cycles (plans <-> tensors, graphs), hand the cached blocks back, say what is still held"""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    log(f"{tag}: {torch.cuda.memory_allocated() / 2 ** 30:.1f} GiB still allocated")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for a 1-GPU box: run the N-rank code path with all ranks on device 0 over gloo (RCCL refuses two
    # ranks on one device).  Never set by the driver; the numbers of such a run are meaningless.
    backend = os.environ.get("CNRMA_BENCH_BACKEND", "nccl")
    if os.environ.get("CNRMA_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    from cnrma_amd import _lib
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=device if backend == "nccl" else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.dense_branch:
        from cnrma_amd import pipeline as _pl
        _pl.DENSE_BRANCH = True
    if args.dense_tuning:
        from cnrma_amd import rma
        rma.dense_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.dense_tuning.split(","))})

    wl, main_block = measure(args.workload, device, rank, world, args, barrier, plugin=args.through_plugin)
    V, C, H, W, dims, stride = wl.V, wl.C, wl.H, wl.W, wl.dims, wl.stride
    result = {
        "metric": "scenes/sec fwd (40-view->192^3 voxel): dense unprojection + RMA + voxelise + FCAF3D + decode",
        "value": main_block["value"], "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_block["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (sparse conv f16x3: 22-bit operands, fp32 accumulate; value_f32_conv = exact fp32)", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {V} views x {C} ch x {H}x{W} fp32 maps ({wl.layout}) -> grid {dims[0]}x{dims[1]}x{dims[2]}, "
                               f"N=300, max_points=500000, FCAF3D MinkResNet34 + head, 1 HIP graph per scene, {len(wl.slots) or 1} in flight",
                   "scenes_per_step": main_block["scenes_per_step"],
                   "description": f"{args.workload}: V={V} views, C={C}, feature maps {H}x{W} (stride {stride}, fp32 [V,C,H,W], " +
                               ("channels-last in memory -- what the plugin's 2D stack, run in torch.channels_last, hands over "
                                "(MultiViewBase.channels_last_2d) --, read in place by reference; block `nchw_input` = the same maps "
                                "handed over NCHW, layout pass inside the timed path" if wl.layout == "channels_last" else
                                "NCHW as the reference's 2D backbone writes them; the layout pass to channels-last is inside the timed "
                                "path") +
                               f"), grid {dims[0]}x{dims[1]}x{dims[2]} @0.04m, N=300 steps, thr=0.05, max_points=500000 (device sampler: "
                               f"a uniformly random subset like np.random.choice, other random stream), FCAF3D MinkResNet34 + "
                               f"head (18 classes); {main_block['distinct_scenes']} distinct scenes per GPU rotated "
                               f"({main_block['input_GB']:.1f} GB of inputs), one HIP graph per scene, {len(wl.slots) or 1} in flight",
                   "step": f"{main_block['scenes_per_step']} scenes per GPU (value = n_gpus*steps*scenes_per_step / median window)",
                   "arithmetic": "geometry / aggregation / epilogues fp32 (bit-exact vs the reference's CPU path); sparse "
                                 "convolutions f16x3: fp32 operands as two fp16 pieces under a per-tensor power-of-two scale, "
                                 "hh+hm+mh on the fp16 MFMA with fp32 accumulation (error <= 3*2^-22 per product); block "
                                 "`f32_conv` = the same run with exact-fp32 MFMA convolutions",
                   "host_cores": os.cpu_count()},
    }
    for k in ("scenes_per_step", "ms_per_scene", "windows_scenes_per_s", "window_s", "plan_violations", "host_cpu_cores_busy",
              "graph_nodes_per_scene"):
        result[k] = main_block[k]
    result["feature_layout"] = wl.layout
    if world > 1:
        # what the process group itself reports (a SCALE run can verify that N ranks took part over RCCL) + every rank's own
        # clock around each window (value uses the MAX over ranks)
        result["dist"] = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), rank=dist.get_rank(),
                              per_rank_window_s=main_block.get("per_rank_window_s"))
    for k in ("M_rows", "M_selected", "M_unique", "level_rows", "head_rows"):
        result["config"][k] = main_block[k]

    if rank == 0 and not args.no_profile:
        log("per-kernel profile (eager pass)")
        result["roofline"] = profile_block(wl, main_block, args.workload)
        for k in ("kernels", "stage_ms", "conv", "conv_layers", "algorithmic_GB_per_scene", "whole_path_hbm", "pmc_sq"):
            result[k] = main_block[k]
    Ms_full = main_block["M_selected"]
    main_layout = wl.layout
    del wl
    _released('block')

    if args.through_plugin:
        result["config"]["driver"] = ("registered RayMarching detector: model(return_loss=False, **data) per scene, "
                                      "{scene}_bbox_raw.npz written per scene (tmpfs), flush inside the timed window")
    if not args.no_secondary and not args.through_plugin:
        # ---- the same workload driven through the plugin API (test.py:205-214 -> RayMarching.forward_test)
        wlp, bp = measure(args.workload, device, rank, world, args, barrier, plugin=True)
        result["through_plugin"] = {k: bp[k] for k in ("value", "ms_per_step", "ms_per_scene", "scenes_per_step",
                                                        "windows_scenes_per_s", "plan_violations", "host_cpu_cores_busy")}
        result["through_plugin"]["ratio_to_value"] = bp["value"] / main_block["value"]
        result["through_plugin"]["note"] = ("the same workload through projects.mvsdetection RayMarching built from "
                                            "projects/configs/mvsdetection/ray_marching_scannet.py's model section (no sampler / "
                                            "graph-path keywords; 2D + Atlas networks not built: their outputs are the inputs; grid "
                                            "and stride of the workload): model(return_loss=False, features=[..], projection=[..], "
                                            "tsdf=.., offset=[..], scene=[..]) per scene, {scene}_bbox_raw.npz written per scene "
                                            "by the writer thread (tmpfs), model.flush() inside the window")
        import shutil
        shutil.rmtree(wlp.save_dir, ignore_errors=True)
        del wlp
        _released('block')
    if not args.no_secondary:
        # ---- the same workload with exact-fp32 MFMA convolutions (every rank takes part: the windows hold collectives)
        wl32, b32 = measure(args.workload, device, rank, world, args, barrier, precision="f32")
        result["f32_conv"] = {k: b32[k] for k in ("value", "ms_per_step", "ms_per_scene", "scenes_per_step", "windows_scenes_per_s",
                                                  "plan_violations")}
        result["f32_conv"]["note"] = "CONV_PRECISION='f32': v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain"
        # first-class: the same workload at the reference's own arithmetic (exact fp32 products in the sparse convolutions)
        result["value_f32_conv"] = b32["value"]
        del wl32
        _released('block')
        if main_layout == "channels_last":
            # ---- the same workload with the maps handed over NCHW (the reference's layout): layout pass inside the timed path
            wln, bn = measure(args.workload, device, rank, world, args, barrier, layout="nchw")
            result["nchw_input"] = {k: bn[k] for k in ("value", "ms_per_step", "ms_per_scene", "scenes_per_step", "windows_scenes_per_s",
                                                       "plan_violations")}
            result["nchw_input"]["note"] = ("feature maps resident as NCHW [V,C,H,W]: cnrma_nchw_to_nhwc_march_f32 (layout pass + march in "
                                            "one launch) runs per scene inside the timed region")
            del wln
            _released('block')
        if args.workload != "S":
            wls, bs = measure("S", device, rank, world, args, barrier)
            if rank == 0 and not args.no_profile:
                bs["roofline"] = profile_block(wls, bs, "S")
            bs["workload"] = "S: ScanNet config shape, V=40, C=32, 120x160 maps (stride 4), grid 192x192x80 (BASELINE configs[1])"
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                log("cpu baseline of the S block")
                bs["cpu_baseline"] = cpu_baseline("S", bs["M_selected"], 32)
            result["S"] = bs
            del wls
            _released('block')
    if not args.no_secondary and not args.through_plugin and args.workload != "St":
        # ---- ScanNet TEST shape (ray_marching_scannet.py:16,19: 50 views, grid 256 x 256 x 96): numbers only
        wlt, bt = measure("St", device, rank, world, args, barrier)
        result["St"] = {k: bt[k] for k in ("value", "ms_per_step", "ms_per_scene", "scenes_per_step", "windows_scenes_per_s",
                                            "plan_violations", "graph_nodes_per_scene", "M_rows", "M_selected", "M_unique", "level_rows", "head_rows")}
        result["St"]["workload"] = "St: ScanNet test shape, V=50, C=32, 120x160 maps (stride 4), grid 256x256x96 (ray_marching_scannet.py:16,19)"
        del wlt
        _released('block')
    if not args.no_secondary and not args.through_plugin:
        # ---- BASELINE configs[2]: the ARKitScenes detector (ray_marching_arkit.py: 17 classes, 8 regression outputs, yaw decode)
        # built from its shipped model section and driven through the plugin API at its own shape (40 views, 192 x 192 x 80)
        try:
            wla, ba = measure("S", device, rank, world, args, barrier, plugin=True, config="ray_marching_arkit.py")
            head = wla.model.detection_head
            result["A"] = {k: ba[k] for k in ("value", "ms_per_step", "ms_per_scene", "scenes_per_step", "windows_scenes_per_s",
                                              "plan_violations", "graph_nodes_per_scene", "M_rows", "M_selected", "M_unique",
                                              "level_rows", "head_rows") if k in ba}
            result["A"].update(n_classes=int(head.n_classes), n_reg_outs=int(head.n_reg_outs),
                               workload="A: projects/configs/mvsdetection/ray_marching_arkit.py model section through the plugin "
                                        "(model(return_loss=False, ...) per scene, result files on tmpfs): V=40, C=32, 120x160 maps "
                                        "(stride 4), grid 192x192x80, oriented boxes (BASELINE configs[2])")
            import shutil
            shutil.rmtree(wla.save_dir, ignore_errors=True)
            del wla, head
        except Exception as e:                                           # noqa: BLE001 -- a secondary block must not cost the line
            if world > 1:
                raise                                                    # the windows hold collectives: no rank may skip them alone
            log(f"A block failed: {e!r}")
            result["A"] = None
        _released('block')
    if rank == 0 and world == 1 and not args.no_secondary and not args.through_plugin:
        log("training step at the ScanNet shape (bf16 autocast)")
        try:
            result["train_S"] = train_block(device)
            log(f"train_S: {result['train_S']['ms_per_step']:.1f} ms per step")
        except Exception as e:                                           # noqa: BLE001 -- a secondary block must not cost the line
            log(f"train_S failed: {e!r}")
            result["train_S"] = None
        _released('block')
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        name = args.workload if args.workload in ("NS", "S", "St") else "tiny"
        log("cpu baseline (oracle on the host cores)")
        result["cpu_baseline"] = cpu_baseline(name, Ms_full, C)
        log("done")
    if rank == 0:
        emit(result, args.detail)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
