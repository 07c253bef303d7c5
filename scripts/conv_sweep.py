"""Per-layer sweep of the sparse-convolution launcher on one scene: every convolution of the network is captured with
its real operands (coordinate sets, features, weights, epilogue) and re-run standalone under forced tile shapes / split
counts / prefetch depths (sparse.conv_tuning); HIP-event time of REPS back-to-back launches each.

    python scripts/conv_sweep.py S            # table: layer class x configuration, best per class
    python scripts/conv_sweep.py NS 64x128:0:1 128x128:0:2 ...      # explicit configurations shape:splits:pf (0 / -1 = auto)
    python scripts/conv_sweep.py S auto:0:0:0x20 auto:0:0:1 auto:0:0:6 ...   # 4th field: ablation mask of the diagnostic kernel
                         (sparse.conv_tuning; 0x20 = diagnostic kernel with nothing switched off; results are not compared)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, synth
from cnrma_amd import sparse as S

REPS = 20
dev = torch.device("cuda:0")
if os.environ.get("GO") is not None:                 # GO=0 / GO=1: the gather-once kernel off / on for this run
    S.GO_CONV = os.environ["GO"] != "0"
wl = sys.argv[1] if len(sys.argv) > 1 else "S"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)

calls = []
orig_conv = S.conv


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    calls.append(dict(x=x, weight=weight, ks=kernel_size, stride=stride, scale=scale, shift=shift, residual=residual, act=act,
                      n_out=y.cs.n, ref=y.F))
    return y


S.conv = rec_conv
sys.modules["cnrma_amd.nn"].S.conv = rec_conv
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
S.conv = orig_conv
sys.modules["cnrma_amd.nn"].S.conv = orig_conv
del feat
torch.cuda.empty_cache()


def run(c):
    return orig_conv(c["x"], c["weight"], c["ks"], c["stride"], c["scale"], c["shift"], c["residual"], c["act"]).F


def timed(c):
    run(c)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        out = run(c)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3, out          # us


if len(sys.argv) > 2:
    cfgs = []
    for spec in sys.argv[2:]:
        sh, sp, pf, ab = (spec.split(":") + ["0", "0", "0"])[:4]
        ws = -1
        if ab.startswith("ws"):                      # 4th field "wsN": the warp-specialised kernel with an N-slot ring
            ws, ab = int(ab[2:]), "0"
        if ab == "xcd":                              # 4th field "xcd": XCD-aware tile order of the stage kernel
            ws, ab = -2, "0"
        cfgs.append((None if sh in ("auto", "") else sh, int(sp) or -1, int(pf) or -1, int(ab, 0), ws))
else:
    cfgs = [(None, -1, -1, 0, -1)]
    for sh in ("64x64", "128x64", "64x128", "128x128"):
        for pf in (1, 2):
            cfgs.append((sh, -1, pf, 0, -1))
    for sp in (1, 3, 9, 14, 27):
        cfgs.append((None, sp, -1, 0, -1))

classes = {}
for c in calls:
    K = c["ks"] ** 3
    key = (c["n_out"], c["x"].F.shape[1], c["ref"].shape[1], K, c["stride"], c["residual"] is not None)
    classes.setdefault(key, c)
print(f"{wl}: {len(calls)} convolutions, {len(classes)} classes; us per launch ({REPS} back to back)")
hdr = ["rows", "Cin", "Cout", "K", "s", "res", "plan"] + [f"{a or 'auto'}:{b}:{c_}" + (f":{d:#x}" if d else "") + (f":ws{e}" if e >= 0 else (":xcd" if e == -2 else ""))
                                                         for a, b, c_, d, e in cfgs]
print(" | ".join(hdr))
tot = {i: 0.0 for i in range(len(cfgs))}
best_tot = 0.0
only = os.environ.get("ONLY")                         # ONLY=Cin,Cout,K,minrows: restrict the sweep to matching layer classes
for key, c in sorted(classes.items(), key=lambda kv: -kv[0][0]):
    n_out, Cin, Cout, K, st, res = key
    if only:
        oc, oo, ok_, omin = (int(v) for v in only.split(","))
        if (Cin, Cout, K) != (oc, oo, ok_) or n_out < omin:
            continue
    mult = sum(1 for d in calls if (d["n_out"], d["x"].F.shape[1], d["ref"].shape[1], d["ks"] ** 3, d["stride"], d["residual"] is not None) == key)
    S.conv_tuning()
    plan = S.conv_plan(n_out, Cin, Cout, K) if Cin % 32 == 0 else None
    ref = run(c).clone()
    row, ts = [], []
    for i, (sh, sp, pf, ab, ws) in enumerate(cfgs):
        if Cin % 32 != 0 and (sh == "64x128" or pf == 2):
            row.append("   -  ")
            ts.append(float("inf"))
            continue
        if Cout <= 32 and sh is not None:
            sh = None
        S.conv_tuning(sh, sp, pf, ab, max(ws, -1), 1 if ws == -2 else (0 if (sh, sp, pf, ab, ws) != (None, -1, -1, 0, -1) else -1))
        try:
            t, out = timed(c)
            err = 0.0 if ab else float((out - ref).abs().max() / (ref.abs().max() + 1e-30))
            row.append(f"{t:6.1f}" + ("" if err < 1e-5 else f"!{err:.0e}"))
            ts.append(t)
            tot[i] += t * mult
        except Exception as e:      # noqa: BLE001
            row.append(" err  ")
            ts.append(float("inf"))
            print("   ", e)
    S.conv_tuning()
    b = min(range(len(cfgs)), key=lambda i: ts[i])
    best_tot += ts[b] * mult
    pl = f"{plan['shape']}/{plan['splits']}/{plan['prefetch']}" if plan else "f32"
    print(f"{n_out:7d} {Cin:4d} {Cout:4d} {K:2d} {st} {int(res)} x{mult:2d} {pl:14s} | " + " ".join(row) +
          f" | best {hdr[7 + b]} {ts[b]:.1f}", flush=True)
print("sum over all launches (us):", " ".join(f"{tot[i]:8.0f}" for i in range(len(cfgs))), "| per-class best:", round(best_tot))
