"""gather-once kernel forms per layer: first form vs second form (work order over the XCDs, weight offsets in flight,
splits); HIP events over REPS back-to-back launches (tile unions cached).  usage: go_forms.py [S|NS]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, synth
from cnrma_amd import sparse as S

REPS = 20
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "S"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
calls, seen = [], set()
orig_conv = S.conv


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    key = (x.cs.n, x.F.shape[1], y.F.shape[1], residual is not None)
    if kernel_size == 3 and stride == 1 and x.F.shape[1] % 32 == 0 and y.F.shape[1] >= 64 and x.cs.compact and key not in seen:
        seen.add(key)
        calls.append(dict(x=x, weight=weight, scale=scale, shift=shift, residual=residual, act=act))
    return y


S.conv = rec_conv
sys.modules["cnrma_amd.nn"].S.conv = rec_conv
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
S.conv = orig_conv
sys.modules["cnrma_amd.nn"].S.conv = orig_conv
del feat
prev_go = S.GO_CONV
S.GO_CONV = True


def timed(c):
    def run():
        return orig_conv(c["x"], c["weight"], 3, 1, c["scale"], c["shift"], c["residual"], c["act"])
    run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3


VARIANTS = [("f1", dict(go=0)), ("f2x0", dict(go=1, xcd=0)), ("f2x1", dict(go=1, xcd=1)), ("f2x2", dict(go=1, xcd=2)),
            ("f2", dict(go=1, nb=10)), ("f2a", dict(go=1, nb=12)), ("f2b4", dict(go=1, nb=4)), ("f3", dict(go=2)), ("f3na", dict(go=2, pf=7)), ("f3occ", dict(go=2, pf=8))]
if os.environ.get("AB"):            # AB=f2,f3: only these variants, interleaved three times (order effects: a slow variant leaves the
    names = os.environ["AB"].split(",")                       # chip in another clock / cache state for the one measured after it)
    VARIANTS = [v for v in VARIANTS if v[0] in names] * 3
tot = {k: 0.0 for k, _ in VARIANTS}
best_tot = 0.0
for c in calls:
    x = c["x"]
    ns = x.F.shape[1] // 32
    res, best = [], (1e9, "")
    for name, kw in VARIANTS:
        S.conv_tuning(**kw)
        t = timed(c)
        tot[name] += t
        res.append(f"{name} {t:6.1f}")
        best = min(best, (t, name))
    if os.environ.get("AB"):
        S.conv_tuning()
        print(f"rows={x.cs.n:7d} Cin={x.F.shape[1]:4d} Cout={c['weight'].shape[-1]:4d} res={int(c['residual'] is not None)} | " + "  ".join(res), flush=True)
        continue
    # splits of the two best-looking forms on the short layers
    extra = []
    if x.cs.n < S.GO_WS_ROWS and ns > 1:
        for name, kw in (("f2", dict(go=1)), ("f3", dict(go=2))):
            for sp in (2, 4, 8, 16):
                if sp <= ns:
                    S.conv_tuning(splits=sp, **kw)
                    t = timed(c)
                    extra.append(f"{name}/s{sp} {t:6.1f}")
                    best = min(best, (t, f"{name}/s{sp}"))
    best_tot += best[0]
    S.conv_tuning()
    print(f"rows={x.cs.n:7d} Cin={x.F.shape[1]:4d} Cout={c['weight'].shape[-1]:4d} res={int(c['residual'] is not None)} | "
          + "  ".join(res) + (" | " + "  ".join(extra) if extra else "") + f" | best {best[1]} {best[0]:.1f}", flush=True)
print("sum over the layer classes (us): " + "  ".join(f"{k} {v:.0f}" for k, v in tot.items()) + f"  per-class best {best_tot:.0f}")
S.GO_CONV = prev_go
