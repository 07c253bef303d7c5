"""shader clock / socket power while one kernel class runs back to back (rocm-smi sampled from a side thread):
    python scripts/clock_probe.py        -> dense unprojection (NS), the 277 k-row 64 -> 64 and 200 k-row 64 -> 128 convolutions in
                                            f16x3 and exact fp32, the MFMA-only peak loop is scripts/mfma_peak.hip
Explains the 1.6-1.9 GHz the counter pass derives from GRBM_GUI_ACTIVE under the f16x3 convolutions (DESIGN.md "Round 6" 1)."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, rma, synth
from cnrma_amd import sparse as S

dev = torch.device("cuda:0")
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            samples.append((time.perf_counter(), card))
        except Exception as e:                                   # noqa: BLE001
            samples.append((time.perf_counter(), {"error": repr(e)}))
        time.sleep(0.05)


def window(name, fn, seconds=2.5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    t1 = time.perf_counter()
    mine = [c for t, c in samples if t0 + 0.5 <= t <= t1]
    def vals(key_part):
        out = []
        for c in mine:
            for k, v in c.items():
                if key_part in k.lower():
                    try:
                        out.append(float(str(v).strip("()MhzW ").split("M")[0]))
                    except ValueError:
                        pass
        return out
    sclk, pw = vals("sclk clock speed"), vals("power")
    print(f"{name:46s} {1e3 * (t1 - t0) / n:8.3f} ms per call   sclk MHz {min(sclk, default=0):.0f}-{max(sclk, default=0):.0f}   "
          f"power W {min(pw, default=0):.0f}-{max(pw, default=0):.0f}   ({len(mine)} samples)", flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(0.3)
print("keys of one rocm-smi sample:", list(samples[-1][1].keys())[:12] if samples else None)
wl = "NS"
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev, channels_last=True)
feat_nchw, proj3, tsdf = sc["features"][:, 0], sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
calls = []
orig_conv = S.conv


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    if kernel_size == 3 and stride == 1 and x.cs.n > 150000:
        calls.append((x, weight, scale, shift, residual, act))
    return y


S.conv = rec_conv
sys.modules["cnrma_amd.nn"].S.conv = rec_conv
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat_nchw, proj3, tsdf, dense=False)
S.conv = orig_conv
sys.modules["cnrma_amd.nn"].S.conv = orig_conv
feat = rma.to_nhwc(feat_nchw)
proj = rma.scale_projection(proj3, stride).to(dev)
window("idle (no kernel)", lambda: None, 1.0)
with torch.no_grad():
    window("dense unprojection (NS)", lambda: rma.backproject_accum(feat, None, dims, 0.04, (0, 0, 0), stride, proj_scaled=proj), 3.0)
    for x, w, sc_, sh, res, act in calls[:2]:
        for prec in ("f16x3", "f32"):
            window(f"conv {x.cs.n} rows {x.F.shape[1]}->{w.shape[-1]} {prec}", lambda: orig_conv(x, w, 3, 1, sc_, sh, res, act, prec))
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        # where the power goes: the FIRST form of the gather-once kernel (experiments library) with phases switched off
        # (results meaningless): 1 MFMAs + fragment reads, 2 union-row loads, 4 weight loads, 8 LDS stores, 16 epilogue stores
        x, w, sc_, sh, res, act = calls[0]
        for go, mask in ((1, 0), (0, 0), (0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (0, 1 | 4), (0, 2 | 8)):
            S.conv_tuning(go=go, ablate=mask)
            window(f"conv {x.cs.n} rows f16x3 form {'second' if go else 'first'} ablate={mask}",
                   lambda: orig_conv(x, w, 3, 1, sc_, sh, None, act, "f16x3"))
        S.conv_tuning()
stop[0] = True
