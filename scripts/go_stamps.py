"""phase timeline of the gather-once kernel's blocks (diagnostic build with s_memtime stamps, conv_tuning(ablate=64)):
per layer class the median share of a block's life spent in [start -> metadata + first barrier], [gather], [offsets], ...
usage: go_stamps.py [S|NS] [go form 1|2] [nb]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cnrma_amd import pipeline, synth
from cnrma_amd import sparse as S

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "S"
form = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
PF = int(sys.argv[4]) if len(sys.argv) > 4 else -1            # third form: 7 = never gather ahead
V, C, H, W, dims, stride = synth.SHAPES[wl]
sc = synth.make_scene(wl, seed=0, boxes=3, device=dev)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
calls, seen = [], set()
orig_conv = S.conv


def rec_conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    y = orig_conv(x, weight, kernel_size, stride, scale, shift, residual, act, precision)
    key = (x.cs.n, x.F.shape[1], y.F.shape[1])
    if kernel_size == 3 and stride == 1 and x.F.shape[1] % 32 == 0 and y.F.shape[1] >= 64 and x.cs.compact and key not in seen \
            and residual is None:
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
S.GO_CONV = True
NBLK = 1 << 16
buf = torch.zeros(NBLK * 16, dtype=torch.int64, device=dev)
S.GO_STAMPS = buf
for c in calls:
    x = c["x"]

    def run():
        return orig_conv(x, c["weight"], 3, 1, c["scale"], c["shift"], c["residual"], c["act"])
    print(f"layer rows={x.cs.n} Cin={x.F.shape[1]} Cout={c['weight'].shape[-1]}", flush=True)
    S.conv_tuning(go=form, nb=nb)
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    t_plain = a.elapsed_time(b) / 10 * 1e3
    S.conv_tuning(go=form, nb=nb, ablate=64, pf=PF)
    print("  diagnostic build", flush=True)
    run()
    torch.cuda.synchronize()
    buf.zero_()
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    t_diag = a.elapsed_time(b) * 1e3
    S.conv_tuning()
    st = buf.view(NBLK, 16).cpu().numpy().astype(np.int64)
    blk = np.arange(NBLK)
    live = (st[:, 0] != 0) & (st[:, 15] != 0)
    st, blk = st[live], blk[live]
    if not len(st):
        print("no stamps", x.cs.n)
        continue
    if form == 2:                                             # third form: stamps of every block's FIRST item; slot 14 = offsets of wave 0
        n_off = st[:, 14].astype(np.float64)
        life = st[:, 15] - st[:, 0]
        n_ph = int((st[0, 1:14] != 0).sum())
        names = ["start->barrier"] + [("offsets", "drain+finish+park", "prefetch+barrier")[(i - 1) % 3] for i in range(1, n_ph)] + ["merge+epilogue"]
        prev, segs = st[:, 0], []
        for i in range(1, n_ph + 1):
            segs.append(st[:, i] - prev)
            prev = st[:, i]
        segs.append(st[:, 15] - prev)
        loops = sum(s_ for n_, s_ in zip(names, segs) if n_ == "offsets")
        print(f"rows={x.cs.n:7d} Cin={x.F.shape[1]:4d} Cout={c['weight'].shape[-1]:4d} blocks={len(st):5d} plain {t_plain:7.1f} us diag {t_diag:7.1f} us; first item: "
              f"median life {np.median(life):8.0f} cycles, offsets of wave 0 {np.median(n_off):.0f}, cycles per offset step {np.median(loops / np.maximum(n_off, 1)):7.0f}")
        print("    " + "  ".join(f"{n} {np.median(s_):7.0f}" for n, s_ in zip(names, segs)) + "   (median cycles)")
        continue
    # s_memtime is not one chip-wide counter: spans are taken inside one XCD (XCC_ID read by the block), the tick from the launch's length
    xcc = st[:, 14] & 0xF
    hw = st[:, 14] >> 32
    st[:, 14] = 0
    agree = float((xcc == blk % 8).mean())
    blk_mod = blk % 8
    blk = xcc                                                  # group by the XCD the block really ran on
    spans = [st[blk % 8 == q][:, 15].max() - st[blk % 8 == q][:, 0].min() for q in range(8) if (blk % 8 == q).any()]
    span = float(np.median(spans))
    tick_us = t_diag / span
    end = st[:, 15]
    life = end - st[:, 0]
    n_ph = int((st[0, 1:14] != 0).sum())
    segs = []
    prev = st[:, 0]
    for i in range(1, n_ph + 1):
        segs.append(st[:, i] - prev)
        prev = st[:, i]
    segs.append(end - prev)
    names = ["meta+sync"] + [("gather", "offsets", "sync")[(i - 1) % 3] for i in range(1, n_ph)] + ["merge+epilogue"]
    starts = np.concatenate([st[blk % 8 == q][:, 0] - st[blk % 8 == q][:, 0].min() for q in range(8) if (blk % 8 == q).any()])
    print(f"rows={x.cs.n:7d} Cin={x.F.shape[1]:4d} Cout={c['weight'].shape[-1]:4d} blocks={len(st):6d} plain {t_plain:7.1f} us, "
          f"diagnostic {t_diag:7.1f} us (span {span:.0f} ticks, {tick_us * 1e3:.2f} ns/tick); block life median {np.median(life) * tick_us:6.1f} us "
          f"(p10 {np.percentile(life, 10) * tick_us:.1f}, p90 {np.percentile(life, 90) * tick_us:.1f}); block starts p50 {np.median(starts) * tick_us:.1f} "
          f"p90 {np.percentile(starts, 90) * tick_us:.1f} us")
    print(f"    XCC_ID == blockIdx % 8 for {agree * 100:.1f} % of the blocks; XCC histogram {np.bincount(xcc, minlength=8).tolist()}; "
          f"spans per XCD (ticks) {[int(v) for v in spans]}; raw median life {np.median(life):.0f} ticks; "
          f"CUs seen {len(np.unique(hw & 0xFF0F))}")
    print("    " + "  ".join(f"{n} {np.median(s_) * tick_us:5.2f}" for n, s_ in zip(names, segs)) + "   (median us per segment, wave 0 of a block)")
S.GO_STAMPS = None
