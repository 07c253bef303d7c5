"""join scripts/mfma_count.py's layer list with the per-dispatch counters of its rocprofv3 pass:
    python scripts/mfma_join.py <rocprof output dir> <layers.json>
Per gather-once convolution: executed MFMA instructions (SQ_INSTS_VALU_MFMA_MOPS_<type> x 512 flops / flops per instruction) against
  algorithmic = products x pairs x Cin x Cout_padded32 / 16384 MACs        (products: 3 for f16x3, 1 for f32 / bf16)
  tile-mask   = the same with 64 rows per (tile, active offset) instead of the pairs
and the matrix pipe's busy cycles against executed x cycles per instruction."""
import csv
import glob
import json
import os
import sys

src, layers_path = sys.argv[1], sys.argv[2]
meta = json.load(open(layers_path))
f32 = meta["precision"] == "f32"
prod = 1 if f32 else 3
macs_per_inst, cyc_per_inst = (32 * 32 * 2, 64) if f32 else (32 * 32 * 16, 32)     # v_mfma_f32_32x32x2_f32: 64 cycles, _32x32x16_f16: 32
disp = {}
for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        d = disp.setdefault(int(r["Dispatch_Id"]), dict(name=r["Kernel_Name"], t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"]), c={}))
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
fam = "sparse_conv_gof_kernel" if f32 else "sparse_conv_go2_kernel"
go = [disp[k] for k in sorted(disp) if fam in disp[k]["name"]]
want = [L for L in meta["layers"] if L["entry"] and "_go_" in L["entry"]]
assert len(go) >= len(want), (len(go), len(want))
go = go[-len(want):]                                          # the recorded scene is the last one
mops = "SQ_INSTS_VALU_MFMA_MOPS_F32" if f32 else "SQ_INSTS_VALU_MFMA_MOPS_F16"
tot = dict(alg=0.0, tile=0.0, exe=0.0, busy=0.0, us=0.0, cu=0.0)
print(f"{meta['workload']} {meta['precision']}: {len(want)} gather-once convolutions ({fam})")
print(" rows    Cin->Cout   us   executed   /algorithmic  /tile-mask   MFMA-busy  busy/(exe*cyc)  busy/(4*CU-busy)  clock(GUI/us)")
for L, d in zip(want, go):
    c = d["c"]
    cp = -(-L["Cout"] // 32) * 32
    alg = prod * L["pairs"] * L["Cin"] * cp / macs_per_inst
    tile = prod * L["tile_offsets"] * 64 * L["Cin"] * cp / macs_per_inst
    exe = c.get(mops, 0.0) * 512 / (2 * macs_per_inst)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    cu = c.get("SQ_BUSY_CU_CYCLES", 0.0)
    us = (d["t1"] - d["t0"]) / 1e3
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    insts = c.get("SQ_INSTS_MFMA")
    for k, v in (("alg", alg), ("tile", tile), ("exe", exe), ("busy", busy), ("us", us), ("cu", cu)):
        tot[k] += v
    print(f"{L['n_out']:7d} {L['Cin']:4d}->{L['Cout']:4d} {us:7.1f} {exe:11.0f} {exe / alg:8.3f} {exe / tile:12.3f} {busy:12.0f} "
          f"{busy / max(exe * cyc_per_inst, 1):10.3f} {busy / max(4 * cu, 1):14.3f} {gui / max(us, 1e-9) / 1e3:12.2f} GHz"
          + (f"  SQ_INSTS_MFMA {insts:.0f}" if insts is not None else ""))
print(f"sum: {tot['us']:.1f} us; executed / algorithmic {tot['exe'] / tot['alg']:.3f}; executed / tile-mask {tot['exe'] / tot['tile']:.3f}; "
      f"busy / (executed x {cyc_per_inst}) {tot['busy'] / (tot['exe'] * cyc_per_inst):.3f}; busy / (4 x CU-busy) {tot['busy'] / (4 * tot['cu']):.3f}")
