"""rocprofv3 counter_collection / kernel_trace CSVs -> per-kernel sums (small CSV for profiles/)
usage: pmc_sum.py <dir with *_counter_collection.csv or *_kernel_trace.csv> <out.csv>"""
import csv, glob, os, sys, collections
src, out = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:160], r["Counter_Name"])
        d = rows.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "Per_Dispatch"])
    for (k, c), (n, s) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, f"{s:.6g}", f"{s / n:.6g}"])
print("wrote", out, len(rows), "rows")
