"""summarise a rocprofv3 --kernel-trace CSV: per-kernel GPU time of the LAST bench step, busy vs wall, gaps"""
import glob, sys
import numpy as np
import pandas as pd
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = pd.read_csv(f).sort_values('Start_Timestamp')
bp = d[d.Kernel_Name.str.contains('backproject_accum')]
starts = bp.Start_Timestamp.values
s0, s1 = starts[-2], starts[-1]
w = d[(d.Start_Timestamp >= s0) & (d.Start_Timestamp < s1)].copy()
w['dur'] = w.End_Timestamp - w.Start_Timestamp
print('step wall (ms)', (s1 - s0) / 1e6, 'kernel busy (ms)', w.dur.sum() / 1e6, 'n kernels', len(w))
w['short'] = w.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace('void ', '').str.slice(0, 70)
g = w.groupby('short').dur.agg(['sum', 'count']).sort_values('sum', ascending=False)
g['sum'] = g['sum'] / 1e6
print(g.head(int(sys.argv[2]) if len(sys.argv) > 2 else 45).to_string())
gaps = (w.Start_Timestamp.values[1:] - w.End_Timestamp.values[:-1])
print('gap total ms', gaps[gaps > 0].sum() / 1e6, 'gaps>20us', (gaps > 20000).sum(), 'largest(us)', np.sort(gaps)[-6:] / 1e3)
