"""Timeline of the LAST potrf in a rocprofv3 kernel trace of tools/bench_stages.py: every kernel with its start, duration
and the idle gap before it (one stream, so gaps are dispatch latency).  Dev tool.
usage: python tools/attic/trace_potrf_small.py <rocprof output dir> [max rows]"""
import sys, glob, re
import pandas as pd
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
df = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
cov = df.index[df.Kernel_Name.str.contains('gpp_cov_tile')]
start = cov[-1]
end = df.index[(df.index > start) & df.Kernel_Name.str.contains('gpp_trmv_lower')][0]
ev = df.iloc[start + 1:end]
lastleaf = ev.index[ev.Kernel_Name.str.contains('leaf')][-1]
pot = df.iloc[start + 1:lastleaf + 1].copy()
t0 = pot.Start_Timestamp.min()
pot['s'] = (pot.Start_Timestamp - t0) / 1e3
pot['d'] = (pot.End_Timestamp - pot.Start_Timestamp) / 1e3
pot['gap'] = (pot.Start_Timestamp - pot.End_Timestamp.shift(1)).fillna(0) / 1e3
def short(n):
    m = re.search(r'gpp_gemm_f64<([^>]*)>', n)
    if m: return 'gemm<' + m.group(1).replace(' ', '') + '>'
    return re.sub(r'\(.*', '', n).replace('(anonymous namespace)::', '')[:28]
pot['k'] = pot.Kernel_Name.map(short)
print('potrf span %.1f us, %d kernels; sum dur %.1f us, sum gaps %.1f us' % ((pot.End_Timestamp.max() - t0) / 1e3, len(pot), pot.d.sum(), pot.gap.sum()))
print(pot.groupby('k').agg(n=('d', 'size'), avg_us=('d', 'mean'), tot_us=('d', 'sum'), avg_gap=('gap', 'mean')).to_string())
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for _, r in pot.head(mx).iterrows():
    print('%8.1f  +%5.1f  %6.1f us  grid %6d  %s' % (r.s, r.gap, r.d, r.Grid_Size_X // max(r.Workgroup_Size_X, 1), r.k))
