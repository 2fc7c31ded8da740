"""Kernel timeline of a rocprofv3 --kernel-trace run between two markers: the LAST occurrence of a kernel whose name contains
<start-substr> up to the end of the trace (or the first later kernel containing <end-substr>).  Prints one line per dispatch
(start ms, duration us, queue, gap on that queue, work-groups, short name) — at most <max-lines> — and per-name totals.
usage: python tools/trace_window.py <rocprof dir> <start-substr> [end-substr|-] [max-lines] [min-us] [t-from-ms] [t-to-ms]"""
import glob, re, sys
import pandas as pd

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
df = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
start_sub = sys.argv[2]
end_sub = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != '-' else None
max_lines = int(sys.argv[4]) if len(sys.argv) > 4 else 400
min_us = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
t_from = float(sys.argv[6]) if len(sys.argv) > 6 else -1.0
t_to = float(sys.argv[7]) if len(sys.argv) > 7 else 1e30
idx = df.index[df.Kernel_Name.str.contains(start_sub, regex=False)]
i0 = idx[-1]
ev = df.iloc[i0:].copy()
if end_sub:
    e = ev.index[ev.Kernel_Name.str.contains(end_sub, regex=False) & (ev.index > i0)]
    if len(e):
        ev = ev.loc[:e[0]]
t0 = ev.Start_Timestamp.min()
ev['s'] = (ev.Start_Timestamp - t0) / 1e6
ev['e'] = (ev.End_Timestamp - t0) / 1e6


def short(n):
    n = re.sub(r'void \(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'(gpp_gemm_f64<[^>]*>)', n)
    if m:
        return m.group(1).replace(' ', '')
    return re.sub(r'\(.*$', '', n)[:60]


ev['name'] = ev.Kernel_Name.map(short)
print('window: %.3f ms, %d dispatches, queues %s' % (ev.e.max(), len(ev), sorted(ev.Queue_Id.unique())))
prev = {}
n = 0
for _, r in ev.iterrows():
    gap = r.s - prev.get(r.Queue_Id, r.s)
    if (r.e - r.s) * 1e3 >= min_us and r.e >= t_from and r.s <= t_to:
        n += 1
    if (r.e - r.s) * 1e3 >= min_us and r.e >= t_from and r.s <= t_to and n <= max_lines:
        print('%9.3f  %8.1f us  q%-2d gap %7.1f  wgs %6d  %s' % (r.s, (r.e - r.s) * 1e3, r.Queue_Id, gap * 1e3,
              (r.Grid_Size_X // max(r.Workgroup_Size_X, 1)) * max(r.Grid_Size_Y, 1) * max(r.Grid_Size_Z, 1), r['name']))
    prev[r.Queue_Id] = r.e
ev['dur'] = ev.e - ev.s
g = ev.groupby('name').dur.agg(['count', 'sum', 'mean']).sort_values('sum', ascending=False)
print('\nper kernel (ms):')
for name, r in g.iterrows():
    print('%6d  %9.3f  %8.4f  %s' % (r['count'], r['sum'], r['mean'], name))
for q, gq in ev.groupby('Queue_Id'):
    print('queue %d: busy %.3f ms (sum of durations), span %.3f..%.3f' % (q, gq.dur.sum(), gq.s.min(), gq.e.max()))
