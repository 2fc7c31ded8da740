"""Dump the kernel timeline of the last potrf in a rocprofv3 kernel trace of tools/bench_stages.py between two times (ms
from the start of that potrf): start, end, duration (us), queue, work-groups, short kernel name.  Dev tool.
usage: python tools/attic/trace_dump.py <rocprof dir> t_from t_to"""
import sys, glob, re
import pandas as pd
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
df = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
cov = df.index[df.Kernel_Name.str.contains('gpp_cov_tile')]
start = cov[-1]
ev = df.iloc[start + 1:].copy()
t0 = ev.Start_Timestamp.min()
ev['s'] = (ev.Start_Timestamp - t0) / 1e6
ev['e'] = (ev.End_Timestamp - t0) / 1e6
a, b = float(sys.argv[2]), float(sys.argv[3])
def short(n):
    n = re.sub(r'void \(anonymous namespace\)::', '', n)
    return re.sub(r'\(.*$', '', n)[:44]
prev = {}
for _, r in ev[(ev.e >= a) & (ev.s <= b)].iterrows():
    gap = r.s - prev.get(r.Queue_Id, r.s)
    print('%8.3f %8.3f  %7.1f us  q%-2d gap %6.1f us  wgs %6d  %s' % (r.s, r.e, (r.e - r.s) * 1e3, r.Queue_Id, gap * 1e3,
          r.Grid_Size_X // max(r.Workgroup_Size_X, 1), short(r.Kernel_Name)))
    prev[r.Queue_Id] = r.e
