"""Per-queue activity of the last potrf in a rocprofv3 kernel trace of tools/bench_stages.py, in 2 ms windows:
busy fraction of each HW queue (union of kernel intervals) and the kernel count.  Dev tool.
usage: python tools/attic/trace_tail.py <rocprof dir>"""
import sys, glob
import pandas as pd, numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
df = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
cov = df.index[df.Kernel_Name.str.contains('gpp_cov_tile')]
start = cov[-1]
end = df.index[(df.index > start) & df.Kernel_Name.str.contains('gpp_trmv_lower')][0]
ev = df.iloc[start + 1:end].copy()
t0 = ev.Start_Timestamp.min()
ev['s'] = (ev.Start_Timestamp - t0) / 1e6
ev['e'] = (ev.End_Timestamp - t0) / 1e6
span = ev.e.max()
print('potrf+trtri span %.2f ms; last leaf ends at %.2f ms' % (span, ev[ev.Kernel_Name.str.contains('leaf')].e.max()))
qs = sorted(ev.Queue_Id.unique())
W = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
print('window(ms)  ' + '  '.join('q%-3d busy  n ' % q for q in qs))
for w0 in np.arange(0, span, W):
    row = '%6.1f     ' % w0
    for q in qs:
        sub = ev[(ev.Queue_Id == q) & (ev.e > w0) & (ev.s < w0 + W)]
        iv = sorted(zip(sub.s.clip(lower=w0), sub.e.clip(upper=w0 + W)))
        tot = 0; ce = -1
        for s, e in iv:
            if s > ce: tot += e - s; ce = e
            elif e > ce: tot += e - ce; ce = e
        row += '  %5.2f %4d ' % (tot / W, len(sub))
    print(row)
