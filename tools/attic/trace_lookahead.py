"""Summarise a rocprofv3 kernel trace of tools/bench_stages.py (last evaluation): busy time per HW queue during potrf
and how much of the potrf span each queue covers."""
import sys, glob
import pandas as pd, numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
df = pd.read_csv(f)
df['dur'] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
df = df.sort_values('Start_Timestamp').reset_index(drop=True)
cov = df.index[df.Kernel_Name.str.contains('gpp_cov_tile')]
start = cov[-1]
_after = df.index[(df.index > start) & df.Kernel_Name.str.contains('gpp_trmv_lower')]
end = _after[0] if len(_after) else len(df)  # (a factorisation-only trace, STAGES_ONLY=build,potrf, has no later stage)
ev = df.iloc[start + 1:end]
lastleaf = ev.index[ev.Kernel_Name.str.contains('leaf')][-1]
pot = df.iloc[start + 1:lastleaf + 1].copy()
t0 = pot.Start_Timestamp.min()
pot['s'] = (pot.Start_Timestamp - t0) / 1e6
pot['e'] = (pot.End_Timestamp - t0) / 1e6
span = pot.e.max()
print('potrf span %.2f ms' % span)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for q, sub in pot.groupby('Queue_Id'):
    print('queue %d: %4d kernels, sum of durations %.2f ms, covered time %.2f ms' % (q, len(sub), sub.dur.sum() / 1e3, union(list(zip(sub.s, sub.e)))))
qs = list(pot.groupby('Queue_Id').groups)
big_q = pot.groupby('Queue_Id').apply(lambda d: d.Grid_Size_X.max()).idxmax()
upd = pot[pot.Queue_Id == big_q]
pan = pot[pot.Queue_Id != big_q]
u_cov = union(list(zip(upd.s, upd.e)))
print('update queue idle inside potrf: %.2f ms' % (span - u_cov))
# gaps on the update queue > 0.1 ms
iv = sorted(zip(upd.s, upd.e)); gaps = []
for (s0, e0), (s1, e1) in zip(iv[:-1], iv[1:]):
    if s1 - e0 > 0.05: gaps.append((round(e0, 2), round(s1 - e0, 2)))
print('update-queue gaps (start, length ms):', gaps)
# per-kernel list of the update queue in the last part of the factorisation
if len(sys.argv) > 2:
    t_from = float(sys.argv[2])
    sub = upd[upd.s >= t_from]
    prev_e = None
    for _, r in sub.iterrows():
        gap = 0.0 if prev_e is None else r.s - prev_e
        print('%7.2f  +%5.2f  %6.3f ms  wgs %6d  %s' % (r.s, gap, r.e - r.s, r.Grid_Size_X // max(r.Workgroup_Size_X, 1), r.Kernel_Name[:60]))
        prev_e = r.e
