#!/bin/bash
# A/B of settings.graphed_segments (GPP_GRAPHED_SEGMENTS=1) on one box: C3, C4, C2 through the plain API, twice each way, + the kernel
# census of a C3 evaluation with segments on.
for rep in 1 2; do
  for c in C3 C4 C2; do
    for g in 0 1; do
      echo "== $c GPP_GRAPHED_SEGMENTS=$g (rep $rep)"
      GPP_GRAPHED_SEGMENTS=$g timeout 600 python3 tools/run_configs.py $c 2>&1 | grep -v "amdgpu\|Warning\|warn" | tail -2
    done
  done
done
export TMPDIR=/tmp
GPP_GRAPHED_SEGMENTS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/census_seg -o C3 -- python3 tools/run_configs.py C3 > /dev/null 2>&1
f=$(find /tmp/census_seg -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lib = sum(int(r["Calls"]) for r in rows if "gpp_" in r["Name"])
oth = sum(int(r["Calls"]) for r in rows if "gpp_" not in r["Name"])
print(f"C3 with segments: {lib} library launches, {oth} others over the run ({len(rows)} distinct kernels)")
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:12]:
    print(f"  {int(r['Calls']):6d}  {r['Name'][:110]}")
PY
