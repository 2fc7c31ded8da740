#!/bin/bash
# Copy the summaries of tools/collect_r06_profiles.sh (gpurun_out/r6prof/, scratch) into profiles/r06_* (tracked).
set -u
cd "$(dirname "$0")/.."
R=gpurun_out/r6prof; P=profiles
cp $R/r06_potrf_pmc.json $R/r06_trtri_pmc.json $R/r06_lauum_pmc.json $P/
cp $R/bench_kernel_stats.csv $P/r06_bench_kernel_stats.csv
grep "^{" $R/bench_line.json > $P/r06_bench_line.json
grep "^{" $R/bench_profiled.json > $P/r06_bench_line_under_rocprof.json
cp $R/pmc_fetch_write.txt $P/r06_pmc_fetch_write.txt
(echo; echo "== FETCH_SIZE pass, whole evaluation"; cat $R/pmcF_ALL_summary.txt; echo; echo "== WRITE_SIZE pass, whole evaluation"; cat $R/pmcW_ALL_summary.txt
 echo; echo "== FETCH_SIZE pass, build + potrf only"; cat $R/pmcF_P_summary.txt; echo; echo "== WRITE_SIZE pass, build + potrf only"; cat $R/pmcW_P_summary.txt) >> $P/r06_pmc_fetch_write.txt
(echo "SQ counters, whole evaluation at N=20000 (tools/bench_stages.py 20000 8 1 under rocprofv3 --pmc, two passes; tools/pmc_summary.py); kernel build $(cat $R/lib_version.txt)"
 echo "GPP_DAG_PHASED=1: the factorisation's ticket list as a sequence of launches of gpp_dag_f64 (counter collection serialises dispatches);"
 echo "its per-launch MfmaUtil is that of chain-only phases and says nothing about the concurrent form — instruction mix, L2 hit rate and bytes do."
 echo "== pass A: instruction mix, wave-cycle split, MFMA busy"; cat $R/pmcA_summary.txt; echo; echo "== pass B: L2 hit rate, LDS"; cat $R/pmcB_summary.txt
 echo; echo "== the phased form's own stage times (tools/bench_stages.py 20000 8 3 with GPP_DAG_PHASED=1)"; grep -v amdgpu $R/stages_20000_phased.txt) > $P/r06_sq_counters.txt
(echo "DAG executor (gpp_dag_f64 + gpp_dag.hip), round 6: per-task stamps of one factorisation (+ inverse) at the C3 / C4 / C2 sizes (TRACE=1 tools/dag_check.py)"
 grep -v amdgpu $R/dag_trace_10000.txt; echo; grep -v amdgpu $R/dag_trace_15000.txt; echo; grep -v amdgpu $R/dag_trace_20000.txt) > $P/r06_dag_traces.txt
(echo "Factorisation + inverse at N=10000 (the C3 size; STAGES_ONLY=build,potrf tools/bench_stages.py 10000 8 2 under rocprofv3 --kernel-trace; tools/trace_window.py, dispatches >= 30 us)."
 echo "q2 = panel stream (32 CUs): gates, panels, signals, filler launches of gpp_dag_f64; q3 = throughput stream (224 CUs): ONE launch of gpp_dag_f64 per factorisation."
 cat $R/timeline_n10000.txt) > $P/r06_timeline_n10000.txt
(echo "Sharded evaluation with ONE rank (the algorithm without communication), tools/run_sharded.py; round 6 (ticket lists, nb = 1024)"
 grep "^N=" $R/sharded_1rank_20000.txt; grep "^N=" $R/sharded_1rank_60000.txt
 echo; echo "== bench.py --mode sharded --n 20000 with GPP_SHARDED_FORCE_COLLECTIVES=1 on ONE rank over gloo (host-staged: the 'comm' block times the calls themselves)"
 grep "^{" $R/sharded_bench_line_20000.json) > $P/r06_restarts_and_sharded_1rank.txt
(echo "Sharded evaluation, ONE rank, bench.py --mode sharded --nb 1024 (tools/shard_list_bench.sh): the per-rank ticket lists of round 6 (GPP_SHARD_LIST=1,"
 echo "default: factor + forward sweep as one list = stage shard_factor; back-substitution as one list) against the launch-per-product path"
 echo "of rounds 2-4 (GPP_SHARD_LIST=0), same box, same build ($(cat $R/lib_version.txt))"
 cat $R/sharded_lists_1rank.txt) > $P/r06_sharded_lists_1rank.txt
grep -v "amdgpu\|Warning" $R/configs.txt > $P/r06_configs_C1_C5_single_gpu.txt
(grep -v amdgpu $R/hbm_probe.txt; echo; echo "per-stage times at N=20000 (tools/bench_stages.py 20000 8 5):"; grep -v amdgpu $R/stages_20000.txt
 echo; echo "C3 through the API (tools/c3_stages.py):"; grep -v "amdgpu\|Warning" $R/stages_c3.txt) > $P/r06_hbm_probe.txt
(echo "Virtual-rank replay (tools/replay_rank.py), round 6, final build $(cat $R/lib_version.txt): every rank of a P = 8 run replayed on ONE MI355X —"
 echo "the rank's real ticket lists (fillers on, one work-group per panel CU), the other ranks' block rows played in by a rate-limited copy kernel with a"
 echo "collective's footprint no earlier than their owners' MEASURED ready times (sweeps over the ranks until those stop moving); tail in pieces of 8192"
 echo "columns, the factor's mirror behind the list.  Per rank: build, factor + forward list, z / alpha, back-substitution list, gradient (ms)."
 echo "Method, and what it cannot contain: profiles/EXPERIMENTS.md (round 6).  First collection of the round (before the pieces): r06_virtual_rank_first.txt"
 echo; echo "==== C5 (N = 60 000, d = 16) ===="; grep -v "running tasks per\|amdgpu.ids" $R/replay_c5.txt
 echo; echo "==== C2 (N = 20 000, d = 8) ===="; grep -v "running tasks per\|amdgpu.ids" $R/replay_c2.txt
 for P_ in 4 2; do for c in c5 c2; do echo; echo "==== $c on $P_ virtual ranks ===="; grep -v "running tasks per\|amdgpu.ids\|^    sweep" $R/replay_${c}_p$P_.txt; done; done) > $P/r06_virtual_rank.txt
(echo "Virtual-rank replay with the owner's side of the PUSH transport (GPP_SHARD_PUSH=1; tools/replay_rank.py --push): a rank's own messages leave"
 echo "without a packing copy — the message's copies read the factor where it lies — everything else as in r06_virtual_rank.txt (same build, same box)."
 echo; echo "==== C5 (N = 60 000, d = 16), 8 virtual ranks ===="; grep -v "running tasks per\|amdgpu.ids\|^    sweep" $R/replay_c5_push.txt
 echo; echo "==== C2 (N = 20 000, d = 8), 8 virtual ranks ===="; grep -v "running tasks per\|amdgpu.ids\|^    sweep" $R/replay_c2_push.txt) > $P/r06_virtual_rank_push.txt
python3 - <<'PY' > $P/r06_virtual_scaling.json
import json
out = {"note": "virtual-rank replay (tools/replay_rank.py) on ONE MI355X: ms per evaluation = max over ranks of factor + forward, + max of the back-substitution, + the small stages; rate = the replayed messages' GB/s (0: every message there when asked for); one rank: profiles/r06_sharded_lists_1rank.txt", "configs": {}}
for c in ("c5", "c2"):
    rows = {}
    for P in (8, 4, 2):
        f = f"gpurun_out/r6prof/replay_{c}.json" if P == 8 else f"gpurun_out/r6prof/replay_{c}_p{P}.json"
        d = json.load(open(f))
        rows[str(P)] = {str(r["rate_gbs"]): {"total_ms": round(r["total_ms"], 2), "ff_ms": round(r["ff_ms"], 2), "back_ms": round(r["back_ms"], 2),
                                             "evals_per_s": round(r["evals_per_s"], 4), "sweeps": r["sweeps"]} for r in d["rates"]}
    out["configs"][c.upper()] = rows
print(json.dumps(out, indent=1))
PY
python3 - <<'PY' > $P/r06_kernel_census.txt
import csv
print("Kernel census of one evaluation through the plain API (tools/run_configs.py under rocprofv3 --kernel-trace --stats; calls per evaluation = total calls / evaluations).")
print("Round 6: what VERDICT r5 item 4 asked to take off the path; the graphed form (settings.graphed_segments) replays the same kernels and was measured slower — profiles/EXPERIMENTS.md.")
for c, ev in (("C3", 6), ("C1", 21)):
    rows = list(csv.DictReader(open(f"gpurun_out/r6prof/census_{c}_kernel_stats.csv")))
    tot = nong = 0.0
    print(f"== {c}: calls per evaluation, mean duration")
    for r in sorted(rows, key=lambda r: -int(r["Calls"])):
        per = int(r["Calls"]) / ev
        tot += per
        nong += per if "gpp_" not in r["Name"] else 0.0
        if per >= 0.9:
            print(f"  {per:6.1f}  {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")
    print(f"  total per evaluation {tot:.0f}, of which not the library's {nong:.0f}")
PY
ls $P | grep r06
