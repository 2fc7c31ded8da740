#!/bin/bash
# Copy the summaries of tools/attic/collect_r05_profiles.sh (gpurun_out/r5prof/, scratch) into profiles/r05_* (tracked).
set -u
cd "$(dirname "$0")/.."
R=gpurun_out/r5prof; P=profiles
cp $R/r05_potrf_pmc.json $R/r05_trtri_pmc.json $R/r05_lauum_pmc.json $P/
cp $R/bench_kernel_stats.csv $P/r05_bench_kernel_stats.csv
grep "^{" $R/bench_line.json > $P/r05_bench_line.json
grep "^{" $R/bench_profiled.json > $P/r05_bench_line_under_rocprof.json
cp $R/pmc_fetch_write.txt $P/r05_pmc_fetch_write.txt
(echo; echo "== FETCH_SIZE pass, whole evaluation"; cat $R/pmcF_ALL_summary.txt; echo; echo "== WRITE_SIZE pass, whole evaluation"; cat $R/pmcW_ALL_summary.txt
 echo; echo "== FETCH_SIZE pass, build + potrf only"; cat $R/pmcF_P_summary.txt; echo; echo "== WRITE_SIZE pass, build + potrf only"; cat $R/pmcW_P_summary.txt) >> $P/r05_pmc_fetch_write.txt
(echo "SQ counters, whole evaluation at N=20000 (tools/bench_stages.py 20000 8 1 under rocprofv3 --pmc, two passes; tools/pmc_summary.py); kernel build $(cat $R/lib_version.txt)"
 echo "GPP_DAG_PHASED=1: the factorisation's ticket list as a sequence of launches of gpp_dag_f64 (counter collection serialises dispatches);"
 echo "its per-launch MfmaUtil is that of chain-only phases and says nothing about the concurrent form — instruction mix, L2 hit rate and bytes do."
 echo "== pass A: instruction mix, wave-cycle split, MFMA busy"; cat $R/pmcA_summary.txt; echo; echo "== pass B: L2 hit rate, LDS"; cat $R/pmcB_summary.txt
 echo; echo "== the phased form's own stage times (tools/bench_stages.py 20000 8 3 with GPP_DAG_PHASED=1)"; grep -v amdgpu $R/stages_20000_phased.txt) > $P/r05_sq_counters.txt
(echo "DAG executor (gpp_dag_f64 + gpp_dag.hip), round 5: per-task stamps of one factorisation (+ inverse) at the C3 / C4 / C2 sizes (TRACE=1 tools/dag_check.py)"
 grep -v amdgpu $R/dag_trace_10000.txt; echo; grep -v amdgpu $R/dag_trace_15000.txt; echo; grep -v amdgpu $R/dag_trace_20000.txt) > $P/r05_dag_traces.txt
(echo "Factorisation + inverse at N=10000 (the C3 size; STAGES_ONLY=build,potrf tools/bench_stages.py 10000 8 2 under rocprofv3 --kernel-trace; tools/trace_window.py, dispatches >= 30 us)."
 echo "q2 = panel stream (32 CUs): gates, panels, signals, filler launches of gpp_dag_f64; q3 = throughput stream (224 CUs): ONE launch of gpp_dag_f64 per factorisation."
 cat $R/timeline_n10000.txt) > $P/r05_timeline_n10000.txt
(echo "Sharded evaluation with ONE rank (the algorithm without communication), tools/run_sharded.py; round 5 (ticket lists, nb = 1024)"
 grep "^N=" $R/sharded_1rank_20000.txt; grep "^N=" $R/sharded_1rank_60000.txt
 echo; echo "== bench.py --mode sharded --n 20000 with GPP_SHARDED_FORCE_COLLECTIVES=1 on ONE rank over gloo (host-staged: the 'comm' block times the calls themselves)"
 grep "^{" $R/sharded_bench_line_20000.json) > $P/r05_restarts_and_sharded_1rank.txt
(echo "Sharded evaluation, ONE rank, bench.py --mode sharded --nb 1024 (tools/shard_list_bench.sh): the per-rank ticket lists of round 5 (GPP_SHARD_LIST=1,"
 echo "default: factor + forward sweep as one list = stage shard_factor; back-substitution as one list) against the launch-per-product path"
 echo "of rounds 2-4 (GPP_SHARD_LIST=0), same box, same build ($(cat $R/lib_version.txt))"
 cat $R/sharded_lists_1rank.txt) > $P/r05_sharded_lists_1rank.txt
grep -v "amdgpu\|Warning" $R/configs.txt > $P/r05_configs_C1_C5_single_gpu.txt
(grep -v amdgpu $R/hbm_probe.txt; echo; echo "per-stage times at N=20000 (tools/bench_stages.py 20000 8 5):"; grep -v amdgpu $R/stages_20000.txt
 echo; echo "C3 through the API (tools/c3_stages.py):"; grep -v "amdgpu\|Warning" $R/stages_c3.txt) > $P/r05_hbm_probe.txt
ls $P | grep r05
