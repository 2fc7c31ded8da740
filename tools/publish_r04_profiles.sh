#!/bin/bash
# Copy the summaries of tools/collect_r04_profiles.sh (gpurun_out/r4prof/, scratch) into profiles/r04_* (tracked).
set -u
cd "$(dirname "$0")/.."
R=gpurun_out/r4prof; P=profiles
cp $R/r04_potrf_pmc.json $R/r04_trtri_pmc.json $R/r04_lauum_pmc.json $P/
cp $R/bench_kernel_stats.csv $P/r04_bench_kernel_stats.csv
grep "^{" $R/bench_profiled.json > $P/r04_bench_line_under_rocprof.json
cp $R/pmc_fetch_write.txt $P/r04_pmc_fetch_write.txt
(echo; echo "== FETCH_SIZE pass, whole evaluation"; cat $R/pmcF_ALL_summary.txt; echo; echo "== WRITE_SIZE pass, whole evaluation"; cat $R/pmcW_ALL_summary.txt) >> $P/r04_pmc_fetch_write.txt
(echo "SQ counters, whole evaluation at N=20000 (tools/bench_stages.py 20000 8 1 under rocprofv3 --pmc, two passes; tools/pmc_summary.py); kernel build $(cat $R/lib_version.txt)"
 echo "== pass A: instruction mix, wave-cycle split, MFMA busy"; cat $R/pmcA_summary.txt; echo; echo "== pass B: L2 hit rate, LDS"; cat $R/pmcB_summary.txt) > $P/r04_sq_counters.txt
(echo "Factorisation at N=20000 (STAGES_ONLY=build,potrf tools/bench_stages.py 20000 8 2 under rocprofv3 --kernel-trace; last evaluation)."
 echo "q2 = panel stream (32 CUs), q3 = masked throughput stream (224 CUs), q5 = stream without a CU mask (bulk of the early trailing updates), q1 = caller's stream."
 grep -v "Traceback\|File \"\|IndexError\|    end = \|    return getitem" $R/trace_lookahead.txt
 echo; echo "== dispatches >= 40 us of the last factorisation (tools/trace_window.py)"; cat $R/trace_potrf_big_kernels.txt) > $P/r04_timeline_potrf.txt
KEEP=$(awk '/== repeatability/{p=1} p' $P/r04_restarts_and_sharded_1rank.txt 2>/dev/null)
(echo "Sharded evaluation with ONE rank (the algorithm without communication), tools/run_sharded.py; round 4 (back-substitution, head/tail row solves)"
 grep "^N=" $R/sharded_1rank_20000.txt; grep "^N=" $R/sharded_1rank_60000.txt
 echo; echo "== dispatches >= 100 us of one evaluation at N = 20000 (rocprofv3 --kernel-trace, tools/trace_window.py)"; cat $R/trace_sharded_big_kernels.txt
 echo; echo "== restart batching (tools/bench_restarts.py; the batched driver replays loss + gradients of each Adam step as one HIP graph,"
 echo "   the 'batched evaluation' lines time BatchedObjective.loss() + backward issued eagerly: replayed, a step is 0.42 ms at N = 100, B = 5;"
 echo "   0.75 ms at N = 500, B = 5; 1.42 ms at N = 500, B = 65)"; grep -v amdgpu $R/restarts.txt
 echo; echo "$KEEP") > $P/r04_restarts_and_sharded_1rank.txt.new && mv $P/r04_restarts_and_sharded_1rank.txt.new $P/r04_restarts_and_sharded_1rank.txt
grep -v "amdgpu\|Warning" $R/configs.txt > $P/r04_configs_C1_C5_single_gpu.txt
(echo "One trailing-update launch C(upper) -= A^T A (gpp_gemm_f64<2,64,64,0,16,2>), isolated launches: rate vs tile count (tools/small_update_probe.py)"
 echo "== all 256 CUs"; grep "K=" $R/update_rate_vs_tiles.txt
 echo "== on the CU-masked throughput stream (224 CUs; GPP_GEMM_ON_UPD=1)"; grep "K=" $R/update_rate_vs_tiles_masked.txt) > $P/r04_update_rate_vs_tiles.txt
(grep -v amdgpu $R/hbm_probe.txt; grep -v amdgpu $R/exp_check.txt; echo; echo "per-stage times at N=20000 (tools/bench_stages.py 20000 8 5):"; grep -v amdgpu $R/stages_20000.txt) > $P/r04_hbm_probe.txt
(echo "Statically scheduled steps of gpp_potrf_ws (gpp_exec_f64 + gpp_plan.hip), round 4: per-task stamps (tools/exec_trace.py), N = 20000 then 15000"
 grep -v amdgpu $R/exec_trace_20000.txt; echo; grep -v amdgpu $R/exec_trace_15000.txt
 echo; echo "== A/B against launches per product and knob sweep on ONE box (tools/sweep_exec.sh; potrf ms at N = 15000 / 20000)"; grep -v amdgpu $R/exec_sweep.txt) > $P/r04_exec_schedule.txt
(echo "Sequential Adam driver (fit_model_torch) with the evaluation replayed as one HIP graph, tools/bench_adam_seq.py"; grep -v amdgpu $R/adam_seq.txt) > $P/r04_adam_seq_replay.txt
ls $P | grep r04
