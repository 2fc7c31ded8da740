#!/bin/bash
# Round-4 profile collection on the GPU box (run through gpurun from the repo root).  Writes gpurun_out/r4prof/; the summaries are
# then copied into profiles/r04_*.  Counter passes are separate runs with --kernel-trace only (no other trace domain).
set -u
OUT=gpurun_out/r4prof
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'.'); from gpplus_amd import _lib; print(_lib.load().gpp_version().decode())" > $OUT/lib_version.txt
# 1) the bench command itself: plain, and under kernel trace + stats
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o b -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
find $OUT/bench_stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
# Counter collection SERIALISES dispatches; the statically scheduled steps are launches that wait for each other through device
# counters (persistent executor <-> gated panels) and cannot run one at a time: every --pmc pass therefore runs the launch-per-product
# driver (GPP_EXEC_SCHED=0), and the traffic records describe that path — same kernels, same tiles, same products per tile.
export GPP_EXEC_SCHED=0
# 2) SQ counters: instruction mix + MFMA busy (pass A), L2 hit / LDS conflicts (pass B), whole evaluation
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmcA -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcA.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/pmcB -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcB.log 2>&1
# 3) HBM-side traffic per stage: FETCH_SIZE and WRITE_SIZE in separate passes
for st in "build,potrf:P" "build,potrf,trtri:PT"; do
  s=${st%%:*}; n=${st##*:}
  STAGES_ONLY=$s timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmcF_$n -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcF_$n.log 2>&1
  STAGES_ONLY=$s timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmcW_$n -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcW_$n.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmcF_ALL -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcF_ALL.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmcW_ALL -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcW_ALL.log 2>&1
timeout 900 python3 tools/pmc_to_json.py $OUT $OUT/pmcF_P $OUT/pmcW_P $OUT/pmcF_PT $OUT/pmcW_PT $OUT/pmcF_ALL $OUT/pmcW_ALL > $OUT/pmc_fetch_write.txt 2>&1
for d in pmcA pmcB pmcF_ALL pmcW_ALL; do python3 tools/pmc_summary.py $OUT/$d > $OUT/${d}_summary.txt 2>&1; done
unset GPP_EXEC_SCHED
# 4) timeline of the factorisation
STAGES_ONLY=build,potrf timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/bench_stages.py 20000 8 2 > $OUT/trace.log 2>&1
timeout 900 python3 tools/trace_lookahead.py $OUT/trace 30 > $OUT/trace_lookahead.txt 2>&1
timeout 900 python3 tools/trace_window.py $OUT/trace gpp_cov_tile - 400 40 > $OUT/trace_potrf_big_kernels.txt 2>&1
# 4b) the statically scheduled steps: per-task time stamps of the executor, A/B against launches per product, knob sweep
timeout 600 python3 tools/exec_trace.py 20000 3 > $OUT/exec_trace_20000.txt 2>&1
timeout 600 python3 tools/exec_trace.py 15000 3 > $OUT/exec_trace_15000.txt 2>&1
timeout 900 bash tools/sweep_exec.sh > $OUT/exec_sweep.txt 2>&1
timeout 600 python3 tools/bench_adam_seq.py > $OUT/adam_seq.txt 2>&1
# 5) sharded evaluation with one rank (algorithm without communication): timing at C2 and C5 size, timeline at C2 size
timeout 900 python3 tools/run_sharded.py 20000 8 1024 2 > $OUT/sharded_1rank_20000.txt 2>&1
timeout 900 python3 tools/run_sharded.py 60000 16 1024 1 > $OUT/sharded_1rank_60000.txt 2>&1
timeout 900 python3 tools/run_sharded.py 60000 16 1536 1 >> $OUT/sharded_1rank_60000.txt 2>&1
timeout 900 python3 tools/run_sharded.py 60000 16 2048 1 >> $OUT/sharded_1rank_60000.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_sh -o t -- python3 tools/run_sharded.py 20000 8 1024 1 > $OUT/trace_sh.log 2>&1
timeout 900 python3 tools/trace_window.py $OUT/trace_sh gpp_cov_tile - 500 100 > $OUT/trace_sharded_big_kernels.txt 2>&1
# 6) all BASELINE configs on one GPU, restart batching, per-launch rate vs tile count, N^2 kernels and the HBM yardstick, exp accuracy
timeout 900 python3 tools/run_configs.py > $OUT/configs.txt 2>&1
timeout 900 python3 tools/bench_restarts.py > $OUT/restarts.txt 2>&1
timeout 900 python3 tools/small_update_probe.py > $OUT/update_rate_vs_tiles.txt 2>&1
GPP_GEMM_ON_UPD=1 timeout 600 python3 tools/small_update_probe.py > $OUT/update_rate_vs_tiles_masked.txt 2>&1
timeout 900 python3 tools/hbm_probe.py > $OUT/hbm_probe.txt 2>&1
timeout 900 python3 tools/exp_check.py > $OUT/exp_check.txt 2>&1
timeout 900 python3 tools/bench_stages.py 20000 8 5 > $OUT/stages_20000.txt 2>&1
# keep the merge under the 64 MiB limit: drop the raw per-dispatch CSVs, keep the summaries and the stats
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
