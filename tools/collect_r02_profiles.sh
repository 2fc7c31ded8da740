#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repo root).  Writes gpurun_out/r2prof/.
# Counter passes are separate runs with --kernel-trace only (no other trace domain), as the pool requires.
set -u
OUT=gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
rocprofv3 -L > $OUT/counters_list.txt 2>&1
# 1) kernel trace + stats of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o b -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
# 2) SQ counters, pass A (instruction mix + wave-cycle split + MFMA busy) and pass B (L2 hit, LDS conflicts), whole evaluation
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmcA -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcA.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/pmcB -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcB.log 2>&1
# 3) HBM-side traffic of the factorisation alone (build + potrf stages only), FETCH_SIZE and WRITE_SIZE in separate passes
STAGES_ONLY=build,potrf rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmcF -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcF.log 2>&1
STAGES_ONLY=build,potrf rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmcW -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcW.log 2>&1
# 3b) the same two passes over the whole evaluation (the LAUUM launch, TAG 1, and the N^2 kernels)
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmcF2 -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcF2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmcW2 -o a -- python3 tools/bench_stages.py 20000 8 1 > $OUT/pmcW2.log 2>&1
# 4) timeline of the factorisation (kernel trace only): per-queue busy time, update-queue gaps, tail windows
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/bench_stages.py 20000 8 2 > $OUT/trace.log 2>&1
python3 tools/trace_lookahead.py $OUT/trace 30 > $OUT/trace_lookahead.txt 2>&1
python3 tools/trace_tail.py $OUT/trace 2 > $OUT/trace_tail.txt 2>&1
for d in pmcA pmcB pmcF pmcW pmcF2 pmcW2; do python3 tools/pmc_summary.py $OUT/$d > $OUT/${d}_summary.txt 2>&1; done
find $OUT -name "*kernel_stats.csv" | head -3
# keep the merge under the 64 MiB limit: drop the raw per-dispatch CSVs, keep the summaries and the stats
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
