#!/bin/bash
# Round-6 profile collection on the GPU box (run through gpurun from the repo root).  Writes gpurun_out/r6prof/; the summaries are
# then copied into profiles/r06_* (tools/publish_r06_profiles.sh).  Counter passes are separate runs with --kernel-trace only.
set -u
OUT=gpurun_out/r6prof
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'.'); from gpplus_amd import _lib; print(_lib.load().gpp_version().decode())" > $OUT/lib_version.txt
# Counter collection SERIALISES dispatches; the DAG executor's launches wait for the panel stream's launches through device
# counters and cannot run one at a time.  Every --pmc pass therefore runs the SAME ticket list with the SAME kernel as a sequence of
# launches that never wait for each other (GPP_DAG_PHASED=1, csrc/gpp_api.hip::potrf_dag): same tiles, same products, same order.
export GPP_DAG_PHASED=1
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
ROUND=r06 timeout 900 python3 tools/pmc_to_json.py $OUT $OUT/pmcF_P $OUT/pmcW_P $OUT/pmcF_PT $OUT/pmcW_PT $OUT/pmcF_ALL $OUT/pmcW_ALL > $OUT/pmc_fetch_write.txt 2>&1
for d in pmcA pmcB pmcF_ALL pmcW_ALL pmcF_P pmcW_P; do python3 tools/pmc_summary.py $OUT/$d > $OUT/${d}_summary.txt 2>&1; done
# the phased form's own timing (how far the profiled form is from the timed one)
timeout 300 python3 tools/bench_stages.py 20000 8 3 > $OUT/stages_20000_phased.txt 2>&1
unset GPP_DAG_PHASED
# 1) the bench command itself, AFTER the counter passes so that its roofline.traffic reads this build's records: the default run
#    (with the CPU baseline, minutes of host time), and under kernel trace + stats
cp $OUT/r06_potrf_pmc.json $OUT/r06_trtri_pmc.json $OUT/r06_lauum_pmc.json profiles/
timeout 900 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o b -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
find $OUT/bench_stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
# 4) timelines: per-task traces of the executor at the C2 / C3 / C4 sizes, kernel timeline at N = 10000
TRACE=1 CHECK=0 timeout 300 python3 tools/dag_check.py 10000 > $OUT/dag_trace_10000.txt 2>&1
TRACE=1 CHECK=0 timeout 300 python3 tools/dag_check.py 15000 > $OUT/dag_trace_15000.txt 2>&1
TRACE=1 CHECK=0 timeout 300 python3 tools/dag_check.py 20000 > $OUT/dag_trace_20000.txt 2>&1
STAGES_ONLY=build,potrf timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace10k -o t -- python3 tools/bench_stages.py 10000 8 2 > $OUT/trace10k.log 2>&1
timeout 900 python3 tools/trace_window.py $OUT/trace10k gpp_cov_tile - 30 100 > $OUT/timeline_n10000.txt 2>&1
# 5) sharded evaluation with one rank (algorithm without communication): timing at C2 and C5 size
timeout 900 python3 tools/run_sharded.py 20000 8 1024 2 > $OUT/sharded_1rank_20000.txt 2>&1
timeout 900 python3 tools/run_sharded.py 60000 16 1024 1 > $OUT/sharded_1rank_60000.txt 2>&1
# ticket lists (default) against the launch-per-product path of rounds 2-4 (GPP_SHARD_LIST=0), one rank, C2 and C5, same box
timeout 1500 bash tools/shard_list_bench.sh 20000 60000 > $OUT/sharded_lists_1rank.txt 2>&1
GPP_SHARDED_FORCE_COLLECTIVES=1 timeout 600 python3 bench.py --mode sharded --n 20000 --steps 3 --warmup 1 > $OUT/sharded_bench_line_20000.json 2> $OUT/sharded_bench_line_20000.err
# 5b) virtual-rank replays: tools/collect_r06_replay.sh (8 / 4 / 2 ranks, checks, the push transport's owner side)
# 5c) kernel census of one evaluation through the plain API (how many launches are not the library's)
for c in C3 C1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/census_$c -o $c -- python3 tools/run_configs.py $c > $OUT/census_$c.log 2>&1
  f=$(find /tmp/census_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/census_${c}_kernel_stats.csv
done
# 6) all BASELINE configs on one GPU, stage tables
timeout 900 python3 tools/run_configs.py > $OUT/configs.txt 2>&1
timeout 900 python3 tools/bench_stages.py 20000 8 5 > $OUT/stages_20000.txt 2>&1
timeout 300 python3 tools/c3_stages.py > $OUT/stages_c3.txt 2>&1
timeout 900 python3 tools/hbm_probe.py > $OUT/hbm_probe.txt 2>&1
# keep the merge under the 64 MiB limit: drop the raw per-dispatch CSVs, keep the summaries and the stats
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
