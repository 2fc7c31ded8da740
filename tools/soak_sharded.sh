#!/bin/bash
# Round 6: the multi-rank tests (ticket lists over gloo on a shared GPU, the one-rank RCCL group, the C program over RCCL, the
# virtual-rank replays) repeated — the hang class of VERDICT r5 ("RCCL start-up time-out seen once in seven runs") counted, not skipped — and the shared-GPU
# artefact of round 6 (one of three ranks' lists starved past the 20 s budget, once in ~40 runs; the test harness repeats such a run ONCE and says so).
# usage (GPU box): bash tools/soak_sharded.sh [repeats] > gpurun_out/soak/soak.txt
R=${1:-4}
mkdir -p gpurun_out/soak
python3 -c "import sys; sys.path.insert(0, '.'); from importlib import import_module; print('library', import_module('gp-plus_amd._lib').load().gpp_version().decode())"
for i in $(seq 1 $R); do
  t0=$(date +%s)
  timeout 1500 python3 -m pytest tests/test_gpu_00_sharded_lists.py tests/test_gpu_sharded.py tests/test_gpu_00_replay.py -q -m gpu -rA > gpurun_out/soak/run_$i.txt 2>&1
  rc=$?
  echo "repeat $i: rc=$rc, $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/soak/run_$i.txt)"
  grep -c "first attempt hung" gpurun_out/soak/run_$i.txt | sed 's/^/  C-program first attempts that hung: /'
  grep -c "^SHARED-GPU-RETRY" gpurun_out/soak/run_$i.txt | sed 's/^/  runs repeated after a ticket-list time-out on the shared GPU (tests\/sharded_launch.py): /'
  grep "^FAILED\|^ERROR" gpurun_out/soak/run_$i.txt
done
