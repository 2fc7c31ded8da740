#!/bin/bash
# Round-6 virtual-rank replays on the GPU box (tools/replay_rank.py): C5 and C2 on 8 virtual ranks with checks and a task trace, and the
# same problems on 2 and 4 ranks — together with the one-rank runs of tools/collect_r06_profiles.sh the scaling curve 1 / 2 / 4 / 8.
set -u
OUT=gpurun_out/r6prof
mkdir -p $OUT
export GPP_SHARD_TIMEOUT_MS=30000
timeout 1500 python3 tools/replay_rank.py --config C5 --P 8 --rates 0,400,150,70,50 --sweeps 8 --check --trace-rank 3 --json $OUT/replay_c5.json > $OUT/replay_c5.txt 2>&1; echo "C5 P=8 rc=$?"
timeout 600 python3 tools/replay_rank.py --config C2 --P 8 --rates 0,400,150,70,50 --sweeps 6 --check --json $OUT/replay_c2.json > $OUT/replay_c2.txt 2>&1; echo "C2 P=8 rc=$?"
for P in 2 4; do
  timeout 1200 python3 tools/replay_rank.py --config C5 --P $P --rates 0,150,70,50 --sweeps 6 --json $OUT/replay_c5_p$P.json > $OUT/replay_c5_p$P.txt 2>&1; echo "C5 P=$P rc=$?"
  timeout 600 python3 tools/replay_rank.py --config C2 --P $P --rates 0,150,70,50 --sweeps 6 --json $OUT/replay_c2_p$P.json > $OUT/replay_c2_p$P.txt 2>&1; echo "C2 P=$P rc=$?"
done
# the push transport's owner side (GPP_SHARD_PUSH=1: no packing copy in front of a message), same ranks and rates
timeout 1500 python3 tools/replay_rank.py --config C5 --P 8 --rates 400,150,70,50 --sweeps 8 --push --json $OUT/replay_c5_push.json > $OUT/replay_c5_push.txt 2>&1; echo "C5 P=8 push rc=$?"
timeout 600 python3 tools/replay_rank.py --config C2 --P 8 --rates 400,150,70,50 --sweeps 6 --push --json $OUT/replay_c2_push.json > $OUT/replay_c2_push.txt 2>&1; echo "C2 P=8 push rc=$?"
grep -E "^virtual|per evaluation|vs the single" $OUT/replay_c*.txt | cut -c1-240
