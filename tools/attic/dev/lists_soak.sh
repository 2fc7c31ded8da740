#!/bin/bash
# dev: repeatability / flake-rate soak of the sharded ticket lists on one GPU: 200 one-rank repetitions (bitwise), 12 three-rank runs
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPP_SHARD_DEBUG=1 GPP_SHARD_TIMEOUT_MS=20000
GPP_SHARDED_FORCE_COLLECTIVES=1 MASTER_PORT=29741 timeout 900 python tools/stress_sharded.py 9000 512 200 nccl 2>&1 | tail -2
ok=0; bad=0
for i in $(seq 1 12); do
  out=$(GPP_SHARD_WORKERS=149 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port $((29760 + i)) \
        tests/workers/sharded_worker.py 10000 5 1024 0 2 2 2>&1 | grep -E "RESULT|sharded rank")
  if echo "$out" | grep -q '"list_evals": 1, "back_list_evals": 1' && ! echo "$out" | grep -q "sharded rank"; then ok=$((ok + 1)); else bad=$((bad + 1)); echo "$out" | cut -c1-300; fi
done
echo "three-rank runs: $ok lists completed, $bad did not"
