#!/bin/bash
# dev: the sharded ticket list against the single-GPU path, ranks sharing the one GPU (usage: tools/attic/shard_list_dev.sh "world N nb" ...)
export GPP_SHARD_DEBUG=1 HSA_ENABLE_IPC_MODE_LEGACY=0 GPP_SHARD_TIMEOUT_MS=${GPP_SHARD_TIMEOUT_MS:-20000}
port=29700
for c in "$@"; do
  set -- $c
  world=$1; N=$2; nb=$3; shift 3
  port=$((port + 1))
  echo "== world $world N $N nb $nb $*"
  env GPP_SHARD_WORKERS=$((448 / world)) "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$world --master-addr 127.0.0.1 \
    --master-port $port tests/workers/sharded_worker.py $N 6 $nb 0 1 0 2>&1 | grep -E "RESULT|same_as|Error|error|Traceback|time|status" | cut -c1-600
done
