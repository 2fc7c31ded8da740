#!/bin/bash
# dev: world-2 sharded list on one GPU with a given number of executor work-groups per rank
export GPP_SHARD_TIMEOUT_MS=8000 GPP_SHARD_DEBUG=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for w in "$@"; do
  echo "== workers $w"
  GPP_SHARD_WORKERS=$w timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29810 + w % 50)) \
    tests/workers/sharded_worker.py 9000 6 1024 0 1 0 2>&1 | grep -E "sharded rank|list_evals" | cut -c1-120 | sed 's/.*list_evals/list_evals/'
done
