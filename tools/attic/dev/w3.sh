#!/bin/bash
export GPP_SHARD_DEBUG=1 HSA_ENABLE_IPC_MODE_LEGACY=0 GPP_SHARD_TIMEOUT_MS=10000
for i in 1 2 3 4 5 6 7 8 9 10; do
GPP_SHARD_WORKERS=149 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port $((29850 + i)) \
    tests/workers/sharded_worker.py 10000 5 1024 2 1 0 2>&1 | grep -E "sharded rank|list_evals|rror" | cut -c1-400 | sed 's/.*\("list_evals.*\)/\1/'
done
