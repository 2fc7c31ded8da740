#!/bin/bash
# dev: does an idle fourth process with a GPU context (like the pytest parent) break the 3-rank list on the shared GPU?
export GPP_SHARD_DEBUG=1 HSA_ENABLE_IPC_MODE_LEGACY=0 GPP_SHARD_TIMEOUT_MS=10000
python - <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from gpplus_amd.backend import get_context, square_buffer
ctx = get_context("cuda:0")
n = 8192
A = square_buffer(n, "cuda"); Li = square_buffer(n, "cuda"); T = square_buffer(n, "cuda")
A.zero_(); A.diagonal().fill_(4.0)
info = torch.zeros(1, dtype=torch.int32, device="cuda")
ctx.potrf(A, Li, info, T); ctx.trtri(A, Li, T)
ss = [torch.cuda.Stream() for _ in range(6)]
for s in ss:
    with torch.cuda.stream(s):
        torch.zeros(4, device="cuda").add_(1)
torch.cuda.synchronize()
print("parent-like process idle with its context", flush=True)
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 60)
PY
PARENT=$!
sleep 12
for i in 1 2; do
env $EXTRA GPP_SHARD_WORKERS=149 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port $((29870 + i)) \
    tests/workers/sharded_worker.py 10000 5 1024 2 1 0 2>&1 | grep -E "sharded rank|list_evals|rror" | cut -c1-300 | sed 's/.*\("list_evals.*\)/\1/'
done
kill $PARENT
