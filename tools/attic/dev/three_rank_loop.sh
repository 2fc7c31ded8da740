#!/bin/bash
# The 3-rank shared-GPU list evaluation (N = 10 000, Matern 5/2) over and over: how often does a list exceed its 20 s budget?
R=${1:-15}
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPP_SHARD_DEBUG=1 GPP_SHARD_TIMEOUT_MS=20000 GPP_SHARD_WORKERS=149
bad=0
for i in $(seq 1 $R); do
  t0=$(date +%s.%N)
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port $((31000 + i)) tests/workers/sharded_worker.py 10000 5 1024 2 1 0 > /tmp/three_$i.log 2>&1
  rc=$?
  dt=$(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0)
  n=$(grep -c "ticket list: status" /tmp/three_$i.log)
  le=$(grep -o '"list_evals": [0-9]*' /tmp/three_$i.log | head -1)
  echo "run $i rc=$rc ${dt}s status-lines=$n $le"
  if [ "$n" != "0" ]; then bad=$((bad+1)); grep "ticket list: status" /tmp/three_$i.log | cut -c1-700; fi
done
echo "runs with a time-out: $bad of $R"
