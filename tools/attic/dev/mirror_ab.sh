#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for m in 1 0; do for n in 20000 60000; do
  echo "== N $n mirror beside the list: $m"
  GPP_SHARD_MIRROR_BESIDE=$m timeout 900 python bench.py --mode sharded --n $n --nb 1024 --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ms_per_step', round(d.get('ms_per_step'), 2), json.dumps(d.get('stages', {}).get('ms')))"
done; done; done
