#!/bin/bash
# dev: one-rank sharded evaluation, ticket list vs launches (usage: tools/shard_list_bench.sh N [N ...])
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in "$@"; do
  for list in 1 0; do
    echo "== N $n GPP_SHARD_LIST=$list"
    GPP_SHARD_LIST=$list timeout 900 python bench.py --mode sharded --n $n --nb 1024 --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ms_per_step', d.get('ms_per_step'), 'value', d.get('value'))
print('stages', json.dumps(d.get('stages', {}).get('ms')))
"
  done
done
