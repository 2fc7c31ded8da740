#!/bin/bash
# dev: how often does the pure-C program hang?  usage: tools/attic/dev/c_prog_loop.sh REPS [N]
cd examples/shard_eval_c && make >/dev/null 2>&1
export GPP_SHARD_TIMEOUT_MS=8000 AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-0}
ok=0; hung=0; other=0
for i in $(seq 1 $1); do
  timeout 40 ./shard_eval ${2:-9000} 1024 > /tmp/c_prog_$i.log 2>&1; rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok + 1)); elif [ $rc -eq 124 ]; then hung=$((hung + 1)); echo "run $i HUNG; last lines:"; tail -4 /tmp/c_prog_$i.log | cut -c1-300; else other=$((other + 1)); echo "run $i rc=$rc"; tail -3 /tmp/c_prog_$i.log | cut -c1-300; fi
done
echo "ok $ok hung $hung other $other"
