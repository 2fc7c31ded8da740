#!/bin/bash
# segments A/B through bench.py at several N (ms per evaluation), alternating, three repetitions
for rep in 1 2 3; do
  for n in 4096 6144 8192 10000 15000 20000; do
    for g in 0 1; do
      ms=$(GPP_GRAPHED_SEGMENTS=$g timeout 600 python3 bench.py --n $n --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys, json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'], 3))")
      echo "rep $rep N $n segments $g: $ms ms"
    done
  done
done
