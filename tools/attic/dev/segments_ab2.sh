#!/bin/bash
# segments A/B, one repetition, + whether PyTorch's AccumulateGrad stream warning still appears
for c in C3 C4 C2; do
  for g in 0 1; do
    echo "== $c GPP_GRAPHED_SEGMENTS=$g"
    GPP_GRAPHED_SEGMENTS=$g timeout 600 python3 tools/run_configs.py $c 2>&1 | grep -v "amdgpu" | grep -c "AccumulateGrad" | sed 's/^/   AccumulateGrad warnings: /'
    GPP_GRAPHED_SEGMENTS=$g timeout 600 python3 tools/run_configs.py $c 2>&1 | grep "^C[0-9]:"
  done
done
