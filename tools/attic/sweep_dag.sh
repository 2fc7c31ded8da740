#!/bin/bash
# Round-5 sweep of the DAG executor's knobs (dev tool): each line = one configuration of tools/dag_check.py, timing only.
out=gpurun_out/r5c; mkdir -p $out
run() { echo "== $*" >> $out/sweep.txt; env "$@" CHECK=0 REPS=5 timeout 600 python tools/dag_check.py $SIZES 2>&1 | grep -v amdgpu.ids >> $out/sweep.txt; }
SIZES="6144 7168 8192 10000 11264"
run GPP_DAG_MIN_N=3840
run GPP_DAG_MIN_N=3840 GPP_DAG_FIRST=512
run GPP_DAG_MIN_N=3840 GPP_DAG_FIRST=256
run GPP_DAG_SCHED=0
SIZES="12288 15000 20000"
run GPP_DAG_SCHED=0
run GPP_DAG_MAX_N=40000
run GPP_DAG_MAX_N=40000 GPP_DAG_FIRST=512
run GPP_DAG_MAX_N=40000 GPP_DAG_INV_MAX=40000
run GPP_DAG_MAX_N=40000 GPP_DAG_INV_MAX=40000 GPP_DAG_FIRST=512
run GPP_DAG_MAX_N=40000 GPP_DAG_NB=2048
run GPP_DAG_MAX_N=40000 GPP_DAG_NB=2048 GPP_DAG_INV_MAX=40000
SIZES="30000"
run GPP_DAG_SCHED=0
run GPP_DAG_MAX_N=40000
run GPP_DAG_MAX_N=40000 GPP_DAG_NB=2048
(GPP_DAG_MAX_N=40000 GPP_DAG_INV_MAX=40000 TRACE=1 CHECK=0 timeout 300 python tools/dag_check.py 20000 2>&1 | grep -v amdgpu.ids > $out/trace_inv_20000.txt)
(GPP_DAG_FIRST=512 GPP_DAG_MAX_N=40000 GPP_DAG_INV_MAX=40000 timeout 600 python tools/dag_check.py 7300 10000 13000 17000 2>&1 | grep -v amdgpu.ids > $out/check.txt)
