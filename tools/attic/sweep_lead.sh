out=gpurun_out/r5h; mkdir -p $out
run() { echo "== $*" >> $out/lead.txt; env "$@" REPS=5 timeout 600 python tools/dag_check.py $SIZES 2>&1 | grep -v amdgpu.ids >> $out/lead.txt; }
SIZES="17000 20000"
run GPP_DAG_INV_LEAD=0 CHECK=0
run GPP_DAG_INV_LEAD=4096 CHECK=0
run GPP_DAG_INV_LEAD=8192 CHECK=1
SIZES="30000"
run GPP_DAG_INV_LEAD=0 CHECK=0
run GPP_DAG_INV_LEAD=8192 CHECK=0
run GPP_DAG_INV_LEAD=16384 CHECK=0
SIZES="20000"
(GPP_DAG_INV_LEAD=8192 TRACE=1 CHECK=0 timeout 300 python tools/dag_check.py 20000 2>&1 | grep -v amdgpu.ids > $out/trace_lead_20000.txt)
cat $out/lead.txt; cat $out/trace_lead_20000.txt
