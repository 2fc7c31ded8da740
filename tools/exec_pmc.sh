# counters of the executor's kernel running ALONE (tools/exec_update_probe.py): separate --pmc passes, kernel trace only
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4exec_pmc; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/A -o a -- python3 tools/exec_update_probe.py 20000 2 > $OUT/A.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/B -o a -- python3 tools/exec_update_probe.py 20000 2 > $OUT/B.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/F -o a -- python3 tools/exec_update_probe.py 20000 2 > $OUT/F.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/W -o a -- python3 tools/exec_update_probe.py 20000 2 > $OUT/W.log 2>&1
for d in A B F W; do echo "== pass $d"; python3 tools/pmc_summary.py $OUT/$d 2>&1 | grep -A3 "counters:\|gpp_exec_f64\|gpp_gemm_f64<2, 64, 64, 0" ; done > $OUT/summary.txt
grep "N=" $OUT/A.log >> $OUT/summary.txt
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
