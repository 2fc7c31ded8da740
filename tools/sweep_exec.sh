export CHECK=0
run() { echo "== $*"; env "$@" timeout 120 python tools/potrf_check.py 15000 20000 2>&1 | grep "N="; }
run GPP_EXEC_SCHED=0
run GPP_EXEC_SCHED=1
run GPP_EXEC_MIN_REM=5500
run GPP_EXEC_MIN_REM=7500
run GPP_EXEC_MIN_REM=8500
run GPP_EXEC_LAF=0.25
run GPP_EXEC_LAF=0.55
run GPP_EXEC_TBLOCK=300
run GPP_EXEC_TBLOCK=1400
run GPP_EXEC_PS=2
run GPP_EXEC_PS=6
run GPP_EXEC_SCHED=0
run GPP_EXEC_SCHED=1
