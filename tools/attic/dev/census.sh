# kernel census of one evaluation of a config: rocprofv3 kernel trace of tools/run_configs.py (model construction + 1 warm-up + reps evaluations)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6h
rm -rf $out; mkdir -p $out
for c in C3 C1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/census_$c -o $c -- python3 tools/run_configs.py $c > $out/$c.log 2>&1
  echo "$c rc=$?"; tail -n 3 $out/$c.log
  f=$(find /tmp/census_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/${c}_kernel_stats.csv
done
ls -la $out
