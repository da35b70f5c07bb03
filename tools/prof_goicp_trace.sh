# Kernel-trace statistics of one full-size Go-ICP run (tools/goicp_sweep.py with GOICP_SETTING, default "2,127").
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_goicp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/goicp_sweep.py ${GOICP_SETTING:-2,127} > $OUT/trace.log 2>&1
find $OUT/trace -name '*_kernel_trace.csv' -delete
head -8 $OUT/trace/*/*_kernel_stats.csv | cut -c1-200
tail -3 $OUT/trace.log
