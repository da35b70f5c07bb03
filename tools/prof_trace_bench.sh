# rocprofv3 kernel-trace stats of the default bench command (the summary goes to profiles/ through tools/summarize_prof.py)
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
rm -rf $OUT/trace; mkdir -p $OUT
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/trace.log 2>&1
find $OUT/trace -name '*_kernel_trace.csv' -delete
tail -1 $OUT/trace.log | cut -c1-200
