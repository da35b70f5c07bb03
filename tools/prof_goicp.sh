# Kernel-trace statistics of the Go-ICP extra of the bench (reference order with look-ahead, then width 8).
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_goicp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 --warmup 1 --nn-points 20000 > $OUT/trace.log 2>&1
find $OUT/trace -name '*_kernel_trace.csv' -delete
head -6 $OUT/trace/*/*_kernel_stats.csv | cut -c1-160
grep -a '^{"metric' $OUT/trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v['time_to_optimum_s'],3), v['launches']) for k,v in d['extra'].items() if isinstance(v,dict)})"
