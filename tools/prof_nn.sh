set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_nn
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 --warmup 2 > $OUT/trace.log 2>&1
head -12 $OUT/trace/*/*_kernel_stats.csv | cut -c1-160
grep k_icp_step $OUT/trace/*/*_kernel_trace.csv | awk -F, '{print $(NF-8), $(NF-7)}' | head -3
