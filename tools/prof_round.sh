# rocprofv3 evidence for the round: kernel-trace stats of the default bench command and, in separate
# runs, the HBM traffic counters of the EDT kernels.
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 5 --warmup 2 > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 5 --warmup 2 > $OUT/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 5 --warmup 2 > $OUT/sq.log 2>&1
# the per-dispatch trace of the whole bench is tens of MiB; the stats files are what is kept
find $OUT/trace -name '*_kernel_trace.csv' -delete
du -sh $OUT
tail -1 $OUT/trace.log | cut -c1-300
