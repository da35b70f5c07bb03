# kernel stats of the bunny pair's 30 iterations through the host class (device loop and host loop)
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_icp_bunny
rm -rf $OUT; mkdir -p $OUT
cd /tmp
python3 $GRAFT_REPO_ROOT/tools/icp_bunny_time.py
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/icp_bunny_time.py > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs']) / 1e6, 3), 'ms, avg', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
PY
rm -rf $OUT/trace
