# rocprofv3 kernel-trace stats of Go-ICP with the inner searches on the device (tools/goicp_dev.py, one run)
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_goicp_dev
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/goicp_dev.py 1 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print(r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs']) / 1e6, 3), 'ms, avg', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
t = glob.glob('$OUT/trace/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(t)) if 'k_search_eval_step' in r['Kernel_Name']]
d = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
import bisect
for lim in (5000, 10000, 20000, 50000, 100000, 10**9):
    k = bisect.bisect_right(d, lim)
    print('eval launches up to', lim, 'ns:', k, 'sum ms', round(sum(d[:k]) / 1e6, 2))
rows = [r for r in csv.DictReader(open(t)) if "k_search_start" in r["Kernel_Name"]]
d = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
for lim in (3000, 5000, 10000, 20000, 10**9):
    k = bisect.bisect_right(d, lim)
    print('step launches up to', lim, 'ns:', k, 'sum ms', round(sum(d[:k]) / 1e6, 2))
PY
find $OUT/trace -name '*_kernel_trace.csv' -delete
tail -1 $OUT/trace.log
