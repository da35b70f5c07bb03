# kernel durations of tools/search_latency.py (one search, 128 steps)
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_search_latency
rm -rf $OUT; mkdir -p $OUT
cd /tmp
python3 $GRAFT_REPO_ROOT/tools/search_latency.py ${1:-1}
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/search_latency.py ${1:-1} > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs']) / 1e6, 3), 'ms, avg', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
t = glob.glob('$OUT/trace/*/*kernel_trace.csv')[0]
rows = sorted((r for r in csv.DictReader(open(t)) if 'k_bnb_eval8_list' in r['Kernel_Name'] or 'k_search_step' in r['Kernel_Name']), key=lambda r: int(r['Start_Timestamp']))
prev = None
out = []
for r in rows[40:60]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append(f"{'eval' if 'eval8' in r['Kernel_Name'] else 'step'} gap {0 if prev is None else s - prev} run {e - s}")
    prev = e
print('; '.join(out))
PY
rm -rf $OUT/trace
