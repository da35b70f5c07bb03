# Kernel-trace statistics of the config-1 ICP (bunny000 -> bunny045, 30 iterations through the host class).
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_icp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/icp_bunny_time.py > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_icp"
import re, collections
per = collections.defaultdict(list)
first = None
for r in csv.DictReader(open(glob.glob(out + "/trace/*/*_kernel_trace.csv")[0])):
    m = re.search(r"\bk_(icp_\w+|reduce_sums)", r["Kernel_Name"])
    if not m: continue
    per[m.group(0)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k in ("k_icp_tile", "k_icp_ball", "k_icp_walk", "k_icp_sums", "k_reduce_sums"):
    d = [(e - s) / 1e3 for s, e in per.get(k, [])]
    print(k, "us, first registration:", [round(x) for x in d[:30]])
    print(k, "us, last registration: ", [round(x) for x in d[-30:]])
t, r = per.get("k_icp_tile", []), per.get("k_reduce_sums", [])
if t and r:
    print("tile start -> reduce end us, last registration:", [round((r[-30 + i][1] - t[-30 + i][0]) / 1e3) for i in range(30)])
    print("reduce end -> next tile start us:", [round((t[-30 + i + 1][0] - r[-30 + i][1]) / 1e3) for i in range(29)])
PY
find $OUT/trace -name '*_kernel_trace.csv' -delete
grep iterations $OUT/trace.log | tail -1
