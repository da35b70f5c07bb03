# Kernel time of every kernel of an ICP pass on the synthetic 1 M x 1 M pair (10 degrees off at the start).
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_1m
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_1m"
rows = list(csv.DictReader(open(glob.glob(out + "/trace/*/*_kernel_trace.csv")[0])))
per = collections.defaultdict(list)
for r in rows:
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    name = m.group(0) if m else r["Kernel_Name"]
    per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in ("k_icp_tile", "k_icp_ball", "k_icp_solo", "k_icp_walk", "k_icp_sums", "k_reduce_sums"):
    d = per.get(k, [])
    print(k, "us per pass:", [round(x) for x in d[:40]])
PY
tail -3 $OUT/trace.log
rm -rf $OUT/trace
