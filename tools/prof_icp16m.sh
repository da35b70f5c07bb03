# Kernel time of every kernel of the first ICP passes on the synthetic 16 M x 16 M pair.
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_16m
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py 16000000 12 > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_16m"
rows = list(csv.DictReader(open(glob.glob(out + "/trace/*/*_kernel_trace.csv")[0])))
per = collections.defaultdict(list)
for r in rows:
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    name = m.group(0) if m else r["Kernel_Name"]
    per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in ("k_icp_tile", "k_icp_ball", "k_icp_solo", "k_icp_walk", "k_icp_sums", "k_reduce_sums", "k_cand_seed", "k_cand_flood", "k_tile_cluster"):
    d = per.get(k, [])
    print(k, "us per launch:", [round(x) for x in d[:14]])
PY
tail -3 $OUT/trace.log
rm -rf $OUT/trace
