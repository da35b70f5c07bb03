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
rows = [r for r in csv.DictReader(open(glob.glob(out + "/trace/*/*_kernel_trace.csv")[0])) if "k_icp_step" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("k_icp_step us, first registration:", [round(x) for x in d[:30]])
print("k_icp_step us, last registration:", [round(x) for x in d[-30:]])
PY
find $OUT/trace -name '*_kernel_trace.csv' -delete
grep iterations $OUT/trace.log | tail -1
