# Kernel time of every k_icp_step of the bench's 1 M x 1 M ICP extra (10 degrees off at the start).
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_1m
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-goicp --no-cpu --steps 5 --warmup 1 > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_1m"
rows = [r for r in csv.DictReader(open(glob.glob(out + "/trace/*/*_kernel_trace.csv")[0])) if "k_icp_step" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("k_icp_step us, 1 M x 1 M:", [round(x) for x in d[:31]])
PY
rm -rf $OUT/trace
