# Kernel trace of the EDT-only bench; prints per-kernel average durations.
set -e -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 20 --warmup 5 > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_edt" in r["Name"]:
        print(r["Name"].split("(")[0][-60:], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1000, 1))
PY
