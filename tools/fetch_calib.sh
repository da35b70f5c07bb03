# Runs build_ab/fetch_calib under the two HBM counters and prints counter / bytes per kernel.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 60 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- $GRAFT_REPO_ROOT/build_ab/fetch_calib > $OUT/$c.log 2>&1 || { echo "$c failed"; tail -3 $OUT/$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/calib"
res = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(res.items()):
    print(f"{k:16s} {c:11s} {sum(v)/len(v):12.0f} KiB = {sum(v)/len(v)*1024/(1<<30):.3f} of the bytes moved")
PY
