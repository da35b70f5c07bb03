set -e -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
ICPX_LIB_DIR=$GRAFT_REPO_ROOT/build_ab/stats timeout -k 10 200 python3 -u tools/icp1m_time.py > gpurun_out/nn_stats.log 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_all
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py > $OUT/p1.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_all"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\bk_\w+", r["Kernel_Name"])
        if not m or not m.group(0).startswith("k_icp"): continue
        res[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/per_pass.txt", "w") as fo:
    for k, c in sorted(res.items()):
        for name, v in sorted(c.items()):
            print(k, name, [round(x / 1e6, 2) for x in v[:40]], file=fo)
PY
rm -rf $OUT/p1
