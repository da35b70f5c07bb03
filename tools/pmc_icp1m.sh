# Counter passes over the kernels of the 1 M x 1 M ICP (tools/icp1m_time.py); prints per-kernel values of the first
# pass (10 degrees off) and averages of the last eight (converged), and writes them to gpurun_out/pmc_1m/icp_pmc.json.
# usage: pmc_icp1m.sh "<counters of pass 1>" "<counters of pass 2>" ...
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_1m
rm -rf $OUT; mkdir -p $OUT
cd /tmp
n=0
for ctrs in "$@"; do
  n=$((n+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$n -- python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py > $OUT/p$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/p$n.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, os, collections, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_1m"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\bk_\w+", r["Kernel_Name"])
        if not m or not (m.group(0).startswith("k_icp") or m.group(0) == "k_reduce_sums"): continue
        res[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
doc = {"first_pass": {}, "converged_pass": {},
       "_note": "per launch; 1 M x 1 M synthetic ICP, 40 passes from a 10 degree offset; converged_pass = mean of the last 8 passes; "
                "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, each from a --pmc run of its own"}
for k, c in sorted(res.items()):
    print(k)
    for name, v in sorted(c.items()):
        tail = v[-8:]
        print(f"   {name:28s} first {v[0]:14.0f}   mean of last {len(tail)}: {sum(tail)/len(tail):14.0f}")
        doc["first_pass"].setdefault(k, {})[name] = v[0]
        doc["converged_pass"].setdefault(k, {})[name] = sum(tail) / len(tail)
json.dump(doc, open(out + "/icp_pmc.json", "w"), indent=1, sort_keys=True)
PY
rm -rf $OUT/p*/
