# Counter passes over the bound / search kernels of one full-size Go-ICP run (tools/goicp_sweep.py, default setting run twice);
# prints per-kernel totals over all launches and writes them to gpurun_out/pmc_goicp/goicp_pmc.json.
# usage: pmc_goicp.sh "<counters of pass 1>" "<counters of pass 2>" ...
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_goicp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
n=0
for ctrs in "$@"; do
  n=$((n+1))
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$n -- python3 $GRAFT_REPO_ROOT/tools/goicp_sweep.py ${GOICP_SETTING:-4,15} > $OUT/p$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/p$n.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, os, collections, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_goicp"
res = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\bk_(bnb|search)\w+", r["Kernel_Name"])
        if not m: continue
        res[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[m.group(0)][r["Counter_Name"]] += 1
doc = {"_note": "totals over all launches of two full-size Go-ICP runs (bunny000 / bun270, 300^3); FETCH_SIZE in KiB as rocprofv3 reports it"}
for k, c in sorted(res.items()):
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:36s} {v:18.0f}   over {calls[k][name]} launches")
        doc.setdefault(k, {})[name] = v
    doc[k]["launches"] = max(calls[k].values())
json.dump(doc, open(out + "/goicp_pmc.json", "w"), indent=1, sort_keys=True)
PY
rm -rf $OUT/p*/
