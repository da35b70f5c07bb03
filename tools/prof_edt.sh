# Kernel trace + two SQ counter passes over the EDT-only bench (program directly after "--").
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 20 --warmup 5 > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 5 --warmup 2 > $OUT/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 5 --warmup 2 > $OUT/pmc2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu --steps 5 --warmup 2 > $OUT/pmc3.log 2>&1 || true
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof"
for d in ("pmc1", "pmc2", "pmc3"):
    fs = glob.glob(f"{out}/{d}/*/*_counter_collection.csv")
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "k_edt_pass" in r["Kernel_Name"]:
            k = "zy" if "pass_zy" in r["Kernel_Name"] else "x"
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        print(d, k, {c: round(sum(v) / len(v)) for c, v in agg[k].items()})
PY
