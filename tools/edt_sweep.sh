# Sweep of the EDT envelope-pass knobs: EDT_SWEEP = list of "groups,levels,chunks" (applied to both passes).
mkdir -p gpurun_out
SWEEP=${EDT_SWEEP:-4,14,64 4,12,90 4,10,110 2,28,64 2,24,120 3,18,64}
for cfg in $SWEEP; do
  echo "tune=$cfg"
  ICPX_EDT_ZY=$cfg ICPX_EDT_X=$cfg timeout -k 10 120 python -u bench.py --no-extra --no-cpu --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k: round(v*1000,1) for k,v in d['roofline']['stage_ms'].items()})" || exit 1
done 2>&1 | tee gpurun_out/edt_sweep.log
