set -e
mkdir -p gpurun_out
for a in 0 1 2 3; do
  echo "== ABLATE=$a" 
  ICPX_EDT_CAP_X=80 ICPX_EDT_ABLATE=$a timeout -k 10 120 python -u bench.py --no-extra --no-cpu --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms'])"
done
