set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_edt.py -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tail -15
for lw in 64 32; do
  echo "== LANES=$lw" 
  ICPX_EDT_LANES=$lw timeout -k 10 120 python -u bench.py --no-extra --no-cpu --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms'])"
done
for a in 1 2 3; do
  echo "== ABLATE=$a" 
  ICPX_EDT_ABLATE=$a timeout -k 10 120 python -u bench.py --no-extra --no-cpu --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms'])"
done
