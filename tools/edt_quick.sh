# EDT parity tests, then the EDT-only bench line.
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_edt.py -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/pytest_edt.log
timeout -k 10 200 python -u bench.py --no-extra --no-cpu --steps 200 --warmup 20 2>&1 | tee gpurun_out/bench_edt.log
