set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/pytest_gpu.log
timeout -k 10 600 python -u bench.py 2>&1 | tee gpurun_out/bench.log
