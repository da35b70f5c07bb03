set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread 2>&1 | tee gpurun_out/pytest_gpu.log
