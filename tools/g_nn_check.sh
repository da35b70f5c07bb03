set -e -o pipefail
timeout -k 10 900 python -u -m pytest tests/test_gpu_nn.py tests/test_gpu_icp_step.py tests/test_gpu_icp_multi.py tests/test_gpu_host_classes.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4
timeout -k 10 600 python -u bench.py --no-cpu --no-goicp 2>&1 | tail -1 > gpurun_out/bench_nn.json
