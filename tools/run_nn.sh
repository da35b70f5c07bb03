set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_nn.py tests/test_gpu_icp_step.py tests/test_gpu_host_classes.py -m gpu -q -x --timeout 300 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 python -u tools/nn_probe.py
