set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u bench.py 2>&1 | tee gpurun_out/bench.log | tail -1 | cut -c1-200
timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 2 --backend gloo --single-device --no-cpu --big-points 0 2>&1 | tee gpurun_out/bench_2rank_rehearsal.log | tail -1 | cut -c1-200
