set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_edt.py -x -q --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/t_edt.log
timeout -k 10 200 python -u tools/nn_dist_profile.py 2>&1 | tee gpurun_out/nn_dist_profile.log
timeout -k 10 100 python -u - <<'PY' 2>&1 | tee gpurun_out/first_build.log
import numpy as np, torch, time, sys
sys.path.insert(0, '.')
from icp_amd.api import Context
from bench import load_scan
tgt = torch.from_numpy(load_scan("bunny045")).cuda()
for rep in range(3):
    ctx = Context(0); ctx.set_target(tgt); ctx.edt_set_timing(True)
    t0 = time.perf_counter(); ctx.edt_build(300); w = time.perf_counter() - t0
    ms, att = ctx.edt_last_build()
    t0 = time.perf_counter(); ctx.edt_build(300); w2 = time.perf_counter() - t0
    ms2, att2 = ctx.edt_last_build()
    print("fresh ctx: wall", round(w*1e3,3), "ms stages", ms.round(4), "sum", ms.sum().round(4), "attempts", att, "| second build: wall", round(w2*1e3,3), ms2.round(4), att2)
    ctx.close()
PY
