"""Wall time of the reference's config 1 (bunny000 -> bunny045, 30 iterations) through the host class (GPU)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icp_amd.host import SingleThreadIcp
src = np.fromfile(os.path.join(ROOT, "tests/golden/bunny000.xyz.f32"), np.float32).reshape(-1, 3)
tgt = np.fromfile(os.path.join(ROOT, "tests/golden/bunny045.xyz.f32"), np.float32).reshape(-1, 3)
icp = SingleThreadIcp(src, tgt, iters=30)
icp.registration()
for _ in range(3):
    t0 = time.perf_counter()
    r = icp.registration()
    dt = time.perf_counter() - t0
    print(f"30 iterations: {dt * 1e3:.2f} ms  ({dt / 30 * 1e6:.0f} us per iteration, {30 * src.shape[0] / dt:.3e} NN queries/s)  final nn mse {r['nn_mse'][-1]:.3e}")
