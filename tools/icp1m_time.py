"""Wall time of every pass of the synthetic 1 M x 1 M ICP (10 degrees off at the start), through icpx_icp_step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icp_amd.api import Context
from icp_amd.host import kabsch_from_sums
from bench import synth_surface, rigid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tgt = synth_surface(n, 42)
Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
src = (synth_surface(n, 43, noise=1e-3).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
ctx = Context(0)
t0 = time.perf_counter(); ctx.set_target(torch.from_numpy(tgt).cuda()); print("set_target s", round(time.perf_counter() - t0, 4))
t0 = time.perf_counter(); ctx.set_source(torch.from_numpy(src).cuda()); print("set_source s", round(time.perf_counter() - t0, 4))
T = np.eye(4); ms = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    t0 = time.perf_counter(); s = ctx.icp_step(T); ms.append((time.perf_counter() - t0) * 1e3)
    T = kabsch_from_sums(s, n) @ T
print("ms per pass:", [round(x, 3) for x in ms])
print("total 30:", round(sum(ms[:30]), 2), "ms; last 10 mean", round(float(np.mean(ms[-10:])), 4), "ms; rms", np.sqrt(s[15] / n))
