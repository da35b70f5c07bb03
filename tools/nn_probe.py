import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icp_amd.api import Context
from bench import synth_surface, rigid
n = 1_000_000
tgt = synth_surface(n, 42)
ctx = Context(0)
t0=time.perf_counter(); ctx.set_target(torch.from_numpy(tgt).cuda()); print('set_target s', time.perf_counter()-t0)
src0 = synth_surface(n, 43, noise=1e-3)
for name, shift in (('near', 0.0), ('2 cells', 0.006), ('6 cells', 0.02), ('15 cells', 0.05), ('rot10', None)):
    if shift is None:
        Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015]); src = (src0.astype(np.float64) @ Tm[:3,:3].T + Tm[:3,3]).astype(np.float32)
    else:
        nrm = src0 / np.linalg.norm(src0, axis=1, keepdims=True); src = (src0 + shift * nrm).astype(np.float32)
    ctx.set_source(torch.from_numpy(src).cuda())
    ctx.icp_step(np.eye(4))
    t0 = time.perf_counter()
    for _ in range(3): s = ctx.icp_step(np.eye(4))
    dt = (time.perf_counter() - t0) / 3
    print(f'{name:10s} {dt*1e3:8.2f} ms/iter  {n/dt/1e6:8.1f} Mq/s  rms NN dist {np.sqrt(s[15]/n):.4f}')
