import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icp_amd.api import Context
from bench import synth_surface, rigid
n = 1_000_000
tgt = synth_surface(n, 42)
src = synth_surface(n, 43, noise=1e-3).astype(np.float64)
Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015]); src = (src @ Tm[:3,:3].T + Tm[:3,3]).astype(np.float32)
full = Context(0); full.set_target(tgt); full.set_source(src)
a = Context(0); a.set_target(tgt); a.set_source(np.ascontiguousarray(src[:n//2]))
b = Context(0); b.set_target(tgt); b.set_source(np.ascontiguousarray(src[n//2:]))
T = np.eye(4)
sf, idx_f = full.icp_step(T, want_idx=True); sa = a.icp_step(T); sb = b.icp_step(T)
print('full vs halves rel diff', np.abs(sf - (sa + sb)).max() / np.abs(sf).max())
idx, d2 = full.nn(src.astype(np.float64))
print('idx equal', np.array_equal(idx, idx_f))
p = src.astype(np.float64); q = tgt[idx].astype(np.float64)
ref = np.concatenate([p.sum(0), q.sum(0), (p.T @ q).reshape(9), [d2.sum()]])
print('sums vs nn-derived rel diff', np.abs(sf - ref).max() / np.abs(ref).max())
# async path on torch stream
sums = torch.zeros(16, dtype=torch.float64, device='cuda')
full.set_stream(torch.cuda.current_stream().cuda_stream)
full.icp_step_async(T, sums); print('async rel diff', np.abs(sums.cpu().numpy() - sf).max() / np.abs(sf).max())
from icp_amd.host import kabsch_from_sums
Tc = np.eye(4)
for it in range(30):
    full.icp_step_async(Tc, sums); s = sums.cpu().numpy(); Tc = kabsch_from_sums(s, n) @ Tc
    if it % 5 == 0 or it == 29:
        ang = np.rad2deg(np.arccos(np.clip((np.trace(Tc[:3,:3] @ Tm[:3,:3]) - 1) / 2, -1, 1))); print(it, 'mse', s[15]/n, 'rot err', ang)
