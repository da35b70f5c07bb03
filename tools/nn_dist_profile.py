"""Per pass of the synthetic 1 M x 1 M ICP: how far the queries are from their nearest target point (in cells of the
search grid) and how far the update moves them. Input for sizing the far-field kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icp_amd.api import Context
from icp_amd.host import kabsch_from_sums
from bench import synth_surface, rigid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cells = float(sys.argv[2]) if len(sys.argv) > 2 else 254.0
tgt = synth_surface(n, 42)
Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
src = (synth_surface(n, 43, noise=1e-3).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
h = float((tgt.max(0) - tgt.min(0)).max()) / cells
print("cell size assumed", h)
ctx = Context(0)
ctx.set_target(torch.from_numpy(tgt).cuda()); ctx.set_source(torch.from_numpy(src).cuda())
T = np.eye(4)
s64 = src.astype(np.float64); t64 = tgt.astype(np.float64)
prev = None
for it in range(30):
    s, idx = ctx.icp_step(T, want_idx=True)
    p = s64 @ T[:3, :3].T + T[:3, 3]
    d = np.linalg.norm(p - t64[idx], axis=1) / h
    mv = np.linalg.norm(p - prev, axis=1) / h if prev is not None else np.zeros(1)
    same = float((idx == pidx).mean()) if prev is not None else 0.0
    prev, pidx = p, idx
    q = np.quantile(d, [0.1, 0.5, 0.9, 0.99, 1.0])
    print(f"pass {it:2d}: NN distance in cells p10 {q[0]:.2f} p50 {q[1]:.2f} p90 {q[2]:.2f} p99 {q[3]:.2f} max {q[4]:.2f}; "
          f"share >1: {(d > 1).mean():.3f} >2: {(d > 2).mean():.3f} >4: {(d > 4).mean():.3f} >8: {(d > 8).mean():.3f} >16: {(d > 16).mean():.3f}; "
          f"moved since last pass: median {np.median(mv):.2f} max {mv.max():.2f}; same pair {same:.3f}", flush=True)
    T = kabsch_from_sums(s, n) @ T
