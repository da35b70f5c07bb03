"""Time of icpx_knn (k = 1, 5, 8) and icpx_nn on the bunny pair and on a 1 M synthetic surface (GPU)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_surface, rigid
from icp_amd.api import Context

def run(name, tgt, q):
    ctx = Context(0)
    ctx.set_target(tgt)
    qd = q   # host arrays: the copies to and from the device are part of every figure
    for k in (0, 1, 5, 8):
        f = (lambda: ctx.nn(qd)) if k == 0 else (lambda: ctx.knn(qd, k))
        f()
        t0 = time.perf_counter()
        for _ in range(5): f()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name}: {'nn' if k == 0 else 'knn k=%d' % k}: {dt * 1e3:.3f} ms for {q.shape[0]} queries ({q.shape[0] / dt:.3e}/s)", flush=True)
    ctx.close()

b0 = np.fromfile(os.path.join(ROOT, "tests/golden/bunny000.xyz.f32"), np.float32).reshape(-1, 3)
b45 = np.fromfile(os.path.join(ROOT, "tests/golden/bunny045.xyz.f32"), np.float32).reshape(-1, 3)
run("bunny000 -> bunny045", b45, b0.astype(np.float64))
tgt = synth_surface(1_000_000, 42)
T = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
src = synth_surface(1_000_000, 43, noise=1e-3).astype(np.float64)
run("1M surface, 10 degrees off", tgt, np.ascontiguousarray(src @ T[:3, :3].T + T[:3, 3]))
