"""Wall time per EDT rebuild with and without per-kernel events (GPU)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from icp_amd.api import Context
pts = np.fromfile("tests/golden/bunny045.xyz.f32", np.float32).reshape(-1, 3)
ctx = Context(0)
ctx.set_target(pts)
ctx.edt_build(300)
K = 300
for _ in range(20):
    ctx.edt_rebuild_async()
ctx.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.edt_rebuild_async()
    ctx.sync()
    plain = (time.perf_counter() - t0) / K * 1e6
    t0 = time.perf_counter()
    st = ctx.edt_rebuild_timed(K, 1)
    timed = (time.perf_counter() - t0) / K * 1e6
    t0 = time.perf_counter()
    st8 = ctx.edt_rebuild_timed(K, 8)
    timed8 = (time.perf_counter() - t0) / K * 1e6
    print(f"plain {plain:.1f} us/build   events on every build {timed:.1f}   on every 8th {timed8:.1f}   kernel sum {st.sum() * 1e3:.1f} / {st8.sum() * 1e3:.1f} us")
