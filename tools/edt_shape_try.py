"""EDT build time and a checksum of the result at the resolutions given on the command line (GPU); run under
different ICPX_EDT_ZY / ICPX_EDT_X settings to compare stack shapes."""
import sys, time, zlib
import numpy as np
sys.path.insert(0, ".")
from icp_amd.api import Context
pts = np.fromfile("tests/golden/bunny045.xyz.f32", np.float32).reshape(-1, 3)
ctx = Context(0)
ctx.set_target(pts)
for div in [int(a) for a in sys.argv[1:]] or [512]:
    ctx.edt_build(div)
    ctx.edt_build(div)
    crc = zlib.crc32(ctx.edt_read().tobytes())
    st = ctx.edt_rebuild_timed(8, 1)
    st = ctx.edt_rebuild_timed(16, 1)
    print(f"div {div:5d}: crc {crc:08x} kernels (ms) splat {st[0]:.3f} census {st[1]:.3f} zy {st[2]:.3f} x {st[3]:.3f}", flush=True)
