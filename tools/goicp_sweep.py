"""Go-ICP on the full bunny pair for several (lookahead, speculate) settings in one process: time, launches, optimum."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_amd.host import GoIcp, normalize_goicp
def load(n): return np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", n + ".xyz.f32"), dtype="<f4").reshape(-1, 3)
m, d, _ = normalize_goicp(load("bunny000"), load("bun270"))
go = GoIcp(m, d, mse=1e-3, dt_div=300)
settings = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(4, 15)]   # lookahead,speculate[,pipeline]
for la, sp, *rest in [settings[0]] + settings:
    go.options(batched=True, frontier_width=1, lookahead=la, speculate=sp, pipeline=bool(rest[0]) if rest else True)
    t0 = time.perf_counter(); r = go.registration(); el = time.perf_counter() - t0
    print(f"lookahead {la:3d} speculate {sp:4d} pipeline {int(bool(rest[0]) if rest else True)}: {el:.3f} s, {r['launches']} launches, {r['inner_evals']} evals, opt {float(r['optError'])!r} lb {float(r['lb'])!r} cubes {r['rot_pops']} used {r['spec_hits']} discarded {r['spec_misses']}", flush=True)
