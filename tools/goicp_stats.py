"""Go-ICP on the bunny pair (the bench's case): host-driven inner searches against searches run on the device.
usage: python tools/goicp_stats.py [stride]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from icp_amd.host import GoIcp, normalize_goicp
from conftest import load_scan
m, d, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = np.ascontiguousarray(d[::stride])
keys = ("optError", "lb", "rot_pops", "spec_hits", "spec_misses")
ref = None
for name, kw in (("host, lookahead 1", dict(lookahead=1, device_search=False)), ("host, lookahead 0", dict(lookahead=0, tail_lookahead=False, device_search=False)),
                 ("device", dict(device_search=True))):
    for rep in range(2):
        go = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, speculate=255, **kw)
        t0 = time.perf_counter()
        r = go.registration()
        dt = time.perf_counter() - t0
        print(f"{name}: {dt:.3f} s, launches {r['launches']}, steps {r['search_steps']}, evals {r['inner_evals']}, " + ", ".join(f"{k} {r[k]}" for k in keys), flush=True)
    if ref is None: ref = r
    else:
        same = all(r[k] == ref[k] for k in keys) and np.array_equal(r["optMat"], ref["optMat"])
        print("   same decisions as the first:", same, flush=True)
