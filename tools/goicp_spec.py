"""Go-ICP on the bunny pair with the inner searches on the device: how many rotation cubes are searched ahead per step.
usage: python tools/goicp_spec.py 255 511 1023 ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from icp_amd.host import GoIcp, normalize_goicp
from conftest import load_scan
m, d, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
for spec in [int(a) for a in sys.argv[1:]] or [255]:
    for rep in range(2):
        go = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, speculate=spec, device_search=True)
        t0 = time.perf_counter()
        r = go.registration()
        print(f"speculate {spec}: {time.perf_counter() - t0:.3f} s, steps {r['search_steps']}, evals {r['inner_evals']}, optError {r['optError']}, rot_pops {r['rot_pops']}, "
              f"hits {r['spec_hits']} misses {r['spec_misses']}", flush=True)
