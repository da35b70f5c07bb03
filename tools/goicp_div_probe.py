import sys, time, numpy as np
sys.path.insert(0, '.')
from bench import load_scan
from icp_amd.host import GoIcp, normalize_goicp
gm, gd, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
for div in (300, 200, 150):
    go = GoIcp(gm, gd, mse=1e-3, dt_div=div, device=0)
    go.options(batched=True, frontier_width=1, lookahead=4, speculate=15)
    t0 = time.perf_counter(); r = go.registration(); el = time.perf_counter() - t0
    print(div, round(el, 3), 's', r["optError"], r["rot_pops"], r["inner_evals"], r["launches"], 'us per 1000 cube evals', round(el / r["inner_evals"] * 1e9, 1))
    del go
