"""Latency of one step of the device searches with a handful of searches running (threshold 0: they run until the queue
overflows or the steps run out). usage: python tools/search_latency.py [searches] [max steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["ICPX_SEARCH_STEPS"] = sys.argv[2] if len(sys.argv) > 2 else "128"
import numpy as np
from icp_amd.api import Context
from icp_amd.host import normalize_goicp
from conftest import load_scan
m, d, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
ctx = Context(0)
ctx.set_target(m); ctx.edt_build(300); ctx.goicp_set_data(d)
rng = np.random.default_rng(5)
Rs = []
for _ in range(ns):
    v = rng.uniform(-1, 1, 3); t = np.linalg.norm(v); k = v / t
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    Rs.append((np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K).astype(np.float32).reshape(9))
cube = np.array([-0.5, -0.5, -0.5, 1.0], np.float32)
for rep in range(3):
    t0 = time.perf_counter()
    opt, have, best, cubes, status, steps = ctx.bnb_search(Rs, [3] * ns, [1e9] * ns, [cube] * ns, 0.0)
    dt = time.perf_counter() - t0
    print(f"{ns} searches: {steps} steps in {dt * 1e3:.2f} ms = {dt / max(steps, 1) * 1e6:.1f} us per step; cubes {cubes.tolist()[:4]} status {status.tolist()[:4]}", flush=True)
ctx.close()
