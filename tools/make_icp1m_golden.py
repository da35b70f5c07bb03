"""CPU: the oracle's kd-tree ICP (SingleThreadIcp restated, nn_mode 0: the reference's kd-tree with its first-visited
tie rule, src/kdtree.cpp:107-150, src/SingleThreadIcp.cpp:293-327) on the seeded synthetic 1 M x 1 M pair of the
benchmark, 30 iterations from the identity. Writes tests/golden/icp_1m_30it.json: the final transform, every
iteration's accumulated transform and SHA-256 of it, and the residual trace. Takes a few minutes on eight cores.
    python tools/make_icp1m_golden.py [points]"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import synth_surface, rigid
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tgt = synth_surface(n, 42)
Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
src = (synth_surface(n, 43, noise=1e-3).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
t0 = time.perf_counter()
r = O.icp(src, tgt, 30, nn_mode=0, threads=os.cpu_count())
print("oracle ICP", round(time.perf_counter() - t0, 1), "s")
# per_iter_T holds the update of every iteration; the accumulated transform after k + 1 of them is what the product's
# trace and the final transform are compared with
acc = np.eye(4); accs = []
for k in range(30):
    acc = r["per_iter_T"][k] @ acc
    accs.append(acc.copy())
assert np.allclose(accs[-1], r["T"], atol=1e-12)
doc = {"points": n, "iterations": 30, "generator": "bench.synth_surface seeds 42 (target) / 43 (source, noise 1e-3), rigid([1,1,1], 10 deg, [0.02,-0.01,0.015])",
       "src_sha256": hashlib.sha256(src.tobytes()).hexdigest(), "tgt_sha256": hashlib.sha256(tgt.tobytes()).hexdigest(),
       "T": r["T"].tolist(), "T_after_iteration": [a.tolist() for a in accs],
       "T_sha256_after_iteration": [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in accs],
       "mse_logged": r["mse"].tolist()}
out = os.path.join(ROOT, "tests", "golden", "icp_1m_30it.json" if n == 1_000_000 else f"icp_{n}_30it.json")
json.dump(doc, open(out, "w"))
print("wrote", out, "rot error vs generator (deg)", float(np.rad2deg(np.arccos(np.clip((np.trace(r["T"][:3, :3] @ Tm[:3, :3]) - 1) / 2, -1, 1)))))
