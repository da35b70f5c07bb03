"""Go-ICP on the reference's bunny pair (model bun000 = bunny000, data bun270), Go-ICP normalisation, DT 300^3."""
import sys, time, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_amd.host import GoIcp
ap = argparse.ArgumentParser()
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--mse", type=float, default=1e-3)
ap.add_argument("--width", type=int, default=1)
ap.add_argument("--max-pops", type=int, default=0)
ap.add_argument("--lookahead", type=int, default=4)
ap.add_argument("--speculate", type=int, default=0)
ap.add_argument("--strict", action="store_true", help="one inner search after the other, sequential float sums (the reference's order)")
a = ap.parse_args()
def load(n): return np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", n + ".xyz.f32"), dtype="<f4").reshape(-1, 3)
model, data = load("bunny000").astype(np.float64), load("bun270").astype(np.float64)
m = model - model.mean(0); d = data - data.mean(0)
s = 1.0 / max(np.abs(m).max(), np.abs(d).max())
m, d = (m * s).astype(np.float32), np.ascontiguousarray((d * s).astype(np.float32)[::a.stride])
import threading
def _beat():
    while True:
        time.sleep(30); print("... running", round(time.perf_counter() - t0), "s", flush=True)
threading.Thread(target=_beat, daemon=True).start()
t0 = time.perf_counter()
go = GoIcp(m, d, mse=a.mse, dt_div=300).options(batched=not a.strict, parity_sums=a.strict, frontier_width=a.width, max_rot_pops=a.max_pops, lookahead=a.lookahead, speculate=a.speculate)
t1 = time.perf_counter()
r = go.registration()
t2 = time.perf_counter()
print(f"Nd={d.shape[0]} Nm={m.shape[0]} setup {t1-t0:.2f}s  search {t2-t1:.2f}s")
print({k: (float(v) if k in ('optError','initError','icpError','lb','sseThresh') else v) for k, v in r.items() if k not in ('optMat', 'resN3')})
print(np.array2string(r['optMat'], precision=6))
