"""Regenerates tests/golden/goicp_bun270_stride8.json with the CPU oracle (about 3.5 minutes on 8 cores)."""
import json, os
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import oracle as O
from conftest import load_scan
model, data = load_scan('bunny000').astype(np.float64), load_scan('bun270').astype(np.float64)
m = model - model.mean(0); d = data - data.mean(0)
s = 1.0 / max(np.abs(m).max(), np.abs(d).max())
m, d = (m * s).astype(np.float32), np.ascontiguousarray((d * s).astype(np.float32)[::8])
mn, cell = O.ldt_box(m, 300); occ,_ = O.ldt_splat(m, 300, mn, cell); sq = O.edt_meijster(occ)
t=time.time(); r = O.goicp(m, d, 1e-3, run_bnb=True, dist_mode=1, sq=sq, mn=mn, cell=cell, threads=8)
print('oracle goicp stride8', time.time()-t, {k:(float(v) if k!='optMat' else None) for k,v in r.items()}); print(r['optMat'])
gold = {"workload": "Go-ICP, model bunny000 (40256), data bun270[::8] (3963), isotropic normalisation, LDT 300^3, mse 1e-3, rotation search enabled",
        "generator": "tools/make_goicp_golden.py (oracle/oracle.cpp orc_goicp, serial best-first search, sequential float sums)",
        "optError": float(r["optError"]), "initError": float(r["initError"]), "icpError": float(r["icpError"]), "rot_pops": int(r["rot_pops"]),
        "inner_evals": int(r["inner_evals"]), "lb": float(r["lb"]), "optMat": r["optMat"].tolist()}
json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "goicp_bun270_stride8.json"), "w"), indent=1)
