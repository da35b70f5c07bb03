"""GPU: query-sharded ICP over two processes (src/SingleThreadIcp.cpp:307-316 is the loop: NN search, covariance, update).
Each rank holds half of the source cloud and the whole target; the 16 sums of every pass are added over the ranks, every
rank solves the same 3 x 3 problem. The final transform must be the one-rank transform (the sums are added in another
order, nothing else differs)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_RANK = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from icp_amd.host import SingleThreadIcp
from bench import synth_surface, rigid
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)                       # both ranks share the one card of the box
n = int(sys.argv[3])
tgt = synth_surface(n, 42)
Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
src = (synth_surface(n, 43, noise=1e-3).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
lo, hi = rank * n // world, (rank + 1) * n // world
calls = [0]
def allreduce(a):
    calls[0] += 1
    t = torch.from_numpy(a); dist.all_reduce(t)
icp = SingleThreadIcp(np.ascontiguousarray(src[lo:hi]), tgt, iters=int(sys.argv[4]), device=0, allreduce=allreduce, rank=rank, world=world)
r = icp.registration()
json.dump(dict(T=r["T"].tolist(), per=r["per_iter_T"].tolist(), mse=r["mse"].tolist(), calls=calls[0]), open(sys.argv[2] + f".{rank}", "w"))
dist.barrier(); dist.destroy_process_group()
"""


def test_icp_two_ranks_share_the_source_cloud(tmp_path):
    from icp_amd.host import SingleThreadIcp
    sys.path.insert(0, ROOT)
    from bench import synth_surface, rigid
    n, iters = 200_000, 12
    tgt = synth_surface(n, 42)
    Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
    src = (synth_surface(n, 43, noise=1e-3).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
    one = SingleThreadIcp(src, tgt, iters=iters).registration()
    script = tmp_path / "rank.py"
    script.write_text(_RANK)
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = str(tmp_path / "res.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), ROOT, out, str(n), str(iters)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(out + f".{k}")) for k in range(2)]
    assert res[0]["T"] == res[1]["T"] and res[0]["per"] == res[1]["per"]          # both ranks solved the same problem, bit for bit
    assert res[0]["calls"] >= iters                                             # one sum over the ranks per pass at least
    np.testing.assert_allclose(np.array(res[0]["T"]), one["T"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.array(res[0]["per"]), one["per_iter_T"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.array(res[0]["mse"]), one["mse"], rtol=1e-10)      # a sum of 2e5 squared distances added in another order
