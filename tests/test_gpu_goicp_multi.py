"""GPU: the pieces of Go-ICP that several GPUs share — bound batches evaluated in shares on the device and summed there,
rotation cubes searched ahead of time — must leave every decision of the search as it is with one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _pair(bunny000, bun270, stride):
    from icp_amd.host import normalize_goicp
    m, d, _ = normalize_goicp(bunny000, bun270)
    return m, np.ascontiguousarray(d[::stride])


def test_bnb_eval_async_shares_sum_to_the_batch(gpu_ctx, oracle, bunny000, bun270):
    """icpx_bnb_eval_async: two shares of one batch, each written with zeros elsewhere, add up (exactly: x + 0) to the
    bounds icpx_bnb_eval returns; icpx_download brings them to the host; the grid pointer is the grid icpx_edt_read copies."""
    import torch
    model, data = _pair(bunny000, bun270, 8)
    gpu_ctx.set_target(model)
    gpu_ctx.edt_build(128)
    gpu_ctx.goicp_set_data(data)
    rng = np.random.default_rng(4)
    n = 37
    Rs = [oracle.goicp_rotmat(rng.uniform(-1.5, 1.5, 3).astype(np.float32)).reshape(9) for _ in range(n)]
    cubes = np.c_[rng.uniform(-0.5, 0.25, (n, 3)), rng.choice([0.5, 0.25, 0.125], n)].astype(np.float32)
    items = gpu_ctx.make_items(Rs, cubes, rng.integers(-1, 20, n))
    ub, lb = gpu_ctx.bnb_eval(items)
    a = torch.full((2 * n,), 7.0, dtype=torch.float64, device="cuda"); b = torch.full((2 * n,), 7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.bnb_eval_async(items, a, 0, 15)
    gpu_ctx.bnb_eval_async(items, b, 15, n - 15)
    ha, hb = gpu_ctx.download(a), gpu_ctx.download(b)
    assert np.all(ha[15:n] == 0) and np.all(ha[n + 15:] == 0) and np.all(hb[:15] == 0) and np.all(hb[n:n + 15] == 0)
    assert np.array_equal((ha + hb)[:n], ub) and np.array_equal((ha + hb)[n:], lb)
    whole = torch.empty(2 * n, dtype=torch.float64, device="cuda")
    gpu_ctx.bnb_eval_async(items, whole)
    assert np.array_equal(gpu_ctx.download(whole), np.concatenate([ub, lb]))
    empty = torch.full((2 * n,), 3.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.bnb_eval_async(items, empty, 5, 0)                    # a rank with nothing to do still clears its buffer
    assert np.all(gpu_ctx.download(empty) == 0)
    # icpx_edt_grid_ptr: the device grid in place
    ptr = gpu_ctx.edt_grid_ptr()
    sq = gpu_ctx.edt_read()
    import ctypes as C
    view = torch.empty(128 ** 3, dtype=torch.int32, device="cuda")
    assert C.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(view.data_ptr()), C.c_void_p(ptr), C.c_size_t(4 * 128 ** 3), 3) == 0
    assert np.array_equal(view.cpu().numpy().reshape(128, 128, 128), sq)


def test_goicp_speculation_keeps_every_decision(bunny000, bun270):
    """Rotation cubes searched ahead of time: the same cubes popped, bit-identical optimum, bound and pose, fewer launches."""
    from icp_amd.host import GoIcp
    m, d = _pair(bunny000, bun270, 8)
    base = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=4, speculate=0).registration()
    for k in (1, 4):
        r = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=4, speculate=k).registration()
        assert r["rot_pops"] == base["rot_pops"] and float(r["optError"]) == float(base["optError"]) and r["lb"] == base["lb"]
        assert np.array_equal(r["optMat"], base["optMat"])
        assert r["launches"] < base["launches"] and r["spec_hits"] > 0


_RANK = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from icp_amd.host import GoIcp, normalize_goicp
from conftest import load_scan
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)                       # both ranks share the one card of the box
m, d, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
d = np.ascontiguousarray(d[::8])
buf = torch.zeros(1 << 14, dtype=torch.float64, device="cuda")
calls = [0]
def allreduce_dev(t):
    calls[0] += 1
    # Two ranks on ONE card cannot use RCCL (one communicator rank per GPU), and gloo's own CUDA path hangs on this
    # stack, so this stand-in for the caller's collective stages through the host; what is under test is the library
    # side: shares evaluated on the device, a device buffer handed to the collective, one download afterwards.
    h = t.cpu(); dist.all_reduce(h); t.copy_(h)
go = GoIcp(m, d, mse=1e-3, dt_div=300, device=0, rank=rank, world=world, allreduce_dev=allreduce_dev, dev_buffer=buf)
go.options(batched=True, lookahead=4, speculate=int(sys.argv[3]), shard_cubes=sys.argv[4] == "cubes")
r = go.registration()
out = dict(optError=float(r["optError"]), lb=float(r["lb"]), rot_pops=r["rot_pops"], optMat=r["optMat"].tolist(), launches=r["launches"],
           collectives=calls[0], counted=r["collectives"], initError=float(r["initError"]), icpError=float(r["icpError"]),
           spec_hits=r["spec_hits"], inner_evals=r["inner_evals"])
json.dump(out, open(sys.argv[2] + f".{rank}", "w"))
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("speculate", [0, 3])
def test_goicp_two_ranks_share_every_batch_on_the_device(tmp_path, bunny000, bun270, speculate):
    """Two processes on cuda:0 (gloo between them): each evaluates its share of every bound batch with
    icpx_bnb_eval_async, the shares are summed by an all-reduce of the device buffer on the context's stream, and both
    ranks take the decisions of the one-rank run: same cubes, bit-identical optimum, bound and pose."""
    from icp_amd.host import GoIcp
    m, d = _pair(bunny000, bun270, 8)
    # one rank: the searches advance as two groups whose batches alternate on the GPU (as with several ranks), or as one group
    # (the searches driven by the host: what the ranks do when they share every batch)
    piped = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=4, speculate=speculate, device_search=False).registration()
    one = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=4, speculate=speculate, pipeline=False, device_search=False).registration()
    for k in ("optError", "lb", "rot_pops", "spec_hits"):     # (how far the last running searches look ahead depends on the group: not inner_evals)
        assert piped[k] == one[k], k
    assert np.array_equal(piped["optMat"], one["optMat"]) and piped["launches"] > one["launches"]
    script = tmp_path / "rank.py"
    script.write_text(_RANK)
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = str(tmp_path / "res.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), ROOT, out, str(speculate), "launches"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(out + f".{k}")) for k in range(2)]
    for x in res:
        assert x["optError"] == float(one["optError"]) and x["lb"] == float(one["lb"]) and x["rot_pops"] == one["rot_pops"]
        assert np.array_equal(np.array(x["optMat"]), one["optMat"])
        assert x["launches"] == piped["launches"] and x["collectives"] >= x["launches"]   # the ranks advance in two groups too
    assert res[0] == res[1]


@pytest.mark.parametrize("speculate", [3, 127])
def test_goicp_two_ranks_deal_out_the_rotation_cubes(tmp_path, bunny000, bun270, speculate):
    """Two processes on cuda:0: the rotation cubes of every step (the popped one and those searched ahead with it) are dealt
    out to the ranks, each rank runs its cubes' inner searches to the end on its own without communication, and one sum
    over the ranks per step spreads the results (src/GoIcp.cpp:203-344 is the loop whose cubes are shared out). Same
    cubes popped, bit-identical optimum, bound and pose as one rank; with 127 cubes searched ahead a collective per 32
    popped cubes at most, where sharing every launch needs one per launch."""
    from icp_amd.host import GoIcp
    m, d = _pair(bunny000, bun270, 8)
    one = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=4, speculate=speculate).registration()
    script = tmp_path / "rank.py"
    script.write_text(_RANK)
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = str(tmp_path / "res.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), ROOT, out, str(speculate), "cubes"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(out + f".{k}")) for k in range(2)]
    for x in res:
        assert x["optError"] == float(one["optError"]) and x["lb"] == float(one["lb"]) and x["rot_pops"] == one["rot_pops"]
        assert np.array_equal(np.array(x["optMat"]), one["optMat"]) and x["spec_hits"] == one["spec_hits"]
        assert x["collectives"] == x["counted"]                     # every sum over the ranks went through the caller's collective
        assert x["collectives"] <= one["rot_pops"] - one["spec_hits"] + 1     # one per step that had a search to run, and the closing one
        if speculate == 127:
            assert x["collectives"] <= one["rot_pops"] // 32
    assert res[0] == res[1]
