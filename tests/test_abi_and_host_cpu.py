"""CPU: the C-ABI library loads and exports every symbol include/icpx.h declares; argument checking that
needs no GPU; the host-side algebra; sharded-sum collectives over gloo (world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "icpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icpx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from icp_amd import lib
    L = lib.icpx()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libicpx.so does not export {n}"
        assert n in lib.ICPX_SYMBOLS, f"icp_amd.lib has no binding for {n}"
    assert sorted(lib.ICPX_SYMBOLS) == names
    # struct layout of icpx_bnb_item: 9 + 4 floats + int32
    assert C.sizeof(lib.BnbItem) == 14 * 4


def test_header_compiles_as_c():
    src = '#include "icpx.h"\nint main(void){ icpx_bnb_item it; (void)it; return ICPX_NSUMS == 16 ? 0 : 1; }\n'
    out = os.path.join(ROOT, "tests", "_hdr_check")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", out],
                   input=src.encode(), check=True)
    assert subprocess.run([out]).returncode == 0
    os.remove(out)


def test_create_fails_loudly_without_gpu_or_bad_device():
    import torch
    from icp_amd import lib
    L = lib.icpx()
    h = C.c_void_p()
    if not torch.cuda.is_available():
        assert L.icpx_create(0, C.byref(h)) == -3
        assert b"no HIP device" in L.icpx_last_error(None)
        from icp_amd.api import Context
        from icp_amd import IcpxError
        with pytest.raises(IcpxError):
            Context(0)
    assert L.icpx_create(0, None) == -1
    assert L.icpx_set_target(None, None, 0, 12, 0) == -1           # NULL context is rejected, not dereferenced
    assert L.icpx_nn(None, None, 0, 0, None, None) == -1
    assert L.icpx_device(None) == -1


def test_host_svd_and_kabsch():
    from icp_amd import host
    rng = np.random.default_rng(1)
    for k in range(500):
        H = rng.normal(size=(3, 3)) * 10 ** rng.uniform(-6, 2)
        if k % 5 == 0:
            H[:, 2] = 0.5 * H[:, 0]            # rank 2
        if k % 11 == 0:
            H[:, 1] = 2.0 * H[:, 0]; H[:, 2] = -H[:, 0]   # rank 1
        U, s, V = host.svd3(H)
        assert np.abs(U @ np.diag(s) @ V.T - H).max() <= 1e-13 * max(np.abs(H).max(), 1e-300)
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-12 and np.abs(V.T @ V - np.eye(3)).max() < 1e-12
        assert s[0] >= s[1] >= s[2] >= 0
        R = host.kabsch_rotation(H)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
    # reflection case: the optimal orthogonal map is a reflection, the result must still be a rotation
    Hr = np.diag([3.0, 2.0, -1.0])
    R = host.kabsch_rotation(Hr)
    assert abs(np.linalg.det(R) - 1) < 1e-12 and np.allclose(R, np.eye(3))


def test_kabsch_from_sums_recovers_a_rigid_motion():
    from icp_amd import host
    from conftest import rigid
    rng = np.random.default_rng(2)
    p = rng.normal(size=(1000, 3))
    T = rigid([0.2, -0.4, 1.0], 33.0, [0.5, -1.0, 0.25])
    q = p @ T[:3, :3].T + T[:3, 3]
    s = np.concatenate([p.sum(0), q.sum(0), (p.T @ q).reshape(9), [0.0]])
    assert np.abs(host.kabsch_from_sums(s, 1000) - T).max() < 1e-12


def _write_ply(path, xyz, binary=True, nan_rows=()):
    xyz = np.array(xyz, np.float32)
    for r in nan_rows:
        xyz[r, 1] = np.nan
    n = xyz.shape[0]
    hdr = ("ply\nformat %s 1.0\ncomment test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property float nx\nproperty float ny\nproperty float nz\nproperty int flags\nelement face 0\n"
           "property list uchar int vertex_indices\nend_header\n") % ("binary_little_endian" if binary else "ascii", n)
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            rec = np.zeros(n, dtype=[("p", "<f4", 3), ("n", "<f4", 3), ("flags", "<i4")])
            rec["p"] = xyz
            f.write(rec.tobytes())
        else:
            for p in xyz:
                f.write(("%r %r %r 0 0 1 0\n" % (float(p[0]), float(p[1]), float(p[2]))).encode())


def test_ply_reader_and_normalisation(tmp_path, bunny045):
    """loadFile of the host library (PLY vertex reader + NaN filter, src/main.cpp:35-51) on the 28-byte
    vertex layout of the reference's scans, binary and ascii, and the Go-ICP normalisation."""
    from icp_amd import host, IcpxError
    pts = bunny045[:5000]
    b, a = str(tmp_path / "b.ply"), str(tmp_path / "a.ply")
    _write_ply(b, pts, binary=True, nan_rows=(3, 77))
    _write_ply(a, pts[:200], binary=False)
    got = host.load_ply(b)
    keep = np.ones(len(pts), bool); keep[[3, 77]] = False
    assert np.array_equal(got, pts[keep])
    assert np.array_equal(host.load_ply(a), pts[:200])
    with pytest.raises(IcpxError):
        host.load_ply(str(tmp_path / "missing.ply"))
    (tmp_path / "bad.ply").write_text("not a ply\n")
    with pytest.raises(IcpxError):
        host.load_ply(str(tmp_path / "bad.ply"))
    m, d, s = host.normalize_goicp(pts, pts[::3] * 2.0 + 1.0)
    assert abs(m.mean(0)).max() < 1e-6 and abs(d.mean(0)).max() < 1e-6
    assert abs(max(np.abs(m).max(), np.abs(d).max()) - 1.0) < 1e-6
    ref = (pts.astype(np.float64) - pts.astype(np.float64).mean(0)) * s
    assert np.abs(m - ref).max() < 1e-6


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from icp_amd import host
from oracle import oracle as O
from conftest import load_scan
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
src, tgt = load_scan("bunny000")[::16], load_scan("bunny045")[::4]
tgt = np.ascontiguousarray(tgt)
def allreduce(a):
    t = torch.from_numpy(a); dist.all_reduce(t)
cb = host.make_allreduce(allreduce)       # the callback type the C++ host classes take
def sums_of(points):
    p = points.astype(np.float64)
    idx, d2 = O.nn_brute(p, tgt)
    q = tgt[idx].astype(np.float64)
    return np.concatenate([p.sum(0), q.sum(0), (p.T @ q).reshape(9), [d2.sum()]])
n = src.shape[0]
lo, hi = rank * n // world, (rank + 1) * n // world
mine = sums_of(src[lo:hi])
import ctypes
cb(mine.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 16)      # in-place sum over the ranks
full = sums_of(src)
assert np.allclose(mine, full, rtol=1e-12, atol=1e-12), (mine, full)
T_sharded = host.kabsch_from_sums(mine, n); T_full = host.kabsch_from_sums(full, n)
assert np.abs(T_sharded - T_full).max() < 1e-12
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_sums_allreduce_gloo_world2(tmp_path):
    """N > 1 ICP path: every rank reduces its slice of the source to 16 sums, one all-reduce, identical
    transform on every rank. Runs on CPU with the gloo backend; the per-slice sums come from the oracle."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    with socket.socket() as sock:      # a port nobody holds right now (a fixed one may still be in TIME_WAIT from an earlier run)
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), ROOT], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
