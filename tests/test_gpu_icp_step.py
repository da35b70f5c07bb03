"""GPU parity: fused transform + NN + covariance sums through the C ABI against the oracle."""
import numpy as np
import pytest

from conftest import rigid

pytestmark = pytest.mark.gpu


def _sums_ref(O, src, tgt, T):
    p = src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    idx, d2 = O.nn_brute(np.ascontiguousarray(p), tgt)
    q = tgt[idx].astype(np.float64)
    s = np.empty(16)
    s[0:3] = p.sum(0); s[3:6] = q.sum(0); s[6:15] = (p.T @ q).reshape(9); s[15] = d2.sum()
    return s, idx


def test_icp_step_sums_and_idx(gpu_ctx, oracle, bunny000, bunny045):
    gpu_ctx.set_target(bunny045)
    gpu_ctx.set_source(bunny000)
    for T in (np.eye(4), rigid([0.2, -1, 0.4], 8.0, [0.003, 0.001, -0.002])):
        sums, idx = gpu_ctx.icp_step(T, want_idx=True)
        ref, ridx = _sums_ref(oracle, bunny000, bunny045, T)
        assert np.array_equal(idx, ridx)
        np.testing.assert_allclose(sums, ref, rtol=1e-12, atol=1e-12)


def test_icp_step_async_on_torch_stream(gpu_ctx, oracle, bunny000, bunny045):
    """The enqueue-only entry point on torch's current stream (the default stream and a side stream):
    the device-resident sums are ordered with torch work on that stream."""
    import torch
    gpu_ctx.set_target(bunny045)
    gpu_ctx.set_source(bunny000)
    T = rigid([1, 0.5, -0.3], 5.0, [0.002, -0.001, 0.003])
    ref = gpu_ctx.icp_step(T)
    try:
        for stream in (torch.cuda.current_stream(), torch.cuda.Stream()):
            with torch.cuda.stream(stream):
                gpu_ctx.set_stream(stream.cuda_stream)
                sums = torch.zeros(16, dtype=torch.float64, device="cuda")
                gpu_ctx.icp_step_async(T, sums)
                doubled = (sums * 2).cpu().numpy()          # torch op queued behind the kernels on the same stream
            assert np.array_equal(doubled, 2 * ref)
    finally:
        gpu_ctx.reset_stream()


def test_icp_step_trimmed(gpu_ctx, oracle, bunny000, bunny045):
    """Sums over the `keep` closest pairs (ICP3D's trimming, src/jly_icp3d.cpp:236-255); keep == n equals the plain step."""
    gpu_ctx.set_target(bunny045)
    gpu_ctx.set_source(bunny000)
    T = rigid([0.2, -1, 0.4], 8.0, [0.003, 0.001, -0.002])
    n = bunny000.shape[0]
    p = bunny000.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    idx, d2 = oracle.nn_brute(np.ascontiguousarray(p), bunny045)
    order = np.argsort(d2, kind="stable")
    for keep in (n, int(n * 0.9), 1000, 1, 0):
        sel = order[:keep]
        q = bunny045[idx[sel]].astype(np.float64)
        ref = np.empty(16)
        ref[0:3] = p[sel].sum(0); ref[3:6] = q.sum(0); ref[6:15] = (p[sel].T @ q).reshape(9); ref[15] = d2[sel].sum()
        got = gpu_ctx.icp_step_trimmed(T, keep)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gpu_ctx.icp_step_trimmed(T, n), gpu_ctx.icp_step(T), rtol=1e-13, atol=1e-13)
    from icp_amd import IcpxError
    with pytest.raises(IcpxError):
        gpu_ctx.icp_step_trimmed(T, n + 1)


def _transform(src, T):
    """T * [src; 1] in double with the kernel's operation order: products added to a zero accumulator, k = 0..3."""
    s = src.astype(np.float64)
    out = np.empty_like(s)
    for r in range(3):
        out[:, r] = ((0.0 + T[r, 0] * s[:, 0]) + T[r, 1] * s[:, 1]) + T[r, 2] * s[:, 2] + T[r, 3]
    return out


def _check_step(ctx, O, src, tgt, T, rng, nsample):
    sums, idx = ctx.icp_step(T, want_idx=True)
    p = _transform(src, T)
    pick = rng.choice(src.shape[0], nsample, replace=False)
    ridx, rd2 = O.nn_brute(np.ascontiguousarray(p[pick]), tgt)
    assert np.array_equal(idx[pick], ridx)                      # exact pairs, lowest index among equals
    assert idx.min() >= 0 and idx.max() < tgt.shape[0]
    q = tgt[idx].astype(np.float64)
    d = p - q
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    assert np.array_equal(d2[pick], rd2)
    want = np.concatenate([p.sum(0), q.sum(0), (p.T @ q).reshape(9), [d2.sum()]])
    np.testing.assert_allclose(sums, want, rtol=1e-12, atol=1e-12)
    return sums


def test_icp_step_million_by_million(oracle):
    """The fused pass on the synthetic 1 M x 1 M pair (source 10 degrees off the target): first from scratch with the
    queries far from the surface, then two passes near alignment that start from the previous pairs — sampled pairs
    against brute force, and the 16 sums against a double-precision recomputation from all returned pairs."""
    from conftest import synth_surface
    from icp_amd.api import Context
    n = 1_000_000
    tgt = synth_surface(n, 42)
    Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
    src = (synth_surface(n, 43, noise=1e-3).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
    rng = np.random.default_rng(12)
    ctx = Context(0)
    try:
        ctx.set_target(tgt)
        ctx.set_source(src)
        s0 = _check_step(ctx, oracle, src, tgt, np.eye(4), rng, 3000)
        back = np.linalg.inv(Tm)
        s1 = _check_step(ctx, oracle, src, tgt, back, rng, 3000)
        assert s1[15] < 0.01 * s0[15]
        nudge = rigid([0, 0, 1], 0.05, [1e-4, 0, -1e-4]) @ back
        _check_step(ctx, oracle, src, tgt, nudge, rng, 3000)
        # the same pass twice gives the same bits (fixed reduction order), whatever the previous pairs were
        a = ctx.icp_step(nudge); b = ctx.icp_step(nudge)
        assert np.array_equal(a, b)
    finally:
        ctx.close()


# ICPX_ICP_TILES (read when a context is created): which kernels a pass uses, by the number of 64-point tiles of the source
PATHS = {"tile + ball + walk": "0,0", "tile + walk": "0,1000000000", "walk only": "1000000000,1000000000"}


@pytest.mark.parametrize("path", list(PATHS))
def test_icp_step_far_from_the_surface_all_pairs(oracle, monkeypatch, path):
    """Queries many cells from the target (where the pass settles whole tiles against the shell of target points inside a
    ball around them), every pair against brute force: first without previous pairs, then from pairs that belong to
    another pose, then near alignment. The target holds 3 000 points twice, so equal distances occur and must go to
    the lower index. Once through each combination of kernels a pass can use (by default the size of the source
    cloud decides)."""
    from conftest import synth_surface
    from icp_amd.api import Context
    n = 40_000
    base = synth_surface(n, 7)
    tgt = np.ascontiguousarray(np.concatenate([base, base[5000:8000]]))
    src = synth_surface(n, 8, noise=1e-3)
    monkeypatch.setenv("ICPX_ICP_TILES", PATHS[path])
    if path == "tile + ball + walk":
        monkeypatch.setenv("ICPX_WALK_WIDE", "0")      # and the walk kernel launched over the whole GPU, as for clouds of 4 M points and more
    ctx = Context(0)
    try:
        ctx.set_target(tgt)
        ctx.set_source(src)
        poses = (rigid([1, 0.3, -0.2], 25.0, [0.15, -0.1, 0.05]), rigid([0, 1, 0.5], -15.0, [-0.05, 0.2, 0.1]),
                 rigid([1, 1, 1], 2.0, [0.004, -0.002, 0.003]), np.eye(4))
        for T in poses:
            sums, idx = ctx.icp_step(T, want_idx=True)
            p = _transform(src, T)
            ridx, rd2 = oracle.nn_brute(np.ascontiguousarray(p), tgt)
            assert np.array_equal(idx, ridx)
            q = tgt[idx].astype(np.float64)
            want = np.concatenate([p.sum(0), q.sum(0), (p.T @ q).reshape(9), [rd2.sum()]])
            np.testing.assert_allclose(sums, want, rtol=1e-12, atol=1e-12)
    finally:
        ctx.close()


@pytest.mark.parametrize("path", list(PATHS))
def test_icp_step_bunny_pair_every_path(oracle, bunny000, bunny045, monkeypatch, path):
    """The scan pair (partial overlap: many source points have no near counterpart) through each combination of kernels:
    same pairs, and the same bits in the 16 sums (their order of addition does not depend on which kernel settled a point)."""
    from icp_amd.api import Context
    monkeypatch.setenv("ICPX_ICP_TILES", PATHS[path])
    ctx = Context(0)
    try:
        ctx.set_target(bunny045)
        ctx.set_source(bunny000)
        T = rigid([0.2, -1, 0.4], 8.0, [0.003, 0.001, -0.002])
        for M in (np.eye(4), T, T):
            sums, idx = ctx.icp_step(M, want_idx=True)
            ref, ridx = _sums_ref(oracle, bunny000, bunny045, M)
            assert np.array_equal(idx, ridx)
            np.testing.assert_allclose(sums, ref, rtol=1e-12, atol=1e-12)
    finally:
        ctx.close()


def test_icp_million_by_million_30_iterations_against_the_kd_tree_trajectory():
    """The synthetic 1 M x 1 M pair from start to end: 30 iterations of the host class against the trajectory of the
    CPU restatement of the reference's loop with its kd-tree (src/SingleThreadIcp.cpp:293-327, src/kdtree.cpp:107-150;
    tools/make_icp1m_golden.py wrote tests/golden/icp_1m_30it.json). The reference breaks exact ties by traversal
    order and this library by lowest index, so the pairs may differ at ties; rotation and translation must agree
    within 1e-5 after every iteration and at the end (observed: 1e-12)."""
    import hashlib, json, os, sys
    from conftest import GOLDEN, ROOT
    sys.path.insert(0, ROOT)
    from bench import synth_surface, rigid
    from icp_amd.host import SingleThreadIcp
    g = json.load(open(os.path.join(GOLDEN, "icp_1m_30it.json")))
    n = g["points"]
    tgt = synth_surface(n, 42)
    Tm = rigid([1, 1, 1], 10.0, [0.02, -0.01, 0.015])
    src = (synth_surface(n, 43, noise=1e-3).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
    if hashlib.sha256(src.tobytes()).hexdigest() != g["src_sha256"] or hashlib.sha256(tgt.tobytes()).hexdigest() != g["tgt_sha256"]:
        pytest.skip("this numpy draws another synthetic cloud than the one the fixture was made from")
    r = SingleThreadIcp(src, tgt, iters=30).registration()
    acc = np.eye(4)
    worst = 0.0
    for k in range(30):
        acc = r["per_iter_T"][k] @ acc
        worst = max(worst, float(np.abs(acc - np.array(g["T_after_iteration"][k])).max()))
    assert worst < 1e-5, worst
    assert np.abs(r["T"] - np.array(g["T"])).max() < 1e-5
    np.testing.assert_allclose(r["mse"], np.array(g["mse_logged"]), rtol=1e-5)      # the residual logged after every update (float terms in the reference)
    print("largest deviation from the kd-tree trajectory:", worst)
