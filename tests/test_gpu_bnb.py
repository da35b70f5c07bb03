"""GPU parity: Go-ICP bound evaluation through the C ABI against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _normalise(model, data):
    """Isotropic Go-ICP normalisation: each cloud about its own centroid, one common scale into [-1,1]^3."""
    m = model.astype(np.float64) - model.astype(np.float64).mean(0)
    d = data.astype(np.float64) - data.astype(np.float64).mean(0)
    s = 1.0 / max(np.abs(m).max(), np.abs(d).max())
    return (m * s).astype(np.float32), (d * s).astype(np.float32)


def test_bounds_match_oracle(gpu_ctx, oracle, bunny000, bun270):
    model, data = _normalise(bunny000, bun270)
    data = np.ascontiguousarray(data[::8])
    gpu_ctx.set_target(model)
    gpu_ctx.edt_build(300)
    info = gpu_ctx.edt_info()
    sq = gpu_ctx.edt_read()
    gpu_ctx.goicp_set_data(data)
    # maxRotDist table
    table = gpu_ctx.goicp_maxrotdist()
    assert np.array_equal(table, oracle.goicp_maxrotdist(data))
    # and against the formula itself, independent of the oracle: InitRot.w is a float (src/GoIcp.cpp:26, :88, :472)
    nrm = np.sqrt((data[:, 0] * data[:, 0] + data[:, 1] * data[:, 1]) + data[:, 2] * data[:, 2]).astype(np.float64)
    w0 = float(np.float32(2 * 3.1415926536))
    for l in range(20):
        tmp = min(w0 / 2.0 ** l / 2.0 * 1.732050808, 3.1415926536)
        assert np.array_equal(table[l], (2 * np.sin(tmp / 2) * nrm).astype(np.float32)), l
    rng = np.random.default_rng(2)
    R = oracle.goicp_rotmat(np.array([0.4, -1.1, 0.7], np.float32))
    rotated = oracle.goicp_rotate(R, data)
    cubes = np.c_[rng.uniform(-0.5, 0.25, (24, 3)), rng.choice([0.5, 0.25, 0.125, 0.0625], 24)].astype(np.float32)
    for level in (-1, 0, 3, 7):
        items = gpu_ctx.make_items([R.reshape(9)] * len(cubes), cubes, [level] * len(cubes))
        ub, lb, md = gpu_ctx.bnb_eval(items, want_mindis=True)
        mr = None if level < 0 else oracle.goicp_maxrotdist(data)[level]
        rub, rlb, rmd = oracle.goicp_bounds(rotated, cubes, sq, info["min"], info["cell"], maxrot=mr, want_mindis=True)
        assert np.array_equal(md, rmd)                       # per-point distances bit for bit
        # sequential float sums of the reference reproduced from the per-point output
        seq_ub = np.array([np.cumsum((row * row).astype(np.float32), dtype=np.float32)[-1] for row in md])
        assert np.array_equal(seq_ub, rub)
        np.testing.assert_allclose(ub, rub.astype(np.float64), rtol=2e-5)
        np.testing.assert_allclose(lb, rlb.astype(np.float64), rtol=2e-5, atol=1e-6)


def test_residuals_match_oracle(gpu_ctx, oracle, bunny000, bun270):
    model, data = _normalise(bunny000, bun270)
    data = np.ascontiguousarray(data[::8])
    gpu_ctx.set_target(model)
    gpu_ctx.edt_build(300)
    info = gpu_ctx.edt_info(); sq = gpu_ctx.edt_read()
    gpu_ctx.goicp_set_data(data)
    T = np.eye(4); T[:3, :3] = oracle.goicp_rotmat(np.array([0.1, 0.2, -0.3], np.float32)); T[:3, 3] = [0.05, -0.02, 0.01]
    got = gpu_ctx.goicp_residuals(T)
    p = data.astype(np.float64)
    o = np.empty_like(p)
    for r in range(3):   # ((a0 x + a1 y) + a2 z) + a3, multiply then add
        o[:, r] = ((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3]
    want = oracle.ldt_lookup(sq, info["min"], info["cell"], o.astype(np.float32))
    assert np.array_equal(got, want)
