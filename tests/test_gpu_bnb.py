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


def _families(rng, nfam, widths):
    """Translation cubes as the search sends them: the eight children of `nfam` parents, in the reference's child order
    (src/GoIcp.cpp:378-390: x = parent.x + (j & 1) w, y with bit 1, z with bit 2, all in float)."""
    cubes = []
    for _ in range(nfam):
        px, py, pz = rng.uniform(-0.9, 0.4, 3).astype(np.float32)
        w = np.float32(rng.choice(widths)) / np.float32(2)
        for j in range(8):
            cubes.append([px + np.float32(j & 1) * w, py + np.float32(j >> 1 & 1) * w, pz + np.float32(j >> 2 & 1) * w, w])
    return np.array(cubes, np.float32)


def test_eight_items_per_workgroup_gives_the_same_bits(oracle, bunny000, bun270, monkeypatch):
    """k_bnb_eval8 (families of eight sibling cubes share the rotated point and the per-axis cell computations) against
    k_bnb_eval (one item per workgroup): bounds and per-point distances bit for bit, for whole families, for families
    mixed with unrelated items, for a batch whose size is no multiple of eight, with cubes reaching outside the grid's
    box; the per-point distances also against the oracle."""
    from icp_amd.api import Context
    model, data = _normalise(bunny000, bun270)
    data = np.ascontiguousarray(data[::4])
    rng = np.random.default_rng(11)
    R1 = oracle.goicp_rotmat(np.array([0.4, -1.1, 0.7], np.float32)).reshape(9)
    R2 = oracle.goicp_rotmat(np.array([-2.0, 0.3, 1.2], np.float32)).reshape(9)
    fam = _families(rng, 6, [1.0, 0.5, 0.25, 0.0625])
    w = np.float32(0.0625)      # the first family far out: most of its points leave the grid's box
    fam[:8] = [[np.float32(-1.6) + np.float32(j & 1) * w, np.float32(-1.5) + np.float32(j >> 1 & 1) * w, np.float32(1.4) + np.float32(j >> 2 & 1) * w, w] for j in range(8)]
    loose = np.c_[rng.uniform(-0.5, 0.25, (13, 3)), rng.choice([0.5, 0.25, 0.125], 13)].astype(np.float32)
    cubes = np.concatenate([fam[:16], loose[:8], fam[16:], loose[8:]])        # 48 + 13 = 61 items
    rots = [R1] * 16 + [R2] * 8 + [R2] * 32 + [R1] * 5
    levels = [-1] * 8 + [2] * 8 + [0] * 8 + [5] * 16 + [-1] * 16 + [3] * 5
    # a family whose members differ in level is no family: items 24..31 share everything but the level of the last one
    levels[31] = 4
    got = {}
    # ICPX_BNB_BRICKS: what the eight-item kernel reads (2, the default: distances as floats in bricks of 4 x 4 x 4 voxels;
    # 1: squared distances as 16-bit values in such bricks; 0: the [x][y][z] grid)
    for name, group_min, bricks in (("eight", "0", "2"), ("eight, 16-bit bricks", "0", "1"), ("eight, plain grid", "0", "0"), ("one", "1000000000", "1")):
        monkeypatch.setenv("ICPX_BNB_GROUP_MIN", group_min)
        monkeypatch.setenv("ICPX_BNB_BRICKS", bricks)
        ctx = Context(0)
        try:
            ctx.set_target(model)
            ctx.edt_build(300)
            info = ctx.edt_info(); sq = ctx.edt_read()
            ctx.goicp_set_data(data)
            items = ctx.make_items(rots, cubes, levels)
            got[name] = ctx.bnb_eval(items, want_mindis=True) + ctx.bnb_eval(items)[:2]
        finally:
            ctx.close()
    for other in ("eight, 16-bit bricks", "eight, plain grid", "one"):
        for a, b in zip(got["eight"], got[other]):
            assert np.array_equal(a, b), other
    assert np.array_equal(got["eight"][0], got["eight"][3]) and np.array_equal(got["eight"][1], got["eight"][4])   # with and without the per-point output
    table = oracle.goicp_maxrotdist(data)
    md = got["eight"][2]
    for k in range(len(cubes)):
        rotated = oracle.goicp_rotate(np.asarray(rots[k]).reshape(3, 3), data)
        mr = None if levels[k] < 0 else table[levels[k]]
        _, _, rmd = oracle.goicp_bounds(rotated, cubes[k:k + 1], sq, info["min"], info["cell"], maxrot=mr, want_mindis=True)
        assert np.array_equal(md[k], rmd[0]), k
    moved = oracle.goicp_rotate(R1.reshape(3, 3), data) + (cubes[0, :3] + cubes[0, 3] / np.float32(2))
    assert (moved < np.asarray(info["min"], np.float32)).any()                 # the overshoot branch was taken


def test_brick_copy_of_the_grid_far_from_the_surface(oracle, monkeypatch):
    """The bound kernel reads brick copies of the grid: the distances as floats, or (ICPX_BNB_BRICKS=1) the squared
    distances as 16-bit values, going back to the 32-bit grid where a value does not fit: a thin target in a 400^3 grid
    has voxels more than 255 cells away. Same bits as the kernel that only knows the 32-bit grid, and per-point distances
    equal to the oracle's."""
    from icp_amd.api import Context
    rng = np.random.default_rng(5)
    target = np.c_[np.linspace(-0.5, 0.5, 200), rng.normal(0, 1e-3, 200), rng.normal(0, 1e-3, 200)].astype(np.float32)
    data = rng.uniform(-1.0, 1.0, (6000, 3)).astype(np.float32)
    R = oracle.goicp_rotmat(np.array([0.05, -0.02, 0.03], np.float32)).reshape(9)
    cubes = np.concatenate([_families(np.random.default_rng(6), 1, [0.125]) * np.float32([0.05, 0.05, 0.05, 1]), _families(np.random.default_rng(7), 1, [0.5]) * np.float32([0.1, 0.1, 0.1, 1])])
    got = {}
    for name, group_min, bricks in (("eight", "0", "2"), ("eight, 16-bit bricks", "0", "1"), ("one", "1000000000", "2")):
        monkeypatch.setenv("ICPX_BNB_GROUP_MIN", group_min)
        monkeypatch.setenv("ICPX_BNB_BRICKS", bricks)
        ctx = Context(0)
        try:
            ctx.set_target(target)
            ctx.edt_build(400)
            info = ctx.edt_info(); sq = ctx.edt_read()
            ctx.goicp_set_data(data)
            got[name] = ctx.bnb_eval(ctx.make_items([R] * 16, cubes, [-1] * 8 + [1] * 8), want_mindis=True)
            # a rebuild discards the copy; the next batch makes it again
            ctx.edt_build(400)
            again = ctx.bnb_eval(ctx.make_items([R] * 16, cubes, [-1] * 8 + [1] * 8), want_mindis=True)
            for a, b in zip(got[name], again):
                assert np.array_equal(a, b)
        finally:
            ctx.close()
    for other in ("eight, 16-bit bricks", "one"):
        for a, b in zip(got["eight"], got[other]):
            assert np.array_equal(a, b), other
    assert int(sq.max()) > 65535
    rotated = oracle.goicp_rotate(R.reshape(3, 3), data)
    _, _, rmd = oracle.goicp_bounds(rotated, cubes[:8], sq, info["min"], info["cell"], maxrot=None, want_mindis=True)
    assert np.array_equal(got["eight"][2][:8], rmd)
    assert (rmd > 256 * info["cell"]).any()            # some looked-up voxels are beyond the 16-bit range


def _children(parents_xyzw):
    """The eight children of every parent cube in the reference's order and float arithmetic (src/GoIcp.cpp:378-390)."""
    out = []
    for x, y, z, w in parents_xyzw.astype(np.float32):
        cw = np.float32(w / 2.0)
        for j in range(8):
            out.append([x + np.float32(j & 1) * cw, y + np.float32(j >> 1 & 1) * cw, z + np.float32(j >> 2 & 1) * cw, cw])
    return np.array(out, np.float32)


@pytest.mark.parametrize("group_min", ["0", "1000000000"])
def test_parent_batches_in_two_slots(oracle, bunny000, bun270, monkeypatch, group_min):
    """icpx_bnb_submit / icpx_bnb_collect: parent cubes stand for their eight children; two batches in flight; a share of
    a batch into a device buffer. Bounds bit for bit those of icpx_bnb_eval on the children, through either kernel."""
    import torch
    from icp_amd.api import Context
    model, data = _normalise(bunny000, bun270)
    data = np.ascontiguousarray(data[::4])
    rng = np.random.default_rng(21)
    Rs = [oracle.goicp_rotmat(rng.uniform(-1.5, 1.5, 3).astype(np.float32)).reshape(9) for _ in range(3)]
    monkeypatch.setenv("ICPX_BNB_GROUP_MIN", group_min)
    ctx = Context(0)
    try:
        ctx.set_target(model)
        ctx.edt_build(300)
        ctx.goicp_set_data(data)
        batches = []
        for np_, seed in ((37, 1), (5, 2), (64, 3)):
            r = np.random.default_rng(seed)
            par = np.c_[r.uniform(-0.9, 0.3, (np_, 3)), r.choice([1.0, 0.5, 0.25, 0.03125], np_)].astype(np.float32)
            par[0] = [-0.5, -0.5, -0.5, 1.0]                      # the root cube of every inner search
            rots = [Rs[k % 3] for k in range(np_)]
            levels = [(-1, 0, 2, 6)[k % 4] for k in range(np_)]
            kids = _children(par)
            want = ctx.bnb_eval(ctx.make_items([rots[k // 8] for k in range(8 * np_)], kids, [levels[k // 8] for k in range(8 * np_)]))
            batches.append((ctx.make_items(rots, par, levels), want))
        # two batches in flight, collected in the order they were submitted, then the slots used again
        ctx.bnb_submit(0, batches[0][0]); ctx.bnb_submit(1, batches[1][0])
        got0 = ctx.bnb_collect(0); ctx.bnb_submit(0, batches[2][0])
        got1 = ctx.bnb_collect(1); got2 = ctx.bnb_collect(0)
        for got, (_, want) in zip((got0, got1, got2), batches):
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # a rank's share of a batch, left in a device buffer: zeros for the other parents
        items, want = batches[0]
        n = len(items)
        out = torch.full((16 * n,), -1.0, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        ctx.bnb_submit(1, items, first=10, count=20, out_dev=out)
        assert ctx.bnb_collect(1) is None
        got = out.cpu().numpy()
        mask = np.zeros(8 * n, bool); mask[80:240] = True
        assert np.array_equal(got[:8 * n][mask], want[0][mask]) and np.array_equal(got[8 * n:][mask], want[1][mask])
        assert not got[:8 * n][~mask].any() and not got[8 * n:][~mask].any()
        # the same with the copy to the host enqueued afterwards (where several ranks put their sum in between)
        ctx.bnb_submit(0, items, first=10, count=20, out_dev=out)
        ctx.sync()
        out.mul_(2.0)                                           # stands for the collective (on torch's stream, hence the two waits)
        torch.cuda.synchronize()
        ctx.bnb_fetch(0, out)
        ub2, lb2 = ctx.bnb_collect(0)
        assert np.array_equal(ub2[mask], 2 * want[0][mask]) and np.array_equal(lb2[mask], 2 * want[1][mask]) and not ub2[~mask].any()
        with pytest.raises(Exception):
            ctx.bnb_collect(1)                                  # nothing in flight
    finally:
        ctx.close()
