"""GPU parity: exact squared EDT + lookup through the C ABI against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build_and_check(ctx, O, pts, div, caps=None):
    ctx.set_target(pts)
    ctx.edt_build(div)
    info = ctx.edt_info()
    mn, cell = O.ldt_box(pts, div)
    assert np.array_equal(info["min"], mn) and info["cell"] == cell      # box rule, bit for bit
    occ, dropped = O.ldt_splat(pts, div, mn, cell)
    assert info["occupied"] == int(occ.sum()) and info["dropped"] == dropped
    sq = ctx.edt_read()
    ref = O.edt_meijster(occ)
    assert np.array_equal(sq, ref), f"{(sq != ref).sum()} voxels differ at div={div}"
    return sq, mn, cell


@pytest.mark.parametrize("div", [32, 64, 100])
def test_edt_bunny_small(gpu_ctx, oracle, bunny045, div):
    _build_and_check(gpu_ctx, oracle, bunny045, div)


def test_edt_bunny_300_and_lookup(gpu_ctx, oracle, bunny045, bunny000):
    sq, mn, cell = _build_and_check(gpu_ctx, oracle, bunny045, 300)
    assert cell == np.float32(0.0011251526)
    # the committed hash of the grid, which the CPU suite checks voxel by voxel against scipy's exact transform
    import hashlib, json, os
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "edt_bunny045.json")))["300"]
    assert hashlib.sha256(sq.astype("<i4").tobytes()).hexdigest() == g["sha256"]
    # lookup: in-box queries (bunny000) and queries far outside the box
    rng = np.random.default_rng(3)
    far = (rng.uniform(-1, 1, (5000, 3))).astype(np.float32)
    q = np.ascontiguousarray(np.concatenate([bunny000, far]))
    got = gpu_ctx.edt_lookup(q)
    want = oracle.ldt_lookup(sq, mn, cell, q)
    assert np.array_equal(got, want)
    # coordinates on and a few float steps around every cell boundary, where a cheaper division would pick the wrong cell
    k = np.arange(0, 301, dtype=np.float32)
    edge = (np.float32(mn[0]) + k * np.float32(cell)).astype(np.float32)
    xs = np.concatenate([np.nextafter(edge, np.float32(np.inf) * s, dtype=np.float32) if n else edge
                         for n, s in ((0, 1), (1, 1), (1, -1))])
    xs = np.concatenate([xs, np.nextafter(np.nextafter(edge, np.float32(np.inf)), np.float32(np.inf)),
                         np.nextafter(np.nextafter(edge, -np.float32(np.inf)), -np.float32(np.inf))]).astype(np.float32)
    for axis in range(3):
        qe = np.tile(bunny000[:1].astype(np.float32), (xs.size, 1))
        qe[:, axis] = xs - np.float32(mn[0]) + np.float32(mn[axis])
        qe = np.ascontiguousarray(qe)
        assert np.array_equal(gpu_ctx.edt_lookup(qe), oracle.ldt_lookup(sq, mn, cell, qe))


def test_edt_dense_random_and_spill(gpu_ctx, oracle):
    """Dense random occupancy drives envelope stacks deep; a tiny LDS capacity forces the spill tier."""
    import ctypes as C
    rng = np.random.default_rng(11)
    div = 96
    pts = rng.uniform(0.0, 1.0, (200000, 3)).astype(np.float32)
    mn = np.zeros(3, np.float32); cell = np.float32(1.0 / div)
    gpu_ctx.set_target(pts)
    gpu_ctx.edt_build(div, box=(mn, cell))
    occ, dropped = oracle.ldt_splat(pts, div, mn, cell)
    ref = oracle.edt_meijster(occ)
    assert np.array_equal(gpu_ctx.edt_read(), ref)
    assert gpu_ctx.edt_info()["dropped"] == dropped


def test_edt_shape_adapted_to_one_cloud_then_another(oracle, bunny045):
    """The stacks are shaped after the first cloud (all levels private, no pool); a cloud with much deeper stacks
    then overflows both passes in its first attempt — pass x runs on whatever the overflowed pass zy left — and
    the repeated build must still be exact (and nothing may hang on the way)."""
    from icp_amd.api import Context
    ctx = Context(0)
    div = 200
    for _ in range(2):
        _build_and_check(ctx, oracle, bunny045, div)
    rng = np.random.default_rng(23)
    for n in (30000, 300000):
        pts = rng.uniform(0.0, 1.0, (n, 3)).astype(np.float32)
        mn = np.zeros(3, np.float32); cell = np.float32(1.0 / div)
        ctx.set_target(pts)
        ctx.edt_build(div, box=(mn, cell))
        occ, _ = oracle.ldt_splat(pts, div, mn, cell)
        assert np.array_equal(ctx.edt_read(), oracle.edt_meijster(occ))
        _build_and_check(ctx, oracle, bunny045, div)      # and back: shaped for the dense cloud, run on the sparse one
        _build_and_check(ctx, oracle, bunny045, div)
    ctx.close()


def test_edt_private_levels_at_two_workgroups_per_cu(oracle, bunny045):
    """Just above the default resolution the stacks of pass x (36 levels at 340) no longer fit three workgroups per
    CU; the builds after the first run pass x with private levels only in a larger workgroup."""
    from icp_amd.api import Context
    ctx = Context(0)
    for _ in range(3):
        sq, _, _ = _build_and_check(ctx, oracle, bunny045, 340)
    ctx.edt_rebuild_async(); ctx.sync()
    assert np.array_equal(ctx.edt_read(), sq)
    ctx.close()


def test_edt_large_div(gpu_ctx, oracle, bunny045, bunny000, bun270):
    """div > 511: ten-bit site indices and values above 2^19 in the packed stack entries. From 512 on the library keeps
    the grid with rows padded to a multiple of 32 values (520 -> 544): the array handed out, the lookups, the DT3D-style
    lookups' grid and the bound kernels (plain grid and both brick copies) must not notice."""
    import ctypes as C
    import torch
    pts = np.ascontiguousarray(bunny045[::3])
    sq, mn, cell = _build_and_check(gpu_ctx, oracle, pts, 520)
    assert gpu_ctx.edt_grid_stride() == 544
    rng = np.random.default_rng(8)
    q = np.ascontiguousarray(np.concatenate([bunny000[::5], rng.uniform(-0.3, 0.3, (4000, 3)).astype(np.float32)]))
    assert np.array_equal(gpu_ctx.edt_lookup(q), oracle.ldt_lookup(sq, mn, cell, q))
    # the grid in place: voxel (x, y, z) at (x div + y) stride + z
    view = torch.empty(520 * 520 * 544, dtype=torch.int32, device="cuda")
    assert C.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(view.data_ptr()), C.c_void_p(gpu_ctx.edt_grid_ptr()), C.c_size_t(4 * view.numel()), 3) == 0
    assert np.array_equal(view.view(520, 520, 544)[:, :, :520].cpu().numpy(), sq)
    # a multiple of 32 keeps its own side as the stride; below 512 nothing is padded
    gpu_ctx.edt_build(300); assert gpu_ctx.edt_grid_stride() == 300
    gpu_ctx.edt_build(512); assert gpu_ctx.edt_grid_stride() == 512
    # bounds over the padded grid: eight-item kernel on the float bricks, the 16-bit bricks and the plain grid, one-item kernel
    from icp_amd.api import Context
    from icp_amd.host import normalize_goicp
    import os
    model, data, _ = normalize_goicp(bunny000, bun270)
    data = np.ascontiguousarray(data[::16])
    R = oracle.goicp_rotmat(np.array([0.3, -0.2, 0.9], np.float32)).reshape(9)
    w = np.float32(0.25)
    cubes = np.array([[np.float32(-0.2) + np.float32(j & 1) * w, np.float32(-0.1) + np.float32(j >> 1 & 1) * w, np.float32(-0.3) + np.float32(j >> 2 & 1) * w, w]
                      for j in range(8)], np.float32)
    got = {}
    for name, group_min, bricks in (("float bricks", "0", "2"), ("16-bit bricks", "0", "1"), ("plain grid", "0", "0"), ("one item", "1000000000", "2")):
        os.environ["ICPX_BNB_GROUP_MIN"] = group_min; os.environ["ICPX_BNB_BRICKS"] = bricks
        try:
            ctx = Context(0)
            ctx.set_target(model); ctx.edt_build(520); ctx.goicp_set_data(data)
            assert ctx.edt_grid_stride() == 544
            got[name] = ctx.bnb_eval(ctx.make_items([R] * 8, cubes, [-1] * 4 + [2] * 4), want_mindis=True)
            if name == "float bricks":
                sqm = ctx.edt_read(); infom = ctx.edt_info()
            ctx.close()
        finally:
            del os.environ["ICPX_BNB_GROUP_MIN"], os.environ["ICPX_BNB_BRICKS"]
    for other in ("16-bit bricks", "plain grid", "one item"):
        for a, b in zip(got["float bricks"], got[other]):
            assert np.array_equal(a, b), other
    rotated = oracle.goicp_rotate(R.reshape(3, 3), data)
    _, _, rmd = oracle.goicp_bounds(rotated, cubes[:4], sqm, infom["min"], infom["cell"], maxrot=None, want_mindis=True)
    assert np.array_equal(got["float bricks"][2][:4], rmd)


@pytest.mark.parametrize("tune", ["4,2,0", "2,7,4", "1,3,2", "3,5,30", "4,14,1"])
def test_edt_pool_and_spill_tiers(oracle, bunny045, tune, monkeypatch):
    """Tiny private LDS capacities push stack levels into the shared pool and, once that is exhausted (or
    absent), into the global spill buffer of a second build attempt. tune = groups,levels,chunks."""
    from icp_amd.api import Context
    monkeypatch.setenv("ICPX_EDT_ZY", tune)
    monkeypatch.setenv("ICPX_EDT_X", tune)
    ctx = Context(0)
    try:
        _build_and_check(ctx, oracle, bunny045, 100)
        rng = np.random.default_rng(1)
        pts = rng.uniform(0.0, 1.0, (60000, 3)).astype(np.float32)
        ctx.set_target(pts)
        ctx.edt_build(72, box=(np.zeros(3, np.float32), np.float32(1.0 / 72)))
        occ, _ = oracle.ldt_splat(pts, 72, np.zeros(3, np.float32), np.float32(1.0 / 72))
        assert np.array_equal(ctx.edt_read(), oracle.edt_meijster(occ))
    finally:
        ctx.close()


def test_edt_maximum_resolution(gpu_ctx, oracle, bunny045):
    """div = ICPX_MAX_DIV (1023): 1.07e9 voxels, 16 z-words per row, ten-bit sites. Too large for the oracle's full
    three-pass run in a test, so: box and splat against the oracle, brute-force squared distances at sampled
    voxels, and two properties that hold for an exact EDT at any size (zero exactly on the occupied voxels;
    sqrt(d) changes by at most 1 between face neighbours)."""
    import torch
    div = 1023
    gpu_ctx.set_target(bunny045)
    gpu_ctx.edt_build(div)
    info = gpu_ctx.edt_info()
    mn, cell = oracle.ldt_box(bunny045, div)
    assert np.array_equal(info["min"], mn) and info["cell"] == cell
    occ, dropped = oracle.ldt_splat(bunny045, div, mn, cell)
    assert info["occupied"] == int(occ.sum()) and info["dropped"] == dropped
    sq = torch.empty((div, div, div), dtype=torch.int32, device="cuda")
    gpu_ctx.edt_read(sq)
    assert int((sq == 0).sum().item()) == info["occupied"]
    occ_idx = np.argwhere(occ)
    assert bool((sq[torch.from_numpy(occ_idx).cuda().unbind(1)] == 0).all().item())
    rng = np.random.default_rng(5)
    near = occ_idx[rng.integers(0, len(occ_idx), 1500)] + rng.integers(-12, 13, (1500, 3))
    vox = np.concatenate([rng.integers(0, div, (1500, 3)), np.clip(near, 0, div - 1)]).astype(np.int32)
    want = oracle.edt_brute_samples(occ, vox)
    got = sq[torch.from_numpy(vox.astype(np.int64)).cuda().unbind(1)].cpu().numpy()
    assert np.array_equal(got, want)
    for axis in range(3):   # |sqrt(a) - sqrt(b)| <= 1  <=>  a <= b + 2 sqrt(b) + 1 (both ways), in float64 per 64-plane slab
        for x0 in range(0, div, 64):
            slab = sq[x0:min(x0 + 65, div)].to(torch.float64) if axis == 0 else sq[x0:min(x0 + 64, div)].to(torch.float64)
            a = slab.narrow(axis, 0, slab.shape[axis] - 1); b = slab.narrow(axis, 1, slab.shape[axis] - 1)
            if a.numel() == 0:
                continue
            assert bool((a <= b + 2 * b.sqrt() + 1 + 1e-6).all().item()) and bool((b <= a + 2 * a.sqrt() + 1 + 1e-6).all().item())


def test_edt_single_voxel_and_brute(gpu_ctx, oracle):
    pts = np.array([[0.31, 0.52, 0.77]], np.float32)
    mn = np.zeros(3, np.float32); cell = np.float32(1.0 / 40)
    gpu_ctx.set_target(pts)
    gpu_ctx.edt_build(40, box=(mn, cell))
    occ, _ = oracle.ldt_splat(pts, 40, mn, cell)
    assert occ.sum() == 1
    assert np.array_equal(gpu_ctx.edt_read(), oracle.edt_brute(occ))


def test_edt_errors(gpu_ctx):
    from icp_amd import IcpxError
    from icp_amd.api import Context
    fresh = Context(0)
    with pytest.raises(IcpxError):
        fresh.edt_build(64)            # no target yet
    fresh.set_target(np.zeros((4, 3), np.float32) + np.arange(4, dtype=np.float32)[:, None])
    with pytest.raises(IcpxError):
        fresh.edt_build(5000)          # above ICPX_MAX_DIV
    with pytest.raises(IcpxError):
        fresh.edt_lookup(np.zeros((1, 3), np.float32))   # nothing built
    fresh.close()
