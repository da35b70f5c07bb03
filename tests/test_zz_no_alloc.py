"""GPU: the allocation-free bound evaluation (icpx_bnb_eval_noalloc) against icpx_bnb_eval, bit for bit, on two
calls in a row through the C ABI, the second with a smaller batch (the fixed partial-sum area then still holds the
first call's values beyond what the second one uses)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _items(ctx, oracle, rng, n):
    Rs = [oracle.goicp_rotmat(rng.uniform(-1.5, 1.5, 3).astype(np.float32)).reshape(9) for _ in range(n)]
    cubes = np.c_[rng.uniform(-0.5, 0.25, (n, 3)), rng.choice([0.5, 0.25, 0.125, 0.0625], n)].astype(np.float32)
    levels = rng.integers(-1, 20, n)
    return ctx.make_items(Rs, cubes, levels)


def test_bnb_eval_noalloc_matches_bnb_eval_twice(oracle, bunny000, bun270):
    import torch
    from icp_amd import IcpxError
    from icp_amd.api import Context
    from icp_amd.host import normalize_goicp
    model, data, _ = normalize_goicp(bunny000, bun270)
    data = np.ascontiguousarray(data[::4])
    ctx = Context(0)
    try:
        ctx.set_target(model)
        ctx.edt_build(128)
        ctx.goicp_set_data(data)
        rng = np.random.default_rng(77)
        for n in (700, 90):                       # the second batch is smaller than the first
            items = _items(ctx, oracle, rng, n)
            ub, lb = ctx.bnb_eval(items)
            raw = np.frombuffer(C.string_at(C.addressof(items), C.sizeof(items)), np.uint8).copy()
            items_dev = torch.from_numpy(raw).cuda()
            out = torch.full((2 * n,), -1.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()              # the tensors are ready before the context's own stream reads them
            ctx.bnb_eval_noalloc(items_dev, n, out)
            ctx.sync()
            got = out.cpu().numpy()
            assert np.array_equal(got[:n], ub) and np.array_equal(got[n:], lb)
            assert np.isfinite(got).all() and (got[:n] >= got[n:]).all()
        # beyond the fixed area: a clear error, nothing launched
        big = torch.zeros(4097 * C.sizeof(items._type_), dtype=torch.uint8, device="cuda")
        out = torch.zeros(2 * 4097, dtype=torch.float64, device="cuda")
        with pytest.raises(IcpxError) as e:
            ctx.bnb_eval_noalloc(big, 4097, out)
        assert e.value.code == -5 and "4096" in str(e.value)
        # a level outside the table cannot be refused by the host (the items are device memory): NaN, no fault
        items = _items(ctx, oracle, rng, 3)
        items[1].level = 25
        raw = np.frombuffer(C.string_at(C.addressof(items), C.sizeof(items)), np.uint8).copy()
        out = torch.zeros(6, dtype=torch.float64, device="cuda")
        dev = torch.from_numpy(raw).cuda()
        torch.cuda.synchronize()
        ctx.bnb_eval_noalloc(dev, 3, out)
        ctx.sync()
        got = out.cpu().numpy()
        assert np.isnan(got[1]) and np.isnan(got[4]) and np.isfinite(got[[0, 2, 3, 5]]).all()
    finally:
        ctx.close()
