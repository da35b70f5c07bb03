"""GPU: the inner searches of Go-ICP run on the device (icpx_bnb_search_run) against the host-driven loop over
icpx_bnb_submit: the same queue order, the same decisions, the same bits (src/GoIcp.cpp:353-457 is the loop)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(bunny000, bun270, stride):
    from icp_amd.host import normalize_goicp
    m, d, _ = normalize_goicp(bunny000, bun270)
    return m, np.ascontiguousarray(d[::stride])


@pytest.mark.parametrize("speculate", [0, 7])
def test_goicp_with_device_searches_takes_the_host_loops_decisions(bunny000, bun270, speculate):
    """Whole Go-ICP runs: cubes popped, optimum, bound, pose, searches found ahead of time and the number of bound
    evaluations (the host loop without looking ahead evaluates exactly the cubes the reference's loop does) are equal."""
    from icp_amd.host import GoIcp
    m, d = _pair(bunny000, bun270, 8)
    host = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=0, tail_lookahead=False, speculate=speculate, device_search=False).registration()
    dev = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, speculate=speculate, device_search=True).registration()
    assert host["search_steps"] == 0 and dev["search_steps"] > 0
    for k in ("optError", "lb", "rot_pops", "spec_hits", "spec_misses", "inner_evals", "initError", "icpError"):
        assert dev[k] == host[k], k
    assert np.array_equal(dev["optMat"], host["optMat"])
    # looking ahead on the host changes the work, not the decisions
    ahead = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=4, speculate=speculate, device_search=False).registration()
    for k in ("optError", "lb", "rot_pops"):
        assert dev[k] == ahead[k], k
    # one search at a time (the reference's order of work): every innerBnB call is a device search of its own
    if speculate == 0:
        serial = GoIcp(m, np.ascontiguousarray(d[::8]), mse=1e-3, dt_div=300).options(batched=False, device_search=True, max_rot_pops=3).registration()
        serial_host = GoIcp(m, np.ascontiguousarray(d[::8]), mse=1e-3, dt_div=300).options(batched=False, device_search=False, lookahead=0, tail_lookahead=False,
                                                                                     max_rot_pops=3).registration()
        for k in ("optError", "lb", "rot_pops", "inner_evals"):
            assert serial[k] == serial_host[k], k


def test_bnb_search_run_against_the_host_loop(gpu_ctx, oracle, bunny000, bun270):
    """icpx_bnb_search_run through the C ABI: a batch of searches with different rotations, levels and starting errors
    against GoIcp::innerBnB driven by the host, one search at a time: best error bit for bit, the same best cube."""
    from icp_amd.host import GoIcp
    m, d = _pair(bunny000, bun270, 16)
    go = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=False, device_search=False, lookahead=0, tail_lookahead=False)
    gpu_ctx.set_target(m)
    gpu_ctx.edt_build(300)
    gpu_ctx.goicp_set_data(d)
    rng = np.random.default_rng(11)
    n = 48
    Rs = [oracle.goicp_rotmat(rng.uniform(-1.8, 1.8, 3).astype(np.float32)).reshape(9) for _ in range(n)]
    levels = rng.integers(-1, 6, n)
    starts = np.where(rng.random(n) < 0.5, 1e9, rng.uniform(20.0, 400.0, n)).astype(np.float32)
    cube = np.array([-0.5, -0.5, -0.5, 1.0], np.float32)
    sse = np.float32(d.shape[0]) * np.float32(1e-3)
    opt, have, best, cubes, status, steps = gpu_ctx.bnb_search(Rs, levels, starts, [cube] * n, sse)
    assert (status == 0).all() and steps >= cubes.max() and cubes.min() >= 1
    improved = 0
    for i in range(n):
        want, tr = go.innerBnB(Rs[i], int(levels[i]), float(starts[i]))
        assert opt[i] == want, (i, opt[i], want)
        if have[i]:
            improved += 1
            assert np.array_equal(best[i], tr), i
        else:
            assert opt[i] == starts[i] and np.array_equal(tr, cube)
    assert 0 < improved
    # the same searches again in one call with others: a search depends on its own record alone
    opt2, have2, best2, cubes2, status2, _ = gpu_ctx.bnb_search(Rs[::-1], levels[::-1], starts[::-1], [cube] * n, sse)
    assert np.array_equal(opt2[::-1], opt) and np.array_equal(best2[::-1], best) and np.array_equal(cubes2[::-1], cubes)


def test_bnb_search_run_gives_up_cleanly(gpu_ctx, oracle, bunny000, bun270):
    """A search that cannot end within the queue's capacity or the step limit (threshold 0: only an empty queue ends it)
    reports status 1 or 2 and the call returns; GoIcp hands such searches to the host loop. Wrong levels are refused."""
    from icp_amd import IcpxError
    m, d = _pair(bunny000, bun270, 64)
    gpu_ctx.set_target(m)
    gpu_ctx.edt_build(128)
    gpu_ctx.goicp_set_data(d)
    R = oracle.goicp_rotmat(np.array([0.2, 0.1, -0.4], np.float32)).reshape(9)
    cube = np.array([-0.5, -0.5, -0.5, 1.0], np.float32)
    opt, have, best, cubes, status, steps = gpu_ctx.bnb_search([R, R], [-1, 3], [1e9, 1e9], [cube, cube], 0.0)
    assert set(status.tolist()) <= {0, 1, 2} and status.max() > 0 and steps <= 4096 and (cubes >= 1).all()
    with pytest.raises(IcpxError):
        gpu_ctx.bnb_search([R], [30], [1e9], [cube], 1.0)


def test_goicp_hands_unfinished_device_searches_to_the_host_loop(monkeypatch, bunny000, bun270):
    """With the device limited to three steps per call most searches come back unfinished (status 2); GoIcp runs those
    from their start through the host loop and ends with the decisions of the host loop alone."""
    from icp_amd.host import GoIcp
    m, d = _pair(bunny000, bun270, 16)
    host = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=0, tail_lookahead=False, speculate=3, device_search=False).registration()
    monkeypatch.setenv("ICPX_SEARCH_STEPS", "3")
    mixed = GoIcp(m, d, mse=1e-3, dt_div=300).options(batched=True, lookahead=0, tail_lookahead=False, speculate=3, device_search=True).registration()
    for k in ("optError", "lb", "rot_pops", "spec_hits"):
        assert mixed[k] == host[k], k
    assert np.array_equal(mixed["optMat"], host["optMat"])
    assert mixed["search_steps"] > 0 and mixed["inner_evals"] > host["inner_evals"]      # the device's unfinished work is counted too
