"""CPU: the oracle (oracle/oracle.cpp) against the figures recorded from runs of the reference itself,
and against definition-level brute force. These pin the restatement that the GPU parity tests use."""
import numpy as np
import pytest


def test_nn_kdtree_vs_brute_and_tie_statistics(oracle, bunny000, bunny045):
    """bunny000 -> bunny045 at identity: the reference kd-tree and lowest-index brute force return the
    same distances; 52 queries have more than one exact argmin and the kd-tree's first-visited rule
    picks a different (equidistant) index than lowest-index in 25 of them (src/kdtree.cpp:83, :142)."""
    q = bunny000.astype(np.float64)
    bidx, bd2 = oracle.nn_brute(q, bunny045)
    kidx, kd2 = oracle.KdTree(bunny045).nn(q)
    assert np.array_equal(bd2, kd2)
    diff = np.nonzero(bidx != kidx)[0]
    assert len(diff) == 25
    t = bunny045.astype(np.float64)
    ties = 0
    for lo in range(0, len(q), 4096):
        qq = q[lo:lo + 4096]
        d = ((qq[:, None, 0] - t[None, :, 0]) ** 2 + (qq[:, None, 1] - t[None, :, 1]) ** 2) + (qq[:, None, 2] - t[None, :, 2]) ** 2
        ties += int(((d == bd2[lo:lo + 4096, None]).sum(1) > 1).sum())
        # definition-level pin of the brute-force oracle: numpy's argmin returns the first (lowest) index of the minimum
        assert np.array_equal(np.argmin(d, axis=1), bidx[lo:lo + 4096]) and np.array_equal(d.min(1), bd2[lo:lo + 4096])
    assert ties == 52
    # the fixture the GPU test compares its indices with. (An FNV-1a64 of these indices, 93463261f448fb72, was noted
    # from an earlier run on the reference's data without saying how they were serialised; no int16/32/64, either
    # byte order, or text encoding tried reproduces it, so the pin is the argmin check above and the committed hash is
    # a regression value.)
    import hashlib, json, os
    from conftest import GOLDEN
    gold = json.load(open(os.path.join(GOLDEN, "nn_bunny000_bunny045.json")))
    assert hashlib.sha256(bidx.astype("<i4").tobytes()).hexdigest() == gold["sha256"] and len(bidx) == gold["n"]
    for i in diff:            # different index, bit-identical distance, kd-tree index is the larger one
        assert kidx[i] > bidx[i]
        assert np.array_equal(t[kidx[i]] - q[i], t[kidx[i]] - q[i])


def test_icp_trajectory_matches_reference_run(oracle, bunny000, bunny045):
    """30 iterations of SingleThreadIcp::registration on bunny000 -> bunny045: first row of transMat and
    the logged MSE trace as recorded from the reference (kd-tree NN, target-mean carry-over)."""
    r = oracle.icp(bunny000, bunny045, 30, nn_mode=0)
    row0 = [0.8447934139625265, 0.026791961891060055, -0.53442144278048831, 0.035047392296106809]
    assert np.abs(r["T"][0] - row0).max() < 1e-13
    assert abs(r["mse"][0] - 3.862e-4) < 1e-7 and abs(r["mse"][1] - 2.086e-4) < 1e-7 and abs(r["mse"][-1] - 9.561e-6) < 1e-9
    # lowest-index NN instead of kd-tree order moves the result by ~1e-15
    b = oracle.icp(bunny000, bunny045, 30, nn_mode=1)
    assert np.abs(b["T"] - r["T"]).max() < 1e-13
    # R is a rotation
    R = r["T"][:3, :3]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12


def test_ldt_box_splat_and_exact_edt(oracle, bunny000, bunny045):
    """div = 300 on bunny045: cellLen 0.00112515, 17459 occupied voxels; the three-pass EDT equals brute
    force at sampled voxels; lookup error against the exact kd-tree distance stays below sqrt(3) cells."""
    mn, cell = oracle.ldt_box(bunny045, 300)
    assert abs(float(cell) - 0.00112515) < 5e-9
    occ, dropped = oracle.ldt_splat(bunny045, 300, mn, cell)
    assert int(occ.sum()) == 17459 and dropped == 0
    sq = oracle.edt_meijster(occ)
    rng = np.random.default_rng(0)
    vox = rng.integers(0, 300, (3000, 3)).astype(np.int32)
    assert np.array_equal(oracle.edt_brute_samples(occ, vox), sq[vox[:, 0], vox[:, 1], vox[:, 2]])
    # every one of the 27e6 voxels against an independent exact transform (scipy's, a different algorithm)
    from scipy import ndimage
    d = ndimage.distance_transform_edt(occ == 0)
    assert np.array_equal(np.rint(d * d).astype(np.int32), sq)
    import hashlib, json, os
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "edt_bunny045.json")))["300"]
    assert hashlib.sha256(sq.astype("<i4").tobytes()).hexdigest() == g["sha256"] and g["occupied"] == 17459
    assert [float(v) for v in mn] == g["min"] and float(cell) == g["cell"]
    dl = oracle.ldt_lookup(sq, mn, cell, bunny000)
    dk = oracle.KdTree(bunny045).distance(bunny000)
    err = np.abs(dl - dk).max()
    assert abs(err - 0.00163) < 1e-5 and err < np.sqrt(3) * cell


@pytest.mark.parametrize("div", [8, 17, 32])
def test_edt_meijster_equals_brute(oracle, div):
    rng = np.random.default_rng(div)
    for density in (0.002, 0.05, 0.6):
        occ = (rng.uniform(size=(div, div, div)) < density).astype(np.uint8)
        occ[rng.integers(div), rng.integers(div), rng.integers(div)] = 1
        assert np.array_equal(oracle.edt_meijster(occ), oracle.edt_brute(occ))


def test_edt_grid_hash_is_stable(oracle, bunny045):
    """Golden: SHA-256 of the 64^3 squared grid of bunny045 (tests/golden/edt_bunny045.json)."""
    import hashlib, json, os
    from conftest import GOLDEN
    gold = json.load(open(os.path.join(GOLDEN, "edt_bunny045.json")))
    for div in (32, 64):
        mn, cell = oracle.ldt_box(bunny045, div)
        occ, _ = oracle.ldt_splat(bunny045, div, mn, cell)
        sq = oracle.edt_meijster(occ)
        g = gold[str(div)]
        assert [float(v) for v in mn] == g["min"] and float(cell) == g["cell"] and int(occ.sum()) == g["occupied"]
        assert hashlib.sha256(sq.astype("<i4").tobytes()).hexdigest() == g["sha256"]


def test_ldt_lookup_out_of_box_and_upper_clamp(oracle, bunny045):
    mn, cell = oracle.ldt_box(bunny045, 32)
    occ, _ = oracle.ldt_splat(bunny045, 32, mn, cell)
    sq = oracle.edt_meijster(occ)
    top = mn + cell * np.float32(32)
    q = np.array([top, mn, mn - 1.0, top + 2.0, [mn[0] - 0.5, top[1] + 0.25, (mn[2] + top[2]) / 2]], np.float32)
    d = oracle.ldt_lookup(sq, mn, cell, q)
    assert np.isfinite(d).all()
    assert d[2] > np.sqrt(3.0) - 1e-3 and d[3] > 2 * np.sqrt(3.0) - 1e-3      # overshoot term dominates


def test_goicp_tables_and_bounds(oracle, bun270):
    data = (bun270[::50] - bun270.mean(0)).astype(np.float32) * 8
    mr = oracle.goicp_maxrotdist(data)
    nrm = np.sqrt((data[:, 0] * data[:, 0] + data[:, 1] * data[:, 1]) + data[:, 2] * data[:, 2])   # float32
    PI, S3 = 3.1415926536, 1.732050808
    # InitRot.w is a float member (src/GoIcp.cpp:26, :88): sigma at :472 starts from 2*PI rounded to float
    w0 = float(np.float32(2 * PI))
    assert w0 != 2 * PI
    differs = 0
    for l in range(20):
        tmp = min(w0 / 2.0 ** l / 2.0 * S3, PI)
        assert np.array_equal(mr[l], (2 * np.sin(tmp / 2) * nrm.astype(np.float64)).astype(np.float32))
        tmp_d = min(2 * PI / 2.0 ** l / 2.0 * S3, PI)
        differs += int((mr[l] != (2 * np.sin(tmp_d / 2) * nrm.astype(np.float64)).astype(np.float32)).sum())
    assert differs > 0          # the double-precision 2*PI gives a different table: this test can tell them apart
    assert np.allclose(mr[0], 2 * nrm, rtol=1e-6)          # sigma_0 * sqrt3 > pi: the bound saturates at the diameter
    R = oracle.goicp_rotmat(np.array([0.5, -0.25, 1.0], np.float32))
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-6 and abs(np.linalg.det(R.astype(np.float64)) - 1) < 1e-6
    assert np.array_equal(oracle.goicp_rotmat(np.zeros(3, np.float32)), np.eye(3, dtype=np.float32))


def test_goicp_oracle_runs_bnb_to_a_certificate(oracle, bunny000, bun270):
    """Tiny subsample: the rotation search (enabled after the point where the committed code returns,
    src/GoIcp.cpp:189) terminates with optError - lb < sseThresh and does not increase the ICP error."""
    m = bunny000.astype(np.float64) - bunny000.astype(np.float64).mean(0)
    d = bun270.astype(np.float64) - bun270.astype(np.float64).mean(0)
    s = 1.0 / max(np.abs(m).max(), np.abs(d).max())
    model, data = (m * s).astype(np.float32)[::80], (d * s).astype(np.float32)[::320]
    mn, cell = oracle.ldt_box(model, 64)
    occ, _ = oracle.ldt_splat(model, 64, mn, cell)
    sq = oracle.edt_meijster(occ)
    r0 = oracle.goicp(model, data, 2e-2, run_bnb=False, dist_mode=1, sq=sq, mn=mn, cell=cell)
    assert r0["rot_pops"] == 0 and r0["optError"] == min(r0["initError"], r0["icpError"])
    r = oracle.goicp(model, data, 2e-2, run_bnb=True, dist_mode=1, sq=sq, mn=mn, cell=cell)
    assert r["optError"] <= r0["optError"] and r["rot_pops"] >= 1
    assert float(r["optError"]) - r["lb"] < data.shape[0] * 2e-2 + 1e-6


def test_knn_icp_restatement_is_a_function_of_its_input(oracle):
    """The icp/ fork's loop restated (oracle.knn_icp): repeatable, the fork's comparator leaves the source order alone
    while all squared distances are below 1, and reorders at a larger scale. Parity unpinned (see oracle.cpp)."""
    from conftest import synth_surface, rigid
    B = synth_surface(1500, 1)
    T = rigid([1, 0.2, 0.1], 3.0, [0.01, -0.02, 0.005])
    A = (synth_surface(1800, 2, noise=1e-3).astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    a = oracle.knn_icp(A, B, 6, 1e-9); b = oracle.knn_icp(A, B, 6, 1e-9, threads=1)
    assert a["iter"] == b["iter"] == 7 and np.array_equal(a["trans"], b["trans"]) and np.array_equal(a["distances"], b["distances"])
    one = oracle.knn_icp(A, B, 1, 0.0)
    idx, d2 = oracle.knn_brute(A[:1500].astype(np.float64), B, 5)
    mean = np.zeros(1500, np.float32)
    for k in range(5):
        mean = (mean + d2[:, k].astype(np.float32)).astype(np.float32)
    assert np.array_equal(one["distances"], mean / np.float32(5))          # source order kept
    big = oracle.knn_icp(A * np.float32(100), B * np.float32(100), 1, 0.0)
    idx2, d2b = oracle.knn_brute((A[:1500] * np.float32(100)).astype(np.float64), B * np.float32(100), 5)
    mean_b = np.zeros(1500, np.float32)
    for k in range(5):
        mean_b = (mean_b + d2b[:, k].astype(np.float32)).astype(np.float32)
    mean_b = mean_b / np.float32(5)
    assert not np.array_equal(big["distances"], mean_b)                       # reordered ...
    assert np.array_equal(np.sort(big["distances"]), np.sort(mean_b))          # ... the same values
    with pytest.raises(ValueError):
        oracle.knn_icp(B, A, 2, 1e-9)
