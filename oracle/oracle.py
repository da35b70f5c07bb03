"""TEST INFRASTRUCTURE ONLY — ctypes bindings for oracle/liboracle.so (the CPU restatement of the
reference's hot path) and, when present, oracle/_ref/libref.so (reference sources compiled in the
build container). Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only;
the product package icp_amd never imports this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICPX_ORACLE_SAN=1 loads the AddressSanitizer / UBSan builds (`make -C oracle asan`, tools/run_asan.sh)
_SAN = os.environ.get("ICPX_ORACLE_SAN", "") not in ("", "0")
_lib = None
_ref = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "_asan", "liboracle.so") if _SAN else os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(path)
        _lib.orc_kdtree_build.restype = C.c_void_p
        _lib.orc_ldt_splat.restype = C.c_int64
    return _lib


def ref():
    """oracle/_ref/libref.so or None (it only exists where /root/reference was present at build time)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_asan" if _SAN else "_ref", "libref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        _ref.ref_dt3d_build.restype = C.c_void_p
        _ref.ref_icp3d_run.restype = C.c_double
        _ref.ref_icp3d_run_ex.restype = C.c_double
    return _ref


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- NN -------------------------------------------------------------------------------------------
def nn_brute(q, tgt, threads=0):
    q, tgt = f64(q), f32(tgt)
    n, m = q.shape[0], tgt.shape[0]
    idx = np.empty(n, np.int32); d2 = np.empty(n, np.float64)
    lib().orc_nn_brute(_vp(q), C.c_int64(n), _vp(tgt), C.c_int64(m), _vp(idx), _vp(d2), C.c_int(threads))
    return idx, d2


def knn_brute(q, tgt, k, threads=0):
    """k nearest by (squared distance, index): idx [n, k] int32 (-1 beyond the cloud), d2 [n, k]."""
    q, tgt = f64(q), f32(tgt)
    n, m = q.shape[0], tgt.shape[0]
    idx = np.empty((n, k), np.int32); d2 = np.empty((n, k), np.float64)
    lib().orc_knn_brute(_vp(q), C.c_int64(n), _vp(tgt), C.c_int64(m), C.c_int(k), _vp(idx), _vp(d2), C.c_int(threads))
    return idx, d2


class KdTree:
    def __init__(self, tgt):
        self.tgt = f32(tgt)
        self.h = C.c_void_p(lib().orc_kdtree_build(_vp(self.tgt), C.c_int64(self.tgt.shape[0])))

    def nn(self, q, threads=0):
        q = f64(q); n = q.shape[0]
        idx = np.empty(n, np.int32); d2 = np.empty(n, np.float64)
        lib().orc_kdtree_nn(self.h, _vp(q), C.c_int64(n), _vp(idx), _vp(d2), C.c_int(threads))
        return idx, d2

    def knn_d2(self, q, k, threads=0):
        """Squared distances of KDTree::k_close's k results, ascending."""
        q = f64(q); n = q.shape[0]
        d2 = np.empty((n, k), np.float64)
        lib().orc_kdtree_knn_d2(self.h, _vp(q), C.c_int64(n), C.c_int(k), _vp(d2), C.c_int(threads))
        return d2

    def distance(self, q, threads=0):
        q = f32(q); n = q.shape[0]
        out = np.empty(n, np.float32)
        lib().orc_kdtree_distance(self.h, _vp(q), C.c_int64(n), _vp(out), C.c_int(threads))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_free(self.h); self.h = None


# ---- SVD ------------------------------------------------------------------------------------------
def svd3(A):
    A = f64(A).reshape(3, 3)
    U = np.empty((3, 3)); s = np.empty(3); V = np.empty((3, 3))
    lib().orc_svd3(_vp(A), _vp(U), _vp(s), _vp(V))
    return U, s, V


def kabsch_R(H):
    H = f64(H).reshape(3, 3); R = np.empty((3, 3))
    lib().orc_kabsch_R(_vp(H), _vp(R))
    return R


# ---- ICP ------------------------------------------------------------------------------------------
def icp(src, tgt, iters, guess=None, nn_mode=1, threads=0):
    """Returns dict(per_iter_T [iters,4,4], T [4,4], mse [iters], idx_last [nd])."""
    src, tgt = f32(src), f32(tgt)
    g = f64(np.eye(4) if guess is None else guess).reshape(16)
    per = np.empty((iters, 16)); fin = np.empty(16); mse = np.empty(iters); idx = np.empty(src.shape[0], np.int32)
    rc = lib().orc_icp(_vp(src), C.c_int64(src.shape[0]), _vp(tgt), C.c_int64(tgt.shape[0]), _vp(g), C.c_int(iters),
                       C.c_int(nn_mode), _vp(per), _vp(fin), _vp(mse), _vp(idx), C.c_int(threads))
    assert rc == 0
    return dict(per_iter_T=per.reshape(iters, 4, 4), T=fin.reshape(4, 4), mse=mse, idx_last=idx)


# ---- DT3D convention ------------------------------------------------------------------------------
def dt3d_box(pts, expand=2.0):
    """dict(SIZE, scale, min[3], max[3]) by the rule of DT3D::Build (src/jly_3ddt.cpp:732-764)."""
    p = f64(pts); info = np.empty(8)
    lib().orc_dt3d_box(_vp(p), C.c_int64(p.shape[0]), C.c_double(expand), _vp(info))
    return dict(SIZE=int(info[0]), scale=info[1], min=info[2:5].copy(), max=info[5:8].copy())


def dt3d_splat(pts, size, mn, scale):
    p = f64(pts); occ = np.empty((size, size, size), np.uint8)
    f = lib().orc_dt3d_splat; f.restype = C.c_int64
    dropped = f(_vp(p), C.c_int64(p.shape[0]), C.c_uint32(size), _vp(f64(mn)), C.c_double(scale), _vp(occ))
    return occ, int(dropped)


def dt3d_lookup(sq, mn, scale, q):
    sq = np.ascontiguousarray(sq, np.int32); q = f64(q); out = np.empty(q.shape[0], np.float32)
    lib().orc_dt3d_lookup(_vp(sq), C.c_uint32(sq.shape[0]), _vp(f64(mn)), C.c_double(scale), _vp(q), C.c_int64(q.shape[0]), _vp(out))
    return out


def icp3d_run(model, data, max_iter=10000, err_diff=0.000001, do_trim=True, trim_fraction=0.0, R=None, t=None, threads=0):
    """ICP3D::Run restated (src/jly_icp3d.cpp:181-295). Returns dict(R, t, err, iterations)."""
    model, data = f64(model), f64(data)
    R = f64(np.eye(3) if R is None else R).copy().reshape(9); t = f64(np.zeros(3) if t is None else t).copy()
    it = C.c_int64()
    f = lib().orc_icp3d_run
    f.restype = C.c_double
    err = f(_vp(model), C.c_int64(model.shape[0]), _vp(data), C.c_int64(data.shape[0]), _vp(R), _vp(t), C.c_int64(max_iter),
            C.c_double(err_diff), C.c_int(int(do_trim)), C.c_double(trim_fraction), C.c_int(threads), C.byref(it))
    return dict(R=R.reshape(3, 3), t=t, err=err, iterations=it.value)


def knn_icp(A, B, max_iteration, tolerance, K=5, remain_percentage=100, threads=0):
    """The icp/ fork's loop restated (icp/src/icp.hpp:398-460, :310-359, :185-253) with a defined order.
    Returns dict(trans [4,4], distances [n] float32, iter)."""
    A, B = f32(A), f32(B)
    trans = np.empty(16); dist = np.empty(min(A.shape[0], B.shape[0]), np.float32); it = C.c_int()
    rc = lib().orc_knn_icp(_vp(A), C.c_int64(A.shape[0]), _vp(B), C.c_int64(B.shape[0]), C.c_int(max_iteration), C.c_float(tolerance),
                           C.c_int(K), C.c_int(remain_percentage), _vp(trans), _vp(dist), C.byref(it), C.c_int(threads))
    if rc != 0:
        raise ValueError("knn_icp: bad arguments")
    return dict(trans=trans.reshape(4, 4), distances=dist, iter=it.value)


# ---- LDT ------------------------------------------------------------------------------------------
def ldt_box(pts, div):
    pts = f32(pts); mn = np.empty(3, np.float32); cl = C.c_float()
    lib().orc_ldt_box(_vp(pts), C.c_int64(pts.shape[0]), C.c_uint32(div), _vp(mn), C.byref(cl))
    return mn, np.float32(cl.value)


def ldt_splat(pts, div, mn, cell):
    pts = f32(pts); occ = np.empty((div, div, div), np.uint8)
    dropped = lib().orc_ldt_splat(_vp(pts), C.c_int64(pts.shape[0]), C.c_uint32(div), _vp(f32(mn)), C.c_float(cell), _vp(occ))
    return occ, int(dropped)


def edt_meijster(occ, threads=0):
    occ = np.ascontiguousarray(occ, np.uint8); div = occ.shape[0]
    sq = np.empty((div, div, div), np.int32)
    lib().orc_edt_meijster(_vp(occ), C.c_uint32(div), _vp(sq), C.c_int(threads))
    return sq


def edt_brute(occ, threads=0):
    occ = np.ascontiguousarray(occ, np.uint8); div = occ.shape[0]
    sq = np.empty((div, div, div), np.int32)
    lib().orc_edt_brute(_vp(occ), C.c_uint32(div), _vp(sq), C.c_int(threads))
    return sq


def edt_brute_samples(occ, vox):
    occ = np.ascontiguousarray(occ, np.uint8); vox = np.ascontiguousarray(vox, np.int32)
    out = np.empty(vox.shape[0], np.int32)
    lib().orc_edt_brute_samples(_vp(occ), C.c_uint32(occ.shape[0]), _vp(vox), C.c_int64(vox.shape[0]), _vp(out))
    return out


def ldt_lookup(sq, mn, cell, q, threads=0):
    sq = np.ascontiguousarray(sq, np.int32); q = f32(q); out = np.empty(q.shape[0], np.float32)
    lib().orc_ldt_lookup(_vp(sq), C.c_uint32(sq.shape[0]), _vp(f32(mn)), C.c_float(cell), _vp(q), C.c_int64(q.shape[0]), _vp(out), C.c_int(threads))
    return out


# ---- Go-ICP ---------------------------------------------------------------------------------------
def goicp_maxrotdist(data):
    data = f32(data); out = np.empty((20, data.shape[0]), np.float32)
    lib().orc_goicp_maxrotdist(_vp(data), C.c_int64(data.shape[0]), _vp(out))
    return out


def goicp_rotmat(v):
    R = np.empty(9, np.float32)
    lib().orc_goicp_rotmat(C.c_float(v[0]), C.c_float(v[1]), C.c_float(v[2]), _vp(R))
    return R.reshape(3, 3)


def goicp_rotate(R, data):
    data = f32(data); out = np.empty_like(data)
    lib().orc_goicp_rotate(_vp(f32(R).reshape(9)), _vp(data), C.c_int64(data.shape[0]), _vp(out))
    return out


def goicp_bounds(rotated, cubes, sq, mn, cell, maxrot=None, want_mindis=False, threads=0):
    """LDT-distance bounds for child cubes (x,y,z,w) under one rotated cloud."""
    rotated, cubes = f32(rotated), f32(cubes).reshape(-1, 4)
    sq = np.ascontiguousarray(sq, np.int32)
    nc, n = cubes.shape[0], rotated.shape[0]
    ub = np.empty(nc, np.float32); lb = np.empty(nc, np.float32)
    md = np.empty((nc, n), np.float32) if want_mindis else None
    mr = None if maxrot is None else f32(maxrot)
    lib().orc_goicp_bounds(C.c_int(1), None, _vp(sq), C.c_uint32(sq.shape[0]), _vp(f32(mn)), C.c_float(cell), _vp(rotated),
                           C.c_int64(n), _vp(mr), _vp(cubes), C.c_int64(nc), _vp(ub), _vp(lb), _vp(md), C.c_int(threads))
    return (ub, lb, md) if want_mindis else (ub, lb)


def goicp(model, data, mse, run_bnb=True, dist_mode=1, sq=None, mn=None, cell=0.0, icp_iters=10, nn_mode=1,
          max_rot_pops=0, threads=0):
    model, data = f32(model), f32(data)
    optMat = np.empty(16); stats = np.zeros(8)
    div = 0
    if dist_mode == 1:
        sq = np.ascontiguousarray(sq, np.int32); div = sq.shape[0]
    lib().orc_goicp(_vp(model), C.c_int64(model.shape[0]), _vp(data), C.c_int64(data.shape[0]), C.c_double(mse),
                    C.c_int(int(run_bnb)), C.c_int(dist_mode), _vp(sq), C.c_uint32(div), _vp(None if mn is None else f32(mn)),
                    C.c_float(cell), C.c_int(icp_iters), C.c_int(nn_mode), C.c_int64(max_rot_pops), _vp(optMat), _vp(stats),
                    C.c_int(threads))
    return dict(optMat=optMat.reshape(4, 4), optError=np.float32(stats[0]), initError=np.float32(stats[1]),
                icpError=np.float32(stats[2]), rot_pops=int(stats[3]), inner_evals=int(stats[4]), lb=stats[5])


# ---- reference pieces (oracle/_ref) ------------------------------------------------------------------
def ref_nanoflann_nn(tgt, q):
    tgt, q = f64(tgt), f64(q)
    idx = np.empty(q.shape[0], np.int64); d2 = np.empty(q.shape[0], np.float64)
    ref().ref_nanoflann_nn(_vp(tgt), C.c_int64(tgt.shape[0]), _vp(q), C.c_int64(q.shape[0]), _vp(idx), _vp(d2))
    return idx, d2


def ref_nanoflann_knn(tgt, q, k):
    tgt, q = f64(tgt), f64(q)
    idx = np.empty((q.shape[0], k), np.int64); d2 = np.empty((q.shape[0], k), np.float64)
    ref().ref_nanoflann_knn(_vp(tgt), C.c_int64(tgt.shape[0]), _vp(q), C.c_int64(q.shape[0]), C.c_int(k), _vp(idx), _vp(d2))
    return idx, d2


def ref_matrix_svd3(A):
    A = f64(A).reshape(3, 3); U = np.empty((3, 3)); W = np.empty(3); V = np.empty((3, 3))
    ref().ref_matrix_svd3(_vp(A), _vp(U), _vp(W), _vp(V))
    return U, W, V


def ref_icp3d_run(model, data, iters, R=None, t=None):
    model, data = f64(model), f64(data)
    R = f64(np.eye(3) if R is None else R).copy().reshape(9); t = f64(np.zeros(3) if t is None else t).copy()
    err = ref().ref_icp3d_run(_vp(model), C.c_int64(model.shape[0]), _vp(data), C.c_int64(data.shape[0]), _vp(R), _vp(t), C.c_int64(iters))
    return R.reshape(3, 3), t, err


def ref_icp3d_run_ex(model, data, max_iter, err_diff, do_trim, trim_fraction, R=None, t=None):
    model, data = f64(model), f64(data)
    R = f64(np.eye(3) if R is None else R).copy().reshape(9); t = f64(np.zeros(3) if t is None else t).copy()
    err = ref().ref_icp3d_run_ex(_vp(model), C.c_int64(model.shape[0]), _vp(data), C.c_int64(data.shape[0]), _vp(R), _vp(t),
                                 C.c_int64(max_iter), C.c_double(err_diff), C.c_int(int(do_trim)), C.c_double(trim_fraction))
    return R.reshape(3, 3), t, err


class RefDT3D:
    def __init__(self, pts, expand=2.0):
        p = f64(pts); x, y, z = (np.ascontiguousarray(p[:, i]) for i in range(3))
        self.h = C.c_void_p(ref().ref_dt3d_build(_vp(x), _vp(y), _vp(z), C.c_int(p.shape[0]), C.c_double(expand)))
        info = np.empty(8); ref().ref_dt3d_info(self.h, _vp(info))
        self.size, self.scale = int(info[0]), info[1]; self.mn = info[2:5].copy(); self.mx = info[5:8].copy()

    def distance(self, q):
        q = f64(q); out = np.empty(q.shape[0], np.float32)
        ref().ref_dt3d_distance(self.h, _vp(q), C.c_int64(q.shape[0]), _vp(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            ref().ref_dt3d_free(self.h); self.h = None
