"""Python handles on the C++ host classes of csrc/host/icpx_host.hpp (KDTree, LDT, SingleThreadIcp,
GoIcp) and the 3x3 host algebra. The classes keep the reference's names and member meaning
(src/kdtree.cpp, src/testLDT.cpp, src/SingleThreadIcp.cpp, src/GoIcp.cpp); everything per-point runs
on the GPU through the C ABI.
"""
import ctypes as C
import numpy as np

from .lib import host as _host, IcpxError

_ALLREDUCE = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int)
_configured = False


def _lib():
    global _configured
    lib = _host()
    if not _configured:
        lib.icpxh_last_error.restype = C.c_char_p
        for name in ("icpxh_kdtree_new", "icpxh_ldt_new", "icpxh_icp_new", "icpxh_goicp_new", "icpxh_icp3d_new", "icpxh_dt3d_new"):
            getattr(lib, name).restype = C.c_void_p
        lib.icpxh_kdtree_new.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        lib.icpxh_ldt_new.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_uint32]
        lib.icpxh_icp_new.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        lib.icpxh_goicp_new.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_uint32]
        for name in ("icpxh_kdtree_free", "icpxh_ldt_free", "icpxh_icp_free", "icpxh_goicp_free"):
            getattr(lib, name).argtypes = [C.c_void_p]; getattr(lib, name).restype = None
        lib.icpxh_kdtree_get_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.icpxh_kdtree_k_close.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.icpxh_kdtree_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.icpxh_ldt_info.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        lib.icpxh_ldt_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.icpxh_ldt_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.icpxh_icp_set_collective.argtypes = [C.c_void_p, C.c_int, C.c_int, _ALLREDUCE]
        lib.icpxh_icp_registration.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        lib.icpxh_dt3d_new.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]
        lib.icpxh_dt3d_free.argtypes = [C.c_void_p]; lib.icpxh_dt3d_free.restype = None
        lib.icpxh_dt3d_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.icpxh_icp3d_new.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        lib.icpxh_icp3d_free.argtypes = [C.c_void_p]; lib.icpxh_icp3d_free.restype = None
        lib.icpxh_icp3d_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_int,
                                        C.c_double, C.c_void_p, C.c_void_p]
        lib.icpxh_knn_icp.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        lib.icpxh_icp_device_loop.argtypes = [C.c_void_p, C.c_int]
        lib.icpxh_goicp_options.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int]
        lib.icpxh_goicp_exact_residuals.argtypes = [C.c_void_p, C.c_int]
        lib.icpxh_goicp_pipeline.argtypes = [C.c_void_p, C.c_int]
        lib.icpxh_goicp_tail_lookahead.argtypes = [C.c_void_p, C.c_int]
        lib.icpxh_goicp_device_search.argtypes = [C.c_void_p, C.c_int]
        lib.icpxh_goicp_set_collective_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, _ALLREDUCE, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.icpxh_goicp_set_collective.argtypes = [C.c_void_p, C.c_int, C.c_int, _ALLREDUCE]
        lib.icpxh_goicp_registration.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.icpxh_goicp_inner.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        lib.icpxh_load_ply.argtypes = [C.c_char_p, C.c_void_p, C.c_long]; lib.icpxh_load_ply.restype = C.c_long
        lib.icpxh_normalize_goicp.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]; lib.icpxh_normalize_goicp.restype = C.c_double
        lib.icpxh_svd3.argtypes = [C.c_void_p] * 4
        lib.icpxh_kabsch.argtypes = [C.c_void_p] * 2
        _configured = True
    return lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ck(rc):
    if rc != 0:
        raise IcpxError(rc, _lib().icpxh_last_error().decode())


def _new(ptr):
    if not ptr:
        raise IcpxError(-1, _lib().icpxh_last_error().decode())
    return C.c_void_p(ptr)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def load_ply(path):
    """Vertex positions of a PLY file (ascii or binary little endian), non-finite vertices dropped:
    the C++ loadFile of csrc/host/icpx_io.cpp, which stands in for src/main.cpp:35-51."""
    n = _lib().icpxh_load_ply(path.encode(), None, 0)
    if n < 0:
        raise IcpxError(-1, _lib().icpxh_last_error().decode())
    out = np.empty((n, 3), np.float32)
    _lib().icpxh_load_ply(path.encode(), _vp(out), n)
    return out


def normalize_goicp(model, data):
    """(model', data', scale): each cloud about its own centroid, one scale into [-1, 1]^3."""
    m, d = _f32(model).copy(), _f32(data).copy()
    s = _lib().icpxh_normalize_goicp(_vp(m), m.shape[0], _vp(d), d.shape[0])
    return m, d, s


def svd3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(3, 3)
    U = np.empty((3, 3)); s = np.empty(3); V = np.empty((3, 3))
    _lib().icpxh_svd3(_vp(A), _vp(U), _vp(s), _vp(V))
    return U, s, V


def kabsch_rotation(H):
    H = np.ascontiguousarray(H, np.float64).reshape(3, 3); R = np.empty((3, 3))
    _lib().icpxh_kabsch(_vp(H), _vp(R))
    return R


def kabsch_from_sums(sums, n):
    """Per-iteration transform from the 16 sums of icpx_icp_step with plain means (no carry-over of the
    previous target mean): used by the benchmark's synthetic ICP loop."""
    pbar = sums[0:3] / n; qbar = sums[3:6] / n
    H = (sums[6:15].reshape(3, 3) - np.outer(pbar, sums[3:6])) / n
    R = kabsch_rotation(H)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = qbar - R @ pbar
    return T


def make_allreduce(fn):
    """Wrap fn(np.ndarray[float64]) -> None (in-place sum over ranks) as the C callback the host classes take."""
    def cb(ptr, n):
        fn(np.ctypeslib.as_array(ptr, shape=(n,)))
    return _ALLREDUCE(cb)


class KDTree:
    def __init__(self, target, device=0):
        self._t = _f32(target)
        self._h = _new(_lib().icpxh_kdtree_new(device, _vp(self._t), self._t.shape[0]))

    def get_closest(self, q, want_d2=False):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        idx = np.empty(q.shape[0], np.int32); d2 = np.empty(q.shape[0], np.float64) if want_d2 else None
        _ck(_lib().icpxh_kdtree_get_closest(self._h, _vp(q), q.shape[0], _vp(idx), _vp(d2)))
        return (idx, d2) if want_d2 else idx

    def k_close(self, q, k):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), np.int32); d2 = np.empty((q.shape[0], k), np.float64)
        _ck(_lib().icpxh_kdtree_k_close(self._h, _vp(q), q.shape[0], k, _vp(idx), _vp(d2)))
        return idx, d2

    def Distance(self, q):
        q = _f32(q).reshape(-1, 3); out = np.empty(q.shape[0], np.float32)
        _ck(_lib().icpxh_kdtree_distance(self._h, _vp(q), q.shape[0], _vp(out)))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().icpxh_kdtree_free(self._h); self._h = None


class LDT:
    def __init__(self, target, div=300, device=0):
        self._t = _f32(target)
        self._h = _new(_lib().icpxh_ldt_new(device, _vp(self._t), self._t.shape[0], div))
        mn = np.empty(3, np.float32); cl = C.c_float(); d = C.c_uint32(); occ = C.c_uint64()
        _lib().icpxh_ldt_info(self._h, _vp(mn), C.byref(cl), C.byref(d), C.byref(occ))
        self._min, self.cellLen, self.div, self.occupied = mn, np.float32(cl.value), d.value, occ.value

    def Distance(self, q):
        q = _f32(q).reshape(-1, 3); out = np.empty(q.shape[0], np.float32)
        _ck(_lib().icpxh_ldt_distance(self._h, _vp(q), q.shape[0], _vp(out)))
        return out

    def grids(self):
        """(g, squared): the float grid the reference exposes as LDT::g and the int32 squared grid."""
        g = np.empty((self.div,) * 3, np.float32); sq = np.empty((self.div,) * 3, np.int32)
        _ck(_lib().icpxh_ldt_grid(self._h, _vp(g), _vp(sq)))
        return g, sq

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().icpxh_ldt_free(self._h); self._h = None


class SingleThreadIcp:
    def __init__(self, source, target, iters=10, device=0, allreduce=None, rank=0, world=1, device_loop=True):
        self._s, self._t = _f32(source), _f32(target)
        self.iters = iters
        self._h = _new(_lib().icpxh_icp_new(device, _vp(self._s), self._s.shape[0], _vp(self._t), self._t.shape[0], iters))
        _lib().icpxh_icp_device_loop(self._h, int(device_loop))     # False: a host round trip per iteration (same bits)
        self._cb = None
        if world > 1:
            self._cb = make_allreduce(allreduce)
            _ck(_lib().icpxh_icp_set_collective(self._h, rank, world, self._cb))

    def registration(self, guess=None, want_points=False, want_idx=False):
        g = np.ascontiguousarray(np.eye(4) if guess is None else guess, np.float64).reshape(16)
        T = np.empty(16); per = np.empty((self.iters, 16)); mse = np.empty(self.iters)
        res = np.empty((self._s.shape[0], 3)) if want_points else None
        idx = np.empty(self._s.shape[0], np.int32) if want_idx else None
        after = np.empty(self.iters)
        _ck(_lib().icpxh_icp_registration(self._h, _vp(g), _vp(T), _vp(per), _vp(mse), _vp(res), _vp(idx), _vp(after)))
        return dict(T=T.reshape(4, 4), per_iter_T=per.reshape(-1, 4, 4), nn_mse=mse, mse=after, resN3=res, idx_last=idx)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().icpxh_icp_free(self._h); self._h = None


class DT3D:
    """icp::DT3D (the reference's src/jly_3ddt.cpp:727-856): Build on a cloud, Distance(queries)."""
    def __init__(self, pts, expandFactor=2.0, device=0):
        p = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        info = np.empty(8)
        self._h = _new(_lib().icpxh_dt3d_new(device, _vp(p), p.shape[0], expandFactor, _vp(info)))
        self.SIZE, self.scale = int(info[0]), info[1]
        self.xMin, self.yMin, self.zMin, self.xMax, self.yMax, self.zMax = info[2:]

    def Distance(self, q):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        out = np.empty(q.shape[0], np.float32)
        _ck(_lib().icpxh_dt3d_distance(self._h, _vp(q), q.shape[0], _vp(out)))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().icpxh_dt3d_free(self._h); self._h = None


def knn_icp(A, B, max_iteration, tolerance, K=5, remain_percentage=100, device=0):
    """icp::knn_icp: the loop of the reference's icp/ fork (icp/src/icp.hpp:398-460) over K exact nearest neighbours, with a
    defined order (see icpx_host.hpp). Returns dict(trans [4,4], distances [n] float32 in the last iteration's order, iter)."""
    A = np.ascontiguousarray(A, np.float32).reshape(-1, 3); B = np.ascontiguousarray(B, np.float32).reshape(-1, 3)
    trans = np.empty(16); dist = np.empty(min(A.shape[0], B.shape[0]), np.float32); it = C.c_int()
    _ck(_lib().icpxh_knn_icp(device, _vp(A), A.shape[0], _vp(B), B.shape[0], int(max_iteration), float(tolerance), int(K), int(remain_percentage),
                             _vp(trans), _vp(dist), C.byref(it)))
    return dict(trans=trans.reshape(4, 4), distances=dist, iter=it.value)


class ICP3D:
    """icp::ICP3D (the reference's src/jly_icp3d.cpp): Build(model) then Run(data, R, t) with optional trimming."""
    def __init__(self, model, device=0):
        self._m = np.ascontiguousarray(model, np.float64).reshape(-1, 3)
        self._h = _new(_lib().icpxh_icp3d_new(device, _vp(self._m), self._m.shape[0]))
        self.max_iter_def, self.err_diff_def, self.trim_fraction, self.do_trim = 10000, 0.000001, 0.0, True

    def Run(self, data, R=None, t=None, max_iter=None, err_diff=None):
        d = np.ascontiguousarray(data, np.float64).reshape(-1, 3)
        R = np.ascontiguousarray(np.eye(3) if R is None else R, np.float64).reshape(9).copy()
        t = np.ascontiguousarray(np.zeros(3) if t is None else t, np.float64).reshape(3).copy()
        err = C.c_double(); it = C.c_size_t()
        _ck(_lib().icpxh_icp3d_run(self._h, _vp(d), d.shape[0], _vp(R), _vp(t), self.max_iter_def if max_iter is None else max_iter,
                                   self.err_diff_def if err_diff is None else err_diff, int(self.do_trim), self.trim_fraction,
                                   C.byref(err), C.byref(it)))
        return dict(R=R.reshape(3, 3), t=t, err=err.value, iterations=it.value)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().icpxh_icp3d_free(self._h); self._h = None


class GoIcp:
    """world > 1: every bound batch is shared by the ranks. Either `allreduce` (in-place sum of a host float64 array over
    the ranks) or, for the device-side path, `allreduce_dev` (in-place sum of a float64 CUDA tensor over the ranks on
    torch's current stream, e.g. torch.distributed.all_reduce over RCCL) together with `dev_buffer`, a float64 CUDA
    tensor the bounds travel in (2 doubles per cube of a batch; larger batches go in pieces)."""
    def __init__(self, model, data, mse=1e-3, dt_div=300, device=0, allreduce=None, rank=0, world=1, allreduce_dev=None, dev_buffer=None):
        self._m, self._d = _f32(model), _f32(data)
        self._h = _new(_lib().icpxh_goicp_new(device, _vp(self._m), self._m.shape[0], _vp(self._d), self._d.shape[0], mse, dt_div))
        self._cb = None
        if world > 1 and allreduce_dev is not None:
            import torch
            self._buf = dev_buffer
            base, n8 = dev_buffer.data_ptr(), dev_buffer.numel()

            def cb(ptr, n):
                off = (C.cast(ptr, C.c_void_p).value - base) // 8
                allreduce_dev(self._buf[off:off + n])
            self._cb = _ALLREDUCE(cb)
            _ck(_lib().icpxh_goicp_set_collective_dev(self._h, rank, world, self._cb, C.c_void_p(base), n8,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        elif world > 1:
            self._cb = make_allreduce(allreduce)
            _ck(_lib().icpxh_goicp_set_collective(self._h, rank, world, self._cb))

    def options(self, run_bnb=True, batched=True, parity_sums=False, frontier_width=1, max_rot_pops=0, lookahead=4, speculate=0,
                exact_residuals=False, pipeline=True, tail_lookahead=True, shard_cubes=True, device_search=True):
        _ck(_lib().icpxh_goicp_options(self._h, int(run_bnb), int(batched), int(parity_sums), frontier_width, max_rot_pops, lookahead, speculate))
        _lib().icpxh_goicp_exact_residuals(self._h, int(exact_residuals))
        _lib().icpxh_goicp_pipeline(self._h, int(pipeline))
        _lib().icpxh_goicp_tail_lookahead(self._h, int(tail_lookahead))
        _lib().icpxh_goicp_shard_cubes(self._h, int(shard_cubes))
        _lib().icpxh_goicp_device_search(self._h, int(device_search))
        return self

    def registration(self, want_points=False):
        M = np.empty(16); st = np.zeros(12)
        res = np.empty((self._d.shape[0], 3)) if want_points else None
        _ck(_lib().icpxh_goicp_registration(self._h, _vp(M), _vp(st), _vp(res)))
        return dict(optMat=M.reshape(4, 4), optError=np.float32(st[0]), initError=np.float32(st[1]), icpError=np.float32(st[2]),
                    rot_pops=int(st[3]), inner_evals=int(st[4]), lb=st[5], launches=int(st[6]), sseThresh=st[7], spec_hits=int(st[8]),
                    spec_misses=int(st[9]), collectives=int(st[10]), search_steps=int(st[11]), resN3=res)

    def innerBnB(self, R, level, optError):
        R = _f32(R).reshape(9); out = C.c_float(); tr = np.empty(4, np.float32)
        _ck(_lib().icpxh_goicp_inner(self._h, _vp(R), level, C.c_float(optError), C.byref(out), _vp(tr)))
        return np.float32(out.value), tr

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().icpxh_goicp_free(self._h); self._h = None
