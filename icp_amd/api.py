"""Thin Python view of the C ABI (include/icpx.h): one Context per GPU.

Arrays may be numpy arrays (host memory) or torch CUDA tensors (device memory, used in place on the
context's GPU). torch is only plumbing here: device memory and streams.
"""
import ctypes as C
import numpy as np

from .lib import icpx, IcpxError, BnbItem

HOST, DEVICE = 0, 1
NSUMS = 16
MAXROTLEVEL = 20


def _is_torch(a):
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda")


def _ptr(a, dtype, ctx=None):
    """(pointer, where) of a C-contiguous numpy array or CUDA tensor of the given dtype. A CUDA tensor may still be
    being written by work on torch's current stream: unless the context launches on that very stream (set_stream), the
    stream is drained first, so that the library's own stream never reads a tensor before torch has produced it."""
    if _is_torch(a):
        import torch
        want = {np.float32: torch.float32, np.float64: torch.float64, np.int32: torch.int32}[dtype]
        if a.dtype != want or not a.is_contiguous() or not a.is_cuda:
            raise TypeError(f"expected a contiguous CUDA tensor of {want}")
        cur = torch.cuda.current_stream(a.device)
        if ctx is None or ctx._stream != cur.cuda_stream:
            cur.synchronize()
        return C.c_void_p(a.data_ptr()), DEVICE
    if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags.c_contiguous:
        raise TypeError(f"expected a C-contiguous numpy array of {np.dtype(dtype)}")
    return a.ctypes.data_as(C.c_void_p), HOST


def _dev(t, ctx):
    """Raw device pointer of a CUDA tensor of any dtype, ordered like _ptr: work still pending on torch's current stream
    (the torch.zeros / .cuda() that produced the tensor) is waited for unless the context launches on that very stream."""
    import torch
    if not t.is_cuda or not t.is_contiguous():
        raise TypeError("expected a contiguous CUDA tensor")
    cur = torch.cuda.current_stream(t.device)
    if ctx._stream != cur.cuda_stream:
        cur.synchronize()
    return C.c_void_p(t.data_ptr())


class Context:
    def __init__(self, device=0):
        self._lib = icpx()
        h = C.c_void_p()
        rc = self._lib.icpx_create(device, C.byref(h))
        if rc != 0:
            raise IcpxError(rc, self._lib.icpx_last_error(None).decode())
        self._h = h
        self._stream = -1              # the context's own stream (not one of torch's)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.icpx_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc):
        if rc != 0:
            raise IcpxError(rc, self._lib.icpx_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr):
        """stream_ptr: a hipStream_t as an integer (torch.cuda.current_stream().cuda_stream); 0 is the default stream."""
        self._ck(self._lib.icpx_set_stream(self._h, C.c_void_p(stream_ptr)))
        self._stream = stream_ptr

    def reset_stream(self):
        self._ck(self._lib.icpx_reset_stream(self._h))
        self._stream = -1

    def sync(self):
        self._ck(self._lib.icpx_sync(self._h))

    # ---- target / NN ----
    def set_target(self, xyz):
        p, where = _ptr(xyz, np.float32, self)
        n, stride = xyz.shape[0], 4 * xyz.shape[1]
        self._ck(self._lib.icpx_set_target(self._h, p, n, stride, where))

    def nn(self, q, want_d2=True):
        p, where = _ptr(q, np.float64, self)
        n = q.shape[0]
        if where == DEVICE:
            import torch
            idx = torch.empty(n, dtype=torch.int32, device=q.device)
            d2 = torch.empty(n, dtype=torch.float64, device=q.device) if want_d2 else None
            pi, pd = C.c_void_p(idx.data_ptr()), (C.c_void_p(d2.data_ptr()) if want_d2 else None)
        else:
            idx = np.empty(n, np.int32)
            d2 = np.empty(n, np.float64) if want_d2 else None
            pi, pd = idx.ctypes.data_as(C.c_void_p), (d2.ctypes.data_as(C.c_void_p) if want_d2 else None)
        self._ck(self._lib.icpx_nn(self._h, p, n, where, pi, pd))
        return idx, d2

    def knn(self, q, k):
        """k nearest target points per query (host arrays): idx [n, k] int32, d2 [n, k] float64, ascending."""
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), np.int32); d2 = np.empty((q.shape[0], k), np.float64)
        self._ck(self._lib.icpx_knn(self._h, q.ctypes.data_as(C.c_void_p), q.shape[0], HOST, k, idx.ctypes.data_as(C.c_void_p),
                                    d2.ctypes.data_as(C.c_void_p)))
        return idx, d2

    def nn_distance(self, q):
        p, where = _ptr(q, np.float32, self)
        n = q.shape[0]
        if where == DEVICE:
            import torch
            out = torch.empty(n, dtype=torch.float32, device=q.device); po = C.c_void_p(out.data_ptr())
        else:
            out = np.empty(n, np.float32); po = out.ctypes.data_as(C.c_void_p)
        self._ck(self._lib.icpx_nn_distance(self._h, p, n, where, po))
        return out

    # ---- ICP step ----
    def set_source(self, xyz):
        p, where = _ptr(xyz, np.float32, self)
        self._ck(self._lib.icpx_set_source(self._h, p, xyz.shape[0], 4 * xyz.shape[1], where))
        self._ns = xyz.shape[0]

    def icp_step(self, T, want_idx=False):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        sums = np.empty(NSUMS, np.float64)
        idx = np.empty(self._ns, np.int32) if want_idx else None
        self._ck(self._lib.icpx_icp_step(self._h, T.ctypes.data_as(C.c_void_p), sums.ctypes.data_as(C.c_void_p),
                                         idx.ctypes.data_as(C.c_void_p) if want_idx else None, HOST))
        return (sums, idx) if want_idx else sums

    def icp_step_trimmed(self, T, keep):
        """Sums of icp_step over the `keep` pairs with the smallest squared distances."""
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        sums = np.empty(NSUMS, np.float64)
        self._ck(self._lib.icpx_icp_step_trimmed(self._h, T.ctypes.data_as(C.c_void_p), int(keep), sums.ctypes.data_as(C.c_void_p)))
        return sums

    def icp_run(self, T0, iterations, mean_tar):
        """icpx_icp_run: `iterations` ICP iterations on the device without the host in between. Returns (finMat 4x4, the
        running target mean, trace [iterations, 18]: curMat (16), residual after the update, residual before it)."""
        T0 = np.ascontiguousarray(T0, np.float64).reshape(16)
        mt = np.ascontiguousarray(mean_tar, np.float64).reshape(3).copy()
        out = np.empty(16); trace = np.empty((max(iterations, 0), 18))
        self._ck(self._lib.icpx_icp_run(self._h, T0.ctypes.data_as(C.c_void_p), int(iterations), mt.ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p), trace.ctypes.data_as(C.c_void_p)))
        return out.reshape(4, 4), mt, trace

    def icp_step_async(self, T, sums_dev):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        p, where = _ptr(sums_dev, np.float64, self)
        assert where == DEVICE
        self._ck(self._lib.icpx_icp_step_async(self._h, T.ctypes.data_as(C.c_void_p), p))

    # ---- EDT ----
    def edt_build(self, div, box=None):
        if box is None:
            self._ck(self._lib.icpx_edt_build(self._h, div))
        else:
            mn = np.ascontiguousarray(box[0], np.float32)
            self._ck(self._lib.icpx_edt_build_box(self._h, div, mn.ctypes.data_as(C.c_void_p), C.c_float(box[1])))

    def edt_build_round(self, div, min3, scale):
        """Grid with DT3D's voxel rule: int((p - min3) * scale + 0.5)."""
        mn = np.ascontiguousarray(min3, np.float64)
        self._ck(self._lib.icpx_edt_build_round(self._h, div, mn.ctypes.data_as(C.c_void_p), C.c_double(scale)))

    def edt_lookup_round(self, q):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 3)
        out = np.empty(q.shape[0], np.float32)
        self._ck(self._lib.icpx_edt_lookup_round(self._h, q.ctypes.data_as(C.c_void_p), q.shape[0], HOST, out.ctypes.data_as(C.c_void_p)))
        return out

    def edt_rebuild_async(self):
        self._ck(self._lib.icpx_edt_rebuild_async(self._h))

    def edt_rebuild_timed(self, reps, sample_every=1):
        """`reps` rebuilds back to back; returns the average kernel milliseconds per stage (splat, census, pass zy,
        pass x) over the builds that carried events (every `sample_every`-th)."""
        ms = (C.c_float * 4)()
        n = C.c_int()
        self._ck(self._lib.icpx_edt_rebuild_timed(self._h, reps, sample_every, ms, C.byref(n)))
        return np.array(list(ms), np.float64) / max(n.value, 1)

    def edt_set_timing(self, on=True):
        self._ck(self._lib.icpx_edt_set_timing(self._h, 1 if on else 0))

    def edt_last_build(self):
        """(kernel ms by stage of the attempt that produced the grid, attempts) of the last edt_build* call made with timing on."""
        ms = (C.c_float * 4)()
        n = C.c_int()
        self._ck(self._lib.icpx_edt_last_build(self._h, ms, C.byref(n)))
        return np.array(list(ms), np.float64), n.value

    def edt_info(self):
        div = C.c_uint32(); mn = (C.c_float * 3)(); cell = C.c_float(); occ = C.c_uint64(); dr = C.c_uint64()
        self._ck(self._lib.icpx_edt_info(self._h, C.byref(div), mn, C.byref(cell), C.byref(occ), C.byref(dr)))
        return dict(div=div.value, min=np.array(list(mn), np.float32), cell=np.float32(cell.value), occupied=occ.value,
                    dropped=dr.value)

    def edt_read(self, out=None):
        div = self.edt_info()["div"]
        if out is None:
            out = np.empty((div, div, div), np.int32)
        p, where = _ptr(out, np.int32, self)
        self._ck(self._lib.icpx_edt_read(self._h, p, where))
        return out

    def edt_lookup(self, q):
        p, where = _ptr(q, np.float32, self)
        n = q.shape[0]
        if where == DEVICE:
            import torch
            out = torch.empty(n, dtype=torch.float32, device=q.device); po = C.c_void_p(out.data_ptr())
        else:
            out = np.empty(n, np.float32); po = out.ctypes.data_as(C.c_void_p)
        self._ck(self._lib.icpx_edt_lookup(self._h, p, n, where, po))
        return out

    # ---- Go-ICP ----
    def goicp_set_data(self, xyz):
        p, where = _ptr(xyz, np.float32, self)
        self._ck(self._lib.icpx_goicp_set_data(self._h, p, xyz.shape[0], 4 * xyz.shape[1], where))
        self._nd = xyz.shape[0]

    def goicp_maxrotdist(self):
        out = np.empty((MAXROTLEVEL, self._nd), np.float32)
        self._ck(self._lib.icpx_goicp_read_maxrotdist(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    @staticmethod
    def make_items(Rs, cubes, levels):
        """Rs: n x 9 float, cubes: n x 4 (x,y,z,w), levels: n ints (-1 = upper-bound form)."""
        n = len(cubes)
        arr = (BnbItem * n)()
        for i in range(n):
            for k in range(9):
                arr[i].R[k] = float(Rs[i][k])
            arr[i].x, arr[i].y, arr[i].z, arr[i].w = (float(v) for v in cubes[i])
            arr[i].level = int(levels[i])
        return arr

    def bnb_eval(self, items, want_mindis=False):
        n = len(items)
        ub = np.empty(n, np.float64); lb = np.empty(n, np.float64)
        md = np.empty((n, self._nd), np.float32) if want_mindis else None
        self._ck(self._lib.icpx_bnb_eval(self._h, C.cast(items, C.c_void_p), n, ub.ctypes.data_as(C.c_void_p),
                                         lb.ctypes.data_as(C.c_void_p), md.ctypes.data_as(C.c_void_p) if want_mindis else None))
        return (ub, lb, md) if want_mindis else (ub, lb)

    def bnb_eval_async(self, items, out_dev, first=0, count=None):
        """icpx_bnb_eval_async: this context's share [first, first + count) of the batch into out_dev (CUDA float64, 2 * len(items),
        [ub..., lb...], zeros for the other items). Only enqueues on the context's stream."""
        n = len(items)
        count = n - first if count is None else count
        self._ck(self._lib.icpx_bnb_eval_async(self._h, C.cast(items, C.c_void_p), n, first, count, _dev(out_dev, self)))

    def bnb_search(self, Rs, levels, opt_errors, cubes, sse_thresh):
        """icpx_bnb_search_run: one inner search per (rotation, level, starting error, starting cube), run on the device to the
        end. Returns (opt_error float32[n], have_best bool[n], best float32[n, 4], cubes int[n], status int[n], steps)."""
        from .lib import BnbFound, BnbSearch
        n = len(levels)
        arr = (BnbSearch * n)()
        for i in range(n):
            arr[i].R[:] = [float(v) for v in np.asarray(Rs[i], np.float32).reshape(9)]
            arr[i].level = int(levels[i]); arr[i].opt_error = float(opt_errors[i])
            arr[i].cube[:] = [float(v) for v in np.asarray(cubes[i], np.float32).reshape(4)]
        out = (BnbFound * n)()
        steps = C.c_longlong(0)
        self._ck(self._lib.icpx_bnb_search_run(self._h, C.cast(arr, C.c_void_p), n, C.c_float(sse_thresh), C.cast(out, C.c_void_p), C.byref(steps)))
        return (np.array([o.opt_error for o in out], np.float32), np.array([bool(o.have_best) for o in out]),
                np.array([list(o.best) for o in out], np.float32).reshape(n, 4), np.array([o.cubes for o in out]),
                np.array([o.status for o in out]), steps.value)

    def bnb_submit(self, slot, parents, first=0, count=None, out_dev=None):
        """icpx_bnb_submit: the eight children of every parent cube in `parents` (make_items records), as batch `slot`."""
        n = len(parents)
        count = n - first if count is None else count
        self._ck(self._lib.icpx_bnb_submit(self._h, slot, C.cast(parents, C.c_void_p), n, first, count,
                                           _dev(out_dev, self) if out_dev is not None else None))
        self._pending = getattr(self, "_pending", {})
        self._pending[slot] = (n, out_dev is None)

    def bnb_fetch(self, slot, out_dev):
        """icpx_bnb_fetch: the bounds a batch left in out_dev (summed over the ranks meanwhile) come to the host after all."""
        self._ck(self._lib.icpx_bnb_fetch(self._h, slot, _dev(out_dev, self)))
        self._pending[slot] = (self._pending[slot][0], True)

    def bnb_collect(self, slot):
        """icpx_bnb_collect: waits for batch `slot`; returns (ub, lb), 8 per parent, when the bounds come to the host."""
        n, to_host = self._pending.pop(slot)
        if not to_host:
            self._ck(self._lib.icpx_bnb_collect(self._h, slot, None, None))
            return None
        ub = np.empty(8 * n, np.float64); lb = np.empty(8 * n, np.float64)
        self._ck(self._lib.icpx_bnb_collect(self._h, slot, ub.ctypes.data_as(C.c_void_p), lb.ctypes.data_as(C.c_void_p)))
        return ub, lb

    def download(self, dev_tensor, nbytes=None):
        """icpx_download: waits for the context's stream, returns the tensor's bytes as a numpy array of its dtype."""
        out = np.empty(tuple(dev_tensor.shape), {8: np.float64, 4: np.int32}[dev_tensor.element_size()])
        self._ck(self._lib.icpx_download(self._h, _dev(dev_tensor, self), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def edt_grid_ptr(self):
        p = C.c_void_p()
        self._ck(self._lib.icpx_edt_grid_ptr(self._h, C.byref(p)))
        return p.value

    def edt_grid_stride(self):
        """icpx_edt_grid_stride: values between consecutive rows of the grid behind edt_grid_ptr."""
        v = C.c_uint32()
        self._ck(self._lib.icpx_edt_grid_stride(self._h, C.byref(v)))
        return v.value

    def bnb_eval_noalloc(self, items_dev, nitems, out_dev):
        """icpx_bnb_eval_noalloc: items_dev a CUDA uint8 tensor holding `nitems` packed icpx_bnb_item records, out_dev a
        CUDA float64 tensor of 2 * nitems ([ub..., lb...]). Only enqueues on the context's stream: sync() before reading."""
        self._ck(self._lib.icpx_bnb_eval_noalloc(self._h, _dev(items_dev, self), nitems, _dev(out_dev, self)))

    def goicp_residuals(self, T):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        out = np.empty(self._nd, np.float32)
        self._ck(self._lib.icpx_goicp_residuals(self._h, T.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out
