"""ctypes loader for the native libraries of icp_amd.

libicpx.so       HIP kernels + the C ABI declared in include/icpx.h
libicpx_host.so  C++ host classes (KDTree, LDT, SingleThreadIcp, GoIcp) over that C ABI

Both are built in-tree (icp_amd/csrc/Makefile, or __graft_entry__.build()). There is no CPU fallback:
a missing library is an error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_icpx = None
_host = None


class IcpxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"icpx error {code}: {msg}")
        self.code = code


class BnbItem(C.Structure):
    """icpx_bnb_item of include/icpx.h."""
    _fields_ = [("R", C.c_float * 9), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float),
                ("level", C.c_int32)]


class BnbSearch(C.Structure):
    """icpx_bnb_search of include/icpx.h."""
    _fields_ = [("R", C.c_float * 9), ("level", C.c_int32), ("opt_error", C.c_float), ("cube", C.c_float * 4)]


class BnbFound(C.Structure):
    """icpx_bnb_found of include/icpx.h."""
    _fields_ = [("opt_error", C.c_float), ("have_best", C.c_int32), ("best", C.c_float * 4), ("cubes", C.c_int32), ("status", C.c_int32)]


# name -> (restype, argtypes); every symbol include/icpx.h declares
ICPX_SYMBOLS = {
    "icpx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "icpx_destroy": (None, [C.c_void_p]),
    "icpx_last_error": (C.c_char_p, [C.c_void_p]),
    "icpx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "icpx_reset_stream": (C.c_int, [C.c_void_p]),
    "icpx_sync": (C.c_int, [C.c_void_p]),
    "icpx_device": (C.c_int, [C.c_void_p]),
    "icpx_set_target": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "icpx_nn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "icpx_knn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "icpx_nn_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "icpx_set_source": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "icpx_icp_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "icpx_icp_step_trimmed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "icpx_icp_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "icpx_icp_step_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "icpx_edt_build": (C.c_int, [C.c_void_p, C.c_uint32]),
    "icpx_edt_build_box": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_float]),
    "icpx_edt_build_round": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_double]),
    "icpx_edt_lookup_round": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "icpx_edt_rebuild_async": (C.c_int, [C.c_void_p]),
    "icpx_edt_rebuild_timed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "icpx_edt_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "icpx_edt_last_build": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "icpx_edt_info": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "icpx_edt_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "icpx_edt_grid_stride": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "icpx_edt_grid_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "icpx_edt_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "icpx_goicp_set_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "icpx_goicp_read_maxrotdist": (C.c_int, [C.c_void_p, C.c_void_p]),
    "icpx_bnb_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "icpx_bnb_eval_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "icpx_bnb_search_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.POINTER(C.c_longlong)]),
    "icpx_bnb_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "icpx_bnb_collect": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "icpx_bnb_fetch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "icpx_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "icpx_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "icpx_bnb_eval_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "icpx_goicp_residuals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
}


def _load(name):
    # ICPX_LIB_DIR: load the libraries from another directory (A/B runs of two builds, tools/edt_ab.py)
    path = os.path.join(os.environ.get("ICPX_LIB_DIR", _HERE), name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C icp_amd/csrc` "
                          f"(or __graft_entry__.build()); icp_amd has no CPU fallback")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def icpx():
    """The HIP library with argument types attached."""
    global _icpx
    if _icpx is None:
        lib = _load("libicpx.so")
        for name, (res, args) in ICPX_SYMBOLS.items():
            fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _icpx = lib
    return _icpx


def host():
    """The C++ host-class library (its extern "C" test/driver surface, see csrc/host/icpx_host.hpp)."""
    global _host
    if _host is None:
        icpx()
        _host = _load("libicpx_host.so")
    return _host
