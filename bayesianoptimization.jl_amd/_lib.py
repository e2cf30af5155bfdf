"""ctypes binding of libbohip.so (include/bohip.h).  No CPU fallback: importing works without a
GPU (so the ABI can be inspected), but every compute entry point raises when the library or the
device is missing."""
from __future__ import annotations

import atexit
import ctypes as C
import importlib.util
import os
import sys
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOHIP_LIB") or os.path.join(_HERE, "csrc", "libbohip.so")   # BOHIP_LIB: measurement builds (csrc/abl)

OK, E_ARG, E_NOTPD, E_HIP, E_NODEVICE, E_STATE, E_UNSUPPORTED, E_COMM = 0, -1, -2, -3, -4, -5, -6, -7
KERN = {"SEArd": 0, "SEIso": 1, "Mat52Ard": 2}
ACQ = {"EI": 0, "PI": 1, "UCB": 2, "MI": 3, "MaxMean": 4, "ThompsonDraw": 5}   # (5: bohip_gp_direct_max only)
INFO_PIVOT, INFO_CAPACITY, INFO_REFITS, INFO_APPENDS = 0, 1, 2, 3
INFO_CHOL_FORM, INFO_CHOL_FALLBACKS, INFO_CHOL_ABORT_TILES, INFO_JITTER_STEPS = 4, 5, 6, 7
INFO_SCORE_LAUNCHES, INFO_SCORE_CHUNK, INFO_KERNEL_CLOCK_MHZ = 8, 9, 10
INFO_COMM_NRANKS, INFO_COMM_EXCHANGES, INFO_COMM_RCCL_VERSION, INFO_CHOL_LOCK_SKIPS = 11, 12, 13, 14
MGP_INFO_DEVICES, MGP_INFO_SHARDS, MGP_INFO_EXCHANGES, MGP_INFO_RCCL_VERSION, MGP_INFO_COMM_NRANKS = 0, 1, 2, 3, 4
UNIQUE_ID_BYTES = 128


class Best(C.Structure):
    _fields_ = [("val", C.c_double), ("idx", C.c_int64)]


class BohipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbohip error {code}: {msg}")
        self.code = code


class NotPositiveDefinite(BohipError):
    pass


_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_gp = C.c_void_p
_mgp = C.c_void_p
_ip = C.POINTER(C.c_int)

# every symbol include/bohip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "bohip_gp_create": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(_gp)]),
    "bohip_gp_destroy": (None, [_gp]),
    "bohip_gp_set_hyper": (C.c_int, [_gp, _dp, C.c_double, C.c_double, C.c_double]),
    "bohip_gp_append": (C.c_int, [_gp, _dp, _dp, C.c_int64]),
    "bohip_gp_refit": (C.c_int, [_gp]),
    "bohip_gp_dims": (C.c_int, [_gp, _i64p, _i64p]),
    "bohip_gp_maxy": (C.c_int, [_gp, _dp]),
    "bohip_gp_get_xy": (C.c_int, [_gp, _dp, _dp]),
    "bohip_gp_mll": (C.c_int, [_gp, _dp]),
    "bohip_gp_mll_grad": (C.c_int, [_gp, _dp, _dp, _dp, _dp]),
    "bohip_gp_set_batch_hint": (C.c_int, [_gp, C.c_int64]),
    "bohip_gp_predict_cov": (C.c_int, [_gp, _dp, C.c_int64, _dp, _dp]),
    "bohip_gp_acquire_max": (C.c_int, [_gp, C.c_int, _dp, _dp, _dp, _dp, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                      _dp, _dp, C.POINTER(Best), _dp, _i64p]),
    "bohip_gp_set_maxtime": (C.c_int, [_gp, C.c_double]),
    "bohip_gp_set_ascent_stop": (C.c_int, [_gp, C.c_double, C.c_double, C.c_double]),
    "bohip_gp_set_jitter": (C.c_int, [_gp, C.c_double, C.c_int]),
    "bohip_gp_predict": (C.c_int, [_gp, _dp, C.c_int64, _dp, _dp]),
    "bohip_gp_score": (C.c_int, [_gp, C.c_int, _dp, _dp, C.c_int64, _dp, C.POINTER(Best)]),
    "bohip_gp_score_grad": (C.c_int, [_gp, C.c_int, _dp, _dp, C.c_int64, _dp, _dp]),
    "bohip_gp_thompson": (C.c_int, [_gp, _dp, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.POINTER(Best)]),
    "bohip_thompson_normal": (C.c_double, [C.c_uint64, C.c_int64, C.c_int64]),
    "bohip_direct_create": (C.c_int, [C.c_int64, _dp, _dp, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "bohip_direct_destroy": (None, [C.c_void_p]),
    "bohip_direct_ask": (C.c_int, [C.c_void_p, _dp, C.c_int64, _i64p]),
    "bohip_direct_tell": (C.c_int, [C.c_void_p, _dp, C.c_int64]),
    "bohip_direct_best": (C.c_int, [C.c_void_p, _dp, _dp, _i64p, _i64p]),
    "bohip_gp_direct_max": (C.c_int, [_gp, C.c_int, _dp, _dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_uint64, _dp, _dp,
                                     _i64p, _i64p]),
    "bohip_gp_score_dev": (C.c_int, [_gp, C.c_int, _dp, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bohip_gp_predict_dev": (C.c_int, [_gp, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bohip_gp_set_stream": (C.c_int, [_gp, C.c_void_p]),
    "bohip_gp_synchronize": (C.c_int, [_gp]),
    "bohip_gp_get_factor": (C.c_int, [_gp, _dp]),
    "bohip_gp_get_alpha": (C.c_int, [_gp, _dp]),
    "bohip_gp_info": (C.c_int, [_gp, C.c_int, _i64p]),
    "bohip_debug_set_chol_inv_g": (C.c_int, [C.c_int]),
    "bohip_gp_enable_timing": (C.c_int, [_gp, C.c_int]),
    "bohip_gp_get_timing": (C.c_int, [_gp, C.POINTER(C.c_char_p), _dp, C.c_int]),
    # multi-GPU: one process, a device list (in-library RCCL)
    "bohip_mgp_create": (C.c_int, [C.c_int64, C.c_int64, C.c_int, _ip, C.c_int, C.c_int, C.POINTER(_mgp)]),
    "bohip_mgp_destroy": (None, [_mgp]),
    "bohip_mgp_set_hyper": (C.c_int, [_mgp, _dp, C.c_double, C.c_double, C.c_double]),
    "bohip_mgp_append": (C.c_int, [_mgp, _dp, _dp, C.c_int64]),
    "bohip_mgp_refit": (C.c_int, [_mgp]),
    "bohip_mgp_score": (C.c_int, [_mgp, C.c_int, _dp, _dp, C.c_int64, _dp, C.POINTER(Best)]),
    "bohip_mgp_set_candidates": (C.c_int, [_mgp, _dp, C.c_int64]),
    "bohip_mgp_score_resident": (C.c_int, [_mgp, C.c_int, _dp, C.POINTER(Best)]),
    "bohip_mgp_thompson": (C.c_int, [_mgp, _dp, C.c_int64, C.c_int64, C.c_uint64, C.POINTER(Best)]),
    "bohip_mgp_acquire_max": (C.c_int, [_mgp, C.c_int, _dp, _dp, _dp, _dp, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                       _dp, _dp, C.POINTER(Best), _dp, _i64p]),
    "bohip_mgp_set_maxtime": (C.c_int, [_mgp, C.c_double]),
    "bohip_mgp_set_ascent_stop": (C.c_int, [_mgp, C.c_double, C.c_double, C.c_double]),
    "bohip_mgp_set_jitter": (C.c_int, [_mgp, C.c_double, C.c_int]),
    "bohip_mgp_handle": (_gp, [_mgp, C.c_int]),
    "bohip_mgp_info": (C.c_int, [_mgp, C.c_int, _i64p]),
    # multi-GPU: one process per device
    "bohip_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int64]),
    "bohip_gp_comm_init": (C.c_int, [_gp, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "bohip_gp_comm_destroy": (C.c_int, [_gp]),
    "bohip_gp_score_sharded_dev": (C.c_int, [_gp, C.c_int, _dp, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                            C.c_void_p]),
    "bohip_gp_thompson_sharded": (C.c_int, [_gp, _dp, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_int64,
                                           C.POINTER(Best)]),
    "bohip_last_error": (C.c_char_p, []),
    "bohip_version": (C.c_char_p, []),
    "bohip_device_count": (C.c_int, []),
}

_lib = None

# Live device objects are closed at interpreter exit BEFORE the HIP / RCCL runtimes run their own static destructors:
# a handle finalised after that (e.g. one kept alive by a traceback until shutdown) would free into a dead runtime.
_live = weakref.WeakSet()


def register(obj):
    _live.add(obj)


@atexit.register
def _close_all():
    for o in list(_live):
        try:
            o.close()
        except Exception:
            pass


def _one_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
    /opt/rocm's); whichever is dlopen'ed first wins for both users, and torch cannot see the GPU when
    the system copy got there first.  So when PyTorch is installed (it is the plumbing for device
    memory, streams and the process group in bench.py / dist.py) and not yet imported, load its copy
    first.  BOHIP_SYSTEM_HIP=1 skips this.  RCCL needs no such care: libbohip binds it with
    dlopen("librccl.so.1") at the first multi-GPU call, which returns the copy the process already holds
    (preloading torch's librccl here made the process abort in a static destructor at exit)."""
    if "torch" in sys.modules or os.environ.get("BOHIP_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    for name in ("libamdhip64.so",):
        cand = os.path.join(os.path.dirname(spec.origin), "lib", name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load():
    """dlopen libbohip.so and bind every declared symbol.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BohipError(E_NODEVICE, f"{LIB_PATH} not built (run __graft_entry__.build()); there is no CPU fallback")
    _one_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == OK:
        return
    msg = load().bohip_last_error().decode()
    if rc == E_NOTPD:
        raise NotPositiveDefinite(rc, msg)
    raise BohipError(rc, msg)
