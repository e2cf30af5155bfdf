"""Multi-GPU model: the candidate set of the multi-restart search (reference src/acquisition.jl:54-68) sharded over a
device list inside ONE process, through the ``bohip_mgp_*`` entry points of libbohip (in-library RCCL: ncclCommInitAll,
one ncclAllGather of the 16-byte arg-max records, device-side reduce).  Same surface as ``ElasticGPE`` where it applies.

The one-process-per-GPU form (``torch.distributed.run``) uses ``ElasticGPE.comm_init`` / ``score_sharded_dev`` instead;
``dist.py`` keeps the torch.distributed fallback used by the CPU (gloo) tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Best, check
from .model import MeanZero, SEArd, _cols, _ptr


def _params(params):
    p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
    if p.size < 2:
        p = np.concatenate([p, np.zeros(2 - p.size)])
    return p


class MultiGPE:
    """``MultiGPE(d, devices=[0, 1, ...], shards_per_device=1; mean, kernel, logNoise, capacity)``: one replica of the
    GP per device (x, y and the hyper-parameters are broadcast; every device factors redundantly), candidates sharded
    contiguously, winners exchanged over RCCL.  ``score`` returns the GLOBAL (value, index), bit-identical to
    ``ElasticGPE.score`` on the whole set."""

    def __init__(self, d, devices=(0,), shards_per_device=1, mean=None, kernel=None, logNoise=-2.0, capacity=1024):
        self.dim = int(d)
        self.mean = mean if mean is not None else MeanZero()
        self.kernel = kernel if kernel is not None else SEArd(np.zeros(d), 0.0)
        self.logNoise = float(logNoise)
        self.devices = [int(v) for v in devices]
        self.shards_per_device = int(shards_per_device)
        self._lib = _lib.load()
        h = C.c_void_p()
        devs = (C.c_int * len(self.devices))(*self.devices)
        check(self._lib.bohip_mgp_create(self.dim, int(capacity), _lib.KERN[self.kernel.kern], devs, len(self.devices),
                                         self.shards_per_device, C.byref(h)))
        self._h = h
        _lib.register(self)
        self._x = np.zeros((self.dim, 0), order="F")
        self._y = np.zeros(0)
        self._push_hyper()

    def close(self):
        """Release every replica, the communicators and the worker threads now (idempotent; run at interpreter exit too)."""
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.bohip_mgp_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _push_hyper(self):
        ll = self.kernel.ll
        ll = np.ascontiguousarray(np.broadcast_to(ll, (self.dim,)) if ll.size == 1 else ll)
        check(self._lib.bohip_mgp_set_hyper(self._h, _ptr(ll), self.kernel.lsigma, self.logNoise, self.mean.beta))

    x = property(lambda self: self._x)
    y = property(lambda self: self._y)
    nobs = property(lambda self: self._y.size)

    @property
    def n_shards(self):
        return self.info(_lib.MGP_INFO_SHARDS)

    def info(self, what):
        v = C.c_int64()
        check(self._lib.bohip_mgp_info(self._h, what, C.byref(v)))
        return v.value

    def append_(self, x, y):
        x = _cols(x, self.dim)
        y = np.ascontiguousarray(np.atleast_1d(np.asarray(y, dtype=np.float64)))
        if x.shape[1] != y.size:
            raise ValueError("x and y disagree on the number of observations")
        rc = self._lib.bohip_mgp_append(self._h, _ptr(x), _ptr(y), y.size)
        if rc in (_lib.OK, _lib.E_NOTPD):
            self._x = np.asfortranarray(np.concatenate([self._x, x], axis=1))
            self._y = np.concatenate([self._y, y])
        check(rc)
        return self

    def fit_(self):
        check(self._lib.bohip_mgp_refit(self._h))
        return self

    def score(self, acq, params, xs, want_scores=True):
        xs = _cols(xs, self.dim)
        R = xs.shape[1]
        p = _params(params)
        sc = np.empty(R) if want_scores else None
        best = Best()
        check(self._lib.bohip_mgp_score(self._h, _lib.ACQ[acq], _ptr(p), _ptr(xs), R, _ptr(sc) if want_scores else None,
                                        C.byref(best)))
        return sc, best.val, best.idx

    def set_candidates(self, xs):
        xs = _cols(xs, self.dim)
        check(self._lib.bohip_mgp_set_candidates(self._h, _ptr(xs), xs.shape[1]))

    def score_resident(self, acq, params):
        p = _params(params)
        best = Best()
        check(self._lib.bohip_mgp_score_resident(self._h, _lib.ACQ[acq], _ptr(p), C.byref(best)))
        return best.val, best.idx

    def thompson(self, xs, S, seed=0):
        xs = _cols(xs, self.dim)
        out = (Best * S)()
        check(self._lib.bohip_mgp_thompson(self._h, _ptr(xs), xs.shape[1], S, seed, out))
        return np.array([b.val for b in out]), np.array([b.idx for b in out], dtype=np.int64)

    def set_maxtime(self, seconds):
        """NLopt's maxtime for the device ascent, on every replica (0 = unlimited)."""
        check(self._lib.bohip_mgp_set_maxtime(self._h, float(seconds)))

    def set_ascent_stop(self, ftol_abs=0.0, xtol_rel=0.0, stopval=float("inf")):
        """NLopt's ftol_abs / xtol_rel / stopval for the device ascent, on every replica"""
        check(self._lib.bohip_mgp_set_ascent_stop(self._h, float(ftol_abs), float(xtol_rel), float(stopval)))

    def set_jitter(self, rel, max_tries=10):
        """see ElasticGPE.set_jitter; applied to every replica"""
        check(self._lib.bohip_mgp_set_jitter(self._h, float(rel), int(max_tries)))

    def ascend(self, acq, params, lowerbounds, upperbounds, starts, maxeval=2000, ftol_rel=1e-10, xtol_abs=1e-10):
        starts = _cols(starts, self.dim)
        R = starts.shape[1]
        p = _params(params)
        lb = np.ascontiguousarray(lowerbounds, dtype=np.float64)
        ub = np.ascontiguousarray(upperbounds, dtype=np.float64)
        f = np.empty(R)
        X = np.empty((self.dim, R), order="F")
        best = Best()
        bx = np.empty(self.dim)
        ev = C.c_int64(0)
        check(self._lib.bohip_mgp_acquire_max(self._h, _lib.ACQ[acq], _ptr(p), _ptr(lb), _ptr(ub), _ptr(starts), R, int(maxeval),
                                              float(ftol_rel), float(xtol_abs), _ptr(X), _ptr(f), C.byref(best), _ptr(bx),
                                              C.byref(ev)))
        return f, X, best.val, best.idx, bx, ev.value

    def direct_max(self, acq, params, lowerbounds, upperbounds, maxeval=2000, stopval=float("inf"), maxtime=0.0, seed=0):
        """:GN_DIRECT_L in one library call (ElasticGPE.direct_max, bohip_gp_direct_max) on the FIRST replica: an iteration of the search is a
        handful of points that depend on the iteration before, so there is nothing to shard -- every device holds the whole model."""
        lb = np.ascontiguousarray(lowerbounds, dtype=np.float64)
        ub = np.ascontiguousarray(upperbounds, dtype=np.float64)
        if lb.size != self.dim or ub.size != self.dim:
            raise ValueError("bounds must have one entry per input dimension")
        p = np.zeros(2)
        if params is not None:
            q = np.atleast_1d(np.asarray(params, dtype=np.float64))
            p[:q.size] = q[:2]
        g = self._lib.bohip_mgp_handle(self._h, 0)
        bf = C.c_double(); bx = np.empty(self.dim); ev = C.c_int64(); dc = C.c_int64()
        check(self._lib.bohip_gp_direct_max(g, _lib.ACQ[acq], _ptr(p), _ptr(lb), _ptr(ub), int(maxeval), float(stopval),
                                            float(maxtime or 0.0), int(seed), C.byref(bf), _ptr(bx), C.byref(ev), C.byref(dc)))
        return float(bf.value), bx, int(ev.value), int(dc.value)

    def replica_factor(self, i):
        """Cholesky factor held by the i-th device (tests: every replica is the same model)."""
        g = self._lib.bohip_mgp_handle(self._h, i)
        L = np.zeros((self.nobs, self.nobs))
        check(self._lib.bohip_gp_get_factor(g, _ptr(L)))
        return L


def comm_unique_id():
    """The bytes rank 0 ships to the other ranks (ncclUniqueId) before ``ElasticGPE.comm_init``."""
    buf = (C.c_char * _lib.UNIQUE_ID_BYTES)()
    check(_lib.load().bohip_comm_unique_id(buf, _lib.UNIQUE_ID_BYTES))
    return bytes(buf)
