"""ctypes wrapper of the C oracle (oracle/_build/libdcoracle.so) and the CPU NUTS
harness (oracle/_build/libnuts_harness.so).  TEST INFRASTRUCTURE -- only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

import dc_oracle as O

_DIR = os.path.dirname(os.path.abspath(__file__))
_lib = None
_har = None


def _build():
    subprocess.run(["make", "-C", _DIR], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(_DIR, "_build", "libdcoracle.so")
        if not os.path.exists(p):
            _build()
        _lib = C.CDLL(p)
        _lib.dco_potential_grad.restype = C.c_int
        _lib.dco_potential_grad.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 10 + [C.c_int]
        _lib.dco_max_threads.restype = C.c_int
    return _lib


def harness():
    global _har
    if _har is None:
        p = os.path.join(_DIR, "_build", "libnuts_harness.so")
        if not os.path.exists(p):
            _build()
        _har = C.CDLL(p)
    return _har


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CFixtures:
    """Arrays in the dtypes the C functions take (kept alive here)."""

    def __init__(self, model, fx: O.Fixtures):
        self.model = model
        self.T = fx.n_teams
        self.K = fx.k if model == O.MODEL_EXTENDED else 0
        self.n = fx.n
        self.h = np.ascontiguousarray(fx.home_idx, dtype=np.uint16)
        self.a = np.ascontiguousarray(fx.away_idx, dtype=np.uint16)
        self.x = np.ascontiguousarray(fx.home_goals, dtype=np.uint8)
        self.y = np.ascontiguousarray(fx.away_goals, dtype=np.uint8)
        self.w = None if fx.weights is None else np.ascontiguousarray(fx.weights, dtype=np.float64)
        self.xs = None
        if self.K:
            self.xs = np.ascontiguousarray(O.standardise_covariates(fx.covariates))
        self.D = O.latent_dim(model, self.T, self.K)


def potential_and_grad(cf: CFixtures, z, nthreads=1):
    z = np.ascontiguousarray(z, dtype=np.float64)
    U = np.zeros(1)
    g = np.zeros(cf.D)
    aux = np.zeros(4)
    rc = lib().dco_potential_grad(cf.model, cf.n, cf.T, cf.K, _p(cf.h), _p(cf.a), _p(cf.x),
                                  _p(cf.y), _p(cf.w), _p(cf.xs), _p(z), _p(U), _p(g), _p(aux),
                                  nthreads)
    assert rc == 0
    return float(U[0]), g, aux


def max_threads():
    return lib().dco_max_threads()


def nuts_dc(cf: CFixtures, warm, samp, key, depth=10, thin=1, z0=None, step_size=1.0):
    kept = samp // thin
    draws = np.zeros((kept, cf.D))
    stats = np.zeros((kept, 4))
    summ = np.zeros(4 + cf.D)
    z0 = None if z0 is None else np.ascontiguousarray(z0, dtype=np.float64)
    f = harness().harness_nuts_dc
    f.restype = C.c_int
    f.argtypes = ([C.c_int, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 4 +
                  [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 3 + [C.c_double])
    rc = f(cf.model, cf.n, cf.T, cf.K, _p(cf.h), _p(cf.a), _p(cf.x), _p(cf.y), _p(cf.w),
           _p(cf.xs), warm, samp, depth, thin, _p(z0), key[0], key[1], _p(draws), _p(stats),
           _p(summ), float(step_size))
    return rc, draws, stats, summ


def nuts_gauss(sd, warm, samp, key, depth=10, thin=1, z0=None, step_size=1.0):
    sd = np.ascontiguousarray(sd, dtype=np.float64)
    D = sd.size
    kept = samp // thin
    draws = np.zeros((kept, D))
    stats = np.zeros((kept, 4))
    summ = np.zeros(4 + D)
    z0 = None if z0 is None else np.ascontiguousarray(z0, dtype=np.float64)
    f = harness().harness_nuts_gauss
    f.restype = C.c_int
    f.argtypes = ([C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_uint32, C.c_uint32] +
                  [C.c_void_p] * 3 + [C.c_double])
    rc = f(D, _p(sd), warm, samp, depth, thin, _p(z0), key[0], key[1], _p(draws), _p(stats),
           _p(summ), float(step_size))
    return rc, draws, stats, summ


# ---------------------------------------------------------------- the timed CPU comparator
_port = None


def port_lib():
    global _port
    if _port is None:
        p = os.path.join(_DIR, "_build", "libdccpuport.so")
        if not os.path.exists(p):
            _build()
        _port = C.CDLL(p)
        _port.dcp_create.restype = C.c_void_p
        _port.dcp_create.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int]
        _port.dcp_destroy.argtypes = [C.c_void_p]
        _port.dcp_eval.argtypes = [C.c_void_p] * 5
        _port.dcp_eval_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _port.dcp_eval_many.restype = C.c_double
        _port.dcp_threads.argtypes = [C.c_void_p]
    return _port


class CpuPort:
    """oracle/dc_cpu_port.c: the HIP kernel's algorithm on host cores (float32 per-fixture
    arithmetic, float64 accumulation, OpenMP, no allocation per evaluation)."""

    def __init__(self, cf: CFixtures, nthreads=0):
        self.cf = cf
        self.w32 = None if cf.w is None else np.ascontiguousarray(cf.w, dtype=np.float32)
        self.h = port_lib().dcp_create(cf.model, cf.n, cf.T, cf.K, _p(cf.h), _p(cf.a), _p(cf.x), _p(cf.y),
                                       _p(self.w32), _p(cf.xs), int(nthreads))
        assert self.h
        self.threads = port_lib().dcp_threads(self.h)

    def eval(self, z):
        z = np.ascontiguousarray(z, dtype=np.float64)
        U, g, aux = np.zeros(1), np.zeros(self.cf.D), np.zeros(4)
        port_lib().dcp_eval(self.h, _p(z), _p(U), _p(g), _p(aux))
        return float(U[0]), g, aux

    def eval_many(self, zs, count):
        zs = np.ascontiguousarray(zs, dtype=np.float64)
        U, g = np.zeros(1), np.zeros(self.cf.D)
        port_lib().dcp_eval_many(self.h, _p(zs), zs.shape[0], int(count), _p(U), _p(g))
        return float(U[0]), g

    def close(self):
        if self.h:
            port_lib().dcp_destroy(self.h)
            self.h = None
