"""ctypes loader for oracle/librbf_oracle.so (checker / cpu_baseline only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librbf_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_phi.restype = ctypes.c_double
        L.orc_phi.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.orc_psi.restype = ctypes.c_double
        L.orc_psi.argtypes = L.orc_phi.argtypes
        L.orc_gram.restype = None
        L.orc_gram.argtypes = [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                               ctypes.c_int, _dp, _dp]
        L.orc_gram_mt.restype = None
        L.orc_gram_mt.argtypes = L.orc_gram.argtypes + [ctypes.c_int]
        L.orc_gram_cols.restype = None
        L.orc_gram_cols.argtypes = [ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, _dp]
        L.orc_fit.restype = ctypes.c_int
        L.orc_fit.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_double,
                              ctypes.c_double, ctypes.c_int, _dp, _dp]
        L.orc_eval_loop.restype = None
        L.orc_eval_loop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def gram(C, kid, a, b, deg, threads=1):
    """Phi, Pi with the per-pair norm(x - c) loop; threads > 1 spreads the columns over OpenMP threads (same arithmetic)."""
    C = np.ascontiguousarray(C, dtype=np.float64)
    n, d = C.shape
    q = 0 if deg < 0 else (1 if deg == 0 else d + 1)
    Phi = np.empty((n, n), order="F")
    Pi = np.empty((n, max(q, 1)), order="F")
    lib().orc_gram_mt(n, d, _p(C), kid, a, b, deg, _p(Phi), _p(Pi), int(threads))
    return Phi, Pi[:, :q]


def gram_cols(C, kid, a, b, cols):
    """first `cols` columns of Phi, faithful single-threaded per-pair loop (bounded sample for the CPU baseline)"""
    C = np.ascontiguousarray(C, dtype=np.float64)
    n, d = C.shape
    out = np.empty((n, cols), order="F")
    lib().orc_gram_cols(n, d, _p(C), kid, a, b, int(cols), _p(out))
    return out


def fit(C, Y, kid, a, b, deg):
    C = np.ascontiguousarray(C, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(C.shape[0], -1)
    n, d = C.shape
    k = Y.shape[1]
    q = 0 if deg < 0 else (1 if deg == 0 else d + 1)
    W = np.empty((n, k))
    Lam = np.empty((max(q, 1), k))
    info = lib().orc_fit(n, d, k, _p(C), _p(Y), kid, a, b, deg, _p(W), _p(Lam))
    return W, Lam[:q], info


def eval_loop(C, W, Lam, kid, a, b, deg, X, want_jac=True):
    C = np.ascontiguousarray(C, dtype=np.float64)
    W = np.ascontiguousarray(W, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = C.shape
    k = W.shape[1]
    m = X.shape[0]
    LamB = np.ascontiguousarray(Lam, dtype=np.float64) if Lam.size else np.zeros((1, k))
    vals = np.empty((m, k))
    jac = np.empty((m, d, k)) if want_jac else None  # per point k x d column-major
    lib().orc_eval_loop(n, d, k, _p(C), _p(W), _p(LamB), kid, a, b, deg, m, _p(X), _p(vals),
                        _p(jac) if want_jac else None)
    return vals, (np.transpose(jac, (0, 2, 1)) if want_jac else None)
