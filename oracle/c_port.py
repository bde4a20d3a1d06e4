"""ctypes wrapper of oracle/libcheb_ref.so (the plain-C restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libcheb_ref.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        h = C.CDLL(_PATH)
        P = C.c_void_p
        h.cheb_ref_threads.restype = C.c_int
        h.cheb_ref_hop.restype = None
        h.cheb_ref_hop.argtypes = [C.c_int64, P, P, P, C.c_int64, C.c_int64, P, P, C.c_float, C.c_float, P, P]
        h.cheb_ref_forward.restype = C.c_int
        h.cheb_ref_forward.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, P, P, P, P, P, P,
                                       C.c_int, P, P, P]
        _lib = h
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def threads():
    return lib().cheb_ref_threads()


def hop(rowptr, col, val, x, z=None, alpha=1.0, beta=0.0):
    nb, n, Cw = x.shape
    y = np.empty_like(x)
    lib().cheb_ref_hop(n, _p(rowptr), _p(col), _p(val), nb, Cw, _p(x), _p(z), alpha, beta, _p(y), None)
    return y


def forward(mode, rowptr, col, val, x3, W, bias, bias_kind):
    """x3 (q,n,C) fp32; W (K,C,N) in the reference basis; returns (q,n,N)."""
    q, n, Cw = x3.shape
    K, _, N = W.shape
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    col = np.ascontiguousarray(col, np.int32)
    val = np.ascontiguousarray(val, np.float32)
    x3 = np.ascontiguousarray(x3, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    out = np.empty((q, n, N), np.float32)
    stack = np.empty(K * q * n * Cw, np.float32)
    prod = np.empty(2 * q * n * Cw, np.float32)
    lib().cheb_ref_forward(mode, K, q, n, Cw, N, _p(rowptr), _p(col), _p(val), _p(x3), _p(W), _p(b), bias_kind, _p(out),
                           _p(stack), _p(prod))
    return out
