"""ctypes loader of the C restatement (oracle/ldw_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libldw_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = C.CDLL(_SO)
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


def max_threads() -> int:
    return int(lib().orc_max_threads())


def mi_block(states, hdw, r, uqe, from_idx, to_idx, ncores=None) -> np.ndarray:
    st = np.ascontiguousarray(states, dtype=np.uint8)
    L, N = st.shape
    hdw = np.ascontiguousarray(hdw, dtype=np.float64)
    r = np.ascontiguousarray(r, dtype=np.float64)
    uqe = np.ascontiguousarray(uqe, dtype=np.float64)
    fi = np.ascontiguousarray(from_idx, dtype=np.int32)
    ti = np.ascontiguousarray(to_idx, dtype=np.int32)
    out = np.zeros(len(fi) * len(ti), dtype=np.float64)
    lib().orc_mi_block(_p(st), C.c_int64(L), C.c_int64(N), _p(hdw), _p(r), _p(uqe), _p(fi), C.c_int64(len(fi)), _p(ti),
                       C.c_int64(len(ti)), _p(out), C.c_int(ncores or max_threads()))
    return out.reshape((len(fi), len(ti)), order="F")


def hamming_weights(states, thresh, want_shared=False, ncores=None):
    st = np.ascontiguousarray(states, dtype=np.uint8)
    L, N = st.shape
    hdw = np.zeros(N)
    shared = np.zeros((N, N), dtype=np.int32) if want_shared else None
    lib().orc_hamming_weights(_p(st), C.c_int64(L), C.c_int64(N), C.c_int32(int(thresh)), _p(hdw),
                              _p(shared) if want_shared else None, C.c_int(ncores or max_threads()))
    return (hdw, shared) if want_shared else hdw


def fast_hadamard(MI, den, uq, pxy, pxpy, RXY, pXrX, pYrY, ncores=1):
    ops = [np.ascontiguousarray(a, dtype=np.float64).reshape(-1) for a in (den, uq, pxy, pxpy, RXY, pXrX, pYrY)]
    assert MI.flags.c_contiguous and MI.dtype == np.float64
    lib().orc_fast_hadamard(_p(MI), *[_p(o) for o in ops], C.c_int64(MI.size), C.c_int(ncores))


def acgtn2num(nv, ref_bytes: bytes, ncores=1):
    assert nv.flags.f_contiguous and nv.dtype == np.float64 and nv.shape[0] == 5
    buf = C.create_string_buffer(ref_bytes, len(ref_bytes))
    lib().orc_acgtn2num(_p(nv), buf, C.c_int64(nv.shape[1]), C.c_int(ncores))
