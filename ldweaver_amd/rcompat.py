"""Base-R behaviours the host side of the path depends on (R itself is the host in the reference).

* ``set.seed(1988); sample(nsnp, k)``   R/computePairwiseMI.R:95-96  -> :class:`RRandom`
* ``stats::quantile(x, p)`` type 7      R/computePairwiseMI.R:354,422 -> :func:`quantile7`
* ``round(x, -3)``                      R/computePairwiseMI.R:69     -> :func:`round_thousands`
* ``write.table`` number formatting     R/computePairwiseMI.R:140,362 -> :func:`format_number`

These are restatements of R's published algorithms (Mersenne Twister MT19937 with
R's seed scrambling and rejection sampling, Hyndman-Fan type 7, formatReal with 15
significant digits); known answers from R are pinned in tests/test_rcompat.py.
"""
from __future__ import annotations

import math

import numpy as np


def r_sample(seed: int, n: int, size: int) -> np.ndarray:
    """``set.seed(seed); sample(n, size)`` through the library's native statement of the stream (``ldw_r_sample``); ``RRandom`` below is
    the Python statement the tests hold it against."""
    from . import _lib as L
    out = np.empty(int(size), dtype=np.int64)
    L.check(L.lib().ldw_r_sample(int(seed) & 0xFFFFFFFF, int(n), int(size), L.ptr(out)))
    return out


class RRandom:
    """R's default RNG stream (Mersenne-Twister, inversion, sample.kind = "Rejection", R >= 3.6)."""

    _N, _M = 624, 397

    def __init__(self, seed: int):
        s = np.uint32(seed & 0xFFFFFFFF)
        mul, one = np.uint32(69069), np.uint32(1)
        with np.errstate(over="ignore"):
            for _ in range(50):          # initial scrambling
                s = s * mul + one
            s = s * mul + one            # dummy[0] (overwritten by the position word)
            st = np.empty(self._N, dtype=np.uint32)
            for j in range(self._N):
                s = s * mul + one
                st[j] = s
        self._mt = st
        self._pos = self._N
        self._buf = None

    def _refill(self):
        mt = self._mt.astype(np.uint64)
        N, M = self._N, self._M
        UP, LO, MAT = 0x80000000, 0x7FFFFFFF, 0x9908B0DF
        # the classic in-place recurrence has a serial dependency through kk+M; run it in scalar chunks
        m = [int(v) for v in mt]
        for kk in range(N - M):
            y = (m[kk] & UP) | (m[kk + 1] & LO)
            m[kk] = m[kk + M] ^ (y >> 1) ^ (MAT if y & 1 else 0)
        for kk in range(N - M, N - 1):
            y = (m[kk] & UP) | (m[kk + 1] & LO)
            m[kk] = m[kk + M - N] ^ (y >> 1) ^ (MAT if y & 1 else 0)
        y = (m[N - 1] & UP) | (m[0] & LO)
        m[N - 1] = m[M - 1] ^ (y >> 1) ^ (MAT if y & 1 else 0)
        self._mt = np.array(m, dtype=np.uint64).astype(np.uint32)
        # temper the whole block at once
        y = self._mt.astype(np.uint64)
        y ^= y >> np.uint64(11)
        y ^= (y << np.uint64(7)) & np.uint64(0x9D2C5680)
        y ^= (y << np.uint64(15)) & np.uint64(0xEFC60000)
        y ^= y >> np.uint64(18)
        self._buf = (y & np.uint64(0xFFFFFFFF)).astype(np.float64) * 2.3283064365386963e-10
        self._pos = 0

    def unif_rand(self) -> float:
        if self._pos >= self._N:
            self._refill()
        v = float(self._buf[self._pos])
        self._pos += 1
        if v <= 0.0:
            return 0.5 * 2.328306437080797e-10
        if 1.0 - v <= 0.0:
            return 1.0 - 0.5 * 2.328306437080797e-10
        return v

    def _unif_index(self, dn: int) -> int:
        if dn <= 0:
            return 0
        bits = int(math.ceil(math.log2(dn)))
        mask = (1 << bits) - 1
        while True:
            v = 0
            n = 0
            while n <= bits:
                v = 65536 * v + int(math.floor(self.unif_rand() * 65536))
                n += 16
            v &= mask
            if v < dn:
                return v

    def sample(self, n: int, size: int) -> np.ndarray:
        """``sample(n, size)`` without replacement; 1-based like R.  ``n > 1e7 and size <= n/2``: R >= 3.6 takes ``do_sample2`` (hashing:
        every element drawn with ``R_unif_index(n) + 1`` and re-drawn while it repeats an earlier one, at most 100 draws)."""
        if n > 10_000_000 and size <= n // 2:
            seen = set()
            out = np.empty(size, dtype=np.int64)
            for i in range(size):
                v = 0
                for _ in range(100):
                    v = self._unif_index(n) + 1
                    if v not in seen:
                        seen.add(v)
                        break
                out[i] = v
            return out
        pool = list(range(1, n + 1))
        out = np.empty(size, dtype=np.int64)
        left = n
        for i in range(size):
            j = self._unif_index(left)
            out[i] = pool[j]
            left -= 1
            pool[j] = pool[left]
        return out


def quantile7(x, prob: float) -> float:
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n == 0:
        return float("nan")
    index = 1.0 + (n - 1) * prob
    lo, hi = int(math.floor(index)), int(math.ceil(index))
    part = np.partition(x, (lo - 1, hi - 1) if hi != lo else (lo - 1,))
    q, xh = float(part[lo - 1]), float(part[hi - 1])
    if index > lo and xh != q:
        h = index - lo
        q = (1.0 - h) * q + h * xh
    return q


def round_thousands(x: float) -> int:
    return int(round(float(x) / 1000.0)) * 1000


def format_number(x, digits: int = 15) -> str:
    """One numeric cell as ``write.table(quote = F)`` prints it."""
    if isinstance(x, (int, np.integer, bool, np.bool_)):
        return str(int(x))
    x = float(x)
    if math.isnan(x):
        return "NA"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "0"
    sign = "-" if x < 0 else ""
    ax = abs(x)
    mant, e = f"{ax:.{digits - 1}e}".split("e")
    e10 = int(e)
    sig = (mant[0] + mant[2:]).rstrip("0") or "0"
    nsig = len(sig)
    wexp = 2 if abs(e10) < 100 else 3
    w_sci = len(sign) + (nsig + 1 if nsig > 1 else 1) + 2 + wexp
    rgt = max(0, nsig - e10 - 1)
    left = e10 + 1 if e10 >= 0 else 1
    w_fix = len(sign) + left + (rgt + 1 if rgt else 0)
    if w_fix <= w_sci:
        return sign + f"{ax:.{rgt}f}"
    body = sig[0] + ("." + sig[1:] if nsig > 1 else "")
    return f"{sign}{body}e{'+' if e10 >= 0 else '-'}{abs(e10):0{wexp}d}"


def circ_len(pos1, pos2, g):
    """len = 0.5*g - abs((pos1 - pos2) %% g - 0.5*g)  (R/computePairwiseMI.R:330)."""
    g = float(g)
    d = np.mod(np.asarray(pos1, dtype=np.float64) - np.asarray(pos2, dtype=np.float64), g)
    return 0.5 * g - np.abs(d - 0.5 * g)
