"""CPU oracle for the LDWeaver all-pairs weighted-MI hot path (numpy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.

PARITY UNPINNED.  The reference (Sudaraka88/LDWeaver v1.5.2) is an R package;
R, Rcpp and MatrixExtra are absent from this image, the reference's native
kernels ``#include <Rcpp.h>`` (so they cannot be compiled here without writing
stand-ins for missing headers, which is not allowed), and the reference's own
test (tests/testthat/test-pipeline.R:32-70) holds no golden values for this
path.  This file is therefore a restatement checked only against (i) a second,
independent per-pair restatement in this same file (``mi_pair_direct``), (ii)
published known answers for the R building blocks it restates (Mersenne
Twister / ``sample()``, ``quantile`` type 7), and (iii) the C restatement in
``oracle/ldw_oracle.c``.

All ``file:line`` citations are relative to the reference checkout.

Conventions: SNP indices are 0-based here, 1-based in R.  ``states`` is a uint8
(L, N) matrix with values 0..4 = A,C,G,T,N (dense equivalent of the five
``snp.matrix_*`` one-hot sparse matrices of ``snp.dat``, R/extractSNPs.R:138-141).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------
# a-2  5-state encoding rule          src/getACGTNsites.cpp:58-70, 229-265
# --------------------------------------------------------------------------
_ENC = np.full(256, 4, dtype=np.uint8)
for _ch, _v in (("A", 0), ("a", 0), ("C", 1), ("c", 1), ("G", 2), ("g", 2), ("T", 3), ("t", 3)):
    _ENC[ord(_ch)] = _v


def encode_states(seqs) -> np.ndarray:
    """char -> state: A/a 0, C/c 1, G/g 2, T/t 3, everything else 4 (N).

    ``seqs``: list of N equal-length byte strings (one per sequence, already
    restricted to the retained SNP columns).  Returns uint8 (L, N)
    (src/getACGTNsites.cpp:229-265 emits the same information as COO triplets).
    """
    arr = np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), -1)
    return np.ascontiguousarray(_ENC[arr].T)


def acgtn_table(states: np.ndarray) -> np.ndarray:
    """5 x L per-state counts (``ACGTN_table``, src/getACGTNsites.cpp:208,238-262)."""
    L = states.shape[0]
    out = np.zeros((5, L), dtype=np.int64)
    for x in range(5):
        out[x] = (states == x).sum(axis=1)
    return out


def uqe_r(states: np.ndarray):
    """``uqe`` (L x 5, 0/1 allele present) and ``r = rowSums(uqe)`` (R/extractSNPs.R:47,141)."""
    uqe = (acgtn_table(states) > 0).T.astype(np.float64)
    return uqe, uqe.sum(axis=1)


def snp_filter(allele_counts, n: int, gap_thresh: float = 0.15, maf_thresh: float = 0.01, filt: int = 0):
    """Literal restatement of the column filter of ``extractAlnParam`` (src/getACGTNsites.cpp:104-166): 1-based POS.
    ``allele_counts``: 5 x L (A,C,G,T,other) as produced at :58-70."""
    ac = np.asarray(allele_counts, dtype=np.float64)
    L = ac.shape[1]
    POS = []
    min_maf = int(n * maf_thresh) if filt == 0 else int(n * (1 - maf_thresh))
    for j in range(L):
        chk = 0
        for k in range(4):
            if ac[k, j] > 0:                       # we need at least one non-gap allele
                chk += 1
                if chk > 1:                        # seems polymorphic
                    if ac[4, j] / n < gap_thresh:  # now check gap content
                        if filt == 0:
                            snp = sorted(ac[:4, j])
                            if snp[2] > min_maf:   # second largest non-gap element
                                POS.append(j + 1)
                        else:
                            if ac[:, j].max() <= min_maf:
                                POS.append(j + 1)
                    break
    return np.array(POS, dtype=np.int32)


# --------------------------------------------------------------------------
# a-3  .ACGTN2num                      src/ACGTN2num_parallel.cpp:10-43
# --------------------------------------------------------------------------
def acgtn2num(nv: np.ndarray, cv) -> None:
    """Zero the reference-allele row of a 5 x L mask IN PLACE.

    ``nv``: float64 (5, L) array whose R memory order is column-major, i.e.
    element (row, c) lives at flat index ``c*5+row``; pass it as a Fortran-
    ordered (5, L) array or as a flat length-5L vector.  ``cv``: sequence of L
    strings, only the first character is inspected; upper-case A/C/G/T, and
    ``N`` or ``-`` -> row 4; anything else leaves the column untouched.
    """
    flat = nv.reshape(-1, order="F") if nv.ndim == 2 else nv
    if nv.ndim == 2 and not nv.flags.f_contiguous:
        raise ValueError("nv must be Fortran-ordered (R column-major) to be mutated in place")
    for c, s in enumerate(cv):
        cc = s[:1]
        if isinstance(cc, bytes):
            cc = cc.decode("latin1")
        if cc == "A":
            flat[c * 5] = 0
        elif cc == "C":
            flat[c * 5 + 1] = 0
        elif cc == "G":
            flat[c * 5 + 2] = 0
        elif cc == "T":
            flat[c * 5 + 3] = 0
        elif cc == "N" or cc == "-":
            flat[c * 5 + 4] = 0


# --------------------------------------------------------------------------
# a-4  estimate_Hamming_distance_weights   R/performPopulationStuctureCorrection.R:20-81
# --------------------------------------------------------------------------
def shared_counts(states: np.ndarray) -> np.ndarray:
    """shared[i,j] = #{snps : state_i == state_j}  (the five crossprods, :49-74). int64 (N, N)."""
    L, N = states.shape
    shared = np.zeros((N, N), dtype=np.int64)
    # chunk over SNPs to bound memory; int32 matmul is exact
    step = max(1, min(L, 4_000_000 // max(N, 1)))
    for lo in range(0, L, step):
        blk = states[lo:lo + step]
        for x in range(5):
            m = (blk == x).astype(np.float32)  # counts < 2^24 per chunk -> exact in f32
            shared += np.rint(m.T @ m).astype(np.int64)
    return shared


def hamming_weights(states: np.ndarray, threshold: float = 0.1) -> np.ndarray:
    """hdw[j] = 1 / (#{i : L - shared[i,j] < as.integer(L*threshold)} + 1)   (:23, :76)."""
    L = states.shape[0]
    thresh = int(L * threshold)  # as.integer truncates
    shared = shared_counts(states)
    cnt = ((L - shared) < thresh).sum(axis=0)
    return 1.0 / (cnt + 1.0)


# --------------------------------------------------------------------------
# a-9  .fastHadamard                    src/computeMI.cpp:11-21
# --------------------------------------------------------------------------
def fast_hadamard(MI, den, uq, pxy, pxpy, RXY, pXrX, pYrY) -> None:
    """MI[c] += uq*pxy/den*log(pxy/(pxpy+RXY+pXrX+pYrY)*den) over the LINEAR index c.

    All operands are read by linear (R column-major) index regardless of their
    dims; pass flat vectors or Fortran-ordered matrices.  Evaluation order as
    written in the reference: ((uq*pxy)/den) * log((pxy/(((pxpy+RXY)+pXrX)+pYrY))*den).
    """
    f = lambda a: np.asarray(a).reshape(-1, order="F")
    MIf = f(MI)
    d = ((f(pxpy) + f(RXY)) + f(pXrX)) + f(pYrY)
    MIf += ((f(uq) * f(pxy)) / f(den)) * np.log((f(pxy) / d) * f(den))
    if MIf.base is None or not np.shares_memory(MIf, MI):
        np.copyto(MI, MIf.reshape(MI.shape, order="F"))


# --------------------------------------------------------------------------
# a-10 / a-12 helpers                   src/computeMI.cpp:25-77, src/fintersect.cpp:6-32
# --------------------------------------------------------------------------
def compare_to_row(x: np.ndarray, y) -> np.ndarray:
    """ret[j] = any(x[j,] in y)."""
    return np.isin(np.asarray(x), np.asarray(y)).any(axis=1)


def vec_pos_match(x, y) -> np.ndarray:
    """1-based position of the first occurrence of each x[i] in y, 0 if absent."""
    y = np.asarray(y)
    out = np.zeros(len(x))
    for i, v in enumerate(x):
        hit = np.nonzero(y == v)[0]
        if hit.size:
            out[i] = hit[0] + 1
    return out


def compare_triplet(MI0X, MI0Z, MI0: float) -> bool:
    """False iff some i has MI0 < MI0X[i] and MI0 < MI0Z[i]."""
    MI0X = np.asarray(MI0X)
    MI0Z = np.asarray(MI0Z)
    return not bool(np.any((MI0 < MI0X) & (MI0 < MI0Z)))


def fast_intersect(A, B):
    """Sorted multiset-style two-pointer intersection of int vectors (duplicates pair off)."""
    Av = sorted(int(a) for a in A)
    Bv = sorted(int(b) for b in B)
    i = j = 0
    out = []
    while i < len(Av) and j < len(Bv):
        if Av[i] < Bv[j]:
            i += 1
        elif Av[i] > Bv[j]:
            j += 1
        else:
            out.append(Av[i])
            i += 1
            j += 1
    return out


# --------------------------------------------------------------------------
# a-6  make_blocks                      R/computePairwiseMI.R:147-165
# --------------------------------------------------------------------------
def r_round_thousands(x: float) -> int:
    """round(x, -3) with R's round-half-even behaviour (R/computePairwiseMI.R:69)."""
    return int(round(x / 1000.0)) * 1000


def make_blocks(nsnp: int, max_blk_sz: int = 10000):
    """Rows (from_s, from_e, to_s, to_e), 1-based inclusive, all i<=j, row-major in i."""
    part1 = math.ceil(nsnp / max_blk_sz)
    fs = [(i - 1) * max_blk_sz + 1 for i in range(1, part1 + 1)]
    fe = [min(i * max_blk_sz, nsnp) for i in range(1, part1 + 1)]
    return [(fs[i], fe[i], fs[j], fe[j]) for i in range(part1) for j in range(i, part1)]


# --------------------------------------------------------------------------
# R building blocks used by the driver (third-party to the reference: base R)
# --------------------------------------------------------------------------
class RMersenneTwister:
    """R's default RNG: MT19937 with R's ``set.seed`` scrambling and ``sample()`` by rejection.

    Restates R's src/main/RNG.c (``RNG_Init``, ``MT_genrand``, ``fixup``,
    ``R_unif_index``/``rbits``) and the no-replacement loop of ``do_sample``
    (src/main/unique.c / random.c) for R >= 3.6.0 (``sample.kind = "Rejection"``).
    Used only for ``set.seed(1988); sample(nsnp, snp_subset)`` at
    R/computePairwiseMI.R:95-96.
    """

    N, M = 624, 397

    def __init__(self, seed: int):
        s = seed & 0xFFFFFFFF
        for _ in range(50):
            s = (69069 * s + 1) & 0xFFFFFFFF
        mt = []
        # i_seed[0] is the position word (overwritten with 624 by FixupSeeds); then 624 state words
        s = (69069 * s + 1) & 0xFFFFFFFF
        for _ in range(self.N):
            s = (69069 * s + 1) & 0xFFFFFFFF
            mt.append(s)
        self.mt = mt
        self.mti = self.N

    def _genrand(self) -> int:
        N, M, mt = self.N, self.M, self.mt
        if self.mti >= N:
            for kk in range(N - M):
                y = (mt[kk] & 0x80000000) | (mt[kk + 1] & 0x7FFFFFFF)
                mt[kk] = mt[kk + M] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            for kk in range(N - M, N - 1):
                y = (mt[kk] & 0x80000000) | (mt[kk + 1] & 0x7FFFFFFF)
                mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            y = (mt[N - 1] & 0x80000000) | (mt[0] & 0x7FFFFFFF)
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.mti = 0
        y = mt[self.mti]
        self.mti += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def unif_rand(self) -> float:
        v = self._genrand() * 2.3283064365386963e-10
        i2_32m1 = 2.328306437080797e-10
        if v <= 0.0:
            return 0.5 * i2_32m1
        if 1.0 - v <= 0.0:
            return 1.0 - 0.5 * i2_32m1
        return v

    def _rbits(self, bits: int) -> float:
        v = 0
        n = 0
        while n <= bits:
            v1 = int(math.floor(self.unif_rand() * 65536))
            v = 65536 * v + v1
            n += 16
        if bits < 64:
            v &= (1 << bits) - 1
        return float(v)

    def unif_index(self, dn: float) -> float:
        if dn <= 0:
            return 0.0
        bits = int(math.ceil(math.log2(dn)))
        while True:
            dv = self._rbits(bits)
            if dn > dv:
                return dv

    def sample(self, n: int, k: int) -> np.ndarray:
        """``sample(n, k)`` (no replacement, 1-based result)."""
        x = list(range(n))
        out = np.empty(k, dtype=np.int64)
        nn = n
        for i in range(k):
            j = int(self.unif_index(nn))
            out[i] = x[j] + 1
            nn -= 1
            x[j] = x[nn]
        return out


def quantile7(x: np.ndarray, prob: float) -> float:
    """``stats::quantile(x, probs = prob)`` type 7 for one probability (R >= 4.0.x code path)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n == 0:
        return float("nan")
    index = 1 + max(n - 1, 0) * prob
    lo = int(math.floor(index))
    hi = int(math.ceil(index))
    part = np.partition(x, sorted({lo - 1, hi - 1}))
    qs = part[lo - 1]
    xhi = part[hi - 1]
    if index > lo and xhi != qs:
        h = index - lo
        qs = (1 - h) * qs + h * xhi
    return float(qs)


def circ_len(pos1, pos2, g):
    """len = 0.5*g - abs((pos1 - pos2) %% g - 0.5*g)  with R's floored ``%%`` (R/computePairwiseMI.R:330)."""
    d = np.mod(np.asarray(pos1, dtype=np.float64) - np.asarray(pos2, dtype=np.float64), float(g))
    return 0.5 * g - np.abs(d - 0.5 * g)


def lr_links_approx(POS: np.ndarray, g: float, sr_dist: float, seed: int = 1988) -> float:
    """R/computePairwiseMI.R:94-97."""
    nsnp = len(POS)
    snp_subset = min(nsnp, int(round(nsnp * 0.1)))
    rng = RMersenneTwister(seed)
    idx = rng.sample(nsnp, snp_subset) - 1
    POSf = np.asarray(POS, dtype=np.float64)
    cnt = 0
    for x in POSf[idx]:
        cnt += int(((0.5 * g - np.abs(np.mod(x - POSf, float(g)) - 0.5 * g)) > sr_dist).sum())
    return cnt / snp_subset * nsnp / 2


# --------------------------------------------------------------------------
# a-7 / a-8  per-block MI
# --------------------------------------------------------------------------
def weighted_onehots(states_rows: np.ndarray, hdw: np.ndarray):
    """tXh (x = 0..4): dense nf x N rows of sqrt(w)-scaled one-hots, and pX = rowSums(tXh^2).

    R/computePairwiseMI.R:238-257.  rowSums accumulates in long double in R;
    numpy's pairwise double sum differs at ~1e-16 relative.
    """
    sq = np.sqrt(np.asarray(hdw, dtype=np.float64))
    tXh, pX = [], []
    for x in range(5):
        m = (states_rows == x).astype(np.float64) * sq[None, :]
        tXh.append(m)
        pX.append(np.sum((m * m).astype(np.longdouble), axis=1).astype(np.float64))
    return tXh, pX


def mi_block_faithful(states, hdw, r, uqe, from_idx, to_idx) -> np.ndarray:
    """Block-faithful MI (oracle B): same operands, shapes and LINEAR-INDEX semantics
    as ``perform_MI_computation_ACGTN`` + ``computeMI_Sprase`` + ``.fastHadamard``
    (R/computePairwiseMI.R:204-298, 390-398; src/computeMI.cpp:11-21).

    Reproduces quirk Q1: ``rft`` is built nt x nf (:261) but read by linear index
    as if nf x nt.  Returns MI as a Fortran-ordered (nf, nt) float64 matrix.
    ``from_idx`` / ``to_idx`` are 0-based index arrays.
    """
    from_idx = np.asarray(from_idx)
    to_idx = np.asarray(to_idx)
    nf, nt = len(from_idx), len(to_idx)
    neff = float(np.sum(hdw))
    fromISto = nf == nt and bool(np.all(from_idx == to_idx))
    rf = np.asarray(r, dtype=np.float64)[from_idx]
    rt = rf if fromISto else np.asarray(r, dtype=np.float64)[to_idx]
    uqf = np.asarray(uqe, dtype=np.float64)[from_idx]
    uqt = uqf if fromISto else np.asarray(uqe, dtype=np.float64)[to_idx]
    tXfh, pXf = weighted_onehots(states[from_idx], hdw)
    if fromISto:
        tYth, pYt = tXfh, pXf
    else:
        tYth, pYt = weighted_onehots(states[to_idx], hdw)

    den = neff + np.outer(rf, rt) * 0.5                      # nf x nt  (:260)
    rft = (np.outer(rf, rt).T * 0.25)                         # nt x nf  (:261)  <- Q1
    RXY_lin = rft.reshape(-1, order="F")                      # read by linear index
    rfh = 0.5 * rf                                            # (:262)
    rth = 0.5 * rt                                            # (:263)
    den_lin = den.reshape(-1, order="F")

    MI = np.zeros(nf * nt, dtype=np.float64)
    for X in range(5):
        for Y in range(5):
            pxy = (tXfh[X] @ tYth[Y].T + 0.5).reshape(-1, order="F")          # :391
            uq = np.outer(uqf[:, X], uqt[:, Y]).reshape(-1, order="F")         # :392
            pXrX = np.outer(pXf[X] * rfh, np.ones(nt)).reshape(-1, order="F")  # :393
            pYrY = np.outer(np.ones(nf), pYt[Y] * rth).reshape(-1, order="F")  # :394
            pxpy = np.outer(pXf[X], pYt[Y]).reshape(-1, order="F")             # :395
            fast_hadamard(MI, den_lin, uq, pxy, pxpy, RXY_lin, pXrX, pYrY)     # :396
    return MI.reshape((nf, nt), order="F")


def joint_counts(states, a: int, b: int) -> np.ndarray:
    """Unweighted integer 5x5 joint histogram of SNPs a (rows) and b (cols). int64."""
    code = states[a].astype(np.int64) * 5 + states[b].astype(np.int64)
    return np.bincount(code, minlength=25).reshape(5, 5)


def mi_pair_direct(states, hdw, r, uqe, a: int, b: int, rxy: float | None = None) -> float:
    """Per-pair direct MI (oracle A), independent of the block machinery.

    Builds the weighted 5x5 table of pair (a = "from"/X side, b = "to"/Y side) with
    per-sequence weight fl(sqrt(w))^2 and applies the formula of
    src/computeMI.cpp:19 cell by cell in the reference's (X outer, Y inner) order.
    ``rxy`` overrides the RXY term (0.25*r_a*r_b when None, i.e. the *intended*
    value; the block path passes the Q1-scrambled value).
    """
    sq = np.sqrt(np.asarray(hdw, dtype=np.float64))
    v = sq * sq
    xa = states[a]
    yb = states[b]
    neff = float(np.sum(hdw))
    ra, rb = float(r[a]), float(r[b])
    den = neff + ra * rb * 0.5
    if rxy is None:
        rxy = ra * rb * 0.25
    rX, rY = 0.5 * ra, 0.5 * rb
    mi = 0.0
    for X in range(5):
        mx = xa == X
        pX = float(np.sum(v[mx].astype(np.longdouble)))
        for Y in range(5):
            my = yb == Y
            pY = float(np.sum(v[my].astype(np.longdouble)))
            pxy = float(np.sum(sq[mx & my] * sq[mx & my])) + 0.5
            uq = float(uqe[a][X]) * float(uqe[b][Y])
            mi += uq * pxy / den * math.log(pxy / (pX * pY + rxy + pX * rX + pY * rY) * den)
    return mi


def q1_rxy(a_loc: int, b_loc: int, nf: int, nt: int, rf, rt) -> float:
    """Effective RXY read at MI[a_loc, b_loc] (0-based within block) under quirk Q1."""
    c = a_loc + b_loc * nf
    return 0.25 * float(rf[c // nt]) * float(rt[c % nt])


# --------------------------------------------------------------------------
# a-7 (second half): pair list, len, sr/lr split, lr filter
# --------------------------------------------------------------------------
def block_pair_index(nf: int, nt: int, fromISto: bool):
    """0-based (row, col) index arrays in the reference's row order (R/computePairwiseMI.R:306-310).

    diag block: which(lower.tri(t(MI))) -> column-major, row > col.
    off-diag:   which(upper.tri(MI)) then which(lower.tri(MI)), each column-major, diag=FALSE.
    """
    if fromISto:
        # t(MI) is nt x nf; square here
        cols, rows = np.nonzero(np.tri(nf, nt, -1, dtype=bool).T)  # iterate column-major
        return rows, cols
    rr = np.arange(nf)[:, None]
    cc = np.arange(nt)[None, :]
    up = (rr < cc)
    lo = (rr > cc)
    cu, ru = np.nonzero(up.T)
    cl, rl = np.nonzero(lo.T)
    return np.concatenate([ru, rl]), np.concatenate([cu, cl])


@dataclass
class BlockLinks:
    sr: dict          # columns pos1,pos2,clust1,clust2,len,MI (+ a,b global 0-based indices)
    lr: dict          # same columns, after the per-block quantile filter
    n_lr_total: int
    disc_thresh: float
    prob: float


def block_links(MI, from_idx, to_idx, POS, paint, g, sr_dist, lr_retain_links, lr_approx,
                sr_only=False) -> BlockLinks:
    """R/computePairwiseMI.R:306-364 for one block (MI is the (nf, nt) block matrix)."""
    from_idx = np.asarray(from_idx)
    to_idx = np.asarray(to_idx)
    nf, nt = len(from_idx), len(to_idx)
    fromISto = nf == nt and bool(np.all(from_idx == to_idx))
    rows, cols = block_pair_index(nf, nt, fromISto)
    POSf = np.asarray(POS, dtype=np.float64)
    pos2 = POSf[from_idx][rows]
    pos1 = POSf[to_idx][cols]
    clust2 = np.asarray(paint)[from_idx][rows]
    clust1 = np.asarray(paint)[to_idx][cols]
    ln = circ_len(pos1, pos2, g)
    mi = np.asarray(MI)[rows, cols]
    tab = dict(pos1=pos1, pos2=pos2, clust1=clust1, clust2=clust2, len=ln, MI=mi,
               a=from_idx[rows], b=to_idx[cols])
    sw = ln <= sr_dist
    sr = {k: v[sw] for k, v in tab.items()}
    lr_all = {k: v[~sw] for k, v in tab.items()}
    n_lr = int((~sw).sum())
    disc = float("nan")
    prob = float("nan")
    lr = {k: v[:0] for k, v in tab.items()}
    if n_lr > 0 and not sr_only:
        prob = max(0.0, 1 - ((lr_retain_links * (n_lr / lr_approx)) / n_lr))
        disc = quantile7(lr_all["MI"], prob)
        keep = lr_all["MI"] >= disc
        lr = {k: v[keep] for k, v in lr_all.items()}
    return BlockLinks(sr=sr, lr=lr, n_lr_total=n_lr, disc_thresh=disc, prob=prob)


# --------------------------------------------------------------------------
# a-11  mergeNsort_sr_links             R/computePairwiseMI.R:400-495
# --------------------------------------------------------------------------
def _beta_mle(x: np.ndarray):
    """``coef(fitdistrplus::fitdist(x, "beta"))``: MLE started from the moment estimates
    (fitdistrplus start values) and maximised with Nelder-Mead like ``optim``'s default.
    Third-party optimiser: agreement with R is to optimiser tolerance only."""
    from scipy import optimize, special
    n = x.size
    m = float(np.mean(x))
    v = (n - 1) / n * float(np.var(x, ddof=1))
    aux = m * (1 - m) / v - 1
    start = np.array([m * aux, (1 - m) * aux])
    slx = float(np.sum(np.log(x)))
    sl1x = float(np.sum(np.log1p(-x)))

    def nll(p):
        a, b = p
        if a <= 0 or b <= 0:
            return np.inf
        return -((a - 1) * slx + (b - 1) * sl1x - n * special.betaln(a, b))

    res = optimize.minimize(nll, start, method="Nelder-Mead",
                            options=dict(xatol=1e-10, fatol=1e-12, maxiter=5000, maxfev=10000))
    # Newton polish on the score equations (the simplex stops at optimiser tolerance; the MLE itself is unique)
    a, b = float(res.x[0]), float(res.x[1])
    dg, tg = special.digamma, lambda z: special.polygamma(1, z)
    for _ in range(60):
        grad = np.array([n * (dg(a + b) - dg(a)) + slx, n * (dg(a + b) - dg(b)) + sl1x])
        H = n * np.array([[tg(a + b) - tg(a), tg(a + b)], [tg(a + b), tg(a + b) - tg(b)]])
        step = np.linalg.solve(H, grad)
        if not np.all(np.isfinite(step)) or a - step[0] <= 0 or b - step[1] <= 0:
            break
        a, b = a - step[0], b - step[1]
        if np.all(np.abs(step) < 1e-13 * np.array([a, b])):
            break
    return float(a), float(b)


def neg_log_beta_sf(x, a: float, b: float) -> np.ndarray:
    """-pbeta(x, a, b, lower.tail = FALSE, log.p = TRUE): scipy where the tail is representable, 60-digit mpmath
    where P(X > x) underflows a double (R's TOMS-708 bratio works in log space there)."""
    from scipy import stats
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide="ignore"):
        out = -stats.beta.logsf(x, a, b)
    deep = (~np.isfinite(out) | (out > 600.0)) & (x < 1)      # denormal tails lose digits before they vanish
    if deep.any():
        import mpmath
        mpmath.mp.dps = 60
        for i in np.nonzero(deep)[0]:   # P(X > x) = I_{1-x}(b, a)
            out[i] = -float(mpmath.log(mpmath.betainc(b, a, 0, 1 - mpmath.mpf(float(x[i])), regularized=True)).real)
    return out


def merge_n_sort_sr_links(sr_links_by_clust, nclust: int, sr_dist: float, srp_cutoff: float, fit_data=None):
    """Returns (sr_links_red, sr_links_ARACNE_check) as dicts of columns
    clust_c,pos1,pos2,clust1,clust2,len,MI,srp_max.  The plot (:440) is not produced; the table saved as
    c<i>_fit_data.rds (maxvls: len, max, fit; :422-439) is appended per cluster to ``fit_data`` when a list is given.
    Reproduces Q5 (``mean_dist[len]`` positional indexing, :448) and Q6 (natural-log srp, :453)."""
    from scipy import stats
    cols = ["pos1", "pos2", "clust1", "clust2", "len", "MI"]
    main = {k: [] for k in ["clust_c"] + cols + ["srp_max"]}
    dup = {k: [] for k in ["clust_c"] + cols + ["srp_max"]}
    for ci in range(1, nclust + 1):
        t = {k: np.asarray(sr_links_by_clust[ci - 1][k]) for k in cols}
        keep = ~np.isnan(t["len"]) & (t["len"] < sr_dist) & (t["len"] > 0)
        t = {k: v[keep] for k, v in t.items()}
        # group_by(len) %>% summarise(max = quantile(MI, 0.95))   (:422)
        ulen, inv = np.unique(t["len"], return_inverse=True)
        order = np.argsort(inv, kind="stable")
        bounds = np.searchsorted(inv[order], np.arange(len(ulen) + 1))
        mx = np.array([quantile7(t["MI"][order[bounds[i]:bounds[i + 1]]], 0.95) for i in range(len(ulen))])
        # fastLm(cbind(log(len), 1), log(max))                    (:428)
        Xd = np.column_stack([np.log(ulen), np.ones(len(ulen))])
        coef, *_ = np.linalg.lstsq(Xd, np.log(mx), rcond=None)
        mean_dist = np.exp(Xd @ coef)                                # (:429)
        if fit_data is not None:
            fit_data.append(dict(len=ulen.astype(np.float64), max=mx, fit=mean_dist))   # saveRDS(maxvls) (:439)
        # diff_dat = MI - mean_dist[len]   positional index by VALUE of len (Q5, :448)
        li = t["len"].astype(np.int64)                                # R truncates toward zero
        ok = (li >= 1) & (li <= len(mean_dist))
        md = np.full(len(li), np.nan)
        md[ok] = mean_dist[li[ok] - 1]
        diff = t["MI"] - md
        idx = np.nonzero(diff > 0)[0]                                 # NA > 0 is dropped by which()
        a_, b_ = _beta_mle(diff[idx])                                 # (:452)
        srp = neg_log_beta_sf(diff[idx], a_, b_)                      # (:453)
        t = {k: v[idx] for k, v in t.items()}
        t["srp_max"] = srp
        t["clust_c"] = np.full(len(idx), ci)
        isdup = t["clust1"] != t["clust2"]
        for k in main:
            main[k].append(t[k][~isdup])
            dup[k].append(t[k][isdup])
    main = {k: np.concatenate(v) if v else np.array([]) for k, v in main.items()}
    dup = {k: np.concatenate(v) if v else np.array([]) for k, v in dup.items()}
    if len(dup["pos1"]) > 0:
        # .I[which.max(srp_max)] by (pos1,pos2,clust1,clust2,len,MI): first max within each key group,
        # groups in order of first appearance (:478-485)
        keys = {}
        for i in range(len(dup["pos1"])):
            k = tuple(dup[c][i] for c in cols)
            if k not in keys or dup["srp_max"][i] > dup["srp_max"][keys[k]]:
                if k not in keys:
                    keys[k] = i
                else:
                    keys[k] = i
        sel = np.array(list(keys.values()), dtype=np.int64)
        main = {k: np.concatenate([main[k], dup[k][sel]]) for k in main}
    red_m = main["srp_max"] > srp_cutoff
    red = {k: v[red_m] for k, v in main.items()}
    if red_m.any():
        chk_m = main["MI"] >= red["MI"].min()
    else:
        chk_m = np.zeros(len(main["MI"]), dtype=bool)
    chk = {k: v[chk_m] for k, v in main.items()}
    return red, chk


# --------------------------------------------------------------------------
# a-12  runARACNE                        R/io_functions.R:101-164
# --------------------------------------------------------------------------
def run_aracne(chk_pos1, chk_pos2, chk_MI, full_pos1, full_pos2, full_MI) -> np.ndarray:
    """Literal restatement (O(n_check * n_full)); use on small inputs."""
    pos_mat = np.column_stack([np.asarray(full_pos1, dtype=np.float64), np.asarray(full_pos2, dtype=np.float64)])
    MIs = np.asarray(full_MI, dtype=np.float64)
    n = len(chk_pos1)
    out = np.ones(n, dtype=bool)
    pX_ = 0
    idX = matX = None
    for i in range(n):
        pX = chk_pos1[i]
        pZ = chk_pos2[i]
        if pX != pX_:
            idX = np.nonzero(compare_to_row(pos_mat, [pX]))[0]
            matX = pos_mat[idX].reshape(-1)           # c(rbind(p1, p2)) interleaves
            matX = matX[matX != pX]
            pX_ = pX
        idZ = np.nonzero(compare_to_row(pos_mat, [pZ]))[0]
        matZ = pos_mat[idZ].reshape(-1)
        matZ = matZ[matZ != pZ]
        com = fast_intersect(matX, matZ)
        if len(com) > 0:
            ix = vec_pos_match(com, matX).astype(np.int64) - 1
            iz = vec_pos_match(com, matZ).astype(np.int64) - 1
            out[i] = compare_triplet(MIs[idX[ix]], MIs[idZ[iz]], float(chk_MI[i]))
    return out


# --------------------------------------------------------------------------
# write.table number formatting (R's formatReal with digits = 15, scipen = 0)
# --------------------------------------------------------------------------
def r_format_double(x: float, digits: int = 15) -> str:
    """How ``write.table`` prints one double (minimal significant digits <= 15 that
    reproduce the 15-digit value; fixed notation unless it is wider than scientific)."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    x = float(x)
    if math.isnan(x):
        return "NA"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0:
        return "0"
    neg = x < 0
    ax = abs(x)
    # significant digits needed (<= digits)
    mant, exp = f"{ax:.{digits - 1}e}".split("e")
    e10 = int(exp)
    sig = mant.replace(".", "").rstrip("0")
    nsig = max(len(sig), 1)
    # scientific width
    mxsl = nsig
    wexp = 2 if abs(e10) < 100 else 3
    wE = (1 if neg else 0) + (mxsl + 1 if mxsl > 1 else mxsl) + 2 + wexp
    # fixed width
    left = e10 + 1 if e10 >= 0 else 1
    rgt = max(0, nsig - e10 - 1)
    wF = (1 if neg else 0) + left + (rgt + 1 if rgt > 0 else 0)
    if wF <= wE:
        s = f"{ax:.{rgt}f}"
    else:
        m = sig[0] + ("." + sig[1:] if nsig > 1 else "")
        s = f"{m}e{'+' if e10 >= 0 else '-'}{abs(e10):0{wexp}d}"
    return ("-" if neg else "") + s


# --------------------------------------------------------------------------
# a-5  perform_MI_computation (driver)   R/computePairwiseMI.R:46-145
# --------------------------------------------------------------------------
@dataclass
class MIResult:
    sr_links_red: dict
    lr_rows: dict
    sr_links_by_clust: list = field(default_factory=list)
    blocks: list = field(default_factory=list)
    lr_links_approx: float = float("nan")
    fit_data: list = field(default_factory=list)      # per cluster: the maxvls table of c<i>_fit_data.rds (:439)


def perform_mi_computation(states, POS, g, r, uqe, hdw, paint, nclust, sr_dist=20000,
                           lr_retain_links=1e6, max_blk_sz=10000, srp_cutoff=3, run_aracne_flag=True,
                           sr_only=False, order_links=True, do_srp=True) -> MIResult:
    """End-to-end oracle of the a-5 loop; returns the rows the reference would write."""
    L = states.shape[0]
    max_blk_sz = r_round_thousands(max_blk_sz)
    blocks = make_blocks(L, max_blk_sz)
    cols = ["pos1", "pos2", "clust1", "clust2", "len", "MI"]
    sr_by_clust = [{k: [] for k in cols} for _ in range(nclust)]
    lr_rows = {k: [] for k in cols}
    approx = None if sr_only else lr_links_approx(POS, g, sr_dist)
    for (fs, fe, ts, te) in blocks:
        from_idx = np.arange(fs - 1, fe)
        to_idx = np.arange(ts - 1, te)
        if sr_only:  # drop sites that form no link < sr_dist with the other side (R/computePairwiseMI.R:179-189)
            POSf = np.asarray(POS, dtype=np.float64)
            ln = np.abs(circ_len(POSf[to_idx][None, :], POSf[from_idx][:, None], g))
            from_idx, to_idx = from_idx[(ln < sr_dist).any(axis=1)], to_idx[(ln < sr_dist).any(axis=0)]
            if len(from_idx) == 0 or len(to_idx) == 0:
                continue
        MI = mi_block_faithful(states, hdw, r, uqe, from_idx, to_idx)
        bl = block_links(MI, from_idx, to_idx, POS, paint, g, sr_dist, lr_retain_links, approx, sr_only)
        for k in cols:
            lr_rows[k].append(bl.lr[k])
        cm1, cm2 = bl.sr["clust1"], bl.sr["clust2"]
        for ci in range(1, nclust + 1):
            sel = (cm1 == ci) | (cm2 == ci)
            for k in cols:
                sr_by_clust[ci - 1][k].append(bl.sr[k][sel])
    lr_rows = {k: np.concatenate(v) for k, v in lr_rows.items()}
    sr_by_clust = [{k: np.concatenate(v) for k, v in d.items()} for d in sr_by_clust]
    res = MIResult(sr_links_red={}, lr_rows=lr_rows, sr_links_by_clust=sr_by_clust, blocks=blocks,
                   lr_links_approx=approx if approx is not None else float("nan"))
    if not do_srp:
        return res
    red, chk = merge_n_sort_sr_links(sr_by_clust, nclust, sr_dist, srp_cutoff, fit_data=res.fit_data)
    if run_aracne_flag:
        red["ARACNE"] = run_aracne(red["pos1"], red["pos2"], red["MI"],
                                   chk["pos1"], chk["pos2"], chk["MI"]).astype(np.float64)
    else:
        red["ARACNE"] = np.ones(len(red["pos1"]))
    if order_links:
        o = np.argsort(-red["srp_max"], kind="stable")   # order(decreasing=T) is stable (radix)
        red = {k: v[o] for k, v in red.items()}
    res.sr_links_red = red
    return res


# --------------------------------------------------------------------------
# §8 f rank 4  analyse_long_range_links (numeric core)      R/lr_analyser.R:72-118
# --------------------------------------------------------------------------
def analyse_long_range_links(lr: dict, sr: dict, min_links: int = 5000, are_lrlinks_ordered: bool = False) -> dict:
    """lr / sr: dicts of arrays pos1, pos2, MI (lr may carry more columns of the same length).  Returns the outlier
    links (rows of lr, re-ordered like the reference: descending MI unless are_lrlinks_ordered), their ARACNE flags,
    the Tukey quantiles / thresholds and whether the "~5000 top links" fallback (:97-102) was taken."""
    mi = np.asarray(lr["MI"], dtype=np.float64)
    q13 = np.array([quantile7(mi, 0.25), quantile7(mi, 0.75)])                     # :72
    iqr = q13[1] - q13[0]                                                           # :73
    thresholds = q13[1] + np.array([1.5, 3.0]) * iqr                                # :74
    keep = mi > thresholds.min()                                                    # :91
    fallback = False
    if keep.sum() < min_links and len(mi) >= min_links:                             # :94
        fallback = True
        # :97; the reference only gets here with >= 5000 rows, where both probabilities are in [0, 1]; a caller-chosen
        # min_links below 5000 may not be, so they are clamped at 0
        thresholds = np.array([quantile7(mi, max(0.0, 1 - (1 / len(mi)) * 4000)), quantile7(mi, max(0.0, 1 - (1 / len(mi)) * 5000))])
        keep = mi > thresholds.min()                                                # :98
    rows = np.nonzero(keep)[0]
    red = {k: np.asarray(v)[rows] for k, v in lr.items()}
    # ARACNE pool: rbind(lr, sr)[MI > min(thresholds)]                              :106-109
    cp1 = np.concatenate([np.asarray(lr["pos1"], dtype=np.float64), np.asarray(sr["pos1"], dtype=np.float64)])
    cp2 = np.concatenate([np.asarray(lr["pos2"], dtype=np.float64), np.asarray(sr["pos2"], dtype=np.float64)])
    cmi = np.concatenate([mi, np.asarray(sr["MI"], dtype=np.float64)])
    m = cmi > thresholds.min()
    aracne = run_aracne(np.asarray(red["pos1"], dtype=np.float64), np.asarray(red["pos2"], dtype=np.float64), red["MI"],
                        cp1[m], cp2[m], cmi[m])                                      # :111
    if not are_lrlinks_ordered:                                                     # :115-117: order(MI, decreasing = T), stable
        o = np.argsort(-np.asarray(red["MI"], dtype=np.float64), kind="stable")
        red = {k: v[o] for k, v in red.items()}
        aracne, rows = aracne[o], rows[o]
    return dict(red=red, rows=rows, ARACNE=aracne, q13=q13, thresholds=thresholds, fallback=fallback, n_pool=int(m.sum()))


# --------------------------------------------------------------------------
# §8 f rank 4  genomewide_LDMap (numeric core)              R/LDSummaryPlot.R:55-106, .mat :176-178, .rescale01 :157-163
# --------------------------------------------------------------------------
def ld_map(lr: dict, sr: dict, reducer=None, from_=None, to=None) -> dict:
    """Returns htm (B x B, rescaled to [0, 1]), pos_vec, reducer.  reducer <= 1 (the reference's unreduced branch) is
    not restated."""
    p_all = np.concatenate([np.asarray(lr["pos1"]), np.asarray(lr["pos2"]), np.asarray(sr["pos1"]), np.asarray(sr["pos2"])]).astype(np.int64)
    pos_vec = np.unique(p_all)                                                      # :56 / :59 sort(unique(...))
    L1, L2, S1, S2 = (np.asarray(lr["pos1"]).astype(np.int64), np.asarray(lr["pos2"]).astype(np.int64),
                      np.asarray(sr["pos1"]).astype(np.int64), np.asarray(sr["pos2"]).astype(np.int64))
    LM, SM = np.asarray(lr["MI"], dtype=np.float64), np.asarray(sr["MI"], dtype=np.float64)
    if from_ is not None:
        pos_vec = pos_vec[(pos_vec < to) & (pos_vec > from_)]                       # :62
        kl = (L1 >= from_) & (L1 <= to) & (L2 >= from_) & (L2 <= to)                # :66
        ks = (S1 >= from_) & (S1 <= to) & (S2 >= from_) & (S2 <= to)                # :67
        L1, L2, LM, S1, S2, SM = L1[kl], L2[kl], LM[kl], S1[ks], S2[ks], SM[ks]
    n = len(pos_vec)
    def idx(p):                                                                     # as.numeric(factor(p, levels = pos_vec)) - 1, NA -> -1
        i = np.searchsorted(pos_vec, p)
        ok = (i < n) & (pos_vec[np.minimum(i, n - 1)] == p)
        return np.where(ok, i, -1)
    i = np.concatenate([idx(L1), idx(L2), idx(S1), idx(S2)])                        # :78
    j = np.concatenate([idx(L2), idx(L1), idx(S2), idx(S1)])                        # :79
    x = np.concatenate([LM, LM, SM, SM])                                            # :80
    ok = (i >= 0) & (j >= 0)       # a link touching a window edge has no level in pos_vec (the reference would stop there)
    i, j, x = i[ok], j[ok], x[ok]
    r = int(np.round(n / 1e3)) if reducer is None else int(np.round(reducer))       # :83-87 (np.round = half-to-even like R)
    if r <= 1:
        raise ValueError("reducer <= 1: unreduced branch not restated")
    B = n // r                                                                      # matrix(..., n, n/r): ncol truncates
    # .mat(n, r): column k is 1 on rows [k r, (k+1) r) — recycling c(rep(1, r), rep(0, n)) down n-row columns shifts the
    # run of ones by r per column; rows beyond B r belong to no column
    bi, bj = i // r, j // r
    keep = (bi < B) & (bj < B)
    red = np.zeros((B, B))
    np.add.at(red, (bi[keep], bj[keep]), x[keep])                                   # crossprod(x, crossprod(sprs, x)) :93
    htm = red / r ** 2                                                              # :94
    htm = np.log10(htm + 1e-5)                                                      # :108
    mv = htm.min()
    htm = (htm - mv) / (htm.max() - mv)                                             # :109 .rescale01
    return dict(htm=htm, pos_vec=pos_vec, reducer=r)
