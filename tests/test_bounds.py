"""The bounds of the default path, brute-forced as FUNCTIONS (pytest -m gpu; BOUNDS.md states each bound and names its test here).

The default path never evaluates most pairs: it dismisses them on bounds derived by hand (BOUNDS.md) — the dual-digit weights' relative
error, the truncation of the approximate GEMM, the fp32 multi-cell bound of the screens, the per-SNP vertex bound behind the tile pruning, the
threshold table of the biallelic pairs.  Verify mode and the fuzzers check them end to end, on the alignments somebody thought of; round 5 found
two holes that way only after five rounds.  Here every bound is called through a test hook of the C ABI (csrc/ldw_debug.hip: the product's own
device functions on caller-made inputs) and compared with the MI formula of src/computeMI.cpp:19 in numpy fp64 — random and extremal joint
tables (empty cells, perfect association and anti-association, states of one sequence), quirk Q1's RXY != r_a r_b / 4, r != number of states,
four weightings.  (The threshold table's brute-force test is test_gpu_parity.py::test_threshold_table_never_dismisses_a_pair_that_reaches_the_level.)
"""
import itertools
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from ldweaver_amd import _lib as L
from ldweaver_amd.engine import Engine
from ldweaver_amd.synth import synth_alignment

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCREEN_EPS = 2e-4   # csrc/ldw_epi.h: what the screens subtract from a block's level before they compare


# ------------------------------------------------------------------------------------------------
# inputs
# ------------------------------------------------------------------------------------------------
def _weighting(kind, N, rng):
    if kind == "few":
        return rng.choice([0.5, 0.25, 1.0 / 3, 1.0 / 7, 0.02], size=N)
    if kind == "distinct":            # 1 / (#neighbours + 1) with every count different: the fine block exponents (per 32 positions)
        return 1.0 / (1.0 + rng.permutation(N).astype(np.float64))
    if kind == "unit":
        return np.ones(N)
    if kind == "wide":                # five orders of magnitude
        return 10.0 ** rng.uniform(-5.0, 0.0, N)
    raise ValueError(kind)


def _engine_with(eng, N, kind, seed):
    """An alignment of N sequences with the weighting `kind` on `eng`; returns (states, hdw).  "hamming": the alignment's own Hamming weights
    (clonal groups: a few dozen weight classes, exponent transitions between macro steps — the SURVEY recipe's case)."""
    rng = np.random.default_rng(seed)
    syn = synth_alignment(500, N, seed=seed)
    st = np.asarray(syn["states"])
    eng.set_engine(L.ENGINE_MFMA)
    eng.set_alignment(st)
    hdw = eng.hamming_weights(50) if kind == "hamming" else _weighting(kind, N, rng)
    cnt = eng.state_counts()
    uqe = (cnt > 0).T.astype(np.float64)
    eng.set_weights(hdw)
    eng.set_snp_meta(uqe.sum(axis=1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
    return st, hdw


def _random_state_pairs(rng, n, N, ka, kb):
    """n virtual SNP pairs over N sequences: a[case, seq] in 0..ka-1, b in 0..kb-1 — marginals from rare (one sequence) to balanced, association from
    independence to perfect (b a function of a) and perfect anti-association, plus states nobody carries."""
    a = np.empty((n, N), dtype=np.uint8)
    b = np.empty((n, N), dtype=np.uint8)
    for k in range(n):
        def marg(kk):
            if kk == 1:
                return np.ones(1)
            style = rng.integers(0, 4)
            if style == 0:      # one dominant state, the others rare (singletons .. a few per cent)
                p = np.concatenate([[1.0], rng.choice([0.5 / N, 1.5 / N, 4.0 / N, 0.01, 0.04], size=kk - 1)])
            elif style == 1:    # comparable frequencies
                p = rng.dirichlet(np.ones(kk))
            elif style == 2:    # one state absent
                p = rng.dirichlet(np.ones(kk))
                p[rng.integers(0, kk)] = 0.0
            else:
                p = rng.dirichlet(np.full(kk, 0.3))
            p = np.maximum(p, 0)
            return p / p.sum()
        pa_, pb_ = marg(ka), marg(kb)
        ak = rng.choice(ka, size=N, p=pa_)
        rho = rng.choice([0.0, 0.0, 0.5, 0.9, 1.0])
        phi = rng.integers(0, kb, size=ka)                       # b = phi(a) where the pair is coupled (kb = 2, phi a bijection, rho = 1: anti / perfect association)
        ind = rng.choice(kb, size=N, p=pb_)
        bk = np.where(rng.random(N) < rho, phi[ak], ind)
        if k % 17 == 0:                                          # exact copies / complements
            bk = (ak if k % 34 == 0 else (ka - 1 - ak)) % kb
        a[k], b[k] = ak, bk
    return a, b


def _sums(a, b, w, ka, kb):
    """S[case, i, j] = sum of w over the sequences with (a, b) = (i, j), exact in float64 (integers below 2^52)."""
    wf = w.astype(np.float64)
    out = np.zeros((len(a), ka, kb))
    for i in range(ka):
        ai = a == i
        for j in range(kb):
            out[:, i, j] = (ai & (b == j)).astype(np.float64) @ wf
    return out


def _mi_formula(n_fix, scale, ra, rb, rxy, neff, mask=None):
    """src/computeMI.cpp:19 with the operands of R/computePairwiseMI.R:390-398, fp64: n_fix[case, i, j] exact fixed-point joint sums (every state of both
    SNPs, whether flagged or not), mask[case, i, j] = uqX (x) uqY."""
    x = n_fix * scale + 0.5
    pX, pY = (n_fix.sum(axis=2) * scale), (n_fix.sum(axis=1) * scale)
    den = neff + 0.5 * ra * rb
    d = pX[:, :, None] * pY[:, None, :] + rxy[:, None, None] + pX[:, :, None] * (0.5 * ra)[:, None, None] + pY[:, None, :] * (0.5 * rb)[:, None, None]
    t = x / den[:, None, None] * np.log(x / d * den[:, None, None])
    if mask is not None:
        t = t * mask
    return t.sum(axis=(1, 2))


def _r_and_rxy(rng, n, ka, kb):
    """r of the two SNPs (mostly their number of states; the caller's `r` is free) and RXY: intended r_a r_b / 4 or, quirk Q1, r of two OTHER SNPs."""
    ra = np.where(rng.random(n) < 0.85, float(ka), rng.choice([2.0, 3.0, 4.0, 5.0], size=n))
    rb = np.where(rng.random(n) < 0.85, float(kb), rng.choice([2.0, 3.0, 4.0, 5.0], size=n))
    q1 = rng.choice([2.0, 2.0, 2.0, 3.0, 4.0, 5.0], size=n) * rng.choice([2.0, 2.0, 2.0, 3.0, 4.0, 5.0], size=n) * 0.25
    rxy = np.where(rng.random(n) < 0.4, ra * rb * 0.25, q1)
    return ra, rb, rxy


def _pack(n, ka, kb, G, mrg_a, mrg_b, pXf, pYf):
    g = np.zeros((n, 4, 4), dtype=np.int64)
    g[:, :kb - 1, :ka - 1] = np.transpose(G[:, :ka - 1, :kb - 1], (0, 2, 1))      # g[case][j][i]
    pa = np.zeros((n, 5), dtype=np.int64)
    pb = np.zeros((n, 5), dtype=np.int64)
    pX = np.zeros((n, 5), dtype=np.float32)
    pY = np.zeros((n, 5), dtype=np.float32)
    pa[:, :ka], pb[:, :kb] = mrg_a, mrg_b
    pX[:, :ka], pY[:, :kb] = pXf, pYf
    return g.reshape(n, 16), pa, pb, pX, pY


WEIGHTINGS = [("hamming", 2000, 11), ("few", 600, 12), ("distinct", 616, 13), ("wide", 400, 14), ("unit", 300, 15)]


# ------------------------------------------------------------------------------------------------
# BOUNDS.md 1-2: dual-digit weights and the truncation of the approximate GEMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,N,seed", WEIGHTINGS)
def test_dual_digit_weights_and_gemm_truncation_are_inside_their_bounds(engine, kind, N, seed):
    """(1) |V'_s / V_s - 1| <= delta for every sequence; (2) every entry of gemm_apx_kernel's output lies in (S' / 2^e_last - lost_units, S' / 2^e_last],
    S' = the exact integer sum of the dual-digit weights over the co-occurring sequences (R/computePairwiseMI.R:391's sum with V' for V) — for all rows
    of 60 SNPs against all rows of 60 others, rare and common, incl. the all-zero padding row."""
    st, hdw = _engine_with(engine, N, kind, seed)
    par, V, Va = engine.debug_apx_params()
    P = dict(zip(Engine.APX_PARAM_NAMES, par))
    if not int(P["flags"]) & 1:
        assert kind == "wide", engine.path_report()["apx_gate"]     # five orders of magnitude: the dual digits are too coarse and the path is OFF by its own gate
        assert P["delta"] > 4e-3
        return
    ok = V > 0
    assert np.all(np.abs(Va[ok] - V[ok]) <= P["delta"] * V[ok] * (1 + 1e-12)) and np.all(Va[~ok] == 0)
    assert P["delta"] <= 4e-3
    row0, meta = engine.debug_rows()
    rng = np.random.default_rng(seed)
    Ls = len(st)
    sn_t, sn_f = rng.choice(Ls, 60, replace=False), rng.choice(Ls, 60, replace=False)

    def rows_of(snps):
        rows, bits = [], []
        for a in snps.tolist():
            m = int(meta[a])
            for i in range(m & 7):
                rows.append(int(row0[a]) + i)
                bits.append(st[a] == ((m >> (8 + 3 * i)) & 7))
        return np.array(rows + [int(row0[-1])], dtype=np.int32), np.array(bits + [np.zeros(N, bool)])    # (+ the padding row R: no sequence)
    rt, bt = rows_of(sn_t)
    rf, bf = rows_of(sn_f)
    G = engine.debug_apx_gemm(rt, rf).astype(np.int64)
    S = (bt.astype(np.float64) * Va.astype(np.float64)) @ bf.astype(np.float64).T     # exact: integers below 2^52
    el = int(P["e_last"])
    lost = S - G * 2.0 ** el
    assert lost.min() >= 0.0, "an entry ABOVE the sum of the approximate weights"
    assert lost.max() <= P["lost_units"] * 2.0 ** el, (lost.max() / 2.0 ** el, P["lost_units"])   # (no exponent transition: every V' is a multiple of 2^e_last, nothing is lost)
    assert np.all(G[-1] == 0) and np.all(G[:, -1] == 0)
    # the approximate marginal the screens derive cells from is floor(sum V' / 2^e_last): one unit of slack, by definition
    print(f"{kind}: delta {P['delta']:.2e}, e_last {el}, lost units bound {P['lost_units']:.3f}, worst observed {lost.max() / 2.0 ** el:.3f}")


# ------------------------------------------------------------------------------------------------
# BOUNDS.md 3: the fp32 multi-cell bound of the approximate screens against the fp64 MI of the EXACT sums
# ------------------------------------------------------------------------------------------------
def _apx_inputs(rng, a, b, ka, kb, V, Va, P, adversarial_loss):
    n = len(a)
    scale = 2.0 ** -int(P["F"])
    el = 2.0 ** int(P["e_last"])
    Sx = _sums(a, b, V, ka, kb)                     # exact fixed-point joint sums
    Sa = _sums(a, b, Va, ka, kb)                    # sums of the dual-digit weights (fixed-point units)
    s = Sa / el                                     # ... in units of 2^e_last: what the GEMM approximates from below
    lost = P["lost_units"]
    loss = np.full(s.shape, np.nextafter(lost, 0)) if adversarial_loss else rng.uniform(0, lost, size=s.shape) if lost > 0 else np.zeros(s.shape)
    G = np.maximum(np.floor(s - loss), 0.0).astype(np.int64)          # any integer in (s - lost, s], never negative (the accumulators only add)
    mrg_a, mrg_b = np.floor(Sa.sum(axis=2) / el).astype(np.int64), np.floor(Sa.sum(axis=1) / el).astype(np.int64)
    pXf = (Sx.sum(axis=2) * scale).astype(np.float32)
    pYf = (Sx.sum(axis=1) * scale).astype(np.float32)
    return Sx, scale, _pack(n, ka, kb, G, mrg_a, mrg_b, pXf, pYf)


@pytest.mark.parametrize("kind,N,seed", WEIGHTINGS)
def test_approximate_screen_bound_is_an_upper_bound_of_the_exact_mi(engine, kind, N, seed):
    """full_cells_screen<NA, NB, APX> (the four straight-line variants of k_mi_screen / k_screen_maybe) and pair_screen_generic<APX> (k_mi_screen_generic):
    for joint tables built from the engine's own V and V', GEMM entries anywhere in their truncation interval and floor marginals, the fp32 value the
    screen compares with `level - SCREEN_EPS` is never below `MI(exact sums) - SCREEN_EPS / 2` — and usually within a few 1e-3 of it (not vacuous)."""
    _engine_with(engine, N, kind, seed)
    par, V, Va = engine.debug_apx_params()
    P = dict(zip(Engine.APX_PARAM_NAMES, par))
    assert int(P["flags"]) & 1
    rng = np.random.default_rng(seed + 100)
    n = 4000
    worst, slack = 0.0, []
    for (ka, kb), adv in itertools.product([(2, 2), (2, 3), (3, 2), (3, 3)], [False, True]):
        a, b = _random_state_pairs(rng, n, N, ka, kb)
        Sx, scale, (g, pa, pb, pX, pY) = _apx_inputs(rng, a, b, ka, kb, V, Va, P, adv)
        ra, rb, rxy = _r_and_rxy(rng, n, ka, kb)
        mi = _mi_formula(Sx, scale, ra, rb, rxy, P["neff"])
        ub = engine.debug_screen_bound(0, ka - 1, kb - 1, g, pa, pb, pX, pY, np.stack([ra, rb, rxy], axis=1), par).astype(np.float64)
        gap = ub - mi
        k = int(np.argmin(gap))
        assert gap[k] > -0.5 * SCREEN_EPS, (kind, ka, kb, adv, gap[k], mi[k], ub[k], Sx[k].tolist(), float(ra[k]), float(rb[k]), float(rxy[k]))
        worst = min(worst, float(gap.min()))
        slack.append(float(np.median(gap)))
    # the predicated screen: 0..4 indicator rows per side, unflagged slots (cells masked by uq = uqX (x) uqY), r free
    for (ka, kb), adv in itertools.product([(1, 2), (2, 4), (4, 3), (5, 5), (3, 1), (4, 4)], [False, True]):
        m = 2500
        a, b = _random_state_pairs(rng, m, N, ka, kb)
        Sx, scale, (g, pa, pb, pX, pY) = _apx_inputs(rng, a, b, ka, kb, V, Va, P, adv)
        ra, rb, rxy = _r_and_rxy(rng, m, ka, kb)
        fa = np.where(rng.random((m, ka)) < 0.85, 1, 0)
        fb = np.where(rng.random((m, kb)) < 0.85, 1, 0)
        mask = fa[:, :, None] * fb[:, None, :]
        ma = (ka - 1) | (fa * (1 << (3 + np.arange(ka)))).sum(axis=1)
        mb = (kb - 1) | (fb * (1 << (3 + np.arange(kb)))).sum(axis=1)
        mi = _mi_formula(Sx, scale, ra, rb, rxy, P["neff"], mask)
        ub = engine.debug_screen_bound(1, 0, 0, g, pa, pb, pX, pY, np.stack([ra, rb, rxy], axis=1), par, masks=np.stack([ma, mb], axis=1)).astype(np.float64)
        gap = ub - mi
        k = int(np.argmin(gap))
        assert gap[k] > -0.5 * SCREEN_EPS, ("generic", kind, ka, kb, adv, gap[k], mi[k], ub[k], Sx[k].tolist(), fa[k].tolist(), fb[k].tolist())
        worst = min(worst, float(gap.min()))
    assert max(slack) < 0.05, slack      # (not vacuous: the median table's bound is a few 1e-3 above its MI — delta x |log| terms + the lost units)
    print(f"{kind}: worst bound - MI {worst:.2e} (allowed > {-0.5 * SCREEN_EPS:.0e}), median slack per shape {['%.1e' % s for s in slack]}")


def test_exact_limb_screens_and_the_fp64_evaluation_match_the_formula(engine):
    """The screens of the limb paths evaluate the MI of the EXACT sums in fp32 (full_cells_screen<.., false>, pair_screen_generic<false>): within SCREEN_EPS / 2
    of the fp64 formula (ldw_epi.h states 1.3e-5).  full_cells_mi — the fp64 value every path EMITS — agrees with numpy to 1e-12 on the same tables, incl.
    RXY != r_a r_b / 4 (quirk Q1) and empty cells."""
    N = 600
    _engine_with(engine, N, "few", 21)
    par, V, Va = engine.debug_apx_params()
    P = dict(zip(Engine.APX_PARAM_NAMES, par))
    rng = np.random.default_rng(77)
    scale = 2.0 ** -int(P["F"])
    n = 5000
    e32 = e64 = 0.0
    for ka, kb in [(2, 2), (2, 3), (3, 2), (3, 3)]:
        a, b = _random_state_pairs(rng, n, N, ka, kb)
        Sx = _sums(a, b, V, ka, kb)
        Gx = Sx.astype(np.int64)
        g, pa, pb, pX, pY = _pack(n, ka, kb, Gx, Gx.sum(axis=2), Gx.sum(axis=1), (Sx.sum(axis=2) * scale).astype(np.float32), (Sx.sum(axis=1) * scale).astype(np.float32))
        ra, rb, rxy = _r_and_rxy(rng, n, ka, kb)
        mi = _mi_formula(Sx, scale, ra, rb, rxy, P["neff"])
        rr = np.stack([ra, rb, rxy], axis=1)
        v32 = engine.debug_screen_bound(2, ka - 1, kb - 1, g, pa, pb, pX, pY, rr, par).astype(np.float64)
        v64 = engine.debug_screen_bound(4, ka - 1, kb - 1, g, pa, pb, pX, pY, rr, par)
        e32, e64 = max(e32, float(np.abs(v32 - mi).max())), max(e64, float(np.abs(v64 - mi).max()))
    for ka, kb in [(1, 3), (4, 2), (5, 5)]:
        a, b = _random_state_pairs(rng, 2000, N, ka, kb)
        Sx = _sums(a, b, V, ka, kb)
        Gx = Sx.astype(np.int64)
        g, pa, pb, pX, pY = _pack(2000, ka, kb, Gx, Gx.sum(axis=2), Gx.sum(axis=1), (Sx.sum(axis=2) * scale).astype(np.float32), (Sx.sum(axis=1) * scale).astype(np.float32))
        ra, rb, rxy = _r_and_rxy(rng, 2000, ka, kb)
        fa, fb = np.where(rng.random((2000, ka)) < 0.85, 1, 0), np.where(rng.random((2000, kb)) < 0.85, 1, 0)
        ma = (ka - 1) | (fa * (1 << (3 + np.arange(ka)))).sum(axis=1)
        mb = (kb - 1) | (fb * (1 << (3 + np.arange(kb)))).sum(axis=1)
        mi = _mi_formula(Sx, scale, ra, rb, rxy, P["neff"], fa[:, :, None] * fb[:, None, :])
        v32 = engine.debug_screen_bound(3, 0, 0, g, pa, pb, pX, pY, np.stack([ra, rb, rxy], axis=1), par, masks=np.stack([ma, mb], axis=1)).astype(np.float64)
        e32 = max(e32, float(np.abs(v32 - mi).max()))
    assert e32 < 0.5 * SCREEN_EPS and e64 < 1e-12, (e32, e64)
    print(f"fp32 screens: max |value - MI| {e32:.2e} (SCREEN_EPS {SCREEN_EPS:.0e}); fp64 evaluation: {e64:.2e}")


@pytest.mark.parametrize("kind,N,seed", [("hamming", 2000, 31), ("distinct", 616, 32), ("wide", 400, 33)])
def test_mixed_precision_screen_margin_covers_the_low_limbs(engine, kind, N, seed):
    """The mixed-precision limb path screens on the sums of V_hi = (V - V_lo) / 2^16 (three high limbs) and widens its margin by lo_bound =
    lo_abs_sum (2 ln(neff + 12.5) + 3) / neff: the fp32 screen value of the HIGH-limb table is never more than lo_bound + SCREEN_EPS / 2 below the MI of the
    exact sums (lo_bound, ldw_mi_items.inc; the path switches itself off above lo_bound = 1e-3)."""
    _engine_with(engine, N, kind, seed)
    par, V, _ = engine.debug_apx_params()
    P = dict(zip(Engine.APX_PARAM_NAMES, par))
    if int(P["nlimbs"]) != 5:
        pytest.skip("the weighting does not take 5 limbs: no mixed-precision path")
    lo = ((V + 32896) % 65536) - 32896                 # two balanced base-256 digits: -32896 .. 32639
    Vhi = (V - lo) // 65536
    assert np.all(Vhi * 65536 + lo == V) and abs(float(np.abs(lo).sum()) * 2.0 ** -int(P["F"]) - P["lo_abs_sum"]) <= 1e-12 * max(1.0, P["lo_abs_sum"])
    tot_hi = int(Vhi.sum())
    shift = max(tot_hi.bit_length() - 31, 0)
    par_hi = par.copy()
    par_hi[13], par_hi[14] = shift, 2.0 ** (shift - int(P["F"]) + 16)
    rng = np.random.default_rng(seed)
    scale = 2.0 ** -int(P["F"])
    n, worst = 4000, 0.0
    for ka, kb in [(2, 2), (2, 3), (3, 2), (3, 3)]:
        a, b = _random_state_pairs(rng, n, N, ka, kb)
        Sx, Sh = _sums(a, b, V, ka, kb), _sums(a, b, Vhi, ka, kb).astype(np.int64)
        g, pa, pb, pX, pY = _pack(n, ka, kb, Sh, Sh.sum(axis=2), Sh.sum(axis=1), (Sx.sum(axis=2) * scale).astype(np.float32), (Sx.sum(axis=1) * scale).astype(np.float32))
        ra, rb, rxy = _r_and_rxy(rng, n, ka, kb)
        mi = _mi_formula(Sx, scale, ra, rb, rxy, P["neff"])
        v = engine.debug_screen_bound(2, ka - 1, kb - 1, g, pa, pb, pX, pY, np.stack([ra, rb, rxy], axis=1), par_hi).astype(np.float64)
        worst = min(worst, float((v - mi).min()))
        assert (v - mi).min() > -(P["lo_bound"] + 0.5 * SCREEN_EPS), (kind, ka, kb, float((v - mi).min()), P["lo_bound"])
    print(f"{kind}: lo_bound {P['lo_bound']:.2e}, worst high-limb screen - MI {worst:.2e}")


# ------------------------------------------------------------------------------------------------
# BOUNDS.md 4: the per-SNP bound behind the tile pruning, against arbitrary partners
# ------------------------------------------------------------------------------------------------
def test_per_snp_bound_holds_for_random_and_extremal_partners(engine):
    """ldw_snp_bounds (k_snp_sup): for SNPs with rare states, NO joint table with the SNP's marginals — partners drawn as row-stochastic maps from near-vertex to
    uniform, every vertex itself, both partner kinds, RXY intended and at / above quirk Q1's floor r_min^2 / 4 — has an MI above the bound; the best vertex reaches it
    (tight to 1e-9).  test_gpu_parity.py::test_snp_bounds_hold_for_every_partner checks the same values against every pair of an alignment."""
    rng = np.random.default_rng(9)
    Ls, N = 400, 300
    st = np.zeros((Ls, N), dtype=np.uint8)
    for a in range(Ls):
        maj, mnr = rng.choice(4, size=2, replace=False)
        st[a] = maj
        st[a, rng.choice(N, int(rng.choice([1, 1, 2, 3, 5, 9, 20])), replace=False)] = mnr
        if a % 3 == 0:
            st[a, rng.choice(N, int(rng.choice([1, 2, 6])), replace=False)] = 4
    cnt = np.stack([(st == x).sum(axis=1) for x in range(5)], axis=1)
    uqe = (cnt > 0).astype(np.float64)
    r = uqe.sum(axis=1)
    hdw = 1.0 / rng.integers(1, 6, size=N).astype(np.float64)
    engine.set_engine(L.ENGINE_MFMA)
    engine.set_alignment(st)
    engine.set_weights(hdw)
    engine.set_snp_meta(r, uqe, np.arange(1, Ls + 1, dtype=np.int32) * 50, np.ones(Ls, np.int32), 50.0 * Ls + 100)
    sup = engine.snp_bounds()
    v = np.sqrt(hdw) ** 2
    neff, rmin = float(hdw.sum()), float(r.min())
    tested = 0
    for a in np.where(sup[:, 0, 0] < 1e299)[0][:80].tolist():
        p = np.array([v[st[a] == x].sum() for x in range(5) if (st[a] == x).any()])
        ka = len(p)
        for kb in (2, 3):
            T = np.concatenate([rng.dirichlet(np.full(kb, al), size=(600, ka)) for al in (0.05, 0.3, 1.0)] +
                               [np.eye(kb)[list(phi)][None] for phi in itertools.product(range(kb), repeat=ka)])
            nj = p[None, :, None] * T                       # joint weights with a's marginals
            pY = nj.sum(axis=1)
            den = neff + 0.5 * ka * kb
            for m, rxys in ((0, [0.25 * ka * kb]), (1, [min(0.25 * ka * kb, 0.25 * rmin * rmin), 0.25 * rmin * (rmin + 1), 0.25 * 25])):
                for rxy in rxys:
                    D = p[None, :, None] * pY[:, None, :] + rxy + (p * 0.5 * ka)[None, :, None] + (pY * 0.5 * kb)[:, None, :]
                    mi = ((nj + 0.5) * np.log((nj + 0.5) * den / D)).sum(axis=(1, 2)) / den
                    assert mi.max() <= sup[a, m, kb - 2] + 1e-9, (a, kb, m, rxy, float(mi.max()), float(sup[a, m, kb - 2]))
                    if rxy == rxys[0]:
                        assert mi.max() >= sup[a, m, kb - 2] - 1e-9     # (the vertices are in T: the bound is attained)
        tested += 1
    assert tested >= 40


# ------------------------------------------------------------------------------------------------
# the fuzzers as part of the suite: >= 400 fixed-seed cases of tools/fuzz_paths.py, >= 100 of tools/fuzz_sr_model.py
# ------------------------------------------------------------------------------------------------
def _run_parallel(jobs, timeout):
    """Each job = an argument list of one tool; three run side by side (pytest + 3 processes on the GPU: below the box's limit of 6)."""
    t0 = time.time()
    pending, running, done = list(jobs), [], []
    while pending or running:
        while pending and len(running) < 3:
            args = pending.pop(0)
            running.append((args, subprocess.Popen([sys.executable] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        for item in list(running):
            args, pr = item
            try:
                out, err = pr.communicate(timeout=2)
            except subprocess.TimeoutExpired:
                if time.time() - t0 > timeout:
                    for _, q in running:
                        q.kill()
                    raise AssertionError(f"fuzz jobs still running after {timeout} s: {[x[0][1:] for x in running]}")
                continue
            running.remove(item)
            done.append((args, pr.returncode, out, err))
    return done


def test_fuzz_paths_400_cases():
    """tools/fuzz_paths.py: default / verify / no-span paths == the plain path bit for bit, cold and warm, zero verify-mode violations, problem after problem on ONE
    context per job — 2 x 100 plain draws of every shape, 2 x ~75 up to 9 000 SNPs and 2 x ~75 partly rewritten alignments with the extra weightings (five
    orders of magnitude, zero weights; up to 9 000 SNPs: the host-side rewriting dominates beyond): ~500 cases, three jobs side by side."""
    tool = os.path.join(ROOT, "tools", "fuzz_paths.py")
    jobs = [[tool, "--cases", "100", "--seed", str(s)] for s in (1101, 1102)] + [[tool, "--cases", "100", "--seed", str(s), "--max-l", "9000"] for s in (1103, 1104)] + \
           [[tool, "--cases", "100", "--seed", str(s), "--max-l", "9000", "--mutate", "mix", "--extra-weights"] for s in (1201, 1202)]
    total = 0
    for args, rc, out, err in _run_parallel(jobs, 500):
        assert rc == 0 and "DIFFERENT" not in out, (args[1:], out[-3000:], err[-1500:])
        total += out.count(": ok")
    assert total >= 400, total


def test_fuzz_sr_model_100_cases():
    """tools/fuzz_sr_model.py: the short-range model + ARACNE, the Tukey analysis and the LD map on contexts RE-USED from problem to problem == a fresh context; the model
    over two re-used contexts == the one-table model.  3 x 35 cases."""
    tool = os.path.join(ROOT, "tools", "fuzz_sr_model.py")
    total = 0
    for args, rc, out, err in _run_parallel([[tool, "--cases", "35", "--seed", str(s)] for s in (81, 82, 83)], 500):
        assert rc == 0 and "DIFFERENT" not in out, (args[1:], out[-3000:], err[-1500:])
        total += out.count(": ok")
    assert total >= 100, total
