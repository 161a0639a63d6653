#!/usr/bin/env python3
"""Randomised A/B of the engine's paths: for random shapes, weightings, block sizes, short-range distances, retention targets, quirk modes and
position layouts, the link tables of the DEFAULT path (approximate GEMM + screens + spans + speculation, cold and warm) must equal the PLAIN path's
(5-limb exact GEMM, fp64 MI of every pair) bit for bit, with zero screen violations in verify mode.  Prints one line per case; exit code 1 on
the first difference (the case's parameters are in the line: rerun with --only K).

    python tools/fuzz_paths.py --cases 40 [--seed 7] [--only K]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd.engine import Engine  # noqa: E402
from ldweaver_amd import _lib as L  # noqa: E402
from ldweaver_amd.mi import lr_links_approx, make_blocks  # noqa: E402
from ldweaver_amd.synth import synth_alignment  # noqa: E402


EXTRA_WEIGHTS = False   # --extra-weights: also weights over five orders of magnitude and weightings with a fifth of the sequences at weight 0 (not in the
                        # default draw: the committed seeds keep their cases)


def case_params(rs):
    Ls = int(rs.choice([700, 1500, 2600, 4100, 6000, 9000, 15000, 24000]))
    N = int(rs.choice([40, 130, 257, 616, 1000, 2100, 5000]))
    B = int(rs.choice([1000, 2000, 3000, 5000, 10000]))
    return dict(L=Ls, N=N, B=B, kind=str(rs.choice(["survey", "survey", "adversarial"])), weights=str(rs.choice(["hamming", "hamming", "distinct", "unit", "few"] + (["wide", "zeros"] if EXTRA_WEIGHTS else []))),
                sr_dist=float(rs.choice([20000.0, 3000.0, 60000.0, 500.5])), retain=float(rs.choice([2e4, 2e5, 1e6, 3e3])), quirk=int(rs.integers(0, 2)),
                pos=str(rs.choice(["recipe", "recipe", "dense", "shuffled_some"])), seed=int(rs.integers(1, 10 ** 6)))


def mutate(st, kind, r2):
    """Edge structure the recipe does not draw by itself (--mutate): every SNP row of `st` (L x N, states 0..4) may be rewritten.
    copies: 5 % of the SNPs become exact copies of another SNP under a random relabelling of its states (perfect association and,
    for two states, perfect anti-association, long-range as well as short-range); gaps: 10 % of the SNPs get 30-70 % gaps; states: 10 % of
    the SNPs get three to five states at comparable frequencies; half: 10 % of the SNPs are rewritten to two states of exactly N / 2
    sequences each (flagged-by-count ties)."""
    Ls, N = st.shape
    st = st.copy()
    pick = lambda frac: r2.choice(Ls, size=max(1, int(frac * Ls)), replace=False)
    if kind in ("copies", "all"):
        dst = pick(0.05)
        src = r2.integers(0, Ls, len(dst))
        for d, s_ in zip(dst.tolist(), src.tolist()):
            st[d] = r2.permutation(5).astype(st.dtype)[st[s_]]
    if kind in ("gaps", "all"):
        for d in pick(0.10).tolist():
            st[d, r2.random(N) < r2.uniform(0.3, 0.7)] = 4
    if kind in ("states", "all"):
        for d in pick(0.10).tolist():
            k = int(r2.integers(3, 6))
            st[d] = r2.permutation(5)[:k].astype(st.dtype)[r2.integers(0, k, N)]
    if kind in ("half", "all"):
        for d in pick(0.10).tolist():
            row = np.zeros(N, dtype=st.dtype)
            row[r2.permutation(N)[: N // 2]] = 1
            st[d] = r2.permutation(4)[:2].astype(st.dtype)[row]
    return st


def tables(eng):
    return eng.links(0), eng.links(1), eng.block_stats()


def same(x, y):
    for w in (0, 1):
        for u, v in zip(x[w], y[w]):
            if not np.array_equal(u, v):
                return False
    for k in ("n_lr_total", "n_lr_kept", "n_sr"):
        if not np.array_equal(x[2][k], y[2][k]):
            return False
    return bool(np.array_equal(x[2]["disc_thresh"], y[2]["disc_thresh"], equal_nan=True))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--start", type=int, default=0, help="skip the cases before this one (their parameters are still drawn)")
    ap.add_argument("--mutate", default="none", choices=["none", "copies", "gaps", "states", "half", "all", "mix"], help="rewrite part of every alignment (see mutate()); mix: a random kind per case")
    ap.add_argument("--max-l", type=int, default=0, help="skip the cases with more SNPs than this (0: none; the host-side rewriting of --mutate is slow on the largest)")
    ap.add_argument("--extra-weights", action="store_true", help="also draw weights over five orders of magnitude and weightings with zeros")
    a = ap.parse_args()
    global EXTRA_WEIGHTS
    EXTRA_WEIGHTS = a.extra_weights
    rs = np.random.default_rng(a.seed)
    bad = 0
    eng = Engine(0)
    for k in range(a.cases):
        p = case_params(rs)
        mkind = a.mutate if a.mutate != "mix" else str(np.random.default_rng(a.seed * 1000 + k).choice(["copies", "gaps", "states", "half", "all"]))
        if (a.only >= 0 and k != a.only) or k < a.start or (a.max_l and p["L"] > a.max_l):
            continue
        t0 = time.time()
        syn = synth_alignment(p["L"], p["N"], seed=p["seed"], kind=p["kind"])
        st, POS, paint, g = syn["states"], syn["POS"].copy(), syn["paint"], float(syn["g"])
        r2 = np.random.default_rng(p["seed"])
        if mkind != "none":
            st = mutate(np.asarray(st), mkind, np.random.default_rng(p["seed"] + 1))
            p = dict(p, mutate=mkind)
        if p["pos"] == "dense":          # a genome barely longer than the SNPs are many: nearly every pair is short-range at the larger distances
            POS = np.sort(r2.choice(np.arange(1, 3 * p["L"]), size=p["L"], replace=False)).astype(POS.dtype)
            g = float(3 * p["L"] + 7)
        elif p["pos"] == "shuffled_some":   # a few positions out of order: blocks whose lists do not ascend take the generic path
            for _ in range(5):
                i, j = r2.integers(0, p["L"], 2)
                POS[i], POS[j] = POS[j], POS[i]
        eng.set_engine(L.ENGINE_MFMA)
        eng.set_alignment(st)
        cnt = eng.state_counts()
        uqe = (cnt > 0).T.astype(np.float64)
        r = uqe.sum(axis=1)
        if p["weights"] == "hamming":
            hdw = eng.hamming_weights(int(p["L"] * 0.1))
        elif p["weights"] == "distinct":
            hdw = 1.0 / (1.0 + r2.permutation(p["N"]).astype(np.float64))
        elif p["weights"] == "unit":
            hdw = np.ones(p["N"])
        elif p["weights"] == "wide":
            hdw = 10.0 ** r2.uniform(-5.0, 0.0, p["N"])
        elif p["weights"] == "zeros":
            hdw = np.where(r2.random(p["N"]) < 0.2, 0.0, r2.choice([1.0, 0.5, 0.25, 1.0 / 3], size=p["N"]))
            hdw[0] = 1.0
        else:
            hdw = r2.choice([0.5, 0.25, 1.0 / 3, 1.0 / 7, 0.02], size=p["N"])
        eng.set_weights(hdw)
        eng.set_snp_meta(r, uqe, POS, paint, g)
        approx = lr_links_approx(POS, g, p["sr_dist"])
        blocks = make_blocks(p["L"], p["B"])
        kw = dict(sr_dist=p["sr_dist"], lr_retain_links=p["retain"], lr_links_approx=approx or 1.0, quirk=p["quirk"])
        res = {}
        for name, (scr, path, mixed, span) in dict(plain=(0, 1, False, False), default=(1, 0, True, True), verify=(2, 0, True, True), nospan=(1, 0, True, False)).items():
            eng.set_screen(scr)
            eng.set_path(path)
            eng.set_mixed(mixed)
            eng.set_span(span)
            c0 = eng.counters()
            eng.reset_speculation()
            eng.mi_all_pairs(blocks, **kw)
            cold = tables(eng)
            eng.mi_all_pairs(blocks, **kw)      # warm: inherits the guesses
            warm = tables(eng)
            c1 = eng.counters()
            res[name] = (cold, warm, {q: c1[q] - c0[q] for q in c1})
        eng.set_screen(1)
        eng.set_path(0)
        eng.set_mixed(True)
        eng.set_span(True)
        eq = {f"{n}.{'cold' if i == 0 else 'warm'}": same(res["plain"][0], res[n][i]) for n in ("plain", "default", "verify", "nospan") for i in (0, 1)}
        ok = all(eq.values())
        if not ok:
            print("   differs from plain.cold:", [k2 for k2, v in eq.items() if not v], flush=True)
            for n in ("default", "verify", "nospan"):
                a_, b_ = res["plain"][0], res[n][0]
                print(f"   {n}: sr rows {len(a_[0][2])} / {len(b_[0][2])}, lr rows {len(a_[1][2])} / {len(b_[1][2])}, kept per block equal {np.array_equal(a_[2]['n_lr_kept'], b_[2]['n_lr_kept'])}, "
                      f"thresholds equal {np.array_equal(a_[2]['disc_thresh'], b_[2]['disc_thresh'], equal_nan=True)}, counters {res[n][2]}", flush=True)
        viol = res["verify"][2]["screen_violations"]
        n_sr, n_lr = len(res["plain"][0][0][2]), len(res["plain"][0][1][2])
        print(f"case {k}: {'ok ' if ok and viol == 0 else 'DIFFERENT'} {p} rows sr {n_sr} lr {n_lr} apx_blocks {res['default'][2]['apx_blocks']} misses {res['default'][2]['spec_misses']} "
              f"violations {viol} gate {eng.path_report()['apx_gate'][:40]!r} {time.time() - t0:.1f} s", flush=True)
        if not ok or viol:
            bad += 1
            break
    eng.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
