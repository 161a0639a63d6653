"""Randomised parity against the oracle on ONE re-used context (r05).  The fixed-input parity tests of test_gpu_parity.py each start from a fresh problem; a
sequence of random small problems — shapes, weightings, block sizes, short-range distances, retention targets, index lists in any order — run one after the other
on the same engine also covers state that survives between problems (tools/fuzz_paths.py found such a bug in r05).  Checked per case, through the C ABI:
Hamming weights (bit-exact), unweighted joint counts of random pairs (bit-exact), the dense MI of a random index-list pair in the reference's quirk mode against
the block-faithful oracle and in the intended mode against the per-pair direct oracle (1e-10), and the link tables of every block pair against the oracle's
selection rule (R/computePairwiseMI.R:306-364) applied to the device's own dense MI: short-range rows in the reference's order with the dense block's bits,
long-range rows equal as a set with the same threshold.  Part of the cases rewrite a share of the alignment first (tools/fuzz_paths.py mutate():
copies and relabelled copies of SNPs, 30-70 % gaps, three to five states at comparable frequencies, two states of exactly N / 2 sequences)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ldw_oracle as orc  # noqa: E402
from ldweaver_amd import _lib as L  # noqa: E402
from ldweaver_amd.synth import synth_alignment  # noqa: E402
from fuzz_paths import mutate  # noqa: E402  (tools/: rewrites part of an alignment — copies and complements of SNPs, heavy gaps, 3-5 states, exact halves)

pytestmark = pytest.mark.gpu


def _case(rs):
    return dict(L=int(rs.choice([230, 517, 900, 1400])), N=int(rs.choice([17, 64, 129, 300])), B=int(rs.choice([1000, 2000])),
                kind=str(rs.choice(["survey", "survey", "adversarial"])), weights=str(rs.choice(["hamming", "hamming", "few", "unit", "distinct", "wide", "zeros"])),
                thr=float(rs.choice([0.1, 0.3])), sr_dist=float(rs.choice([20000.0, 3000.0, 100000.0])), retain=float(rs.choice([2e3, 2e4, 1e6])),
                quirk=int(rs.integers(0, 2)), seed=int(rs.integers(1, 10 ** 6)), mutate=str(rs.choice(["none", "none", "copies", "gaps", "states", "half", "all"])))


_SEEDS = [int(x) for x in os.environ.get("LDW_FUZZ_SEEDS", "3,14").split(",")]   # (more seeds for a longer hunt: LDW_FUZZ_SEEDS=1,2,3,...)


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_problems_on_one_context_against_the_oracle(engine, seed):
    rs = np.random.default_rng(seed)
    n_lr_checked = n_sr_checked = 0
    for k in range(14):
        p = _case(rs)
        tag = (k, p)
        syn = synth_alignment(p["L"], p["N"], seed=p["seed"], kind=p["kind"])
        st, POS, paint, g = syn["states"], syn["POS"], syn["paint"], float(syn["g"])
        r2 = np.random.default_rng(p["seed"])
        if p["mutate"] != "none":
            st = mutate(st, p["mutate"], np.random.default_rng(p["seed"] + 1))
        uqe, r = orc.uqe_r(st)
        engine.set_engine(L.ENGINE_MFMA)
        engine.set_alignment(st)
        # Hamming weights: bit-exact
        hd = engine.hamming_weights(int(p["L"] * p["thr"]))
        assert np.array_equal(hd, orc.hamming_weights(st, p["thr"])), tag
        hdw = {"hamming": hd, "unit": np.ones(p["N"]), "few": r2.choice([0.5, 0.25, 1.0 / 3, 0.02], size=p["N"]),
               "distinct": 1.0 / (1.0 + r2.permutation(p["N"])), "wide": 10.0 ** r2.uniform(-5.0, 0.0, p["N"]),
               "zeros": np.where(r2.random(p["N"]) < 0.2, 0.0, r2.choice([1.0, 0.5, 0.25, 1.0 / 3], size=p["N"]))}[p["weights"]].astype(np.float64)
        hdw[0] = max(hdw[0], 0.25)   # (never all zero)
        engine.set_weights(hdw)
        engine.set_snp_meta(r, uqe, POS, paint, g)
        # joint counts of random pairs: bit-exact
        pa, pb = r2.integers(0, p["L"], 6), r2.integers(0, p["L"], 6)
        cnt, _, _ = engine.joint_tables(pa, pb)
        for q in range(6):
            assert np.array_equal(cnt[q], orc.joint_counts(st, int(pa[q]), int(pb[q]))), tag
        # dense MI of a random (ragged, unordered) index-list pair
        fi = r2.choice(p["L"], size=int(r2.integers(3, 120)), replace=False)
        ti = r2.choice(p["L"], size=int(r2.integers(3, 120)), replace=False)
        MI = engine.mi_block(fi, ti, quirk=L.QUIRK_REFERENCE)
        assert np.abs(MI - orc.mi_block_faithful(st, hdw, r, uqe, fi, ti)).max() < 1e-10, tag
        MIi = engine.mi_block(fi, ti, quirk=L.QUIRK_INTENDED)
        for _ in range(5):
            i, j = int(r2.integers(0, len(fi))), int(r2.integers(0, len(ti)))
            assert abs(MIi[i, j] - orc.mi_pair_direct(st, hdw, r, uqe, int(fi[i]), int(ti[j]))) < 1e-10, tag
        # link tables of every block pair against the oracle's rule on the device's own dense MI
        blocks = np.array(orc.make_blocks(p["L"], p["B"]), dtype=np.int32)
        approx = orc.lr_links_approx(POS, g, p["sr_dist"]) or 1.0
        engine.reset_speculation()
        engine.mi_all_pairs(blocks, p["sr_dist"], p["retain"], approx, quirk=p["quirk"])
        stt = engine.block_stats()
        (sa, sb, smi), (la, lb, lmi) = engine.links(0), engine.links(1)
        so = np.concatenate([[0], np.cumsum(stt["n_sr"])])
        lo = np.concatenate([[0], np.cumsum(stt["n_lr_kept"])])
        for bi, (fs, fe, ts, te) in enumerate(blocks.tolist()):
            f_idx, t_idx = np.arange(fs - 1, fe), np.arange(ts - 1, te)
            dense = engine.mi_block(f_idx, t_idx, quirk=p["quirk"])
            want = orc.block_links(dense, f_idx, t_idx, POS, paint, g, p["sr_dist"], p["retain"], approx)
            s0, s1 = so[bi], so[bi + 1]
            assert np.array_equal(sa[s0:s1], want.sr["a"]) and np.array_equal(sb[s0:s1], want.sr["b"]) and np.array_equal(smi[s0:s1], want.sr["MI"]), (tag, bi)
            assert stt["n_lr_total"][bi] == want.n_lr_total, (tag, bi)
            l0, l1 = lo[bi], lo[bi + 1]
            got = {(int(x), int(y)): float(m) for x, y, m in zip(la[l0:l1], lb[l0:l1], lmi[l0:l1])}
            ref = {(int(x), int(y)): float(m) for x, y, m in zip(want.lr["a"], want.lr["b"], want.lr["MI"])}
            assert got == ref, (tag, bi, len(got), len(ref))
            if want.n_lr_total:
                assert stt["disc_thresh"][bi] == want.disc_thresh, (tag, bi)
            n_sr_checked += s1 - s0
            n_lr_checked += l1 - l0
    assert n_sr_checked > 1000 and n_lr_checked > 1000
