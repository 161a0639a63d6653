#!/usr/bin/env python3
"""GPU bring-up checks (run on the GPU box): python tests/bringup_gpu.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ldw_oracle as orc  # noqa: E402
from ldweaver_amd import _lib as L  # noqa: E402
from ldweaver_amd.engine import Engine  # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, "tests/golden/snp_sample_states.npz"))
    o = np.load(os.path.join(ROOT, "tests/golden/snp_sample_oracle.npz"))
    states, POS = g["states"], g["POS"]
    hdw, r, uqe, paint, G = o["hdw"], o["r"], o["uqe"], o["paint"], float(o["g"])
    Ls, Ns = states.shape
    eng = Engine(0)
    eng.set_alignment(states)
    assert np.array_equal(eng.get_alignment(), states), "alignment round trip"
    cnt = eng.state_counts()
    assert np.array_equal(cnt, orc.acgtn_table(states)), "state counts"
    print("alignment + counts OK")

    # Hamming weights
    t0 = time.time()
    hw, shared = eng.hamming_weights(int(Ls * 0.1), want_shared=True)
    print("hamming ms", (time.time() - t0) * 1e3, eng.last_timing())
    sh_ref = orc.shared_counts(states)
    print("shared exact:", np.array_equal(shared, sh_ref), "hdw exact:", np.array_equal(hw, hdw))

    eng.set_weights(hdw)
    eng.set_snp_meta(r, uqe, POS, paint, G)

    # joint tables
    rng = np.random.default_rng(1)
    pa = rng.integers(0, Ls, 300)
    pb = rng.integers(0, Ls, 300)
    cntT, fixT, fb = eng.joint_tables(pa, pb)
    ok = True
    for k in range(len(pa)):
        ok &= np.array_equal(cntT[k], orc.joint_counts(states, pa[k], pb[k]))
    print("joint counts exact:", ok, "frac_bits", fb)
    sq = np.sqrt(hdw)
    V = np.rint(np.ldexp(sq * sq, fb)).astype(np.int64)
    ok = True
    for k in range(len(pa)):
        code = states[pa[k]].astype(np.int64) * 5 + states[pb[k]]
        ref = np.bincount(code, weights=None, minlength=25) * 0
        ref = np.array([V[code == c].sum() for c in range(25)]).reshape(5, 5)
        ok &= np.array_equal(fixT[k], ref)
    print("fixed-point joint sums exact:", ok)

    # single block MI, both engines, both quirk modes
    idx = np.arange(Ls)
    for engine in (L.ENGINE_MFMA, L.ENGINE_HIST):
        eng.set_engine(engine)
        t0 = time.time()
        MI = eng.mi_block(idx, idx)
        dt = time.time() - t0
        sub = MI[np.ix_(o["sub_r"], o["sub_c"])]
        print("engine", engine, "single block: max|dMI| sub", np.abs(sub - o["MI_single_sub"]).max(),
              "colsum", np.abs(MI.sum(axis=0) - o["MI_single_colsum"]).max(), "ms", dt * 1e3, eng.last_timing())
        for bi, (fs, fe, ts, te) in enumerate(o["blocks"]):
            Mb = eng.mi_block(np.arange(fs - 1, fe), np.arange(ts - 1, te))
            print("  blk", bi, Mb.shape, "max|d|", np.abs(Mb[::7, ::5] - o[f"MI_blk{bi}_sub"]).max(),
                  "rowsum", np.abs(Mb.sum(axis=1) - o[f"MI_blk{bi}_rowsum"]).max())
    eng.set_engine(L.ENGINE_MFMA)

    # full pipeline vs oracle digests
    for tag, mb, retain in (("single", 10000, 1e6), ("multi", 1000, 1e5)):
        blocks = np.array(orc.make_blocks(Ls, mb), dtype=np.int32)
        approx = float(o[f"{tag}_lr_approx"])
        eng.mi_all_pairs(blocks, 20000.0, retain, approx)
        a, b, mi = eng.links(1)
        st = eng.block_stats()
        print(tag, "lr rows", len(mi), "oracle", int(o[f"{tag}_lr_n"]), st)
        if len(mi) == int(o[f"{tag}_lr_n"]):
            pos1, pos2 = POS[b].astype(float), POS[a].astype(float)
            print("   lr head pos1", np.array_equal(pos1[:200], o[f"{tag}_lr_pos1_head"]), "pos2",
                  np.array_equal(pos2[:200], o[f"{tag}_lr_pos2_head"]), "MI", np.abs(mi[:200] - o[f"{tag}_lr_MI_head"]).max(),
                  "tail", np.abs(mi[-200:] - o[f"{tag}_lr_MI_tail"]).max(), "sum", abs(mi.sum() - float(o[f"{tag}_lr_MI_sum"])))
        a, b, mi = eng.links(0)
        pos1, pos2 = POS[b].astype(float), POS[a].astype(float)
        c1, c2 = paint[b], paint[a]
        tot = 0
        for ci in (1, 2, 3):
            sel = (c1 == ci) | (c2 == ci)
            n_ref = int(o[f"{tag}_sr{ci}_n"])
            good = sel.sum() == n_ref
            if good:
                good &= np.array_equal(pos1[sel][:200], o[f"{tag}_sr{ci}_pos1_head"])
                good &= np.array_equal(pos2[sel][-200:], o[f"{tag}_sr{ci}_pos2_tail"])
                dm = np.abs(mi[sel][:200] - o[f"{tag}_sr{ci}_MI_head"]).max()
                ds = abs(mi[sel].sum() - float(o[f"{tag}_sr{ci}_MI_sum"]))
            else:
                dm = ds = float("nan")
            print("   sr clust", ci, "rows", int(sel.sum()), "oracle", n_ref, "order ok", bool(good), "dMI", dm, "dsum", ds)
        print("   timing", eng.last_timing())
    print("BRINGUP DONE")


if __name__ == "__main__":
    main()
