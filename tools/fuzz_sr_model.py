#!/usr/bin/env python3
"""Randomised check for STATE that survives between problems on one context, downstream of the MI pass: for a sequence of random problems the short-range model + ARACNE
(ldw_sr_len_quantiles ... ldw_aracne_device), the long-range Tukey analysis and the LD map on a context that is RE-USED from case to case must equal, bit for bit, what a
fresh context gives for the same problem; and the model over two re-used contexts (LDW_MI_SR_ROWS_STAY + ldw_sr_*_multi) must equal the one-table model.

    python tools/fuzz_sr_model.py --cases 30 [--seed 5]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd.engine import Engine, EngineGroup  # noqa: E402
from ldweaver_amd.mi import lr_links_approx, make_blocks  # noqa: E402
from ldweaver_amd.srp import merge_n_sort_sr_links_device  # noqa: E402
from ldweaver_amd.synth import synth_alignment  # noqa: E402


def setup(e, st, hdw, r, uqe, POS, paint, g):
    e.set_alignment(st)
    e.set_weights(hdw)
    e.set_snp_meta(r, uqe, POS, paint, g)


def model(view, nclust, sr_dist, cut, POS, paint, g, block_rows):
    try:
        red, flags, aux = merge_n_sort_sr_links_device(view, nclust, sr_dist, cut, POS, paint, g, run_aracne=True, order_links=True, block_rows=block_rows)
    except ValueError as e:   # (a cluster without two links above its fitted decay: the reference's fitdist stops the same way)
        return ("error", str(e)[:60])
    return red, flags, aux


def same_model(x, y):
    if x[0] == "error" or y[0] == "error":
        return x[0] == y[0]
    for k in ("a", "b", "MI", "clust_c", "srp_max", "first_clust", "dup"):
        if not np.array_equal(x[0][k], y[0][k]):
            return False
    return bool(np.array_equal(x[1], y[1]) and np.array_equal(x[2]["shape"], y[2]["shape"]) and np.array_equal(x[2]["mean_dist"], y[2]["mean_dist"], equal_nan=True))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    rs = np.random.default_rng(a.seed)
    A, C = Engine(0), Engine(0)       # re-used from case to case
    bad = 0
    for k in range(a.cases):
        Ls = int(rs.choice([1200, 2500, 4000, 7000, 12000]))
        N = int(rs.choice([60, 200, 616, 1500]))
        B = int(rs.choice([1000, 2000, 3000]))
        nclust = int(rs.choice([2, 3, 4, 6]))
        sr_dist = float(rs.choice([20000.0, 3000.0, 700.0, 60000.0]))
        cut = float(rs.choice([3.0, 1.0, 0.0, 8.0]))
        seed = int(rs.integers(1, 10 ** 6))
        t0 = time.time()
        syn = synth_alignment(Ls, N, seed=seed)
        st, POS, g = syn["states"], syn["POS"], float(syn["g"])
        paint = (np.arange(Ls) * nclust // Ls + 1).astype(syn["paint"].dtype) if rs.random() < 0.5 else ((syn["paint"] - 1) % nclust + 1).astype(syn["paint"].dtype)
        paint[:nclust] = np.arange(1, nclust + 1)
        blocks = make_blocks(Ls, B)
        approx = lr_links_approx(POS, g, sr_dist) or 1.0
        kw = dict(sr_dist=sr_dist, lr_retain_links=5e4, lr_links_approx=approx)
        Bf = Engine(0)                # fresh for this case
        try:
            Bf.set_alignment(st)
            cnt = Bf.state_counts()
            uqe = (cnt > 0).T.astype(np.float64)
            r = uqe.sum(axis=1)
            hdw = Bf.hamming_weights(int(Ls * 0.1))
            for e in (A, C, Bf):
                setup(e, st, hdw, r, uqe, POS, paint, g)
            out = {}
            for name, e in (("reused", A), ("fresh", Bf)):
                e.reset_speculation()
                e.mi_all_pairs(blocks, **kw)
                tabs = (e.links(0), e.links(1))
                m = model(e, nclust, sr_dist, cut, POS, paint, g, e.block_stats()["n_sr"])
                e.mi_all_pairs(blocks, **kw)   # (the model above left the kept links' state; the tables again for the consumers below)
                tuk = None
                try:
                    if m[0] != "error" and len(m[0]["MI"]):
                        info = e.lr_tukey(300, sr=(m[0]["a"], m[0]["b"], m[0]["MI"]))
                        lr = e.lr_reduced()
                        tuk = (info["q13"], info["thresholds"], lr["row"], e.aracne_device())
                except Exception as ex:   # noqa: BLE001
                    tuk = ("error", str(ex)[:60])
                try:
                    ld = e.ldmap(7)[0]
                except Exception as ex:   # noqa: BLE001
                    ld = np.array([hash(str(ex)[:40]) % 1000])
                out[name] = (tabs, m, tuk, ld)
            ok = all(np.array_equal(u, v) for w in (0, 1) for u, v in zip(out["reused"][0][w], out["fresh"][0][w]))
            ok = ok and same_model(out["reused"][1], out["fresh"][1])
            tr, tf = out["reused"][2], out["fresh"][2]
            ok = ok and ((tr is None and tf is None) or (tr is not None and tf is not None and len(tr) == len(tf) and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(tr, tf))))
            ok = ok and np.allclose(out["reused"][3], out["fresh"][3], rtol=0, atol=1e-12, equal_nan=True)
            # the model over two re-used contexts
            A.reset_speculation()
            C.reset_speculation()
            Engine.mi_all_pairs_multi([A, C], blocks, sr_rows_stay=True, **kw)
            mg = model(EngineGroup([A, C]), nclust, sr_dist, cut, POS, paint, g, None)
            ok2 = same_model(mg, out["fresh"][1])
            n_red = "error" if out["fresh"][1][0] == "error" else len(out["fresh"][1][0]["MI"])
            print(f"case {k}: {'ok ' if ok and ok2 else 'DIFFERENT'} L {Ls} N {N} B {B} nclust {nclust} sr_dist {sr_dist} cut {cut} seed {seed}: sr rows {len(out['fresh'][0][0][2])} kept {n_red} "
                  f"reused==fresh {ok} two-contexts==one-table {ok2} {time.time() - t0:.1f} s", flush=True)
            if not (ok and ok2):
                bad += 1
                break
        finally:
            Bf.close()
    A.close()
    C.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
