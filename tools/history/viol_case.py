#!/usr/bin/env python3
"""Reproduce one case of tools/fuzz_paths.py in verify mode and print the pairs the screen would have lost (ldw_debug_violations) with their weighted 2 x 2 tables."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd import _lib as L
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment
p = dict(L=24000, N=130, B=10000, kind='survey', sr_dist=500.5, retain=20000.0, quirk=1, seed=783690)
syn = synth_alignment(p["L"], p["N"], seed=p["seed"], kind=p["kind"])
st, POS, paint, g = syn["states"], syn["POS"], syn["paint"], float(syn["g"])
with Engine(0) as e:
    e.set_alignment(st)
    cnt = e.state_counts(); uqe = (cnt > 0).T.astype(np.float64); r = uqe.sum(axis=1)
    hdw = e.hamming_weights(int(p["L"] * 0.1))
    e.set_weights(hdw); e.set_snp_meta(r, uqe, POS, paint, g)
    approx = lr_links_approx(POS, g, p["sr_dist"])
    blocks = make_blocks(p["L"], p["B"])
    e.set_screen(2)
    e.reset_speculation()
    e.mi_all_pairs(blocks, p["sr_dist"], p["retain"], approx, quirk=p["quirk"])
    e.mi_all_pairs(blocks, p["sr_dist"], p["retain"], approx, quirk=p["quirk"])
    out = np.zeros(65)
    L.check(L.lib().ldw_debug_violations(e._ctx, L.ptr(out)))
    print("violations", int(out[0]), "neff", float(hdw.sum()), "distinct weights", len(np.unique(hdw)), "apx", e.apx_info())
    seen = set()
    for k in range(min(16, int(out[0]))):
        a, b, mi, lo = int(out[1 + 4 * k]), int(out[2 + 4 * k]), out[3 + 4 * k], out[4 + 4 * k]
        if (a, b) in seen: continue
        seen.add((a, b))
        c5, fx, F = e.joint_tables([a], [b])
        W = fx[0].astype(np.float64) / 2.0 ** F
        print(f"pair from {a} to {b}: MI {mi!r} level {lo!r}; r {r[a]} {r[b]}; counts\n{c5[0][np.ix_(cnt[:, a] > 0, cnt[:, b] > 0)]}\nweighted\n{W[np.ix_(cnt[:, a] > 0, cnt[:, b] > 0)]}")
        print("   marginals from", W.sum(axis=1)[cnt[:, a] > 0], "to", W.sum(axis=0)[cnt[:, b] > 0], "states", np.nonzero(cnt[:, a] > 0)[0], np.nonzero(cnt[:, b] > 0)[0])
