#!/usr/bin/env python3
"""Find the pairs the default path loses against the plain path on the adversarial alignment and print their joint tables."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd import mi as MIH
from ldweaver_amd.engine import Engine
from ldweaver_amd.synth import synth_alignment

Ls, N = 20000, 2000
syn = synth_alignment(Ls, N, seed=1988, device="cuda", as_numpy=False, kind="adversarial")
with Engine(0) as e:
    e.set_alignment(syn["states"])
    cnt = e.state_counts()
    uqe = (cnt > 0).T.astype(np.float64)
    r = uqe.sum(axis=1)
    hdw = e.hamming_weights(int(Ls * 0.1))
    e.set_weights(hdw)
    POS, g = syn["POS"], float(syn["g"])
    e.set_snp_meta(r, uqe, POS, syn["paint"], g)
    approx = MIH.lr_links_approx(POS, g, 20000.0)
    blocks = MIH.make_blocks(Ls, 5000)
    e.set_mixed(False); e.set_screen(0); e.set_path(1)
    e.mi_all_pairs(blocks, 20000.0, 2e5, approx)
    pa, pb, pm = e.links(1)
    stp = e.block_stats()
    e.set_mixed(True); e.set_screen(1); e.set_path(0)
    for cold in (True, False):
        if cold: e.reset_speculation()
        e.mi_all_pairs(blocks, 20000.0, 2e5, approx)
        da, db, dm = e.links(1)
        std = e.block_stats()
        P = set(zip(pa.tolist(), pb.tolist())); D = set(zip(da.tolist(), db.tolist()))
        print("cold" if cold else "warm", "plain rows", len(pa), "default rows", len(da), "missing", len(P - D), "extra", len(D - P))
        print(" kept per block plain", stp["n_lr_kept"], "default", std["n_lr_kept"])
        print(" thr plain", stp["disc_thresh"][:5], "default", std["disc_thresh"][:5])
        miss = sorted(P - D)[:5]
        for (a, b) in miss:
            k = [i for i in range(len(pa)) if pa[i] == a and pb[i] == b][0]
            cn, fx, _ = e.joint_tables([a], [b])
            print("  missing pair", a, b, "MI", pm[k], "r", r[a], r[b], "counts\n", cn[0])
