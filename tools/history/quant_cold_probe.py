#!/usr/bin/env python3
"""Where does the first ldw_sr_len_quantiles of a context spend its extra ~20 ms?  Cold context -> MI pass -> [optionally a quantile call on a
small imported table first] -> the quantile call on the real table, each timed (LDW_HOST_TIMING=1 prints the library's own phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment

L, N = 100000, 5000
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
with Engine(0) as e:
    e.set_alignment(syn["states"])
    cnt = e.state_counts()
    uqe = (cnt > 0).T.astype(np.float64)
    e.set_weights(e.hamming_weights(int(L * 0.1)))
    e.set_snp_meta(uqe.sum(1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
    e.mi_all_pairs(make_blocks(L, 10000), 20000.0, 1e6, lr_links_approx(syn["POS"], float(syn["g"]), 20000.0))
    e.sync()
    if "--small-first" in sys.argv:
        a, b, mi = e.links(0)
        keep = slice(0, 3_000_000)
        sr = (a.copy(), b.copy(), mi.copy())
        e.links_import(0, a[keep], b[keep], mi[keep])
        t = time.perf_counter(); e.sr_len_quantiles(3, 20000.0, 0.95); print("small table first: %.2f ms" % ((time.perf_counter() - t) * 1e3))
        e.links_import(0, *sr)
    idle = float(sys.argv[sys.argv.index("--idle-ms") + 1]) * 1e-3 if "--idle-ms" in sys.argv else 0.0
    for k in range(3):
        time.sleep(idle)          # (a GPU left idle, as during the host's formatting of lr_links.tsv)
        t = time.perf_counter(); e.sr_len_quantiles(3, 20000.0, 0.95); print("call %d: %.2f ms" % (k, (time.perf_counter() - t) * 1e3))
