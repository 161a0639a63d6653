#!/usr/bin/env python3
"""Measurement build only (-DLDW_SCREEN_STATS): how many pairs / (wave, column) groups of the multi-cell screen would pass a cheaper first-level
bound (log form with separable denominators; chi-square form without logarithms) next to the bound that is used."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ldweaver_amd import _lib as L
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment

Ln, N = 100000, 5000
kind = sys.argv[1] if len(sys.argv) > 1 else "survey"
syn = synth_alignment(Ln, N, seed=1988, device="cuda", as_numpy=False, kind=kind)
with Engine(0) as e:
    e.set_alignment(syn["states"])
    cnt = e.state_counts(); uqe = (cnt > 0).T.astype(np.float64)
    e.set_weights(e.hamming_weights(int(Ln * 0.1)))
    e.set_snp_meta(uqe.sum(1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
    e.mi_all_pairs(make_blocks(Ln, 10000), 20000.0, 1e6, lr_links_approx(syn["POS"], float(syn["g"]), 20000.0))
    e.sync()
    out = (C.c_ulonglong * 16)()
    lib = C.CDLL(L.LIB_PATH)
    assert lib.ldw_debug_screen_stats(out) == 0
    v = list(out)
    print(kind, dict(pairs=v[0], full=v[1], log_pre=v[2], chi_pre=v[3], columns=v[4], col_full=v[5], col_log=v[6], col_chi=v[7], viol_log=v[8], viol_chi=v[9]))
    print("per pair: full %.3e  log %.3e  chi %.3e ; per column: full %.3f log %.3f chi %.3f" % (v[1] / v[0], v[2] / v[0], v[3] / v[0], v[5] / v[4], v[6] / v[4], v[7] / v[4]))
