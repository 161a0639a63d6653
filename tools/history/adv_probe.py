#!/usr/bin/env python3
"""Verify-mode probe on the adversarial alignment (tests/test_gpu_parity.py::test_adversarial_alignment_default_equals_plain): screen violations
under the diagnostic switches given in the environment.  usage: [LDW_NO_TAB11=1 ...] python tools/adv_probe.py [hamming|distinct] [L] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd import mi as MIH
from ldweaver_amd.engine import Engine
from ldweaver_amd.synth import synth_alignment

w = sys.argv[1] if len(sys.argv) > 1 else "hamming"
Ls = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
syn = synth_alignment(Ls, N, seed=1988, device="cuda", as_numpy=False, kind="adversarial")
with Engine(0) as e:
    e.set_alignment(syn["states"])
    cnt = e.state_counts()
    uqe = (cnt > 0).T.astype(np.float64)
    r = uqe.sum(axis=1)
    hdw = e.hamming_weights(int(Ls * 0.1))
    if w == "distinct":
        u = ((np.arange(N, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 32)).astype(np.float64) / float(1 << 32)
        hdw = 1.0 / (1.0 + 49.0 * u)
    e.set_weights(hdw)
    POS, g = syn["POS"], float(syn["g"])
    e.set_snp_meta(r, uqe, POS, syn["paint"], g)
    approx = MIH.lr_links_approx(POS, g, 20000.0)
    blocks = MIH.make_blocks(Ls, 5000)
    for quirk in (0, 1):
        e.set_screen(1)
        e.reset_speculation()
        e.mi_all_pairs(blocks, 20000.0, 2e5, approx, quirk=quirk)
        e.set_screen(2)
        c0 = e.counters()
        e.mi_all_pairs(blocks, 20000.0, 2e5, approx, quirk=quirk)
        c1 = e.counters()
        st = e.block_stats()
        print("quirk", quirk, "violations", c1["screen_violations"] - c0["screen_violations"], "apx", c1["apx_blocks"] - c0["apx_blocks"], "thr", st["disc_thresh"][:4], e.apx_info(),
              {k: v for k, v in os.environ.items() if k.startswith("LDW_")}, flush=True)
