#!/usr/bin/env python3
"""Cold passes of the default path over the adversarial alignment (synth kind 'adversarial', 100k x 5k) — for rocprofv3 (tools/prof_cmd.sh):
    python tools/adv_pass.py [--weights hamming|distinct] [--passes 5] [--no-overlap]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=100_000)
ap.add_argument("--N", type=int, default=5_000)
ap.add_argument("--weights", default="hamming")
ap.add_argument("--passes", type=int, default=5)
ap.add_argument("--no-overlap", action="store_true")
a = ap.parse_args()
syn = synth_alignment(a.L, a.N, seed=1988, device="cuda", as_numpy=False, kind="adversarial")
POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
blocks = make_blocks(a.L, 10000)
approx = lr_links_approx(POS, g, 20000.0)
with Engine(0) as e:
    e.set_alignment(syn["states"])
    cnt = e.state_counts()
    uqe = (cnt > 0).T.astype(np.float64)
    r = uqe.sum(axis=1)
    if a.weights == "hamming":
        hdw = e.hamming_weights(int(a.L * 0.1))
    else:
        u = ((np.arange(a.N, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 32)).astype(np.float64) / float(1 << 32)
        hdw = 1.0 / (1.0 + 49.0 * u)
    e.set_weights(hdw)
    e.set_snp_meta(r, uqe, POS, paint, g)
    e.set_overlap(not a.no_overlap)
    for k in range(a.passes + 1):
        e.reset_speculation()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.mi_all_pairs(blocks, 20000.0, 1e6, approx)
        torch.cuda.synchronize()
        print(f"pass {k}: {(time.perf_counter() - t0) * 1e3:.1f} ms", e.last_timing(), flush=True)
    print(e.path_report(), e.prune_report(), e.overflow_report(), e.counters())
