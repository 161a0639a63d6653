#!/usr/bin/env python3
"""At which kept fraction lr_retain_links / lr_links_approx does the speculative path (approximate GEMM + screen + pair lists) stop paying
against the plain path (5-limb GEMM + fp64 MI of every pair)?  Warm passes, ms per pass, default vs plain."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment

L, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30000, 2000)
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
eng = Engine(0)
eng.set_alignment(syn["states"])
uqe = (eng.state_counts() > 0).T.astype(np.float64)
eng.set_weights(eng.hamming_weights(int(L * 0.1)))
eng.set_snp_meta(uqe.sum(1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
approx = lr_links_approx(syn["POS"], float(syn["g"]), 20000.0)
blocks = make_blocks(L, 10000)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.mi_all_pairs(blocks, 20000.0, keep, approx)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for frac in (2e-4, 2e-3, 5e-3, 1e-2, 2e-2, 5e-2, 1e-1):
    keep = frac * approx
    out = []
    for mode in ("default", "plain"):
        eng.set_mixed(mode == "default"); eng.set_screen(1 if mode == "default" else 0); eng.set_path(0 if mode == "default" else 1)
        run(2)
        out.append(run(4))
    print(f"kept fraction {frac:7.4f}: default {out[0]:7.2f} ms   plain {out[1]:7.2f} ms   links {eng.links_count(1)}")
