#!/usr/bin/env python3
"""Per-block trace of a cold pass and of the warm pass after it on the bench workload (LDW_BLOCK_TRACE=1): which blocks cost
what, with which bucket guess, how many candidates.  usage: LDW_BLOCK_TRACE=1 python tools/cold_probe.py [L N] 2> trace.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment

L, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 5000)
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
eng = Engine(0)
eng.set_alignment(syn["states"])
cnt = eng.state_counts()
uqe = (cnt > 0).T.astype(np.float64)
hdw = eng.hamming_weights(int(L * 0.1))
eng.set_weights(hdw)
eng.set_snp_meta(uqe.sum(1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
approx = lr_links_approx(syn["POS"], float(syn["g"]), 20000.0)
blocks = make_blocks(L, 10000)
eng.mi_all_pairs(blocks, 20000.0, 1e6, approx)          # allocations
for tag, reset in (("cold", True), ("warm", False), ("cold", True), ("warm", False)):
    if reset:
        eng.reset_speculation()
    torch.cuda.synchronize()
    print(f"==== {tag} pass", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    eng.mi_all_pairs(blocks, 20000.0, 1e6, approx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"==== {tag} pass: {dt:.2f} ms, links {eng.links_count(0)} / {eng.links_count(1)}, {eng.path_report()}", file=sys.stderr, flush=True)
    print(tag, round(dt, 2))
