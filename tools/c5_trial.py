#!/usr/bin/env python3
"""One pass of the hot path at BASELINE config 5's shape (500k SNPs x 10k seqs) on ONE GPU: robustness / timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment
L, N = int(os.environ.get("C5_L", 500000)), int(os.environ.get("C5_N", 10000))
t0 = time.time()
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
eng = Engine(0)
eng.set_alignment(syn["states"])
del syn["states"]; torch.cuda.empty_cache()
cnt = eng.state_counts(); uqe = (cnt > 0).T.astype(np.float64)
print("setup s", time.time() - t0, flush=True)
t0 = time.time(); hdw = eng.hamming_weights(int(L * 0.1)); print("hamming s", time.time() - t0, "neff", hdw.sum(), "distinct", len(np.unique(hdw)), flush=True)
eng.set_weights(hdw); eng.set_snp_meta(uqe.sum(1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
approx = lr_links_approx(syn["POS"], float(syn["g"]), 20000.0)
blocks = make_blocks(L, 10000)
t0 = time.time(); eng.mi_all_pairs(blocks, 20000.0, 1e6, approx); dt = time.time() - t0
st = eng.block_stats()
pairs = L * (L - 1) // 2
print(f"blocks {len(blocks)} step {dt:.2f} s  {pairs / dt:.3e} pairs/s  sr rows {eng.links_count(0)}  lr rows {eng.links_count(1)}  timing {eng.last_timing()}")
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
print("thresh", np.round(st["disc_thresh"], 6).tolist()); print("kept", st["n_lr_kept"].tolist()); print("nlr", st["n_lr_total"].tolist()); print(eng.counters())
