#!/usr/bin/env python3
"""What streaming lr_links.tsv costs a pass (r05): fresh engine per mode, first (allocating) pass and two more cold passes, with and without
ldw_lr_stream_begin around ldw_mi_all_pairs.  usage: python tools/lr_stream_probe.py [--L 100000 --N 5000]"""
import argparse
import os
import sys
import tempfile
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
a = ap.parse_args()
syn = synth_alignment(a.L, a.N, seed=1988, device="cuda", as_numpy=False)
POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
blocks = make_blocks(a.L, 10000)
approx = lr_links_approx(POS, g, 20000.0)
tmp = tempfile.mkdtemp()
hdw = None
for mode in ("plain", "stream", "plain", "stream"):
    with Engine(0) as e:
        e.set_alignment(syn["states"])
        cnt = e.state_counts()
        uqe = (cnt > 0).T.astype(np.float64)
        r = uqe.sum(axis=1)
        if hdw is None:
            hdw = e.hamming_weights(int(a.L * 0.1))
        e.set_weights(hdw)
        e.set_snp_meta(r, uqe, POS, paint, g)
        ts = []
        for k in range(3):
            f = os.path.join(tmp, f"{mode}{k}.tsv")
            e.reset_speculation()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "stream":
                e.lr_stream_begin(f, append=False)
            e.mi_all_pairs(blocks, 20000.0, 1e6, approx)
            t1 = time.perf_counter()
            if mode == "stream":
                e.lr_stream_end()
            else:
                e.write_links_tsv(1, f, append=False)
            t2 = time.perf_counter()
            ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1)))
        print(mode, "pass ms / tsv-after ms:", ts, flush=True)
