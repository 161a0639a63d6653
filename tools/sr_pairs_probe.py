#!/usr/bin/env python3
"""Time of ldw_sr_pairs_fill (Engine.sr_pairs: the index columns of the assembled short-range table from positions alone) on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment
L_, N = 100_000, 5_000
syn = synth_alignment(L_, N, seed=1988, device="cuda", as_numpy=False)
POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
blocks = make_blocks(L_, 10000)
with Engine(0) as e:
    e.set_alignment(syn["states"])
    cnt = e.state_counts(); uqe = (cnt > 0).T.astype(np.float64); r = uqe.sum(axis=1)
    e.set_weights(e.hamming_weights(int(L_ * 0.1))); e.set_snp_meta(r, uqe, POS, paint, g)
    e.mi_all_pairs(blocks, 20000.0, 1e6, lr_links_approx(POS, g, 20000.0))
    a0, b0, _ = e.links(0)
    for host in (False, True):   # r05: intervals built on the device (default) / by the host loop (LDW_SR_PAIRS_HOST=1)
        if host:
            os.environ["LDW_SR_PAIRS_HOST"] = "1"
        for k in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pa, pb = e.sr_pairs(blocks, 20000.0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            print(f"sr_pairs ({'host' if host else 'device'} intervals): {dt:.2f} ms, rows {len(pa)}, equal to the pass's own: {bool(np.array_equal(pa.cpu().numpy(), a0) and np.array_equal(pb.cpu().numpy(), b0))}", flush=True)
    os.environ.pop("LDW_SR_PAIRS_HOST", None)
