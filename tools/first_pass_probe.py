#!/usr/bin/env python3
"""Where does the first pass of an engine spend its extra time?  Times, on the C4 workload: ensure_rows (forced through a one-pair
ldw_joint_tables call), then cold passes 1, 2, 3 of the same engine (LDW_HOST_TIMING=1 prints the host phases of each)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment

L, N = 100_000, 5_000
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
t = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
with Engine(0) as e:
    t0 = t(); e.set_alignment(syn["states"]); e.sync(); t1 = t()
    counts = e.state_counts(); uqe = (counts > 0).T.astype(np.float64); r = uqe.sum(axis=1)
    t2 = t(); hdw = e.hamming_weights(int(L * 0.1)); t3 = t()
    e.set_weights(hdw); t4 = t()
    e.set_snp_meta(r, uqe, POS, paint, g); t5 = t()
    e.joint_tables([0], [1]); t6 = t()
    approx = lr_links_approx(POS, g, 20000.0)
    blocks = make_blocks(L, 10000)
    print(f"set_alignment {1e3*(t1-t0):.1f} ms  hamming {1e3*(t3-t2):.1f}  set_weights {1e3*(t4-t3):.1f}  set_snp_meta {1e3*(t5-t4):.1f}  ensure_rows (+ a joint table) {1e3*(t6-t5):.1f}")
    for k in range(4):
        e.reset_speculation()
        a = t(); e.mi_all_pairs(blocks, 20000.0, 1e6, approx); b = t()
        print(f"cold pass {k}: {1e3*(b-a):.1f} ms", flush=True)
