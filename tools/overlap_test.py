#!/usr/bin/env python3
"""Do two independent block pipelines (GEMM -> epilogue), each on its own stream, overlap on one GPU?  Two engines run
the same 10k x 10k off-diagonal block K times from two host threads; compare with one engine doing 2K blocks."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldweaver_amd.engine import Engine
from ldweaver_amd.synth import synth_alignment

L, N, B, K = 20000, 5000, 10000, int(os.environ.get("OV_K", 12))
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
engs = []
for _ in range(2):
    e = Engine(0)
    e.set_alignment(syn["states"])
    cnt = e.state_counts(); uqe = (cnt > 0).T.astype(np.float64)
    hdw = e.hamming_weights(int(L * 0.1))
    e.set_weights(hdw); e.set_snp_meta(uqe.sum(1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
    engs.append(e)
outs = [torch.empty(B * B, dtype=torch.float64, device="cuda") for _ in range(2)]
fi, ti = np.arange(0, B), np.arange(B, 2 * B)

def run(e, out, k):
    for _ in range(k):
        e.mi_block(fi, ti, out=out)

for e, o in zip(engs, outs):
    run(e, o, 2)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(engs[0], outs[0], 2 * K); torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(engs[i], outs[i], K)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize(); t_par = time.perf_counter() - t0
print(f"sequential {t_seq / (2 * K) * 1e3:.3f} ms/block   two streams {t_par / (2 * K) * 1e3:.3f} ms/block   ratio {t_par / t_seq:.3f}  last {engs[0].last_timing()}")
