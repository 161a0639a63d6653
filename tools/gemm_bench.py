#!/usr/bin/env python3
"""Time the co-occurrence GEMM + epilogue of one 10k x 10k off-diagonal block (C4 geometry). GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldweaver_amd.engine import Engine
from ldweaver_amd.synth import synth_alignment

L, N, B = int(os.environ.get("GB_L", 20000)), int(os.environ.get("GB_N", 5000)), int(os.environ.get("GB_B", 10000))
nl = int(os.environ.get("GB_LIMBS", 5))
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
eng = Engine(0)
eng.set_alignment(syn["states"])
cnt = eng.state_counts()
uqe = (cnt > 0).T.astype(np.float64)
hdw = eng.hamming_weights(int(L * 0.1))
eng.set_weights(hdw, nl)
eng.set_snp_meta(uqe.sum(1), uqe, syn["POS"], syn["paint"], float(syn["g"]))
out = torch.empty(B * B, dtype=torch.float64, device="cuda")
fi, ti = np.arange(0, B), np.arange(B, 2 * B)
res = []
for it in range(6):
    eng.mi_block(fi, ti, out=out)
    res.append(eng.last_timing())
rows = int(((cnt > 0).sum(0)[:B] - 1).sum()), int(((cnt > 0).sum(0)[B:2 * B] - 1).sum())
g = np.median([r["gemm_ms"] for r in res[1:]]); e = np.median([r["epilogue_ms"] for r in res[1:]])
Npad = (N + 127) // 128 * 128
ops = 2.0 * (-(-rows[0] // 128) * 128) * (-(-rows[1] // 128) * 128) * Npad * nl
print(f"rows {rows} gemm {g:.3f} ms = {ops / g / 1e9:.0f} TOP/s ({ops / g / 1e9 / 5000 * 100:.1f}% of 5 POPS)  epilogue {e:.3f} ms = {B * B / e / 1e6:.2f} Gpairs/s")
