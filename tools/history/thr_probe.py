#!/usr/bin/env python3
"""Why a block's device threshold can differ from quantile7 on the dense device MI by one ulp: print both, the two order statistics,
h, and the candidate roundings of (1 - h) q + h xh."""
import os, sys, math
from fractions import Fraction
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import ldw_oracle as orc
from ldweaver_amd.engine import Engine
from ldweaver_amd import mi as MIH
from ldweaver_amd.synth import synth_alignment

Ls, N = 100_000, 5_000
syn = synth_alignment(Ls, N, seed=1988, device="cuda", as_numpy=False)
eng = Engine(0)
eng.set_alignment(syn["states"])
cnt = eng.state_counts()
uqe = (cnt > 0).T.astype(np.float64)
r = uqe.sum(axis=1)
hdw = eng.hamming_weights(int(Ls * 0.1))
eng.set_weights(hdw)
POS, g = syn["POS"], float(syn["g"])
eng.set_snp_meta(r, uqe, POS, syn["paint"], g)
approx = MIH.lr_links_approx(POS, g, 20000.0)
blocks = MIH.make_blocks(Ls, 10000)
sub = blocks[[30]]
fs, fe, ts, te = sub[0].tolist()
fi, ti = np.arange(fs - 1, fe), np.arange(ts - 1, te)
Md = eng.mi_block(fi, ti)
rr, cc = orc.block_pair_index(len(fi), len(ti), False)
P = np.asarray(POS, dtype=np.float64)
lrm = orc.circ_len(P[ti][cc], P[fi][rr], g) > 20000.0
vals = Md[rr[lrm], cc[lrm]]
n = len(vals)
prob = max(0.0, 1 - ((1e6 * (n / approx)) / n))
index = 1.0 + (n - 1) * prob
lo, hi = int(math.floor(index)), int(math.ceil(index))
part = np.partition(vals, (lo - 1, hi - 1))
q, xh = float(part[lo - 1]), float(part[hi - 1])
h = index - lo
py = (1.0 - h) * q + h * xh
exact = Fraction(1.0 - h) * Fraction(q) + Fraction(h) * Fraction(xh)
print("n", n, "prob", repr(prob), "index", repr(index), "lo", lo, "hi", hi, "h", repr(h))
print("q", repr(q), "xh", repr(xh))
print("python (no fma)", repr(py), " exact->double", repr(float(exact)))
for mode in ("cold", "warm"):
    if mode == "cold":
        eng.reset_speculation()
    eng.mi_all_pairs(sub, 20000.0, 1e6, approx)
    st = eng.block_stats()
    la, lb, lmi = eng.links(1)
    keep = vals >= py
    wa, wb, wmi = fi[rr[lrm][keep]], ti[cc[lrm][keep]], vals[keep]
    print(mode, "device thr", repr(float(st["disc_thresh"][0])), "n_lr_total", int(st["n_lr_total"][0]), "kept dev", len(lmi), "kept py", int(keep.sum()),
          "rows equal", np.array_equal(la, wa) and np.array_equal(lb, wb), "MI bits equal", np.array_equal(lmi, wmi) if len(lmi) == len(wmi) else None)
    srt = np.sort(lmi)
    print("   two smallest kept dev", repr(float(srt[0])), repr(float(srt[1])), " xh in kept:", bool((lmi == xh).any()), " q in dense:", bool((vals == q).any()))
