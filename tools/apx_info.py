import sys, numpy as np, torch
sys.path.insert(0, '.')
from ldweaver_amd.engine import Engine
from ldweaver_amd.synth import synth_alignment
L, N = 100000, 5000
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
eng = Engine(0); eng.set_alignment(syn["states"])
hdw = eng.hamming_weights(int(L*0.1))
eng.set_weights(hdw)
print(eng.apx_info())
u, c = np.unique(hdw, return_counts=True)
print("distinct", len(u), "sizes sorted", sorted(c.tolist(), reverse=True)[:60], "min w", u.min(), "max w", u.max(), "neff", hdw.sum())
