"""Fixed cost of one ldw_mi_all_pairs call (what every gather phase of a multi-GPU run pays): time of calls over k blocks."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment
L, N = 100000, 5000
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
eng = Engine(0); eng.set_alignment(syn["states"])
cnt = eng.state_counts(); uqe = (cnt > 0).T.astype(np.float64); r = uqe.sum(1)
hdw = eng.hamming_weights(int(L * 0.1)); eng.set_weights(hdw); eng.set_snp_meta(r, uqe, syn["POS"], syn["paint"], float(syn["g"]))
approx = lr_links_approx(syn["POS"], float(syn["g"]), 20000.0); blocks = make_blocks(L, 10000)
eng.mi_all_pairs(blocks, 20000.0, 1e6, approx); eng.sync()
off = blocks[[2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16]]      # far off-diagonal block pairs
for k in (1, 2, 3, 4, 7, 12):
    ts = []
    for _ in range(5):
        t = time.perf_counter(); eng.mi_all_pairs(off[:k], 20000.0, 1e6, approx); eng.sync(); ts.append(time.perf_counter() - t)
    print(f"{k:2d} blocks: {min(ts) * 1e3:7.2f} ms  ({min(ts) * 1e3 / k:.2f} per block)")
