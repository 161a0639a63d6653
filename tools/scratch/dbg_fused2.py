import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ldw_oracle as orc
from ldweaver_amd.engine import Engine
s = np.load(os.path.join(ROOT, "tests/golden/synth_c2slice.npz")); d = {k: s[k] for k in s.files}; d["g"] = float(d["g"])
eng = Engine(0)
eng.set_alignment(d["states"]); eng.set_weights(d["hdw"], 0); eng.set_snp_meta(d["r"], d["uqe"], d["POS"], d["paint"], d["g"])
approx = orc.lr_links_approx(d["POS"], d["g"], 20000.0)
blocks = np.array(orc.make_blocks(512, 150), dtype=np.int32)
out = {}
for fused in (False, True):
    eng.set_fused(fused)
    for rep in range(2):
        eng.mi_all_pairs(blocks, 20000.0, 1e12, approx)
    print(fused, eng.counters())
    out[fused] = eng.links(1)
(a0, b0, m0), (a1, b1, m1) = out[False], out[True]
print(len(m0), len(m1), np.array_equal(a0, a1), np.array_equal(b0, b1))
bad = np.nonzero(np.abs(m0 - m1) > 1e-12)[0]
print("bad", len(bad))
cnt = eng.state_counts()
nrows = (cnt > 0).sum(axis=0) - 1
import collections
print("bad a", collections.Counter(a0[bad].tolist()).most_common(10))
print("bad b", collections.Counter(b0[bad].tolist()).most_common(10))
for k in bad[:10]:
    print(a0[k], b0[k], m0[k], m1[k], "rows", nrows[a0[k]], nrows[b0[k]], "uqe", d["uqe"][a0[k]], d["uqe"][b0[k]], "cnt", cnt[:, a0[k]], cnt[:, b0[k]])
eng.close()
