import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import ldw_oracle as orc
from ldweaver_amd.engine import Engine
from ldweaver_amd import _lib as L
s = np.load("tests/golden/synth_c2slice.npz")
d = {k: s[k] for k in s.files}
POS = d["POS"][np.random.default_rng(77).permutation(512)]
eng = Engine(0)
eng.set_alignment(d["states"]); eng.set_weights(d["hdw"]); eng.set_snp_meta(d["r"], d["uqe"], POS, d["paint"], float(d["g"]))
approx = orc.lr_links_approx(POS, float(d["g"]), 60000.0)
blocks = np.array(orc.make_blocks(512, 150), dtype=np.int32)
eng.mi_all_pairs(blocks, 60000.0, 4000.0, approx)
print("ok", eng.links_count(0), eng.links_count(1), eng.path_report())
