import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ldw_oracle as orc
from ldweaver_amd.engine import Engine
from ldweaver_amd import _lib as L
s = np.load(os.path.join(ROOT, "tests/golden/synth_c2slice.npz")); d = {k: s[k] for k in s.files}; d["g"] = float(d["g"])
eng = Engine(0)
eng.set_alignment(d["states"]); eng.set_weights(d["hdw"], 0); eng.set_snp_meta(d["r"], d["uqe"], d["POS"], d["paint"], d["g"])
approx = orc.lr_links_approx(d["POS"], d["g"], 20000.0)
blocks = np.array(orc.make_blocks(512, 150), dtype=np.int32)
out = {}
for fused in (False, True):
    eng.set_fused(fused)
    for rep in range(2):
        eng.mi_all_pairs(blocks, 20000.0, 3000.0, approx)
        st = eng.block_stats()
        print("fused", fused, "rep", rep, eng.counters())
        print(" n_sr", st["n_sr"].tolist()); print(" lr_tot", st["n_lr_total"].tolist()); print(" lr_kept", st["n_lr_kept"].tolist())
        print(" thr", repr(st["disc_thresh"][0]), "lowest kept of block 0:", np.sort(eng.links(1)[2][:st["n_lr_kept"][0]])[:18].tolist())
    out[fused] = (eng.links(0), eng.links(1))
for which in (0, 1):
    (a0, b0, m0), (a1, b1, m1) = out[False][which], out[True][which]
    print("which", which, len(m0), len(m1))
    k0 = set(zip(a0.tolist(), b0.tolist())); k1 = set(zip(a1.tolist(), b1.tolist()))
    print(" only unfused", sorted(k0 - k1)[:20]); print(" only fused", sorted(k1 - k0)[:20])
    if len(m0) == len(m1):
        print(" order equal", np.array_equal(a0, a1) and np.array_equal(b0, b1), "max dMI", np.abs(m0 - m1).max())
    else:
        d0 = dict(zip(zip(a0.tolist(), b0.tolist()), m0.tolist())); d1 = dict(zip(zip(a1.tolist(), b1.tolist()), m1.tolist()))
        common = set(d0) & set(d1)
        print(" common", len(common), "max dMI", max(abs(d0[k] - d1[k]) for k in common))
        import collections
        c1 = collections.Counter(zip(a1.tolist(), b1.tolist())); print(" dups fused", [k for k, v in c1.items() if v > 1][:10])
eng.close()
import c_oracle
fi = np.arange(0, 150)
Mb = c_oracle.mi_block(d["states"], d["hdw"], d["r"], d["uqe"], fi, fi)
bl = orc.block_links(Mb, fi, fi, d["POS"], d["paint"], d["g"], 20000.0, 3000.0, approx)
ko = set(zip(bl.lr["a"].tolist(), bl.lr["b"].tolist()))
for fused in (False, True):
    a, b, m = out[fused][1]
    n = 261 if not fused else 275
    kk = set(zip(a[:n].tolist(), b[:n].tolist()))
    print("fused", fused, "oracle n", len(ko), "only oracle", sorted(ko - kk)[:20], "only gpu", sorted(kk - ko)[:20])
print("thr oracle", bl.disc_thresh)
for bb in (6, 7, 17):
    print("MI(65,%d) oracle" % bb, Mb[65, bb], "sr?", orc.circ_len(d["POS"][bb], d["POS"][65], d["g"]) if hasattr(orc, "circ_len") else None)
eng2 = Engine(0)
eng2.set_alignment(d["states"]); eng2.set_weights(d["hdw"], 0); eng2.set_snp_meta(d["r"], d["uqe"], d["POS"], d["paint"], d["g"])
M = eng2.mi_block(fi, fi)
print("dense gpu MI(65,6)", M[65, 6], M[6, 65], "max diff vs oracle", np.abs(M - Mb).max())
cnt = eng2.state_counts()
print("cnt65", cnt[:, 65], "uqe65", d["uqe"][65], "r", d["r"][65])
eng2.close()
