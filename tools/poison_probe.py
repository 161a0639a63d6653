import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import ldw_oracle as orc
for N in (40, 100, 128, 130, 257):
    syn = synth_alignment(1500, N, seed=5)
    st, POS, paint, g = syn["states"], syn["POS"], syn["paint"], float(syn["g"])
    with Engine(0) as e:
        e.set_alignment(st)
        cnt = e.state_counts(); uqe = (cnt > 0).T.astype(np.float64); r = uqe.sum(axis=1)
        hdw = e.hamming_weights(150)
        ok_h = np.array_equal(hdw, orc.hamming_weights(st, 0.1))
        e.set_weights(hdw); e.set_snp_meta(r, uqe, POS, paint, g)
        idx = np.arange(300)
        MI = e.mi_block(idx, idx)
        ref = orc.mi_block_faithful(st, hdw, r, uqe, idx, idx)
        err = np.nanmax(np.abs(MI - ref)); nn = int(np.isnan(MI).sum())
        cj, _, _ = e.joint_tables([0, 5, 299], [1, 200, 17])
        okj = all(np.array_equal(cj[k], orc.joint_counts(st, a, b)) for k, (a, b) in enumerate(((0, 1), (5, 200), (299, 17))))
        print(f"N {N}: hamming ok {ok_h}, mi_block max err {err:.2e} NaNs {nn}, joint counts ok {okj}", flush=True)
