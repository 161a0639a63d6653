#!/usr/bin/env python3
"""cProfile of the product entry point perform_MI_computation on the bench workload: where the host time goes.
Default: the second call on a warm engine; --cold: the FIRST call of a fresh engine (what bench.py's job leg times)."""
import cProfile
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd.cpushare import limit_thread_pools  # noqa: E402

limit_thread_pools()   # (before numpy / torch: keep their pools inside the cgroup's CPU share)
import numpy as np

from ldweaver_amd import mi as MIH
from ldweaver_amd.engine import Engine
from ldweaver_amd.snpdat import CdsVar, SnpDat
from ldweaver_amd.synth import synth_alignment

L, N = 100000, 5000
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
st_host = syn["states"].cpu().numpy()
tmp = tempfile.mkdtemp(prefix="ldw_prof_")
with Engine(0) as e:
    e.set_alignment(st_host)
    sd = SnpDat.from_states(st_host, syn["POS"], float(syn["g"]), counts=e.state_counts())
    hdw = MIH.estimate_Hamming_distance_weights(sd, threshold=0.1, engine=e, alignment_resident=True, verbose=False)
    kw = dict(lr_save_path=os.path.join(tmp, "lr.tsv"), sr_save_path=os.path.join(tmp, "sr.tsv"), plt_folder=os.path.join(tmp, "P"), engine=e,
              alignment_resident=True, verbose=False, return_aux=True)
    if "--cold" not in sys.argv:
        MIH.perform_MI_computation(sd, hdw, CdsVar(paint=syn["paint"], nclust=3), **kw)
    pr = cProfile.Profile()
    pr.enable()
    red, aux = MIH.perform_MI_computation(sd, hdw, CdsVar(paint=syn["paint"], nclust=3), **kw)
    pr.disable()
    print(aux["stages_s"])
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
