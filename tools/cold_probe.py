import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment
L,N=100000,5000
syn=synth_alignment(L,N,seed=1988,device="cuda",as_numpy=False)
eng=Engine(0); eng.set_alignment(syn["states"])
cnt=eng.state_counts(); uqe=(cnt>0).T.astype(np.float64); r=uqe.sum(1)
hdw=eng.hamming_weights(int(L*0.1)); eng.set_weights(hdw); eng.set_snp_meta(r,uqe,syn["POS"],syn["paint"],float(syn["g"]))
approx=lr_links_approx(syn["POS"],float(syn["g"]),20000.0); blocks=make_blocks(L,10000)
eng.mi_all_pairs(blocks,20000.0,1e6,approx); eng.sync()
for rep in range(2):
    eng.set_weights(hdw); c0=eng.counters(); t=time.perf_counter(); eng.mi_all_pairs(blocks,20000.0,1e6,approx); eng.sync(); dt=time.perf_counter()-t
    c1=eng.counters(); print("cold pass ms", round(dt*1e3,1), {k:c1[k]-c0[k] for k in c1})
    c0=eng.counters(); t=time.perf_counter(); eng.mi_all_pairs(blocks,20000.0,1e6,approx); eng.sync(); dt=time.perf_counter()-t
    c1=eng.counters(); print("warm pass ms", round(dt*1e3,1), {k:c1[k]-c0[k] for k in c1})
