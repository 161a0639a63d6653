import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ldweaver_amd.engine import Engine
from ldweaver_amd.synth import synth_alignment
syn = synth_alignment(100000, 5000, seed=1988, device="cuda", as_numpy=False)
for rep in range(3):
    t0=time.perf_counter()
    e = Engine(0)
    t1=time.perf_counter()
    e.set_alignment(syn["states"]); torch.cuda.synchronize()
    t2=time.perf_counter()
    cnt = e.state_counts(); torch.cuda.synchronize()
    t3=time.perf_counter()
    hdw = e.hamming_weights(10000); torch.cuda.synchronize()
    t4=time.perf_counter()
    hdw = e.hamming_weights(10000); torch.cuda.synchronize()
    t5=time.perf_counter()
    print(f"rep {rep}: create {1e3*(t1-t0):.1f} ms, set_alignment {1e3*(t2-t1):.1f}, counts {1e3*(t3-t2):.1f}, hamming {1e3*(t4-t3):.1f}, again {1e3*(t5-t4):.1f}", flush=True)
    e.close()
