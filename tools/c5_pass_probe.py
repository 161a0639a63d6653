import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ldweaver_amd.engine import Engine
from ldweaver_amd.mi import lr_links_approx, make_blocks
from ldweaver_amd.synth import synth_alignment
L, N = 500_000, 10_000
syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
with Engine(0) as e:
    e.set_alignment(syn["states"]); cnt = e.state_counts(); uqe = (cnt > 0).T.astype(np.float64); r = uqe.sum(axis=1)
    hdw = e.hamming_weights(int(L * 0.1)); e.set_weights(hdw); e.set_snp_meta(r, uqe, syn["POS"], syn["paint"], float(syn["g"]))
    approx = lr_links_approx(syn["POS"], float(syn["g"]), 20000.0); blocks = make_blocks(L, 10000)
    for k in range(3):
        e.reset_speculation(); c0 = e.counters(); s0 = e.span_report()
        torch.cuda.synchronize(); t0 = time.perf_counter(); e.mi_all_pairs(blocks, 20000.0, 1e6, approx); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c1 = e.counters(); s1 = e.span_report()
        print(f"C5 cold pass {k}: {dt*1e3:.0f} ms, misses {c1['spec_misses'] - c0['spec_misses']}, spans {s1['spans'] - s0['spans']} redone {s1['redone'] - s0['redone']}, pairs listed {c1['apx_pairs_listed'] - c0['apx_pairs_listed']}, links {e.links_count(0)} {e.links_count(1)}", flush=True)
