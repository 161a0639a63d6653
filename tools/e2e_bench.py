#!/usr/bin/env python3
"""Whole job on ONE GPU, stage by stage: synthetic alignment (resident) -> Hamming weights -> all-pairs MI with
link selection (a COLD pass: ldw_reset_speculation before every repetition, as a job runs it) -> lr_links.tsv by the native writer ->
short-range model (quantiles, excess statistics, p-values) -> ARACNE -> kept links on the host.
Prints one JSON line; `--out` also writes it to a file (profiles/).

    python tools/e2e_bench.py --L 100000 --N 5000           # BASELINE config 4 shape
    python tools/e2e_bench.py --L 500000 --N 10000          # config 5 shape (single GPU: ~10 s of MI)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldweaver_amd.cpushare import limit_thread_pools  # noqa: E402

limit_thread_pools()   # (before numpy / torch: keep their pools inside the cgroup's CPU share)
import torch  # noqa: E402

from ldweaver_amd import srp  # noqa: E402
from ldweaver_amd.engine import Engine  # noqa: E402
from ldweaver_amd.mi import lr_links_approx, make_blocks  # noqa: E402
from ldweaver_amd.synth import synth_alignment  # noqa: E402


class Stages:
    def __init__(self, eng):
        self.eng, self.t, self.out = eng, time.perf_counter(), {}

    def lap(self, name):
        self.eng.sync()
        now = time.perf_counter()
        self.out[name] = round((now - self.t) * 1e3, 3)
        self.t = now


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=100000)
    ap.add_argument("--N", type=int, default=5000)
    ap.add_argument("--sr-dist", type=float, default=20000.0)
    ap.add_argument("--srp-cutoff", type=float, default=3.0)
    ap.add_argument("--repeat", type=int, default=2, help="the last repetition is reported (first one warms allocations)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    L, N = a.L, a.N
    syn = synth_alignment(L, N, seed=1988, device="cuda", as_numpy=False)
    eng = Engine(0)
    eng.set_alignment(syn["states"])
    del syn["states"]
    torch.cuda.empty_cache()
    cnt = eng.state_counts()
    uqe = (cnt > 0).T.astype(np.float64)
    POS, paint, g = syn["POS"], syn["paint"], float(syn["g"])
    blocks = make_blocks(L, 10000)
    res = None
    for _ in range(a.repeat):
        st = Stages(eng)
        hdw = eng.hamming_weights(int(L * 0.1))
        st.lap("hamming_weights_ms")
        eng.set_weights(hdw)
        eng.set_snp_meta(uqe.sum(1), uqe, POS, paint, g)
        approx = lr_links_approx(POS, g, a.sr_dist)
        st.lap("setup_and_lr_approx_ms")
        eng.reset_speculation()
        eng.mi_all_pairs(blocks, a.sr_dist, 1e6, approx)
        st.lap("mi_all_pairs_ms")
        lr_rows, lr_bytes = eng.write_links_tsv(1, "/tmp/ldw_e2e_lr_links.tsv", append=False)
        st.lap("lr_tsv_write_ms")
        n_sr, n_lr = eng.links_count(0), eng.links_count(1)
        qlo, qhi, cn = eng.sr_len_quantiles(3, a.sr_dist, 0.95)
        st.lap("sr_len_quantiles_ms")
        t0 = time.perf_counter()
        S = qlo.shape[1]
        md = np.full((3, S), np.nan)
        lens = np.arange(1, S + 1, dtype=np.float64)
        for ci in range(3):
            has = cn[ci] > 0
            n = cn[ci][has].astype(np.float64)
            idx = 1 + (n - 1) * 0.95
            h = idx - np.floor(idx)
            lo, hi = qlo[ci][has], qhi[ci][has]
            mx = np.where((h > 0) & (hi != lo), (1 - h) * lo + h * hi, lo)
            fit = srp.fit_decay(lens[has], mx)
            md[ci, :len(fit)] = fit
        st.lap("host_decay_fit_ms")
        stats = eng.sr_excess_stats(md)
        st.lap("sr_excess_stats_ms")
        shape = np.empty((3, 3))
        import sys as _sys
        for ci in range(3):
            _t = time.perf_counter()
            sa, sb = srp.beta_mle_stats(*stats[ci])
            shape[ci] = sa, sb, srp._betaln(sa, sb)
            print(f"[e2e] beta MLE cluster {ci}: stats {[repr(float(x)) for x in stats[ci]]} -> ({sa!r}, {sb!r}) in {(time.perf_counter() - _t) * 1e3:.2f} ms, "
                  f"simplex fallback loaded: {'scipy.optimize' in _sys.modules}", file=_sys.stderr)
        st.lap("host_beta_mle_ms")
        n_red, n_pool, min_mi = eng.sr_pvalues(md, shape, a.srp_cutoff)
        st.lap("sr_pvalues_ms")
        flags = eng.aracne_device()
        st.lap("aracne_device_ms")
        red = eng.sr_reduced()
        la, lb, lmi = eng.links(1)
        st.lap("fetch_kept_links_ms")
        pairs = L * (L - 1) // 2
        res = dict(workload=f"synthetic {L} SNPs x {N} seqs, 1 GPU, seed 1988", pairs=pairs, blocks=len(blocks), n_sr=n_sr, n_lr=n_lr,
                   n_red=n_red, n_pool=n_pool, aracne_direct=int(flags.sum()), min_mi_kept=min_mi, stages_ms=st.out,
                   total_ms=round(sum(st.out.values()), 3), mi_pairs_per_s=pairs / (st.out["mi_all_pairs_ms"] * 1e-3),
                   sr_model_rows_per_s=n_sr / max(1e-9, (st.out["sr_len_quantiles_ms"] + st.out["sr_excess_stats_ms"] + st.out["sr_pvalues_ms"]) * 1e-3),
                   beta_shapes=shape[:, :2].round(6).tolist(), hbm_peak_GB=round(torch.cuda.max_memory_allocated() / 1e9, 2),
                   lr_tsv_rows=lr_rows, lr_tsv_bytes=lr_bytes, path=eng.path_report())
    line = json.dumps(res)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(line + "\n")


if __name__ == "__main__":
    main()
