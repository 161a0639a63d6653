#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the BUILD container only (reads the reference's bundled DATA files
/root/reference/inst/extdata/snp_sample.fa.gz and snp_sample.pos; nothing under
/root/reference exists on the GPU box).  The fixtures are data: the 5-state
matrix of the reference's own sample alignment and oracle outputs on it and on
a seeded synthetic case.  No reference source text is stored.

    python tests/golden/make_golden.py

Outputs
  snp_sample_states.npz   states uint8 (1268, 400), POS int32[1268], seq case-folded per
                          src/getACGTNsites.cpp:229-265 (A/a..T/t -> 0..3, else 4)
  snp_sample_oracle.npz   hdw, r, uqe, neff, strided sub-matrix + row/col sums of the single-block MI,
                          the same for the forced multi-block run (max_blk_sz = 1000: blocks 1000 + 268),
                          lr_links_approx, link-table digests (counts, head/tail rows, checksums) of both runs
  synth_c2slice.npz       seeded synthetic 512 SNPs x 1000 seqs (recipe of SURVEY.md §8d) + oracle MI
  kat_small.npz           known-answer vectors for fast_hadamard / acgtn2num / aracne helpers
"""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import ldw_oracle as orc  # noqa: E402

REF = "/root/reference/inst/extdata"


def read_fasta_gz(path):
    names, seqs, cur = [], [], []
    with gzip.open(path, "rb") as fh:
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                    cur = []
                names.append(line[1:].split()[0].decode())
            elif line:
                cur.append(line)
    if cur:
        seqs.append(b"".join(cur))
    return names, seqs


def paint_windows(POS, seed=1988, width=10_000, probs=(0.6, 0.3, 0.1)):
    """Synthetic cds_var$paint: cluster ids 1..3 over contiguous 10-kb windows."""
    rng = np.random.default_rng(seed)
    nwin = int(POS.max() // width) + 1
    win = rng.choice(len(probs), size=nwin, p=probs) + 1
    return win[(POS // width).astype(np.int64)].astype(np.int32)


def main():
    names, seqs = read_fasta_gz(os.path.join(REF, "snp_sample.fa.gz"))
    states = orc.encode_states(seqs)
    POS = np.loadtxt(os.path.join(REF, "snp_sample.pos"), dtype=np.int64).astype(np.int32)
    assert states.shape == (1268, 400) and POS.shape == (1268,)
    np.savez_compressed(os.path.join(HERE, "snp_sample_states.npz"), states=states, POS=POS)

    g = 50000  # full-alignment path sets g = alignment length (R/extractSNPs.R:140); sample is 1-50000
    uqe, r = orc.uqe_r(states)
    hdw = orc.hamming_weights(states, 0.1)
    paint = paint_windows(POS, width=1000)
    L = states.shape[0]
    allidx = np.arange(L)
    MI_single = orc.mi_block_faithful(states, hdw, r, uqe, allidx, allidx)
    sub_r = np.arange(0, L, 7)
    sub_c = np.arange(3, L, 11)
    shared = orc.shared_counts(states)
    out = dict(hdw=hdw, r=r, uqe=uqe, neff=hdw.sum(), paint=paint, g=g, sub_r=sub_r, sub_c=sub_c,
               MI_single_sub=MI_single[np.ix_(sub_r, sub_c)], MI_single_colsum=MI_single.sum(axis=0),
               MI_single_rowsum=MI_single.sum(axis=1), shared_sub=shared[::9, ::13].astype(np.int32),
               shared_colsum=shared.sum(axis=0))
    # forced multi-block: blocks of 1000 and 268 -> exercises Q1 on a non-square block and Q3
    blocks = orc.make_blocks(L, 1000)
    out["blocks"] = np.array(blocks, dtype=np.int32)
    for bi, (fs, fe, ts, te) in enumerate(blocks):
        Mb = orc.mi_block_faithful(states, hdw, r, uqe, np.arange(fs - 1, fe), np.arange(ts - 1, te))
        out[f"MI_blk{bi}_sub"] = Mb[::7, ::5]
        out[f"MI_blk{bi}_colsum"] = Mb.sum(axis=0)
        out[f"MI_blk{bi}_rowsum"] = Mb.sum(axis=1)
    for tag, mb in (("single", 10000), ("multi", 1000)):
        res = orc.perform_mi_computation(states, POS, g, r, uqe, hdw, paint, 3, sr_dist=20000,
                                         lr_retain_links=1e5 if tag == "multi" else 1e6,
                                         max_blk_sz=mb, do_srp=False)
        out[f"{tag}_lr_approx"] = res.lr_links_approx
        # link tables are large: keep row counts, head/tail rows and column checksums
        def digest(prefix, d):
            n = len(d["MI"])
            out[f"{prefix}_n"] = n
            for k, v in d.items():
                v = np.asarray(v, dtype=np.float64)
                out[f"{prefix}_{k}_head"] = v[:200]
                out[f"{prefix}_{k}_tail"] = v[-200:]
                out[f"{prefix}_{k}_sum"] = v.sum()
                out[f"{prefix}_{k}_wsum"] = (v * (np.arange(n) % 1009 + 1)).sum()
        digest(f"{tag}_lr", res.lr_rows)
        for ci, d in enumerate(res.sr_links_by_clust):
            digest(f"{tag}_sr{ci + 1}", d)
    np.savez_compressed(os.path.join(HERE, "snp_sample_oracle.npz"), **out)

    # seeded synthetic C2-shaped slice
    from ldweaver_amd.synth import synth_alignment
    syn = synth_alignment(L=512, N=1000, seed=1988)
    s_states, s_POS, s_paint, s_g = syn["states"], syn["POS"], syn["paint"], syn["g"]
    s_uqe, s_r = orc.uqe_r(s_states)
    s_hdw = orc.hamming_weights(s_states, 0.1)
    sidx = np.arange(s_states.shape[0])
    s_MI = orc.mi_block_faithful(s_states, s_hdw, s_r, s_uqe, sidx, sidx)
    jc = np.stack([orc.joint_counts(s_states, a, b) for a, b in ((0, 1), (5, 300), (511, 17), (100, 100))])
    np.savez_compressed(os.path.join(HERE, "synth_c2slice.npz"), states=s_states, POS=s_POS, paint=s_paint,
                        g=s_g, hdw=s_hdw, r=s_r, uqe=s_uqe, MI_sub=s_MI[::3, ::5], MI_colsum=s_MI.sum(axis=0),
                        MI_rowsum=s_MI.sum(axis=1), joint_pairs=np.array([[0, 1], [5, 300], [511, 17], [100, 100]]),
                        joint_counts=jc)

    # small known-answer vectors for the element-wise / helper twins
    rng = np.random.default_rng(1988)
    n = 257
    ops = {k: rng.uniform(0.5, 3.0, n) for k in ("den", "pxy", "pxpy", "RXY", "pXrX", "pYrY")}
    ops["uq"] = rng.integers(0, 2, n).astype(np.float64)
    MI0 = rng.uniform(0, 1, n)
    MI1 = MI0.copy()
    orc.fast_hadamard(MI1, ops["den"], ops["uq"], ops["pxy"], ops["pxpy"], ops["RXY"], ops["pXrX"], ops["pYrY"])
    ref_chars = list("ACGTN-acgtnRYKM.*AACCGGTT")
    nv = np.ones((5, len(ref_chars)), order="F")
    orc.acgtn2num(nv, ref_chars)
    np.savez_compressed(os.path.join(HERE, "kat_small.npz"), MI0=MI0, MI1=MI1, nv=nv,
                        ref_chars=np.array(ref_chars), **{f"op_{k}": v for k, v in ops.items()})
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
