"""Seeded synthetic alignments of the shape BASELINE.json quotes the metric on.

Recipe of SURVEY.md §8(d) (seed 1988 everywhere): per SNP a major allele uniform
in {A,C,G,T}, a minor allele != major, MAF ~ Beta(0.5, 2) clipped to [0.02, 0.5];
15 % of SNPs carry gaps (state 4) at a rate U(0.002, 0.10); 0.5 % carry a third
allele at 1-2 %.  Population structure (so that the Hamming weights are
non-trivial): ceil(N/25) clonal groups with Zipf sizes, LD blocks of 50 SNPs,
2-4 block haplotypes per LD block, one per group, then 1 % per-site noise.
POS: L distinct sorted positions in 1..g (g = 2,221,315); paint: 3 clusters over
contiguous 10-kb windows with probabilities (0.6, 0.3, 0.1).

Generated with torch so the big configurations can be made directly in HBM;
the stream differs between CPU and GPU generators, which is fine: committed
fixtures store the generated matrix itself.
"""
from __future__ import annotations

import math

import numpy as np
import torch

G_DEFAULT = 2_221_315


def zipf_group_sizes(N: int, G: int) -> np.ndarray:
    w = 1.0 / np.arange(1, G + 1)
    sizes = np.maximum(1, np.floor(N * w / w.sum()).astype(np.int64))
    # fix the total by adjusting the largest groups
    diff = N - int(sizes.sum())
    i = 0
    while diff != 0:
        step = 1 if diff > 0 else -1
        if sizes[i % G] + step >= 1:
            sizes[i % G] += step
            diff -= step
        i += 1
    return sizes


@torch.no_grad()
def synth_alignment(L: int, N: int, seed: int = 1988, g: int = G_DEFAULT, device="cpu",
                    chunk: int = 4096, as_numpy: bool | None = None, kind: str = "survey") -> dict:
    """Returns dict(states (L,N) uint8, POS int32[L], paint int32[L], g, nclust=3).

    ``device='cuda'`` keeps ``states`` on the GPU (torch tensor) unless ``as_numpy``.
    ``kind='adversarial'``: the same recipe with the two properties the default path profits from taken away —
    MAF uniform in [0.2, 0.5] (no rare minor states: nothing for the marginal-only tile pruning) and NO clonal groups (every
    sequence draws its own block haplotypes: no down-weighted clones, N_eff near N instead of near N / 25).
    """
    dev = torch.device(device)
    if as_numpy is None:
        as_numpy = dev.type == "cpu"
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    rs = np.random.default_rng(seed)

    # --- per-SNP parameters (host, O(L)) ---
    major = rs.integers(0, 4, L)
    minor = (major + rs.integers(1, 4, L)) % 4
    maf = np.clip(rs.beta(0.5, 2.0, L), 0.02, 0.5)
    if kind == "adversarial":
        maf = rs.uniform(0.2, 0.5, L)
    elif kind != "survey":
        raise ValueError(f"unknown synthetic alignment kind {kind!r}")
    has_gap = rs.random(L) < 0.15
    gap_rate = np.where(has_gap, rs.uniform(0.002, 0.10, L), 0.0)
    has_third = rs.random(L) < 0.005
    third_rate = np.where(has_third, rs.uniform(0.01, 0.02, L), 0.0)
    # third allele: a base that is neither major nor minor
    cand = (major + 1) % 4
    cand = np.where(cand == minor, (cand + 1) % 4, cand)
    third = cand

    # --- population structure ---
    G = N if kind == "adversarial" else max(1, math.ceil(N / 25))
    sizes = np.ones(N, dtype=np.int64) if kind == "adversarial" else zipf_group_sizes(N, G)
    group_of_seq = np.repeat(np.arange(G), sizes)
    rs.shuffle(group_of_seq)
    nb = math.ceil(L / 50)
    nhap = rs.integers(2, 5, nb)                                  # 2..4 haplotypes per LD block
    choice = (rs.integers(0, 1 << 30, (G, nb)) % nhap[None, :])  # group's haplotype in each block

    # positions and paint
    POS = np.sort(rs.choice(g, size=L, replace=False) + 1).astype(np.int32)
    nwin = g // 10_000 + 1
    win = rs.choice(3, size=nwin, p=(0.6, 0.3, 0.1)) + 1
    paint = win[(POS // 10_000)].astype(np.int32)

    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=dev)
    major_t, minor_t, third_t = t(major, torch.uint8), t(minor, torch.uint8), t(third, torch.uint8)
    maf_t, gap_t, thr_t = t(maf, torch.float32), t(gap_rate, torch.float32), t(third_rate, torch.float32)
    grp_t = t(group_of_seq, torch.int64)
    choice_t = t(choice, torch.int64)
    blk_of_snp = torch.arange(L, device=dev) // 50

    states = torch.empty((L, N), dtype=torch.uint8, device=dev)
    for lo in range(0, L, chunk):
        hi = min(L, lo + chunk)
        n = hi - lo
        # haplotype minor flags: (n, 4)
        hap_minor = torch.rand((n, 4), generator=gen, device=dev) < maf_t[lo:hi, None]
        # haplotype index per (snp, seq): choice[group[s], block[l]]
        hidx = choice_t[:, blk_of_snp[lo:hi]].t()[:, grp_t]          # (n, N)
        is_minor = torch.gather(hap_minor, 1, hidx)
        # 1 % per-site noise: resample from the SNP's marginal
        noise = torch.rand((n, N), generator=gen, device=dev) < 0.01
        resamp = torch.rand((n, N), generator=gen, device=dev) < maf_t[lo:hi, None]
        is_minor = torch.where(noise, resamp, is_minor)
        st = torch.where(is_minor, minor_t[lo:hi, None], major_t[lo:hi, None])
        u = torch.rand((n, N), generator=gen, device=dev)
        st = torch.where(u < thr_t[lo:hi, None], third_t[lo:hi, None], st)
        u = torch.rand((n, N), generator=gen, device=dev)
        st = torch.where(u < gap_t[lo:hi, None], torch.full_like(st, 4), st)
        # guarantee polymorphism: force one minor and one major carrier
        st[:, 0] = major_t[lo:hi]
        st[:, N - 1] = minor_t[lo:hi]
        states[lo:hi] = st
    out = dict(POS=POS, paint=paint, g=g, nclust=3)
    out["states"] = states.cpu().numpy() if as_numpy else states
    return out
