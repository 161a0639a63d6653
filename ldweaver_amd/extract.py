"""FASTA -> SNP filter -> 5-state matrix: host side of ``parse_fasta_alignment`` / ``parse_fasta_SNP_alignment``
(R/extractSNPs.R:23-142, 168-281) on top of the device scan / encoder (``ldw_alignment_scan``,
``ldw_encode_alignment``) that replace ``.extractAlnParam`` / ``.extractSNPs`` (src/getACGTNsites.cpp:13-291).
SURVEY.md §8(f) rank 1 ("next"): it removes the five sparse matrices and their dense re-expansion from the path.
"""
from __future__ import annotations

import warnings

import numpy as np

from .engine import Engine
from .snpdat import SnpDat, read_fasta


def snp_filter(allele_counts: np.ndarray, n: int, gap_thresh: float = 0.15, maf_thresh: float = 0.01, filt: int = 0) -> np.ndarray:
    """1-based retained columns.  ``allele_counts``: 5 x L (A,C,G,T,other).  Rule of src/getACGTNsites.cpp:104-166:
    at least two non-gap alleles present, gap fraction ``counts[4]/n < gap_thresh`` and
    * filt 0 (spydrpick default): second-largest non-gap count > ``int(n*maf_thresh)``
    * filt 1 (relaxed):           largest of the five counts <= ``int(n*(1-maf_thresh))``."""
    ac = np.asarray(allele_counts, dtype=np.float64)
    poly = (ac[:4] > 0).sum(axis=0) > 1
    gap_ok = ac[4] / n < gap_thresh
    if filt == 0:
        min_maf = int(n * maf_thresh)
        second = np.sort(ac[:4], axis=0)[2]
        keep = poly & gap_ok & (second > min_maf)
    else:
        min_maf = int(n * (1 - maf_thresh))
        keep = poly & gap_ok & (ac.max(axis=0) <= min_maf)
    return (np.nonzero(keep)[0] + 1).astype(np.int32)


def _method_to_filter(method: str) -> int:
    if method == "default":
        return 0
    if method == "relaxed":
        return 1
    warnings.warn("Unkown filtering method, using default...")
    return 0


def _parse(aln_path, gap_freq, maf_freq, method, engine, keep_on_device):
    names, chars = read_fasta(aln_path)          # raises on ragged / empty input like the reference's stop()s
    n, ltot = chars.shape
    own = engine is None
    eng = engine or Engine(0)
    try:
        counts = eng.alignment_scan(chars)
        pos = snp_filter(counts, n, gap_freq, maf_freq, _method_to_filter(method))
        if len(pos) == 0:
            raise ValueError("File does not contain any SNPs")
        table = eng.encode_alignment(None, pos)
        states = None if keep_on_device else eng.get_alignment()
    finally:
        if own:
            eng.close()
    return names, ltot, pos, table, states


def parse_fasta_alignment(aln_path, gap_freq=0.15, maf_freq=0.01, method="default", mega_dset=False, *,
                          engine: Engine | None = None, keep_on_device: bool = False) -> SnpDat:
    """Mirror of ``parse_fasta_alignment``: ``g`` = alignment length, ``POS`` = retained columns.
    With ``engine`` given and ``keep_on_device`` the state matrix stays resident there (``states`` is None)."""
    names, ltot, pos, table, states = _parse(aln_path, gap_freq, maf_freq, method, engine, keep_on_device)
    uqe = (table > 0).T.astype(np.float64)
    return SnpDat(states=states, POS=pos, g=float(ltot), uqe=uqe, r=uqe.sum(axis=1), seq_names=names)


def parse_fasta_SNP_alignment(aln_path, pos, gap_freq=0.15, maf_freq=0.01, method="default", mega_dset=False, *,
                              engine: Engine | None = None, keep_on_device: bool = False) -> SnpDat:
    """Mirror of ``parse_fasta_SNP_alignment`` (SNP-only alignment + positions file): ``g`` is NULL until patched
    from the annotation (R/BacGWES.R:338-345), ``POS = pos[retained]``."""
    names, ltot, kept, table, states = _parse(aln_path, gap_freq, maf_freq, method, engine, keep_on_device)
    pos = np.asarray(pos)
    if len(pos) != ltot:
        raise ValueError("Error! Number of positions do not match the fasta sequence length")
    uqe = (table > 0).T.astype(np.float64)
    return SnpDat(states=states, POS=pos[kept - 1].astype(np.int32), g=None, uqe=uqe, r=uqe.sum(axis=1), seq_names=names)
