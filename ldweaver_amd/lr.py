"""Consumers of the link tables (SURVEY.md §8 f rank 4): the numeric cores of ``analyse_long_range_links``
(R/lr_analyser.R:29-190) and ``genomewide_LDMap`` (R/LDSummaryPlot.R:25-125) on the device-resident tables of the
engine.  Plots, SnpEff annotation and the tanglegram are out of scope (SURVEY.md §8 f)."""
from __future__ import annotations

import warnings

import numpy as np
import pandas as pd

from .rcompat import circ_len


SR_TSV_COLS = ["clust_c", "pos1", "pos2", "clust1", "clust2", "len", "MI", "srp_max", "ARACNE"]   # R/computePairwiseMI.R:140


def positions_to_snp_index(POS, p) -> np.ndarray:
    """0-based SNP index of every position in ``p`` — ``snp.dat$POS`` in ANY order (the reference imposes none: R/computePairwiseMI.R:176-177;
    r03 assumed ascending positions here); a position held by several SNPs maps to the first of them.  Raises on a position that is no SNP's."""
    POS = np.asarray(POS)
    p = np.asarray(p)
    order = np.argsort(POS, kind="stable")
    srt = POS[order]
    k = np.searchsorted(srt, p, side="left")
    if len(p) and not np.array_equal(srt[np.minimum(k, len(srt) - 1)], p):
        raise ValueError("sr_links holds positions that are not SNP positions of snp_dat")
    return order[np.minimum(k, len(srt) - 1)] if len(p) else np.zeros(0, dtype=np.int64)


def analyse_long_range_links(eng, snp_dat, sr_links, cds_var=None, are_lrlinks_ordered: bool = False, min_links: int = 5000) -> dict:
    """Tukey outlier analysis + ARACNE of the long-range links the engine holds after ``perform_MI_computation`` /
    ``mi_all_pairs`` (the reference reads them back from lr_links.tsv).  ``sr_links`` is the short-range part of the ARACNE
    pool exactly as the reference has it: the contents of sr_links.tsv (R/lr_analyser.R:67), i.e. the REDUCED frame
    ``perform_MI_computation`` returned (srp_max > srp_cutoff) — pass that frame (pos1, pos2, MI columns) or the path of the
    tsv.  Returns the reference's ``lr_links_red`` (pos1 pos2 [clust1 clust2] len MI ARACNE, descending MI unless
    ``are_lrlinks_ordered``) plus the thresholds.  Everything O(#links) runs on the device (ldw_lr_tukey, ldw_aracne_device)."""
    if isinstance(sr_links, (str, bytes)) or hasattr(sr_links, "__fspath__"):
        sr_links = pd.read_csv(sr_links, sep="\t", header=None, names=SR_TSV_COLS)
    POS_ = np.asarray(snp_dat.POS)
    p1, p2 = np.asarray(sr_links["pos1"]), np.asarray(sr_links["pos2"])
    sb_, sa_ = positions_to_snp_index(POS_, p1), positions_to_snp_index(POS_, p2)      # pos1 = to side (b), pos2 = from side (a)
    info = eng.lr_tukey(min_links, sr=(sa_, sb_, np.asarray(sr_links["MI"], dtype=np.float64)))
    if info["fallback"]:   # R/lr_analyser.R:96
        warnings.warn("Not enough lr links pass the Tukey criteria, ~5000 top links were retained instead")
    red = eng.lr_reduced()
    flags = eng.aracne_device()
    POS = np.asarray(snp_dat.POS)
    a, b = red["a"], red["b"]
    pos1, pos2 = POS[b].astype(np.int64), POS[a].astype(np.int64)     # to side = pos1, from side = pos2 (R/computePairwiseMI.R:319-320)
    cols = dict(pos1=pos1, pos2=pos2)
    if cds_var is not None:
        paint = np.asarray(cds_var.paint)
        cols.update(clust1=paint[b], clust2=paint[a])
    cols.update(len=circ_len(pos1.astype(float), pos2.astype(float), float(snp_dat.g)), MI=red["MI"], ARACNE=flags.astype(int))
    df = pd.DataFrame(cols)
    if not are_lrlinks_ordered:    # :115-117
        df = df.iloc[np.argsort(-df["MI"].to_numpy(), kind="stable")].reset_index(drop=True)
    return dict(lr_links_red=df, q13=info["q13"], thresholds=info["thresholds"], fallback=info["fallback"], n_pool=info["n_pool"])


def genomewide_LDMap(eng, snp_dat, reducer=None, from_=None, to=None) -> dict:
    """Numeric core of ``genomewide_LDMap``: the reduced, log10-scaled, 0..1-rescaled LD matrix ``htm`` with its row /
    column labels (the reference's ``nms``: pos_vec[seq(1, n, by = reducer - 1)][1:B], R/LDSummaryPlot.R:95-96)."""
    if reducer is not None and reducer < 0:     # :30-35
        warnings.warn("<reducer> for genomewide_LDMap should be >0, set to default")
        reducer = None
    if (from_ is None) != (to is None):         # :37-38
        raise ValueError("If <from> is provided, <to> must be provided as well!" if to is None else
                         "If <to> is provided, <from> must be provided as well!")
    if from_ is not None:                       # :43-47
        if to <= from_:
            raise ValueError("<to> must be greater than <from>!")
        if from_ < 0 or to < 0:
            raise ValueError("<from> and <to> must be positive values")
        from_, to = int(round(from_)), int(round(to))
    r = 0 if reducer is None else int(np.round(reducer))
    htm, n_pos, r = eng.ldmap(r, from_ or 0, to or 0)
    return dict(htm=htm, n_pos=n_pos, reducer=r)
