"""Host side of the hot path with the reference's own interface.

``perform_MI_computation``, ``estimate_Hamming_distance_weights`` and ``ACGTN2num`` keep the argument names,
meaning and error behaviour of the R functions (R/computePairwiseMI.R:46-48,
R/performPopulationStuctureCorrection.R:20, R/RcppExports.R:4-6); all numerics run in
libldweaver_amd.so on the GPU.  R is absent from this image, so this module is the Python mirror of the
R shim kept as source in r_shim/ (see INTEGRATION.md).
"""
from __future__ import annotations

import math
import os
import time
import warnings

import numpy as np
import pandas as pd

from . import _lib as L
from . import rcompat
from .engine import Engine, aracne, write_table_tsv
from .snpdat import CdsVar, SnpDat
from .srp import COLS, merge_n_sort_sr_links, merge_n_sort_sr_links_device


# ---------------------------------------------------------------------------------------------
# a-6 make_blocks                                                  R/computePairwiseMI.R:147-165
# ---------------------------------------------------------------------------------------------
def make_blocks(nsnp: int, max_blk_sz: int = 10000) -> np.ndarray:
    if max_blk_sz <= 0:
        raise ValueError("max_blk_sz rounds to 0 (round(max_blk_sz, -3)); use a block size >= 500")
    part1 = math.ceil(nsnp / max_blk_sz)
    fs = [(i - 1) * max_blk_sz + 1 for i in range(1, part1 + 1)]
    fe = [min(i * max_blk_sz, nsnp) for i in range(1, part1 + 1)]
    return np.array([(fs[i], fe[i], fs[j], fe[j]) for i in range(part1) for j in range(i, part1)], dtype=np.int32)


def lr_links_approx(POS, g, sr_dist, seed: int = 1988) -> float:
    """R/computePairwiseMI.R:94-97: 10 % of the SNPs drawn with ``set.seed(1988); sample()``."""
    POS = np.asarray(POS, dtype=np.float64)
    nsnp = len(POS)
    snp_subset = min(nsnp, int(round(nsnp * 0.1)))
    idx = rcompat.r_sample(seed, nsnp, snp_subset) - 1
    total = 0
    gi, si = float(g), float(sr_dist)
    if (nsnp > 2000 and np.all(np.diff(POS) >= 0) and gi == int(gi) and np.all(POS == np.rint(POS)) and 2 * si < gi
            and POS[0] >= 0 and POS[-1] <= gi):
        # POS ascending, everything integral: len > sr_dist  <=>  the partner is outside [x - sr, x + sr] and outside
        # the two wrap-around windows; count the complement with binary searches (identical result, O(k log n))
        x = POS[idx]
        centre = np.searchsorted(POS, x + si, "right") - np.searchsorted(POS, x - si, "left")
        wrap_lo = np.searchsorted(POS, x + si - gi, "right")            # partners <= x + sr - g
        wrap_hi = nsnp - np.searchsorted(POS, x - si + gi, "left")      # partners >= x - sr + g
        total = int((nsnp - centre - wrap_lo - wrap_hi).sum())
    else:
        step = max(1, 4_000_000 // nsnp)
        for lo in range(0, snp_subset, step):
            x = POS[idx[lo:lo + step]]
            total += int((rcompat.circ_len(x[:, None], POS[None, :], g) > sr_dist).sum())
    return total / snp_subset * nsnp / 2


# ---------------------------------------------------------------------------------------------
# a-4 estimate_Hamming_distance_weights            R/performPopulationStuctureCorrection.R:20-81
# ---------------------------------------------------------------------------------------------
def estimate_Hamming_distance_weights(snp_dat: SnpDat, threshold: float = 0.1, mega_dset: bool = False,
                                      engine: Engine | None = None, alignment_resident: bool = False, group=None,
                                      verbose: bool = True, engines=None) -> np.ndarray:
    """``engines`` (r05): several engines of this process, one per GPU — the sequence x sequence comparison is cut into one strip per
    engine inside the library (``ldw_hamming_weights_multi``); bit-identical weights."""
    t0 = time.time()
    thresh = int(snp_dat.nsnp * threshold)  # as.integer() truncates
    if engines is not None:
        if engine is not None or len(engines) < 1:
            raise ValueError("pass either `engine` or a non-empty `engines` list")
        engine = engines[0]
    own = engine is None
    eng = engine or Engine(0)
    inproc = engines is not None and len(engines) > 1
    try:
        if not alignment_resident:
            for e in (engines if inproc else [eng]):
                e.set_alignment(snp_dat.states)
        world = 1
        forced = os.environ.get("LDW_FORCE_COLLECTIVE", "0") not in ("", "0")
        try:
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized():
                world = tdist.get_world_size(group)
            else:
                forced = False
        except ImportError:
            forced = False
        if world > 1 or forced:   # one process per GPU: every rank counts a strip of the sequence x sequence comparison
            from .dist import hamming_weights_sharded
            hdw = hamming_weights_sharded(eng, thresh, group=group)
        elif inproc:
            hdw = Engine.hamming_weights_multi(engines, thresh)
        else:
            hdw = eng.hamming_weights(thresh)
    finally:
        if own:
            eng.close()
    if verbose:
        print(f"Done in {round(time.time() - t0, 2)} s")
    return hdw


# ---------------------------------------------------------------------------------------------
# a-3 .ACGTN2num                                            src/ACGTN2num_parallel.cpp:10-43
# ---------------------------------------------------------------------------------------------
def ACGTN2num(nv: np.ndarray, cv, ncores: int = 1, engine: Engine | None = None) -> None:
    """In place; returns None like the reference (invisible NULL)."""
    own = engine is None
    eng = engine or Engine(0)
    try:
        eng.acgtn2num(nv, cv)
    finally:
        if own:
            eng.close()


# ---------------------------------------------------------------------------------------------
# tsv output                                                R/computePairwiseMI.R:140,362
# ---------------------------------------------------------------------------------------------
def append_table(path: str, columns: list, nthreads: int = 0) -> int:
    """write.table(append = T, quote = F, row.names = F, col.names = F, sep = '\\t') by the native writer
    (ldw_write_table_tsv: host threads, R's 15-significant-digit rule); integer arrays print as integers.  Returns the
    bytes written.  ``rcompat.format_number`` is the Python statement of the same rule (tests compare the two)."""
    return write_table_tsv(path, columns, append=True, nthreads=nthreads)


def _pos_int(v):
    """snp.dat$POS is an INTEGER vector in the reference (src/getACGTNsites.cpp:97,173; R/extractSNPs.R:200), so the pos1 / pos2
    columns print as integers (100000, not 1e+05 as a double would)."""
    return np.asarray(v).astype(np.int64)


def links_frame(a, b, mi, POS, paint, g) -> pd.DataFrame:
    """(a, b, MI) index triples -> the reference's MI_df columns (R/computePairwiseMI.R:319-331)."""
    pos2 = np.asarray(POS, dtype=np.float64)[a]      # from side
    pos1 = np.asarray(POS, dtype=np.float64)[b]      # to side
    return pd.DataFrame({"pos1": pos1, "pos2": pos2, "clust1": np.asarray(paint)[b], "clust2": np.asarray(paint)[a],
                         "len": rcompat.circ_len(pos1, pos2, g), "MI": mi})


def _run_blocks(eng: Engine, blocks: np.ndarray, mine, kw: dict, POS, g: float) -> dict:
    """Compute the block pairs ``blocks[mine]`` (make_blocks order) into the engine's link tables and return their
    per-block statistics, one entry per block of ``mine``.  ``kw['sr_only']``: sites that form no link < sr_dist with the
    other side of their block are dropped first and blocks left empty are skipped (R/computePairwiseMI.R:179-189)."""
    mine = np.asarray(mine, dtype=np.int64)
    keys = ("n_lr_total", "n_lr_kept", "n_sr", "disc_thresh")
    out = {k: (np.full(len(mine), np.nan) if k == "disc_thresh" else np.zeros(len(mine), dtype=np.int64)) for k in keys}
    if not kw["sr_only"]:
        if len(mine):
            eng.mi_all_pairs(blocks[mine], **kw)
            return eng.block_stats()
        eng.links_begin(1)   # a rank without blocks still holds (empty) link tables
        eng.links_end()
        return out
    eng.links_begin(max(1, len(mine)))
    POSf = np.asarray(POS, dtype=np.float64)
    done = []
    for j, (fs, fe, ts, te) in enumerate(blocks[mine]):
        fi, ti = np.arange(fs - 1, fe), np.arange(ts - 1, te)
        near = _near_mask(POSf[fi], POSf[ti], g, kw["sr_dist"])
        fi, ti = fi[near[0]], ti[near[1]]
        if len(fi) == 0 or len(ti) == 0:
            continue
        eng.mi_block_links(fi, ti, **kw)
        done.append(j)
    eng.links_end()
    st = eng.block_stats()
    for k in keys:
        out[k][done] = st[k]
    return out


def _near_mask(pf: np.ndarray, pt: np.ndarray, g: float, sr_dist: float):
    """(keep_from, keep_to): sites with at least one partner on the other side at circular distance < sr_dist
    (the kp_f / kp_t of R/computePairwiseMI.R:182-183), in strips so that the nf x nt distance matrix is never whole."""
    kf = np.zeros(len(pf), dtype=bool)
    kt = np.zeros(len(pt), dtype=bool)
    step = max(1, 8_000_000 // max(1, len(pt)))
    for lo in range(0, len(pf), step):
        near = np.abs(rcompat.circ_len(pt[None, :], pf[lo:lo + step, None], g)) < sr_dist
        kf[lo:lo + step] = near.any(axis=1)
        kt |= near.any(axis=0)
    return kf, kt


# ---------------------------------------------------------------------------------------------
# a-5 perform_MI_computation                                      R/computePairwiseMI.R:46-145
# ---------------------------------------------------------------------------------------------
def perform_MI_computation(snp_dat: SnpDat, hdw, cds_var: CdsVar, ncores: int = 1, lr_save_path=None, sr_save_path=None,
                           plt_folder=None, sr_dist=20000, lr_retain_links=1e6, max_blk_sz=10000, srp_cutoff=3,
                           runARACNE=True, perform_SR_analysis_only=False, order_links=True, mega_dset=False, *,
                           engine: Engine | None = None, alignment_resident: bool = False,
                           quirk_mode: int = L.QUIRK_REFERENCE, nlimbs: int = 0, verbose: bool = True,
                           return_aux: bool = False, sr_model: str = "device", group=None, engines=None, stream_lr: bool = True,
                           sr_tail: str = "gather"):
    """Returns the short-range link data.frame (clust_c,pos1,pos2,clust1,clust2,len,MI,srp_max,ARACNE);
    long-range links are appended to ``lr_save_path`` and the returned frame to ``sr_save_path``.

    Multi-GPU: when ``torch.distributed`` is initialised with more than one rank (one process per GPU, every rank calling
    this function with the same arguments and its own ``engine``), the block pairs of ``make_blocks`` are dealt over the
    ranks, every rank computes its share on its GPU, ONE variable-length gather assembles the link tables on rank 0
    (``dist.gather_link_tables``), rank 0 adopts them (``ldw_links_import``) and runs the short-range model / ARACNE on
    them and writes the files; the other ranks return None.

    ``sr_tail`` (r05, several ranks): "gather" assembles both link tables on rank 0 as above; "dist" leaves the short-range rows — 99 % of the
    links — on the ranks that computed them: only the long-range table is gathered and the short-range model + ARACNE run over the ranks
    (``dist_srp.merge_n_sort_sr_links_dist``: per-group bounds, ~7 % of the MI column, block sums, the kept links and the pool travel).  Same
    frame and files; needs ``sr_model="device"``.

    ``engines`` (r05): several engines of THIS process, one per GPU (``[Engine(d) for d in devices]``) — the route a host that cannot
    start one process per GPU takes (R through .Call: ``options(ldwamd.devices = 0:7)`` in r_shim/).  Every engine receives the alignment,
    weights and meta data, ``ldw_mi_all_pairs_multi`` deals the block pairs over them inside the library (worker threads, peer-to-peer
    gather into ``engines[0]``), and the short-range model, ARACNE and the files run on ``engines[0]``.  SR-only passes (per-block site
    filters, :179-189) run on ``engines[0]`` alone.

    ``stream_lr`` (r05, default): on one engine ``lr_links.tsv`` is appended while the block loop runs (``ldw_lr_stream_begin`` / ``_end``) like the
    reference's per-block ``write.table(append = T)`` (:362); ``False`` writes the table after the pass, beside the short-range model (r04)."""
    t000 = time.time()
    say = print if verbose else (lambda *a, **k: None)
    if lr_save_path is None:
        lr_save_path = os.path.join(os.getcwd(), "lr_links.tsv")
    if sr_save_path is None:
        sr_save_path = os.path.join(os.getcwd(), "sr_links.tsv")
    if plt_folder is None:
        plt_folder = os.path.join(os.getcwd(), "PLOTS")
    os.makedirs(plt_folder, exist_ok=True)
    if snp_dat.g is None:
        raise ValueError("snp.dat$g is NULL: set the genome length first (R/BacGWES.R:338-345)")
    say("Begin MI computation... ")
    max_blk_sz = rcompat.round_thousands(max_blk_sz)
    blocks = make_blocks(snp_dat.nsnp, max_blk_sz)
    POS, g, paint = snp_dat.POS, float(snp_dat.g), np.asarray(cds_var.paint)
    # (snp.dat$POS may be in any order, like in the reference: blocks whose lists do not ascend take the library's generic path)
    approx = None if perform_SR_analysis_only else lr_links_approx(POS, g, sr_dist)

    if engines is not None:
        if engine is not None or len(engines) < 1:
            raise ValueError("pass either `engine` or a non-empty `engines` list")
        engine = engines[0]
    own = engine is None
    eng = engine or Engine(0)
    inproc = engines is not None and len(engines) > 1 and not perform_SR_analysis_only
    stages = {"lr_links_approx_s": time.time() - t000}
    lr_stream, streamed = False, None
    dist_tail, inproc_rows_stay = None, False
    try:
        def setup():
            for e in (engines if inproc else [eng]):
                if not alignment_resident:   # pass alignment_resident=True when `engine` (every one of `engines`) already holds snp_dat.states
                    e.set_alignment(snp_dat.states)
                e.set_weights(hdw, nlimbs)
                e.set_snp_meta(snp_dat.r, snp_dat.uqe, POS, paint, g)

        kw = dict(sr_dist=sr_dist, lr_retain_links=lr_retain_links, lr_links_approx=approx or 1.0,
                  sr_only=perform_SR_analysis_only, quirk=quirk_mode)
        world, rank = 1, 0
        forced = os.environ.get("LDW_FORCE_COLLECTIVE", "0") not in ("", "0")
        try:
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized():
                world, rank = tdist.get_world_size(group), tdist.get_rank(group)
            else:
                forced = False
        except ImportError:
            forced = False
        if world > 1 or forced:
            from .dist import agree, deal_blocks, gather_block_stats, gather_link_tables
            mine = deal_blocks(blocks, world)[rank]
            err, my_stats, local = None, None, None
            try:
                setup()
                my_stats = _run_blocks(eng, blocks, mine, kw, POS, g)
                local = {"sr": eng.links(0, device_tensors=True), "lr": eng.links(1, device_tensors=True)}
            except Exception as e:   # the other ranks must learn of it before they enter the gather
                err = e
            try:
                agree(err is None, group, "perform_MI_computation")
            except RuntimeError as e:
                raise e from err
            # r04: the index columns of the short-range rows are a function of the positions alone (R/computePairwiseMI.R:306-333): with ascending
            # POS and whole blocks only their MI column travels and rank 0 rebuilds (a, b) (Engine.sr_pairs); SR-only passes (filtered site
            # lists) and unsorted positions send all three columns
            mi_only = (not perform_SR_analysis_only) and bool(np.all(np.diff(np.asarray(POS, dtype=np.int64)) >= 0))
            if sr_tail == "dist":
                if sr_model != "device":
                    raise ValueError("sr_tail='dist' needs sr_model='device'")
                import torch
                e_ = lambda dt: torch.empty(0, dtype=dt, device=local["lr"][2].device)
                out = gather_link_tables({"sr": (e_(torch.int32), e_(torch.int32), e_(torch.float64)), "lr": local["lr"]}, mine,
                                         {"sr": np.zeros(len(mine), dtype=np.int64), "lr": my_stats["n_lr_kept"]}, len(blocks), group=group)
                stats = gather_block_stats(my_stats, mine, len(blocks), group=group)
                dist_tail = (mine, stats["n_sr"])
                if rank != 0:   # this rank's rows take part in the model below; nothing comes back to it
                    from .dist_srp import merge_n_sort_sr_links_dist
                    merge_n_sort_sr_links_dist(eng, cds_var.nclust, sr_dist, srp_cutoff, POS, paint, g, mine, stats["n_sr"], run_aracne=runARACNE,
                                               order_links=order_links, group=group)
                    return None
                eng.links_import(1, *out["lr"])
            elif sr_tail != "gather":
                raise ValueError("sr_tail must be 'gather' or 'dist'")
            else:
                out = gather_link_tables(local, mine, {"sr": my_stats["n_sr"], "lr": my_stats["n_lr_kept"]}, len(blocks), group=group,
                                         sr_pairs=(lambda n: eng.sr_pairs(blocks, sr_dist, n)) if mi_only else None)
                stats = gather_block_stats(my_stats, mine, len(blocks), group=group)
                if rank != 0:
                    return None
                eng.links_import(0, *out["sr"])
                eng.links_import(1, *out["lr"])
        else:
            t_s = time.time()
            setup()
            stages["setup_s"] = time.time() - t_s
            lr_stream = stream_lr and not inproc and not perform_SR_analysis_only
            t_s = time.time()
            if inproc:
                if sr_tail not in ("gather", "dist"):
                    raise ValueError("sr_tail must be 'gather' or 'dist'")
                inproc_rows_stay = sr_tail == "dist" and sr_model == "device"
                stages["inproc"] = Engine.mi_all_pairs_multi(engines, blocks, sr_rows_stay=inproc_rows_stay, **kw)
                stages["inproc"]["owner"] = stages["inproc"]["owner"].tolist()
                stats = eng.block_stats()
            else:
                # r05: lr_links.tsv is appended WHILE the pass runs, item by item, as the reference appends it block by block
                # (R/computePairwiseMI.R:362): a pass that dies leaves the finished blocks' rows, and the file is complete when the pass is
                if lr_stream:
                    eng.lr_stream_begin(lr_save_path, append=True)
                try:
                    stats = _run_blocks(eng, blocks, np.arange(len(blocks)), kw, POS, g)
                except BaseException:
                    if lr_stream:
                        eng.lr_stream_end()   # (on failure: what was finished is on disk before the error travels on)
                    raise
                # (the writer's last batch is formatted beside the short-range model below, like the r04 table-at-once writer: lr_stream_end follows it)
            stages["mi_all_pairs_s"] = time.time() - t_s
        # lr_links.tsv (R/computePairwiseMI.R:362) straight from the device-resident table: fetched, derived (pos, clust, len) and
        # formatted by the library's host threads
        # r04: the table is fetched here and formatted / written by host threads WHILE the short-range model below runs on the GPU (the model
        # does not touch the long-range table: R/computePairwiseMI.R:119-126); `lr_tsv_s` = what the job waits for it (fetch + the final join)
        t_w = time.time()
        n_lr_rows = 0
        if not perform_SR_analysis_only and not lr_stream:
            eng.write_links_tsv_begin(1, lr_save_path, append=True)
        tsv_s = time.time() - t_w
        t_s = time.time()
        if sr_model == "device" and dist_tail is not None:
            from .dist_srp import merge_n_sort_sr_links_dist
            redd, flags, model_aux = merge_n_sort_sr_links_dist(eng, cds_var.nclust, sr_dist, srp_cutoff, POS, paint, g, dist_tail[0], dist_tail[1],
                                                                run_aracne=runARACNE, order_links=order_links, group=group)
            pool = eng.sr_pool() if (return_aux and runARACNE and len(redd["MI"])) else None   # (the pool of all ranks, as rank 0 adopted it)
            fit_data = model_aux["fit_data"]
            sa, sb, smi = redd["a"], redd["b"], redd["MI"]
            stages["sr_tail_bytes_sent"] = model_aux["bytes_sent"]
        elif sr_model == "device" and inproc_rows_stay:
            # r05: the short-range rows stayed on the engines that computed them; the model's reductions run over them inside the library
            from .engine import EngineGroup
            grp = EngineGroup(engines)
            redd, flags, model_aux = merge_n_sort_sr_links_device(grp, cds_var.nclust, sr_dist, srp_cutoff, POS, paint, g,
                                                                  run_aracne=runARACNE, order_links=order_links)
            pool = grp.sr_pool() if return_aux else None
            fit_data = model_aux["fit_data"]
            sa, sb, smi = redd["a"], redd["b"], redd["MI"]
        elif sr_model == "device":
            # mergeNsort_sr_links + runARACNE on the device-resident table; only the kept links come back
            redd, flags, model_aux = merge_n_sort_sr_links_device(eng, cds_var.nclust, sr_dist, srp_cutoff, POS, paint, g,
                                                                  run_aracne=runARACNE, order_links=order_links, block_rows=stats["n_sr"])
            pool = eng.sr_pool() if return_aux else None
            fit_data = model_aux["fit_data"]
            sa, sb, smi = redd["a"], redd["b"], redd["MI"]
        elif sr_model == "host":
            sa, sb, smi = eng.links(0)
        else:
            raise ValueError("sr_model must be 'device' or 'host'")
        stages["sr_model_aracne_s"] = time.time() - t_s
        t_w = time.time()
        if not perform_SR_analysis_only and not lr_stream:
            n_lr_rows, _ = eng.write_links_tsv_end()
        elif lr_stream:
            streamed = eng.lr_stream_end()
            n_lr_rows = streamed[0]
        tsv_s += time.time() - t_w
        stages["lr_tsv_s"] = tsv_s
        path_report = eng.path_report()
    finally:
        if lr_stream:
            try:
                eng.lr_stream_end()   # (no-op after the regular end above; an error on the way here must not leave the writer open on the caller's engine)
            except Exception:
                pass
        if own:
            eng.close()
    if not path_report["apx_gate"].startswith("ok") and not perform_SR_analysis_only:
        say(f"note: the approximate-GEMM path is off for these weights ({path_report['apx_gate']}); limb paths used")
    for bi in range(len(stats["n_sr"])):
        say(f"Block {bi + 1} of {len(blocks)} ... Adding {stats['n_lr_kept'][bi]} LR links with MI>"
            f"{round(float(stats['disc_thresh'][bi]), 3)} to file ... Adding {stats['n_sr'][bi]} SR links to list ...")


    sr = links_frame(sa, sb, smi, POS, paint, g)
    if sr_model == "device":
        red = sr
        red.insert(0, "clust_c", redd["clust_c"].astype(np.int64))
        red["srp_max"] = redd["srp_max"]
        chk = links_frame(*pool, POS, paint, g) if pool is not None else None
        if runARACNE:
            say(f"Running ARACNE on {len(red)} links... ")
            red["ARACNE"] = flags.astype(np.float64)
    else:
        sr_links = [sr[(sr["clust1"] == ci) | (sr["clust2"] == ci)] for ci in range(1, cds_var.nclust + 1)]
        fit_data = []
        red, chk = merge_n_sort_sr_links(sr_links, cds_var.nclust, sr_dist, srp_cutoff, fit_data=fit_data)
        if runARACNE:
            say(f"Running ARACNE on {len(red)} links... ")
            red["ARACNE"] = aracne(red["pos1"], red["pos2"], red["MI"], chk["pos1"], chk["pos2"], chk["MI"]).astype(np.float64)
    # the data of c<i>_fit_data.rds (saveRDS(maxvls), R/computePairwiseMI.R:439) as a tsv with a header; the png is out of scope
    for ci, fd in enumerate(fit_data):
        with open(os.path.join(plt_folder, f"c{ci + 1}_fit_data.tsv"), "w") as fh:
            fh.write("len\tmax\tfit\n")
        append_table(os.path.join(plt_folder, f"c{ci + 1}_fit_data.tsv"), [fd["len"].to_numpy(), fd["max"].to_numpy(), fd["fit"].to_numpy()])
    if not runARACNE:
        warnings.warn("ARACNE not run, all values will be set to 1")
        red["ARACNE"] = 1.0
    if order_links and sr_model != "device":   # (the device path has ordered the columns before the frame was built)
        red = red.iloc[np.argsort(-red["srp_max"].to_numpy(), kind="stable")].reset_index(drop=True)
    t_w = time.time()
    append_table(sr_save_path, [_pos_int(red[c]) if c in ("pos1", "pos2") else red[c].to_numpy() for c in ["clust_c"] + COLS + ["srp_max", "ARACNE"]])
    tsv_s += time.time() - t_w
    stages["sr_tsv_s"] = time.time() - t_w
    stages["total_s"] = time.time() - t000
    say(f"All done in {round((time.time() - t000) / 60, 2)} mins ")
    if return_aux:   # not part of the reference's return value: the ARACNE pool and per-block statistics
        return red, dict(sr_links_ARACNE_check=chk, block_stats=stats, lr_links_approx=approx, fit_data=fit_data, tsv_write_s=tsv_s,
                         lr_rows_written=n_lr_rows, path=path_report, stages_s=stages)
    return red
