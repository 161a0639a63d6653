"""The short-range model of ``mergeNsort_sr_links`` (R/computePairwiseMI.R:400-495) + ``runARACNE`` over the ranks of a multi-GPU job
WITHOUT assembling the short-range table on one GPU (r05; docs/HISTORY.md 7b).

The reference sees one table.  Here every rank keeps the rows of the block pairs it computed (99 % of a job's links: 1.44 GB at C4,
36 GB at C5) and the ranks exchange only what the model's four data reductions need:

  1. per (cluster, len) the two order statistics of quantile(MI, 0.95) (:422) — every rank's local count and lower order statistic;
     ``L = min over ranks`` of the latter is a lower bound of the global one (rank r has at most floor(0.95 (n_r - 1)) members below its own,
     and the sum of those is <= floor(0.95 (n - 1))), so the global order statistics lie among the rows >= L: each rank sends those
     (``ldw_sr_tail_extract``: 5-9 % of its MI column, values only, grouped) and rank 0 selects by rank from the top
     (``ldw_sr_quantiles_merge``; a group whose statistic were NOT among the candidates is counted and raises);
  2. the five sums per cluster of the positive excesses (:444-452) — per reference BLOCK (``ldw_sr_excess_stats_blocks``), summed over the
     blocks in ``make_blocks`` order: bit-identical for any deal of the blocks over any number of ranks;
  3. p-values per row where the row lies (:453); the kept links travel, the smallest kept MI is a minimum over ranks;
  4. the ARACNE pool (:489-490) from every rank's rows for THAT minimum; rank 0 runs ARACNE on kept links x pool.

The log-log least-squares fit (:428) and the beta MLE (:452) stay O(1)-sized host steps as in ``srp.merge_n_sort_sr_links_device``.
Collectives: all-reduce (counts, bounds, block sums, minimum), broadcast (fitted decay), variable-length gathers to rank 0 (candidates,
kept links, pool).  Backends: "nccl" (= RCCL) and "gloo".
"""
from __future__ import annotations

import numpy as np

from .srp import _betaln, beta_mle_stats, fit_decay, stable_argsort_desc

RED_DT = np.dtype([("row", "<i8"), ("MI", "<f8"), ("srp_max", "<f8"), ("a", "<i4"), ("b", "<i4"), ("clust_c", "<i4"),
                   ("first_clust", "<i4"), ("dup", "u1"), ("pad", "u1", (7,))])
POOL_DT = np.dtype([("MI", "<f8"), ("a", "<i4"), ("b", "<i4")])


class Comm:
    """The three exchanges the protocol needs, on numpy arrays, over a torch.distributed group (or alone: world 1)."""

    def __init__(self, group=None, device=None):
        self.group, self.world, self.rank, self.dev = group, 1, 0, None
        self.bytes_sent = 0
        self.alone = True    # no exchange at all (one rank, collectives not forced)
        try:
            import os
            import torch
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
                if dist.get_backend(group) == "nccl":
                    self.dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
                else:
                    self.dev = torch.device("cpu")
                # LDW_FORCE_COLLECTIVE=1 (dist.py): a group of ONE rank runs every collective anyway — the all-reduces and broadcasts on the backend's
                # tensors, the gathers as a loop-back isend / irecv — so the RCCL branch of this file executes on a box with one GPU (tests/rccl_worker.py)
                self.alone = self.world == 1 and os.environ.get("LDW_FORCE_COLLECTIVE", "0") in ("", "0")
        except ImportError:
            pass

    def _t(self, arr):
        import torch
        return torch.from_numpy(np.array(arr, copy=True, order="C")).to(self.dev)   # (a copy: from_numpy shares memory and the collectives work in place)

    def all_reduce(self, arr: np.ndarray, op: str) -> np.ndarray:
        """op: 'sum' | 'min'; same shape and dtype on every rank."""
        if self.alone:
            return np.array(arr, copy=True)
        import torch.distributed as dist
        t = self._t(arr)
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN}[op], group=self.group)
        self.bytes_sent += t.numel() * t.element_size()   # (what a ring moves per rank, to within a factor (world - 1) / world x 2)
        return t.cpu().numpy()

    def bcast(self, arr, shape, dtype, src: int = 0) -> np.ndarray:
        if self.alone:
            return np.asarray(arr, dtype=dtype).reshape(shape)
        import torch
        import torch.distributed as dist
        t = self._t(np.asarray(arr, dtype=dtype).reshape(shape)) if self.rank == src else torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=self.dev)
        dist.broadcast(t, src=src, group=self.group)
        if self.rank == src:
            self.bytes_sent += t.numel() * t.element_size()
        return t.cpu().numpy()

    def agree(self, err, what: str) -> None:
        """Every rank learns whether a step failed on ANY rank before it enters the next exchange (``err``: this rank's exception or None)."""
        if self.alone:
            if err is not None:
                raise err
            return
        from .dist import agree
        try:
            agree(err is None, self.group, what, force_collective=True if self.world == 1 else None)
        except RuntimeError as e:
            raise e from err

    def gatherv(self, arr: np.ndarray, dst: int = 0):
        """1-D arrays of one dtype and any length -> list of every rank's array on ``dst`` (None elsewhere)."""
        arr = np.ascontiguousarray(arr)
        if self.alone:
            return [arr]
        import torch
        import torch.distributed as dist
        nb = np.zeros(self.world, dtype=np.int64)
        nb[self.rank] = arr.nbytes
        nb = self.all_reduce(nb, "sum")
        mine = self._t(arr.view(np.uint8).reshape(-1))
        if self.world == 1:   # collectives forced on one rank: the buffer travels rank 0 -> rank 0 through the same grouped isend / irecv
            buf = torch.empty(int(nb[0]), dtype=torch.uint8, device=self.dev)
            if nb[0] > 0:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, 0, self.group), dist.P2POp(dist.isend, mine, 0, self.group)]):
                    w.wait()
                if self.dev.type == "cuda":
                    torch.cuda.current_stream(self.dev).synchronize()
            self.bytes_sent += int(nb[0])
            return [buf.cpu().numpy().view(arr.dtype)]
        if self.rank == dst:
            bufs = [mine if r == dst else torch.empty(int(nb[r]), dtype=torch.uint8, device=self.dev) for r in range(self.world)]
            ops = [dist.P2POp(dist.irecv, bufs[r], r, self.group) for r in range(self.world) if r != dst and nb[r] > 0]
        else:
            bufs = None
            ops = [dist.P2POp(dist.isend, mine, dst, self.group)] if nb[self.rank] > 0 else []
            self.bytes_sent += int(nb[self.rank])
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if self.dev is not None and self.dev.type == "cuda":
            torch.cuda.current_stream(self.dev).synchronize()
        if self.rank != dst:
            return None
        return [b.cpu().numpy().view(arr.dtype) for b in bufs]


def _gatherv_device(cm: "Comm", t, dst: int = 0):
    """Comm.gatherv for ONE-dimensional float64 tensors that already lie on the backend's device (RCCL): no pass through host memory."""
    import torch
    import torch.distributed as dist
    nb = np.zeros(cm.world, dtype=np.int64)
    nb[cm.rank] = t.numel()
    nb = cm.all_reduce(nb, "sum")
    torch.cuda.current_stream(t.device).synchronize()
    if cm.world == 1:    # collectives forced on one rank: loop-back
        buf = torch.empty(int(nb[0]), dtype=t.dtype, device=t.device)
        if nb[0] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, 0, cm.group), dist.P2POp(dist.isend, t, 0, cm.group)]):
                w.wait()
            torch.cuda.current_stream(t.device).synchronize()
        cm.bytes_sent += int(nb[0]) * 8
        return [buf]
    if cm.rank == dst:
        bufs = [t if r == dst else torch.empty(int(nb[r]), dtype=t.dtype, device=t.device) for r in range(cm.world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, cm.group) for r in range(cm.world) if r != dst and nb[r] > 0]
    else:
        bufs = None
        ops = [dist.P2POp(dist.isend, t, dst, cm.group)] if nb[cm.rank] > 0 else []
        cm.bytes_sent += int(nb[cm.rank]) * 8
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    torch.cuda.current_stream(t.device).synchronize()
    return bufs


class ThreadGroup:
    """Ranks that are THREADS of one process, one engine (one GPU) each: the exchanges of ``Comm`` through shared memory.  The library's calls
    release the interpreter lock, so the ranks' kernels run side by side."""

    def __init__(self, world: int):
        import threading
        self.world = int(world)
        self.barrier = threading.Barrier(self.world)
        self.slots = [None] * self.world

    def comm(self, rank: int) -> "ThreadComm":
        return ThreadComm(self, rank)


class ThreadComm(Comm):
    def __init__(self, tg: ThreadGroup, rank: int):
        self.group, self.world, self.rank, self.dev, self.bytes_sent, self.tg, self.alone = None, tg.world, int(rank), None, 0, tg, False

    def _exchange(self, obj):
        self.tg.slots[self.rank] = obj
        self.tg.barrier.wait()
        got = list(self.tg.slots)
        self.tg.barrier.wait()   # (nobody overwrites a slot before everybody has read them)
        return got

    def all_reduce(self, arr, op):
        got = self._exchange(np.array(arr, copy=True))
        out = got[0].copy()
        for x in got[1:]:        # rank order: the same sum on every rank
            out = out + x if op == "sum" else np.minimum(out, x)
        self.bytes_sent += out.nbytes
        return out

    def bcast(self, arr, shape, dtype, src=0):
        got = self._exchange(np.asarray(arr, dtype=dtype).reshape(shape) if self.rank == src else None)
        if self.rank == src:
            self.bytes_sent += got[src].nbytes
        return np.array(got[src], copy=True)

    def agree(self, err, what):
        bad = [r for r, e in enumerate(self._exchange(err)) if e is not None]
        if bad:
            raise RuntimeError(f"{what} failed on rank(s) {bad} (this is rank {self.rank})") from err

    def gatherv(self, arr, dst=0):
        arr = np.ascontiguousarray(arr)
        got = self._exchange(arr)
        if self.rank != dst:
            self.bytes_sent += arr.nbytes
            return None
        return got


def merge_n_sort_sr_links_dist(eng, nclust: int, sr_dist: float, srp_cutoff: float, POS, paint, g, my_blocks, n_sr_blocks, run_aracne=True,
                               order_links=False, group=None, comm: Comm | None = None):
    """``srp.merge_n_sort_sr_links_device`` for a job whose short-range rows lie on several ranks.

    eng:          this rank's engine; its short-range table holds the rows of ``my_blocks`` (indices into the job's block list, ascending) in
                  that order, as ``perform_MI_computation``'s block loop leaves it.
    n_sr_blocks:  short-range rows of EVERY block of the job (``dist.gather_block_stats(...)["n_sr"]``: known on all ranks).
    Every rank calls it (it contains collectives).  Rank 0 returns what ``merge_n_sort_sr_links_device`` returns — the kept links in the
    reference's order, their ARACNE flags, side results (+ ``bytes_sent`` per exchange) —; the other ranks return (None, None, aux).
    With one rank the same steps run without any exchange: the result does not depend on the number of ranks, bit for bit."""
    cm = comm or Comm(group, getattr(eng, "device", None))
    if int(np.max(paint)) > nclust or int(np.min(paint)) < 1:
        raise ValueError("Cluster mismatch detected, stopping!")
    my_blocks = np.asarray(my_blocks, dtype=np.int64)
    n_sr_blocks = np.asarray(n_sr_blocks, dtype=np.int64)
    rows_mine = n_sr_blocks[my_blocks]
    import time
    sent, ms = {}, {}
    mark = [cm.bytes_sent, time.perf_counter()]

    def took(what):   # bytes this rank sent and wall clock since the last mark (local work + the exchange + waiting for the slowest rank)
        now = time.perf_counter()
        sent[what] = cm.bytes_sent - mark[0]
        ms[what] = (now - mark[1]) * 1e3
        mark[0], mark[1] = cm.bytes_sent, now

    def local(what, fn):
        """A rank-local stage between two exchanges: run it, then let every rank learn whether it failed ANYWHERE before anybody enters the
        next exchange (a rank that raises — out of memory at config 5, a refused argument — would otherwise leave the others blocked in an
        all-reduce / gather until the backend's timeout).  One int32 all-reduce per stage (``dist.agree``); nothing when there is one rank."""
        out, err = None, None
        try:
            out = fn()
        except Exception as e:
            err = e
        cm.agree(err, f"the short-range model's '{what}' stage")
        return out

    # 1. order statistics per (cluster, len)
    qlo, qhi, cnt = local("len quantiles", lambda: eng.sr_len_quantiles(nclust, sr_dist, 0.95))
    S = qlo.shape[1]
    n_total = cm.all_reduce(cnt, "sum")
    lower = cm.all_reduce(np.where(cnt > 0, qlo, np.inf), "min")
    lower = np.where(np.isfinite(lower), lower, np.nan)            # groups without a member anywhere: nothing to send
    # under RCCL the candidates stay on the device from the extraction to the merge (at config 5 they are 180 MB per rank of eight: through host
    # memory twice on every rank and once more on rank 0 would be most of the step); gloo and the in-process ranks use host arrays
    on_dev = (not cm.alone) and cm.dev is not None and cm.dev.type == "cuda" and hasattr(eng, "_ctx")
    if on_dev:
        tcnt, tmi = local("tail extract", lambda: eng.sr_tail_extract(lower, on_device=True))
        took("bounds")
        cnts, mis = cm.gatherv(tcnt.reshape(-1)), _gatherv_device(cm, tmi)
    else:
        tcnt, tmi = local("tail extract", lambda: eng.sr_tail_extract(lower))
        took("bounds")
        cnts, mis = cm.gatherv(tcnt.reshape(-1)), cm.gatherv(tmi)
    took("candidates")
    md = None
    fit_data = []
    lens = np.arange(1, S + 1, dtype=np.float64)
    err = None
    if cm.rank == 0:
        try:
            import pandas as pd
            gq_lo, gq_hi, viol = eng.sr_quantiles_merge(0.95, [c.reshape(S, nclust) for c in cnts], mis, n_total)
            if viol:
                raise RuntimeError(f"short-range quantiles: {viol} groups whose order statistic is not among the candidates the ranks sent")
            md = np.full((nclust, S), np.nan)
            for ci in range(nclust):
                has = n_total[ci] > 0
                n = n_total[ci][has].astype(np.float64)
                index = 1 + (n - 1) * 0.95                      # quantile type 7 (stats::quantile)
                h = index - np.floor(index)
                lo, hi = gq_lo[ci][has], gq_hi[ci][has]
                maxvls = np.where((h > 0) & (hi != lo), (1 - h) * lo + h * hi, lo)
                mean_dist = fit_decay(lens[has], maxvls)
                md[ci, :len(mean_dist)] = mean_dist             # looked up by the VALUE of len (Q5)
                fit_data.append(pd.DataFrame({"len": lens[has], "max": maxvls, "fit": mean_dist}))
        except Exception as e:   # the other ranks wait in the broadcast below: they must learn of it first
            err = e
    cm.agree(err, "the short-range quantile merge on rank 0")
    md = cm.bcast(md, (nclust, S), np.float64)
    took("fitted_decay")
    # 2. excess statistics: per block, summed in make_blocks order (adding the zeros of the other ranks' blocks is exact)
    def block_sums():
        part = np.zeros((len(n_sr_blocks), nclust, 5))
        if len(my_blocks):
            part[my_blocks] = eng.sr_excess_stats_blocks(md, rows_mine)
        return part
    part = cm.all_reduce(local("block sums", block_sums), "sum")
    took("block_sums")
    stats = np.zeros((nclust, 5))
    for b in range(len(n_sr_blocks)):
        stats += part[b]
    # 3. p-values where the rows lie; 4. the pool for the minimum over ranks
    def pvalues():
        shape = np.empty((nclust, 3))
        for ci in range(nclust):
            a_, b_ = beta_mle_stats(*stats[ci])
            shape[ci] = a_, b_, _betaln(a_, b_)
        return (shape,) + tuple(eng.sr_pvalues_local(md, shape, srp_cutoff))
    shape, n_red_local, min_local = local("p-values", pvalues)
    min_mi = float(cm.all_reduce(np.array([min_local if n_red_local else np.inf]), "min")[0])

    def kept_and_pool():
        n_pool_local = eng.sr_pool_build(min_mi) if np.isfinite(min_mi) else 0
        red = eng.sr_reduced()
        pa, pb, pmi = eng.sr_pool() if n_pool_local else (np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0))
        # rows of this rank's table -> rows of the job's table in make_blocks order
        loc_off = np.concatenate([[0], np.cumsum(rows_mine)])
        glo_off = np.concatenate([[0], np.cumsum(n_sr_blocks)])
        bi = np.searchsorted(loc_off, red["row"], side="right") - 1
        rec = np.zeros(n_red_local, dtype=RED_DT)
        rec["row"] = glo_off[my_blocks[bi]] + (red["row"] - loc_off[bi])
        for k in ("MI", "srp_max", "a", "b", "clust_c", "first_clust"):
            rec[k] = red[k]
        rec["dup"] = red["dup"]
        prec = np.zeros(n_pool_local, dtype=POOL_DT)
        prec["MI"], prec["a"], prec["b"] = pmi, pa, pb
        return rec, prec
    rec, prec = local("kept links and pool", kept_and_pool)
    took("minimum")
    reds, pools = cm.gatherv(rec), cm.gatherv(prec)
    took("kept_links_and_pool")
    aux = dict(mean_dist=md, shape=shape, stats=stats, min_mi=min_mi if np.isfinite(min_mi) else np.nan, counts=n_total, fit_data=fit_data, bytes_sent=sent, ms=ms,
               candidates=int(tmi.numel() if hasattr(tmi, "numel") else len(tmi)), local_rows=int(rows_mine.sum()))
    if cm.rank != 0:
        return None, None, aux
    allr, allp = np.concatenate(reds), np.concatenate(pools)
    n_red = len(allr)
    aux["n_pool"] = len(allp)
    if run_aracne and n_red:
        eng.sr_reduced_import(allr["a"], allr["b"], allr["MI"], allp["a"], allp["b"], allp["MI"])
        flags = eng.aracne_device()
    else:
        flags = np.ones(n_red, dtype=bool)
    took("aracne_on_rank0")
    # reference row order: per cluster the links inside one cluster, then the cross-cluster links in order of first appearance (:470-486)
    dup = allr["dup"].astype(bool)
    key_cl = np.where(dup, allr["first_clust"], allr["clust_c"])
    order = np.lexsort((allr["row"], key_cl, dup))
    if order_links:
        order = order[stable_argsort_desc(allr["srp_max"][order])]   # sr_links_red[order(-srp_max)] (:126; order() is stable)
    out = {k: np.ascontiguousarray(allr[k][order]) for k in ("row", "a", "b", "MI", "clust_c", "first_clust", "srp_max")}
    out["dup"] = dup[order]
    return out, flags[order], aux
