"""Multi-GPU layer: the L x L pair space is block-partitioned over the ranks of one node.

The unit of work is a block pair of ``make_blocks`` (R/computePairwiseMI.R:147-165).  Blocks are independent
given the replicated state matrix (<= 5 GB of 288 GB per GPU), and the long-range filter is per block
(R/computePairwiseMI.R:352-358), so sharding by reference blocks reproduces the reference's retained set
exactly.  There is no data-path collective; the only exchange is ONE variable-length gather of the link
tables to rank 0: an all-gather of the per-block row counts followed by a grouped send/recv (RCCL has no
gatherv; a 7 -> 1 gather uses every peer's own xGMI link to GPU 0 concurrently).  The gather can be started in PHASES
(gather_begin per subset of a rank's blocks, gather_end once): the rows of the blocks that are done travel while the
rank computes its next ones, so only the last phase's transfer is exposed.

Works with backend "nccl" (= RCCL, GPU tensors) and "gloo" (CPU tensors, used by the CPU tests).

``force_collective`` (argument, or LDW_FORCE_COLLECTIVE=1): a group of ONE rank normally short-cuts every exchange; with the
switch the collectives run anyway — the all-reduce on the device table and a loop-back isend/irecv pair of the packed
byte buffer — so the whole RCCL path executes on a single GPU (tests/test_gpu_parity.py::test_rccl_single_rank_walk).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def _forced(flag) -> bool:
    return bool(flag) if flag is not None else os.environ.get("LDW_FORCE_COLLECTIVE", "0") not in ("", "0")


def _world_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _collective_device(group, fallback: torch.device) -> torch.device:
    """Tensors of a collective live on the GPU under RCCL and on the host under gloo."""
    if dist.get_backend(group) == "nccl":
        return fallback if fallback.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def agree(ok: bool, group=None, what: str = "", force_collective=None) -> None:
    """Error agreement: every rank reports whether its share succeeded (ONE all-reduce of a flag) BEFORE any rank enters a
    gather, so that a rank that failed (out of memory, a refused argument) cannot leave the others blocked in a collective.
    Raises RuntimeError on every rank when any rank failed; the failing rank chains its own exception to it."""
    world, rank = _world_rank(group)
    if world == 1 and not (_forced(force_collective) and dist.is_initialized()):
        if not ok:
            raise RuntimeError(f"{what or 'the computation'} failed")
        return
    dev = _collective_device(group, torch.device("cpu"))
    flag = torch.zeros(world, dtype=torch.int32, device=dev)
    flag[rank] = 0 if ok else 1
    dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
    bad = np.nonzero(flag.cpu().numpy())[0]
    if len(bad):
        raise RuntimeError(f"{what or 'the computation'} failed on rank(s) {bad.tolist()} (this is rank {rank})")


def block_cost(blocks: np.ndarray, diag_factor: float = 3.3) -> np.ndarray:
    """Relative GPU time of the block pairs.  An off-diagonal pair costs its nf * nt pairs.  A diagonal pair has half the pairs
    (half the block-wide GEMM and screen) but holds the dense short-range band, whose units take the exact 5-limb GEMM of the
    band's tiles and the whole-unit fp64 kernel: measured on the C4 shape (LDW_BLOCK_TRACE, r03: 1.40 ms per diagonal block against
    0.85 ms per off-diagonal one, pass after pass) a diagonal pair costs 1.65 off-diagonal ones, i.e. ``diag_factor`` = 3.3 times
    its own pair count (r02 assumed 2.0, which left the ranks holding two diagonal blocks ~1 ms late at N = 8)."""
    b = np.asarray(blocks, dtype=np.int64).reshape(-1, 4)
    nf, nt = b[:, 1] - b[:, 0] + 1, b[:, 3] - b[:, 2] + 1
    diag = (b[:, 0] == b[:, 2]) & (b[:, 1] == b[:, 3])
    return np.where(diag, (nf * (nf - 1) // 2 * diag_factor).astype(np.int64), nf * nt).astype(np.int64)


def deal_blocks(blocks: np.ndarray, world: int, cost: np.ndarray | None = None) -> list:
    """Cost-weighted longest-processing-time deal (``cost``: per block, default ``block_cost``); every rank keeps its blocks in
    make_blocks order."""
    cost = block_cost(blocks) if cost is None else np.asarray(cost, dtype=np.int64)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(cost), dtype=np.int64)
    for bi in order:
        rk = int(np.argmin(load))
        owner[bi] = rk
        load[rk] += cost[bi]
    return [np.nonzero(owner == rk)[0] for rk in range(world)]


class _Phase:
    """One started gather: the count table of its blocks, the packed buffers and the outstanding transfers."""
    __slots__ = ("tab", "owner", "rows", "bufs", "works", "mine", "out_dev", "rank", "world", "dst", "sr_index")


def gather_begin(local: dict, my_blocks: np.ndarray, counts: dict, nblocks: int, group=None, dst: int = 0,
                 force_collective=None, sr_index: bool = True) -> _Phase:
    """Start the gather of the link tables of a SUBSET of blocks (this rank's ``my_blocks``, any rank's may be empty) and
    return without waiting for the transfers: the caller goes on computing its next blocks while the rows travel, and
    hands all its phases to ``gather_end``.  Every rank must call it the same number of times (it contains a collective).

    local:   {"sr": (a, b, mi), "lr": (a, b, mi)} tensors of this rank (int32, int32, float64), rows grouped by
             block in the order of ``my_blocks``.
    counts:  {"sr": int64[len(my_blocks)], "lr": ...} rows per owned block.
    sr_index=False (r04): the index columns (a, b) of the SHORT-RANGE rows stay at home — they are a pure function of the positions and
             the block geometry (R/computePairwiseMI.R:306-333), the destination rebuilds them (``gather_end(..., sr_pairs=...)``,
             ``Engine.sr_pairs``) — and only their MI column travels: 8 instead of 16 bytes per row of the table that is 99 % of the gather.
    """
    ph = _Phase()
    ph.sr_index = bool(sr_index)
    world, rank = _world_rank(group)
    loopback = world == 1 and _forced(force_collective) and dist.is_initialized()
    ph.world, ph.rank, ph.dst = world, rank, dst
    ph.out_dev = local["sr"][2].device
    # gloo cannot move GPU tensors point to point: stage through the host in that case (CPU tests, or a
    # multi-process run on one GPU); RCCL ("nccl") exchanges device memory directly over xGMI
    via_host = world > 1 and dist.get_backend(group) == "gloo" and ph.out_dev.type != "cpu"
    if via_host:
        local = {k: tuple(t.cpu() for t in v) for k, v in local.items()}
    dev = local["sr"][2].device
    kinds = ("sr", "lr")
    # 1) everyone learns the row counts and owner of the blocks of this phase
    table = torch.zeros((nblocks, 3), dtype=torch.int64, device=dev)  # sr rows, lr rows, owner+1
    if len(my_blocks):
        idx = torch.as_tensor(np.asarray(my_blocks), dtype=torch.int64, device=dev)
        table[idx, 0] = torch.as_tensor(np.asarray(counts["sr"]), dtype=torch.int64, device=dev)
        table[idx, 1] = torch.as_tensor(np.asarray(counts["lr"]), dtype=torch.int64, device=dev)
        table[idx, 2] = rank + 1
    if world > 1 or loopback:
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)  # blocks are disjoint: a sum is a gather here
    tab = table.cpu().numpy()
    ph.tab, ph.owner = tab, tab[:, 2] - 1

    # 2) one packed byte buffer per rank: [sr_a | sr_b | sr_mi | lr_a | lr_b | lr_mi]
    def pack(tabs):
        parts = []
        for k in kinds:
            cols = tabs[k] if (k != "sr" or ph.sr_index) else tabs[k][2:]   # [sr_mi] alone when the index columns stay at home
            parts += [t.contiguous().view(torch.uint8) for t in cols if t.numel()]   # (an empty tensor may have stride 0)
        return torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=dev)

    ph.rows = {k: np.array([tab[ph.owner == rk, i].sum() for rk in range(world)]) for i, k in enumerate(kinds)}
    sr_b = 16 if ph.sr_index else 8
    nbytes = [int(sr_b * ph.rows["sr"][rk] + 16 * ph.rows["lr"][rk]) for rk in range(world)]
    ph.mine = pack(local)
    assert ph.mine.numel() == nbytes[rank], (ph.mine.numel(), nbytes[rank])
    if ph.mine.is_cuda:
        # `local` may alias the engine's own tables (Engine.links_view): the pack above is an asynchronous read on torch's current
        # stream, and the caller is about to overwrite those tables from the library's streams (the next phase's blocks).  The
        # packed copy must be complete before this function returns; the compute of the phase has already been waited for
        # (ldw_links_end), so this costs the duration of the copy itself
        torch.cuda.current_stream(ph.mine.device).synchronize()
    ph.bufs, ph.works = None, []
    if world > 1:
        if rank == dst:
            ph.bufs = [ph.mine if rk == dst else torch.empty(nbytes[rk], dtype=torch.uint8, device=dev) for rk in range(world)]
            ops = [dist.P2POp(dist.irecv, ph.bufs[rk], rk, group) for rk in range(world) if rk != dst and nbytes[rk] > 0]
        else:
            ops = [dist.P2POp(dist.isend, ph.mine, dst, group)] if nbytes[rank] > 0 else []
        if ops:
            ph.works = dist.batch_isend_irecv(ops)
    elif loopback and nbytes[0] > 0:
        # one rank, collectives forced: the packed buffer travels rank 0 -> rank 0 through the same grouped isend / irecv
        ph.bufs = [torch.empty(nbytes[0], dtype=torch.uint8, device=dev)]
        ph.works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, ph.bufs[0], rank, group), dist.P2POp(dist.isend, ph.mine, rank, group)])
    else:
        ph.bufs = [ph.mine]
    return ph


def gather_end(phases: list, nblocks: int, sr_pairs=None):
    """Wait for the transfers of all phases and assemble the global link tables on the destination rank in make_blocks
    order (None elsewhere).  Phases begun with ``sr_index=False`` need ``sr_pairs``: a callable ``n_rows -> (a, b)`` giving the index
    columns of the assembled short-range table (``Engine.sr_pairs`` over the whole block list)."""
    for ph in phases:
        for w in ph.works:
            w.wait()
        if ph.works and ph.mine.is_cuda:   # RCCL: wait() only orders the current torch stream behind the transfer
            torch.cuda.current_stream(ph.mine.device).synchronize()
    ph0 = phases[0]
    if ph0.rank != ph0.dst:
        return None
    kinds = ("sr", "lr")
    owned = np.zeros(nblocks, dtype=bool)
    # per phase: unpack the per-rank buffers, then cut them into per-block segments
    segs = {k: [[None, None, None] for _ in range(nblocks)] for k in kinds}
    for ph in phases:
        per_rank = []
        for rk in range(ph.world):
            o, d = 0, {}
            for k in kinds:
                n = int(ph.rows[k][rk])
                if k == "sr" and not ph.sr_index:
                    a = b = None
                else:
                    a = ph.bufs[rk][o:o + 4 * n].view(torch.int32); o += 4 * n
                    b = ph.bufs[rk][o:o + 4 * n].view(torch.int32); o += 4 * n
                mi = ph.bufs[rk][o:o + 8 * n].view(torch.float64); o += 8 * n
                d[k] = (a, b, mi)
            per_rank.append(d)
        for ki, k in enumerate(kinds):
            cursor = [0] * ph.world
            for bi in range(nblocks):
                rk = int(ph.owner[bi])
                if rk < 0:
                    continue
                owned[bi] = True
                n = int(ph.tab[bi, ki])
                c0 = cursor[rk]
                for j in range(3):
                    col = per_rank[rk][k][j]
                    segs[k][bi][j] = col[c0:c0 + n] if col is not None else None
                cursor[rk] = c0 + n
    if not owned.all():
        raise RuntimeError("some blocks were processed by no rank")
    out = {}
    rebuild_sr = any(not ph.sr_index for ph in phases)
    if rebuild_sr and not all(not ph.sr_index for ph in phases):
        raise RuntimeError("gather_end: every phase must be begun with the same sr_index")
    for k in kinds:
        cols = []
        for j, dt in enumerate((torch.int32, torch.int32, torch.float64)):
            if k == "sr" and rebuild_sr and j < 2:
                cols.append(None)
                continue
            parts = [segs[k][bi][j] for bi in range(nblocks) if segs[k][bi][j] is not None and len(segs[k][bi][j])]
            cols.append((torch.cat(parts) if parts else torch.empty(0, dtype=dt)).to(ph0.out_dev))
        if k == "sr" and rebuild_sr:
            if sr_pairs is None:
                raise RuntimeError("gather_end: phases begun with sr_index=False need sr_pairs")
            a, b = sr_pairs(int(cols[2].numel()))
            if a.numel() != cols[2].numel() or b.numel() != cols[2].numel():
                raise RuntimeError(f"gather_end: sr_pairs gave {a.numel()} rows, the gathered table has {cols[2].numel()}")
            cols[0], cols[1] = a.to(ph0.out_dev), b.to(ph0.out_dev)
        out[k] = tuple(cols)
    return out


def gather_link_tables(local: dict, my_blocks: np.ndarray, counts: dict, nblocks: int, group=None, dst: int = 0,
                       force_collective=None, sr_pairs=None):
    """Assemble the global link tables on rank ``dst`` in make_blocks order (one phase: see gather_begin / gather_end).
    Returns the same dict of global tensors on rank dst, None elsewhere."""
    world, _ = _world_rank(group)
    if world == 1 and not (_forced(force_collective) and dist.is_initialized()):
        # one rank owns every block, in make_blocks order already: nothing to move
        if len(my_blocks) != nblocks:
            raise RuntimeError("some blocks were processed by no rank")
        return {k: tuple(local[k]) for k in ("sr", "lr")}
    return gather_end([gather_begin(local, my_blocks, counts, nblocks, group=group, dst=dst, force_collective=force_collective,
                                    sr_index=sr_pairs is None)], nblocks, sr_pairs=sr_pairs)


def gather_block_stats(my_stats: dict, my_blocks: np.ndarray, nblocks: int, group=None, force_collective=None) -> dict:
    """Per-block diagnostics (Engine.block_stats) of all ranks in make_blocks order, on every rank."""
    world, _ = _world_rank(group)
    collective = world > 1 or (_forced(force_collective) and dist.is_initialized())
    keys = ("n_lr_total", "n_lr_kept", "n_sr", "disc_thresh")
    tab = torch.zeros((nblocks, 4), dtype=torch.float64)
    if len(my_blocks):
        idx = torch.as_tensor(np.asarray(my_blocks), dtype=torch.int64)
        for j, k in enumerate(keys):
            v = np.asarray(my_stats[k], dtype=np.float64)
            tab[idx, j] = torch.as_tensor(np.where(np.isnan(v), 0.0, v))   # NaN threshold (no lr links) travels as 0
    if collective:
        t = tab.to(_collective_device(group, tab.device))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)   # blocks are disjoint: a sum is a gather here
        tab = t.cpu()
    a = tab.numpy()
    return dict(n_lr_total=a[:, 0].astype(np.int64), n_lr_kept=a[:, 1].astype(np.int64), n_sr=a[:, 2].astype(np.int64),
                disc_thresh=a[:, 3].copy())


def hamming_tile_strips(nseq: int, world: int) -> list:
    """Row-tile strips [t0, t1) of the symmetric N x N sequence comparison for the ranks: only pairs (t, f), t <= f, are
    computed, so tile row k costs ~(ntiles - k); the strips are cut at equal cumulative cost."""
    ntiles = (nseq + 127) // 128
    cost = np.arange(ntiles, 0, -1, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [int(np.searchsorted(cum, cum[-1] * k / world, side="left")) for k in range(world + 1)]
    cuts[0], cuts[-1] = 0, ntiles
    return [(cuts[k], max(cuts[k], cuts[k + 1])) for k in range(world)]


def hamming_weights_sharded(eng, thresh: int, group=None, force_collective=None) -> np.ndarray:
    """estimate_Hamming_distance_weights over the ranks of an initialised process group: every rank (its engine holding the
    same alignment) counts its strip's neighbours (ldw_hamming_counts), ONE all-reduce of N integers, hdw = 1 / (n + 1)
    on every rank — integers in, so the weights are bit-identical everywhere."""
    world, rank = _world_rank(group)
    t0, t1 = hamming_tile_strips(eng.N, world)[rank]
    err = None
    try:
        cnt = eng.hamming_counts(thresh, t0, t1) if t1 > t0 else np.zeros(eng.N, dtype=np.int64)
    except Exception as e:   # agree before the all-reduce: the other ranks must not wait for this one
        err, cnt = e, np.zeros(eng.N, dtype=np.int64)
    try:
        agree(err is None, group, "the Hamming strip", force_collective)
    except RuntimeError as e:
        raise e from err
    t = torch.as_tensor(cnt)
    if world > 1 or (_forced(force_collective) and dist.is_initialized()):
        t = t.to(_collective_device(group, torch.device("cuda", eng.device) if torch.cuda.is_available() else t.device))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t = t.cpu()
    return 1.0 / (t.numpy().astype(np.float64) + 1.0)
