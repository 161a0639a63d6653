"""Multi-GPU layer: the L x L pair space is block-partitioned over the ranks of one node.

The unit of work is a block pair of ``make_blocks`` (R/computePairwiseMI.R:147-165).  Blocks are independent
given the replicated state matrix (<= 5 GB of 288 GB per GPU), and the long-range filter is per block
(R/computePairwiseMI.R:352-358), so sharding by reference blocks reproduces the reference's retained set
exactly.  There is no data-path collective; the only exchange is ONE variable-length gather of the link
tables to rank 0: an all-gather of the per-block row counts followed by a grouped send/recv (RCCL has no
gatherv; a 7 -> 1 gather uses every peer's own xGMI link to GPU 0 concurrently).

Works with backend "nccl" (= RCCL, GPU tensors) and "gloo" (CPU tensors, used by the CPU tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def block_cost(blocks: np.ndarray) -> np.ndarray:
    b = np.asarray(blocks, dtype=np.int64).reshape(-1, 4)
    nf, nt = b[:, 1] - b[:, 0] + 1, b[:, 3] - b[:, 2] + 1
    diag = (b[:, 0] == b[:, 2]) & (b[:, 1] == b[:, 3])
    return np.where(diag, nf * (nf - 1) // 2, nf * nt).astype(np.int64)


def deal_blocks(blocks: np.ndarray, world: int) -> list:
    """Cost-weighted longest-processing-time deal; every rank keeps its blocks in make_blocks order."""
    cost = block_cost(blocks)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(cost), dtype=np.int64)
    for bi in order:
        rk = int(np.argmin(load))
        owner[bi] = rk
        load[rk] += cost[bi]
    return [np.nonzero(owner == rk)[0] for rk in range(world)]


def gather_link_tables(local: dict, my_blocks: np.ndarray, counts: dict, nblocks: int, group=None, dst: int = 0):
    """Assemble the global link tables on rank ``dst`` in make_blocks order.

    local:   {"sr": (a, b, mi), "lr": (a, b, mi)} tensors of this rank (int32, int32, float64), rows grouped by
             block in the order of ``my_blocks``.
    counts:  {"sr": int64[len(my_blocks)], "lr": ...} rows per owned block.
    Returns the same dict of global tensors on rank dst, None elsewhere.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:   # one rank owns every block, in make_blocks order already: nothing to move
        if len(my_blocks) != nblocks:
            raise RuntimeError("some blocks were processed by no rank")
        return {k: tuple(local[k]) for k in ("sr", "lr")}
    out_dev = local["sr"][2].device
    # gloo cannot move GPU tensors point to point: stage through the host in that case (CPU tests, or a
    # multi-process run on one GPU); RCCL ("nccl") exchanges device memory directly over xGMI
    via_host = world > 1 and dist.get_backend(group) == "gloo" and out_dev.type != "cpu"
    if via_host:
        local = {k: tuple(t.cpu() for t in v) for k, v in local.items()}
    dev = local["sr"][2].device
    kinds = ("sr", "lr")
    # 1) everyone learns every block's row counts and owner
    table = torch.zeros((nblocks, 3), dtype=torch.int64, device=dev)  # sr rows, lr rows, owner+1
    if len(my_blocks):
        idx = torch.as_tensor(np.asarray(my_blocks), dtype=torch.int64, device=dev)
        table[idx, 0] = torch.as_tensor(np.asarray(counts["sr"]), dtype=torch.int64, device=dev)
        table[idx, 1] = torch.as_tensor(np.asarray(counts["lr"]), dtype=torch.int64, device=dev)
        table[idx, 2] = rank + 1
    if world > 1:
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)  # blocks are disjoint: a sum is a gather here
    tab = table.cpu().numpy()
    owner = tab[:, 2] - 1
    if (owner < 0).any():
        raise RuntimeError("some blocks were processed by no rank")

    # 2) one packed byte buffer per rank: [sr_a | sr_b | sr_mi | lr_a | lr_b | lr_mi]
    def pack(tabs):
        parts = []
        for k in kinds:
            a, b, mi = tabs[k]
            parts += [a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8), mi.contiguous().view(torch.uint8)]
        return torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=dev)

    rows = {k: np.array([tab[owner == rk, i].sum() for rk in range(world)]) for i, k in enumerate(kinds)}
    nbytes = [int(16 * (rows["sr"][rk] + rows["lr"][rk])) for rk in range(world)]
    mine = pack(local)
    assert mine.numel() == nbytes[rank], (mine.numel(), nbytes[rank])
    bufs = None
    if world > 1:
        if rank == dst:
            bufs = [mine if rk == dst else torch.empty(nbytes[rk], dtype=torch.uint8, device=dev) for rk in range(world)]
            ops = [dist.P2POp(dist.irecv, bufs[rk], rk, group) for rk in range(world) if rk != dst and nbytes[rk] > 0]
        else:
            ops = [dist.P2POp(dist.isend, mine, dst, group)] if nbytes[rank] > 0 else []
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
    else:
        bufs = [mine]
    if rank != dst:
        return None

    # 3) unpack and interleave the per-rank segments back into make_blocks order
    out = {}
    per_rank = []
    for rk in range(world):
        o, d = 0, {}
        for k in kinds:
            n = int(rows[k][rk])
            a = bufs[rk][o:o + 4 * n].view(torch.int32); o += 4 * n
            b = bufs[rk][o:o + 4 * n].view(torch.int32); o += 4 * n
            mi = bufs[rk][o:o + 8 * n].view(torch.float64); o += 8 * n
            d[k] = (a, b, mi)
        per_rank.append(d)
    for ki, k in enumerate(kinds):
        cursor = [0] * world
        segs = ([], [], [])
        for bi in range(nblocks):
            rk, n = int(owner[bi]), int(tab[bi, ki])
            if n:
                c0 = cursor[rk]
                for j in range(3):
                    segs[j].append(per_rank[rk][k][j][c0:c0 + n])
                cursor[rk] = c0 + n
        out[k] = tuple((torch.cat(s) if s else per_rank[0][k][j][:0]).to(out_dev) for j, s in enumerate(segs))
    return out


def gather_block_stats(my_stats: dict, my_blocks: np.ndarray, nblocks: int, group=None) -> dict:
    """Per-block diagnostics (Engine.block_stats) of all ranks in make_blocks order, on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    keys = ("n_lr_total", "n_lr_kept", "n_sr", "disc_thresh")
    tab = torch.zeros((nblocks, 4), dtype=torch.float64)
    if len(my_blocks):
        idx = torch.as_tensor(np.asarray(my_blocks), dtype=torch.int64)
        for j, k in enumerate(keys):
            v = np.asarray(my_stats[k], dtype=np.float64)
            tab[idx, j] = torch.as_tensor(np.where(np.isnan(v), 0.0, v))   # NaN threshold (no lr links) travels as 0
    if world > 1:
        t = tab.cuda() if dist.get_backend(group) == "nccl" else tab
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)   # blocks are disjoint: a sum is a gather here
        tab = t.cpu()
    a = tab.numpy()
    return dict(n_lr_total=a[:, 0].astype(np.int64), n_lr_kept=a[:, 1].astype(np.int64), n_sr=a[:, 2].astype(np.int64),
                disc_thresh=a[:, 3].copy())
