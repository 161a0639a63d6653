"""CPU: the N > 1 path (block deal + variable-length gather of link tables) with world_size 2 and 8 over gloo (8 = the rank count
of the driver's scaling run: 21 block pairs over 8 ranks leaves ranks with 2-3 blocks, ragged phases and empty ones)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ldweaver_amd.dist import deal_blocks, gather_begin, gather_block_stats, gather_end, gather_link_tables
from ldweaver_amd.mi import make_blocks


def _fake_block_links(bi, kind):
    """Deterministic fake link rows of block bi (so every rank can recompute the expected global table)."""
    rng = np.random.default_rng(1000 * bi + (7 if kind == "lr" else 3))
    n = int(rng.integers(0, 50)) if bi % 5 else 0   # some blocks contribute nothing
    return rng.integers(0, 10 ** 6, n).astype(np.int32), rng.integers(0, 10 ** 6, n).astype(np.int32), rng.random(n)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blocks = make_blocks(23000, 4000)          # 6 from-blocks -> 21 block pairs
        mine = deal_blocks(blocks, world)[rank]
        local, counts = {}, {}
        for kind in ("sr", "lr"):
            segs = [_fake_block_links(int(bi), kind) for bi in mine]
            counts[kind] = np.array([len(s[2]) for s in segs], dtype=np.int64)
            cat = lambda j, dt: torch.as_tensor(np.concatenate([s[j] for s in segs]) if segs else np.zeros(0), dtype=dt)
            local[kind] = (cat(0, torch.int32), cat(1, torch.int32), cat(2, torch.float64))
        out = gather_link_tables(local, mine, counts, len(blocks))
        # the same tables through a gather in 3 phases (rows of finished blocks travel while the rank computes on): the
        # chunks are ragged and one of rank 1's phases is empty
        cuts = [0, min(2, len(mine)), len(mine) if rank == 0 else min(2, len(mine)), len(mine)]
        phases = []
        for p0, p1 in zip(cuts[:-1], cuts[1:]):
            sub, lo, cn = mine[p0:p1], {}, {}
            for kind in ("sr", "lr"):
                segs = [_fake_block_links(int(bi), kind) for bi in sub]
                cn[kind] = np.array([len(s[2]) for s in segs], dtype=np.int64)
                cat = lambda j, dt: torch.as_tensor(np.concatenate([s[j] for s in segs]) if segs else np.zeros(0), dtype=dt)
                lo[kind] = (cat(0, torch.int32), cat(1, torch.int32), cat(2, torch.float64))
            phases.append(gather_begin(lo, sub, cn, len(blocks)))
        out3 = gather_end(phases, len(blocks))
        if rank == 0:
            for kind in ("sr", "lr"):
                for j in range(3):
                    assert torch.equal(out3[kind][j], out[kind][j])
        else:
            assert out3 is None
        # r04: the same phases with the short-range index columns left at home (sr_index=False): only sr_mi travels — 8 instead of 16 bytes
        # per short-range row — and the destination rebuilds (a, b) itself (here: from the fake generator; in the product: Engine.sr_pairs)
        phases_mi, sent16, sent8 = [], 0, 0
        for p0, p1 in zip(cuts[:-1], cuts[1:]):
            sub, lo, cn = mine[p0:p1], {}, {}
            for kind in ("sr", "lr"):
                segs = [_fake_block_links(int(bi), kind) for bi in sub]
                cn[kind] = np.array([len(s[2]) for s in segs], dtype=np.int64)
                cat = lambda j, dt: torch.as_tensor(np.concatenate([s[j] for s in segs]) if segs else np.zeros(0), dtype=dt)
                lo[kind] = (cat(0, torch.int32), cat(1, torch.int32), cat(2, torch.float64))
            ph = gather_begin(lo, sub, cn, len(blocks), sr_index=False)
            phases_mi.append(ph)
            sent8 += int(ph.mine.numel())
            sent16 += 16 * int(cn["sr"].sum() + cn["lr"].sum())
            assert int(ph.mine.numel()) == 8 * int(cn["sr"].sum()) + 16 * int(cn["lr"].sum())

        def fake_pairs(n_rows):
            exp = [_fake_block_links(bi, "sr") for bi in range(len(blocks))]
            a = torch.as_tensor(np.concatenate([e[0] for e in exp]), dtype=torch.int32)
            b = torch.as_tensor(np.concatenate([e[1] for e in exp]), dtype=torch.int32)
            assert len(a) == n_rows
            return a, b
        out_mi = gather_end(phases_mi, len(blocks), sr_pairs=fake_pairs if rank == 0 else None)
        if rank == 0:
            for kind in ("sr", "lr"):
                for j in range(3):
                    assert torch.equal(out_mi[kind][j], out[kind][j]), (kind, j)
        else:
            assert out_mi is None and (sent16 == 0 or sent8 < sent16)
        # per-block diagnostics travel the same way: every rank ends up with all blocks' rows
        fake = lambda bi: (1000 + bi, 10 + bi, 5 * bi, float("nan") if bi % 4 == 0 else 0.25 + bi)
        st = {k: np.array([fake(int(bi))[j] for bi in mine], dtype=np.float64 if k == "disc_thresh" else np.int64)
              for j, k in enumerate(("n_lr_total", "n_lr_kept", "n_sr", "disc_thresh"))}
        allst = gather_block_stats(st, mine, len(blocks))
        exp = [fake(bi) for bi in range(len(blocks))]
        assert allst["n_lr_total"].tolist() == [e[0] for e in exp] and allst["n_sr"].tolist() == [e[2] for e in exp]
        assert np.allclose(allst["disc_thresh"], [0.0 if np.isnan(e[3]) else e[3] for e in exp])
        if rank == 0:
            ok = True
            for kind in ("sr", "lr"):
                exp = [_fake_block_links(bi, kind) for bi in range(len(blocks))]
                for j in range(3):
                    ok &= np.array_equal(out[kind][j].numpy(), np.concatenate([e[j] for e in exp]))
            q.put(bool(ok))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gather_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, q)) for rk in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get() is True


def test_deal_covers_every_block_once_for_1_to_8_ranks():
    """The deal the driver's 1/2/4/8-GPU runs use on the bench's 55 block pairs: a partition, in make_blocks order per rank, and
    balanced (LPT: no rank above the mean load by more than one block's cost)."""
    from ldweaver_amd.dist import block_cost
    blocks = make_blocks(100_000, 10_000)
    cost = block_cost(blocks)
    for world in (1, 2, 4, 8):
        deal = deal_blocks(blocks, world)
        allb = np.concatenate(deal)
        assert sorted(allb.tolist()) == list(range(len(blocks)))
        assert all((np.diff(d) > 0).all() for d in deal if len(d) > 1)
        loads = np.array([cost[d].sum() for d in deal])
        assert loads.max() - loads.mean() <= cost.max()


def test_native_deal_equals_python_deal():
    """ldw_deal_blocks (the deal of ldw_mi_all_pairs_multi, host only: no GPU needed) == dist.deal_blocks block for block — one deal,
    whichever way the GPUs of a node are driven — on the bench's blocks, a ragged last block column and a single block."""
    from ldweaver_amd import _lib as LL
    for Ls, B in ((100_000, 10_000), (85_000, 10_000), (500_000, 10_000), (1268, 1000), (300, 1000)):
        blocks = np.ascontiguousarray(make_blocks(Ls, B), dtype=np.int32)
        for world in (1, 2, 3, 4, 8):
            owner = np.full(len(blocks), -1, dtype=np.int32)
            LL.check(LL.lib().ldw_deal_blocks(LL.ptr(blocks), len(blocks), world, LL.ptr(owner)))
            want = np.empty(len(blocks), dtype=np.int32)
            for rk, ids in enumerate(deal_blocks(blocks, world)):
                want[ids] = rk
            assert np.array_equal(owner, want), (Ls, B, world)
    bad = np.array([[5, 4, 1, 3]], dtype=np.int32)
    assert LL.lib().ldw_deal_blocks(LL.ptr(bad), 1, 2, LL.ptr(np.zeros(1, dtype=np.int32))) == LL.LDW_ERR_ARG


def test_gather_single_process():
    blocks = make_blocks(9000, 4000)
    mine = np.arange(len(blocks))
    local, counts = {}, {}
    for kind in ("sr", "lr"):
        segs = [_fake_block_links(int(bi), kind) for bi in mine]
        counts[kind] = np.array([len(s[2]) for s in segs], dtype=np.int64)
        local[kind] = tuple(torch.as_tensor(np.concatenate([s[j] for s in segs]), dtype=dt)
                            for j, dt in enumerate((torch.int32, torch.int32, torch.float64)))
    out = gather_link_tables(local, mine, counts, len(blocks))
    for kind in ("sr", "lr"):
        for j in range(3):
            assert torch.equal(out[kind][j], local[kind][j])
