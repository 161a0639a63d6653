"""The short-range model over ranks (ldweaver_amd/dist_srp.py): every rank keeps its rows, the ranks exchange bounds,
~7 % of the MI column, block sums, kept links and the pool — against the one-table model (R/computePairwiseMI.R:400-495).

CPU part: the PROTOCOL (bounds, offsets, row order, the merge's rank arithmetic) on a numpy stand-in for the engine's reductions, ranks as
threads and as two gloo processes, against the oracle's model on the assembled table.  GPU part: the library's entry points themselves, three
engines on the box's one GPU as three ranks, against one engine holding the whole table."""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ldw_oracle as orc  # noqa: E402
from ldweaver_amd import dist_srp  # noqa: E402
from ldweaver_amd import srp as srp_host  # noqa: E402


# ------------------------------------------------------------------------------------------------------------------
# numpy stand-in for the engine's short-range reductions (tests only): the semantics of include/ldweaver_amd.h (7), (7b)
# ------------------------------------------------------------------------------------------------------------------
class NumpyRank:
    def __init__(self, a, b, mi, POS, paint, g):
        self.a, self.b, self.mi = np.asarray(a), np.asarray(b), np.asarray(mi, dtype=np.float64)
        self.POS, self.paint, self.g = np.asarray(POS, dtype=np.float64), np.asarray(paint), float(g)
        self.len_all = orc.circ_len(self.POS[self.b], self.POS[self.a], self.g).astype(np.int64)
        self.c1, self.c2 = self.paint[self.b].astype(np.int64), self.paint[self.a].astype(np.int64)

    def _members(self):
        """(row, cluster, group) of every membership: a row of two clusters is a member of both (R/computePairwiseMI.R:411-414)."""
        ok = self.len > 0
        r1, r2 = np.nonzero(ok)[0], np.nonzero(ok & (self.c2 != self.c1))[0]
        rows, cl = np.concatenate([r1, r2]), np.concatenate([self.c1[r1], self.c2[r2]])
        return rows, cl, (self.len[rows] - 1) * self.nclust + (cl - 1)

    def sr_len_quantiles(self, nclust, sr_dist, prob):
        self.S = S = int(np.ceil(sr_dist)) - 1
        self.nclust = nclust
        self.len = np.where((self.len_all > 0) & (self.len_all < sr_dist), self.len_all, 0)
        rows, cl, grp = self._members()
        G = S * nclust
        o = np.lexsort((self.mi[rows], grp))
        v, n = self.mi[rows][o], np.bincount(grp, minlength=G)
        st = np.concatenate([[0], np.cumsum(n)])[:-1]
        has = n > 0
        idx = 1 + (n[has] - 1).astype(float) * prob
        qlo, qhi = np.full(G, np.nan), np.full(G, np.nan)
        qlo[has] = v[st[has] + np.floor(idx).astype(np.int64) - 1]
        qhi[has] = v[st[has] + np.ceil(idx).astype(np.int64) - 1]
        lm = lambda x: np.ascontiguousarray(x.reshape(S, nclust).T)     # group order is len-major; the API's tables are (nclust, S)
        return lm(qlo), lm(qhi), lm(n).astype(np.int64)

    def sr_tail_extract(self, lower):
        nclust, S = lower.shape
        rows, cl, grp = self._members()
        keep = self.mi[rows] >= lower[cl - 1, self.len[rows] - 1]
        grp, v = grp[keep], self.mi[rows][keep]
        o = np.argsort(grp, kind="stable")
        return np.bincount(grp, minlength=S * nclust).reshape(S, nclust).astype(np.int64), v[o]

    def sr_quantiles_merge(self, prob, cnts, mis, n_total):
        nclust, S = n_total.shape
        G = S * nclust
        grp = np.concatenate([np.repeat(np.arange(G), np.asarray(c).reshape(-1)) for c in cnts])
        v = np.concatenate(mis)
        o = np.lexsort((v, grp))
        v, here = v[o], np.bincount(grp, minlength=G)
        st = np.concatenate([[0], np.cumsum(here)])[:-1]
        n = np.ascontiguousarray(n_total.T).reshape(-1)
        qlo, qhi = np.full(G, np.nan), np.full(G, np.nan)
        idx = 1 + (np.maximum(n, 1) - 1).astype(float) * prob
        lo, hi, below = np.floor(idx).astype(np.int64) - 1, np.ceil(idx).astype(np.int64) - 1, n - here
        bad = ((n == 0) & (here > 0)) | ((n > 0) & ((lo < below) | (here > n)))
        ok = (n > 0) & ~bad
        qlo[ok], qhi[ok] = v[st[ok] + lo[ok] - below[ok]], v[st[ok] + hi[ok] - below[ok]]
        lm = lambda x: np.ascontiguousarray(x.reshape(S, nclust).T)
        return lm(qlo), lm(qhi), int(bad.sum())

    def _excess(self, md):
        rows, cl, grp = self._members()
        return rows, cl, self.mi[rows] - md[cl - 1, self.len[rows] - 1]

    def sr_excess_stats_blocks(self, md, rows_per_block):
        nb, nclust = len(rows_per_block), md.shape[0]
        off = np.concatenate([[0], np.cumsum(rows_per_block)])
        rows, cl, d = self._excess(md)
        pos = d > 0
        rows, cl, x = rows[pos], cl[pos], d[pos]
        key = (np.searchsorted(off, rows, side="right") - 1) * nclust + (cl - 1)
        out = np.zeros((nb * nclust, 5))
        for k, w in enumerate((np.ones_like(x), x, x * x, np.log(x), np.log1p(-x))):
            out[:, k] = np.bincount(key, weights=w, minlength=nb * nclust)
        return out.reshape(nb, nclust, 5)

    def sr_pvalues_local(self, md, shape, cutoff):
        n = len(self.mi)
        rows, cl, d = self._excess(md)
        pos = d > 0
        rows, cl, d = rows[pos], cl[pos], d[pos]
        srp = np.empty(len(d))
        for c in range(1, md.shape[0] + 1):
            m = cl == c
            srp[m] = srp_host.neg_log_beta_sf(d[m], shape[c - 1, 0], shape[c - 1, 1])
        # per row: the maximum over its clusters, ties to the smaller cluster id; the first cluster (ascending id) with a positive excess
        best, cc, first = np.full(n, -np.inf), np.zeros(n, np.int64), np.full(n, 1 << 30, np.int64)
        for r_, c_, s_ in sorted(zip(rows.tolist(), cl.tolist(), srp.tolist()), key=lambda t: (t[0], t[1])):
            if s_ > best[r_]:
                best[r_], cc[r_] = s_, c_
            first[r_] = min(first[r_], c_)
        keep = (cc > 0) & (best > cutoff)
        row = np.nonzero(keep)[0]
        self._red = dict(row=row, a=self.a[row], b=self.b[row], MI=self.mi[row], clust_c=cc[row], first_clust=first[row],
                         dup=(self.c1 != self.c2)[row], srp_max=best[row])
        self._anypos = cc > 0
        return len(row), (self.mi[row].min() if len(row) else np.nan)

    def sr_pool_build(self, min_mi):
        self._pool = np.nonzero(self._anypos & (self.mi >= min_mi))[0]
        return len(self._pool)

    def sr_reduced(self):
        return self._red

    def sr_pool(self):
        return self.a[self._pool], self.b[self._pool], self.mi[self._pool]

    def sr_reduced_import(self, a, b, mi, pa, pb, pmi):
        self._imp = (a, b, mi, pa, pb, pmi)

    def aracne_device(self):
        a, b, mi, pa, pb, pmi = self._imp
        P = self.POS
        return np.asarray(orc.run_aracne(P[b], P[a], mi, P[pb], P[pa], pmi)).astype(bool)


def _job(seed, L=900, nblk=6, nclust=3, sr_dist=300.0):
    """A small synthetic short-range table in make_blocks-like block order: rows grouped by block, MI decaying with len plus noise."""
    rng = np.random.default_rng(seed)
    g = 3000.0
    POS = np.sort(rng.choice(np.arange(1, int(g)), size=L, replace=False)).astype(np.float64)
    paint = rng.integers(1, nclust + 1, size=L)
    a = rng.integers(0, L, size=60000)
    b = rng.integers(0, L, size=60000)
    ln = orc.circ_len(POS[b], POS[a], g)
    ok = (ln > 0) & (ln < sr_dist) & (a != b)
    a, b, ln = a[ok], b[ok], ln[ok]
    mi = np.clip(0.3 * ln ** -0.5 * rng.lognormal(0.0, 0.5, size=len(ln)) + 1e-4, 1e-6, 0.95)
    mi[rng.integers(0, len(mi), size=200)] = mi[0]              # ties across ranks
    blk = rng.integers(0, nblk, size=len(mi))
    blk[ln > 0.8 * sr_dist] = nblk - 1                          # groups that exist on ONE rank only (the others' local statistics are NaN there)
    o = np.argsort(blk, kind="stable")
    a, b, mi, blk = a[o], b[o], mi[o], blk[o]
    n_sr_blocks = np.bincount(blk, minlength=nblk)
    return dict(a=a, b=b, mi=mi, POS=POS, paint=paint, g=g, n_sr_blocks=n_sr_blocks, blk=blk, nclust=nclust, sr_dist=sr_dist)


def _run_ranks(job, owners, make_rank, world):
    """Ranks as threads: rank r holds the rows of the blocks owners == r, in block order."""
    tg = dist_srp.ThreadGroup(world)
    res, errs = [None] * world, [None] * world

    def work(rk):
        try:
            mine = np.nonzero(owners == rk)[0]
            sel = np.isin(job["blk"], mine)
            eng = make_rank(job["a"][sel], job["b"][sel], job["mi"][sel])
            res[rk] = dist_srp.merge_n_sort_sr_links_dist(eng, job["nclust"], job["sr_dist"], 2.0, job["POS"], job["paint"], job["g"], mine,
                                                          job["n_sr_blocks"], run_aracne=True, order_links=True, comm=tg.comm(rk))
        except BaseException as e:   # noqa: BLE001
            errs[rk] = e
            tg.barrier.abort()

    th = [threading.Thread(target=work, args=(rk,)) for rk in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    return res


def _same(x, y):
    (rx, fx, ax), (ry, fy, ay) = x, y
    assert set(rx) == set(ry)
    for k in rx:
        assert np.array_equal(rx[k], ry[k]), k
    assert np.array_equal(fx, fy)
    for k in ("mean_dist", "shape", "stats", "counts"):
        assert np.array_equal(ax[k], ay[k], equal_nan=True), k


@pytest.mark.parametrize("seed", [1, 2])
def test_protocol_on_numpy_ranks_equals_the_oracle_model(seed):
    """1, 2 and 4 thread-ranks over a numpy stand-in: bit-identical results for every rank count and deal (an idle rank included), and the
    same kept links / srp / ARACNE flags as the ORACLE's mergeNsort_sr_links + runARACNE on the assembled table."""
    job = _job(seed)
    mk = lambda a, b, mi: NumpyRank(a, b, mi, job["POS"], job["paint"], job["g"])
    nblk = len(job["n_sr_blocks"])
    one = _run_ranks(job, np.zeros(nblk, int), mk, 1)[0]
    red, flags, aux = one
    assert len(red["MI"]) > 50 and 0 < flags.sum() <= len(flags)
    for world, owners in ((2, np.arange(nblk) % 2), (4, np.array([3, 0, 0, 3, 1, 0])[:nblk]), (3, np.array([1, 1, 1, 1, 1, 1])[:nblk])):
        got = _run_ranks(job, owners, mk, world)
        _same(one, got[0])
        assert all(g[0] is None for g in got[1:])
        if world > 1:
            sent = [sum(g[2]["bytes_sent"].values()) for g in got]
            assert max(sent[1:]) < 0.6 * 8 * len(job["mi"])     # (a small job: the counts per group dominate; the GPU test prices the real shape)
    # the oracle on the whole table
    a, b, mi, POS, paint, g = job["a"], job["b"], job["mi"], job["POS"], job["paint"], job["g"]
    tab = dict(pos1=POS[b], pos2=POS[a], clust1=paint[b], clust2=paint[a], len=orc.circ_len(POS[b], POS[a], g), MI=mi)
    by_clust = [{k: v[(tab["clust1"] == ci) | (tab["clust2"] == ci)] for k, v in tab.items()} for ci in range(1, job["nclust"] + 1)]
    ored, ochk = orc.merge_n_sort_sr_links(by_clust, job["nclust"], job["sr_dist"], 2.0)
    o = np.argsort(-np.asarray(ored["srp_max"]), kind="stable")
    assert len(ored["MI"]) == len(red["MI"])
    assert np.array_equal(np.asarray(ored["pos1"])[o], POS[red["b"]]) and np.array_equal(np.asarray(ored["pos2"])[o], POS[red["a"]])
    assert np.array_equal(np.asarray(ored["MI"])[o], red["MI"]) and np.array_equal(np.asarray(ored["clust_c"])[o], red["clust_c"])
    np.testing.assert_allclose(red["srp_max"], np.asarray(ored["srp_max"])[o], rtol=1e-6)
    of = orc.run_aracne(ored["pos1"], ored["pos2"], ored["MI"], ochk["pos1"], ochk["pos2"], ochk["MI"])
    assert np.array_equal(np.asarray(of)[o].astype(bool), flags)


def test_a_wrong_bound_is_caught():
    """The merge counts groups whose order statistic is not among the candidates: a rank that under-reports (here: a bound above the true
    statistic) raises on every rank instead of returning a wrong quantile."""
    job = _job(3)

    class Liar(NumpyRank):
        def sr_tail_extract(self, lower):
            return super().sr_tail_extract(lower + 0.2)

    mk = lambda a, b, mi: Liar(a, b, mi, job["POS"], job["paint"], job["g"])
    with pytest.raises(RuntimeError, match="order statistic|failed on rank"):
        _run_ranks(job, np.arange(len(job["n_sr_blocks"])) % 2, mk, 2)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        job = _job(1)
        owners = np.arange(len(job["n_sr_blocks"])) % world
        mine = np.nonzero(owners == rank)[0]
        sel = np.isin(job["blk"], mine)
        eng = NumpyRank(job["a"][sel], job["b"][sel], job["mi"][sel], job["POS"], job["paint"], job["g"])
        red, flags, aux = dist_srp.merge_n_sort_sr_links_dist(eng, job["nclust"], job["sr_dist"], 2.0, job["POS"], job["paint"], job["g"], mine,
                                                              job["n_sr_blocks"], run_aracne=True, order_links=True)
        q.put((rank, red, flags, {k: aux[k] for k in ("mean_dist", "shape", "stats", "counts", "bytes_sent")}))
    finally:
        dist.destroy_process_group()


def test_protocol_over_two_gloo_processes():
    """The same exchanges through torch.distributed (gloo, world_size 2): equal to the one-rank run bit for bit."""
    import torch.multiprocessing as mp
    job = _job(1)
    mk = lambda a, b, mi: NumpyRank(a, b, mi, job["POS"], job["paint"], job["g"])
    one = _run_ranks(job, np.zeros(len(job["n_sr_blocks"]), int), mk, 1)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(2):
        r, red, flags, aux = q.get(timeout=300)
        got[r] = (red, flags, aux)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    _same(one, got[0])
    assert got[1][0] is None and got[1][2]["bytes_sent"]["candidates"] > 0


# ------------------------------------------------------------------------------------------------------------------
# GPU: the library's entry points, three engines on one GPU as three ranks
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_sr_model_over_ranks_equals_one_table(tmp_path):
    """VERDICT r04 item 6.  12 000 SNPs x 1 000 sequences in 10 blocks.  (i) ldw_sr_tail_extract / ldw_sr_quantiles_merge /
    ldw_sr_excess_stats_blocks / ldw_sr_pool_build against numpy on the fetched table; (ii) the model over 1, 2 and 3 ranks (engines of this
    process, ranks as threads) — bit-identical kept links, srp, ARACNE flags, fitted decay and shapes for every rank count; (iii) against the one-table
    device model (same rows and order, srp to 1e-9: its sums run in another order); (iv) what each rank sent."""
    from ldweaver_amd import mi as MIH
    from ldweaver_amd.engine import Engine
    from ldweaver_amd.synth import synth_alignment
    Ls, N, B = 12_000, 1_000, 3_000
    syn = synth_alignment(Ls, N, seed=1988, device="cuda", as_numpy=False)
    POS, g, paint = syn["POS"], float(syn["g"]), syn["paint"]
    blocks = MIH.make_blocks(Ls, B)
    approx = MIH.lr_links_approx(POS, g, 20000.0)
    kw = dict(sr_dist=20000.0, lr_retain_links=2e5, lr_links_approx=approx)
    engs = [Engine(0) for _ in range(3)]
    try:
        for e in engs:
            e.set_alignment(syn["states"])
        cnt = engs[0].state_counts()
        uqe = (cnt > 0).T.astype(np.float64)
        r = uqe.sum(axis=1)
        hdw = engs[0].hamming_weights(int(Ls * 0.1))
        for e in engs:
            e.set_weights(hdw)
            e.set_snp_meta(r, uqe, POS, paint, g)
        nclust = int(paint.max())
        cut = 2.0
        engs[0].mi_all_pairs(blocks, **kw)
        n_sr_blocks = engs[0].block_stats()["n_sr"]
        a, b, mi = engs[0].links(0)
        assert len(mi) == n_sr_blocks.sum() > 100_000

        # (i) the new reductions one by one against numpy
        ref = NumpyRank(a, b, mi, POS, paint, g)
        qlo, qhi, qn = engs[0].sr_len_quantiles(nclust, 20000.0, 0.95)
        lower = np.where(qn > 0, qlo, np.nan)
        lower[:, ::3] = -np.inf                                           # every member of a third of the groups
        tc, tm = engs[0].sr_tail_extract(lower)
        ref.sr_len_quantiles(nclust, 20000.0, 0.95)
        rc, rm = ref.sr_tail_extract(lower)
        assert np.array_equal(tc, rc) and len(tm) == tc.sum() == len(rm)
        off = np.concatenate([[0], np.cumsum(tc.reshape(-1))])
        roff = off
        for gi in np.nonzero(tc.reshape(-1))[0][:4000]:
            assert np.array_equal(np.sort(tm[off[gi]:off[gi + 1]]), np.sort(rm[roff[gi]:roff[gi + 1]])), gi
        # candidates of "two ranks" = a split of this table's candidates: the merge gives the table's own order statistics
        lo2 = np.where(qn > 0, qlo, np.nan)
        c2, m2 = engs[0].sr_tail_extract(lo2)
        o2 = np.concatenate([[0], np.cumsum(c2.reshape(-1))])
        ca, cb = c2.reshape(-1) // 2, c2.reshape(-1) - c2.reshape(-1) // 2
        ma = np.concatenate([m2[o2[k]:o2[k] + ca[k]] for k in range(len(ca))])
        mb = np.concatenate([m2[o2[k] + ca[k]:o2[k + 1]] for k in range(len(ca))])
        glo, ghi, viol = engs[1].sr_quantiles_merge(0.95, [ca.reshape(c2.shape), cb.reshape(c2.shape)], [ma, mb], qn)
        assert viol == 0 and np.array_equal(glo, qlo, equal_nan=True) and np.array_equal(ghi, qhi, equal_nan=True)
        _, _, viol = engs[1].sr_quantiles_merge(0.95, [ca.reshape(c2.shape)], [ma], qn)     # half the candidates missing: counted, not returned
        assert viol > 0
        md = np.full((nclust, 19999), np.nan)
        md[:, :3000] = 0.05 * np.arange(1, 3001, dtype=float) ** -0.4
        per_block = engs[0].sr_excess_stats_blocks(md, n_sr_blocks)
        want = ref.sr_excess_stats_blocks(md, n_sr_blocks)
        np.testing.assert_allclose(per_block, want, rtol=1e-11, atol=1e-300)
        assert np.array_equal(per_block[:, :, 0], want[:, :, 0])
        np.testing.assert_allclose(per_block.sum(axis=0), engs[0].sr_excess_stats(md), rtol=1e-12)
        with pytest.raises(Exception, match="rows"):
            engs[0].sr_excess_stats_blocks(md, n_sr_blocks[:-1])
        # six clusters: the kernels for any cluster count (k_sr_stats per block, k_sr_tail, the merge) on the same table under another paint
        paint6 = ((paint - 1) * 2 + (np.arange(Ls) // 7) % 2 + 1).astype(paint.dtype)
        engs[0].set_snp_meta(r, uqe, POS, paint6, g)
        ref6 = NumpyRank(a, b, mi, POS, paint6, g)
        q6 = engs[0].sr_len_quantiles(6, 20000.0, 0.95)
        r6 = ref6.sr_len_quantiles(6, 20000.0, 0.95)
        for u, v in zip(q6, r6):
            assert np.array_equal(u, v, equal_nan=True)
        md6 = np.full((6, 19999), np.nan)
        md6[:, :3000] = 0.05 * np.arange(1, 3001, dtype=float) ** -0.4
        pb6, want6 = engs[0].sr_excess_stats_blocks(md6, n_sr_blocks), ref6.sr_excess_stats_blocks(md6, n_sr_blocks)
        np.testing.assert_allclose(pb6, want6, rtol=1e-11, atol=1e-300)
        assert np.array_equal(pb6[:, :, 0], want6[:, :, 0]) and pb6[:, 3:, 0].sum() > 0
        lo6 = np.where(q6[2] > 0, q6[0], np.nan)
        c6, m6 = engs[0].sr_tail_extract(lo6)
        rc6, rm6 = ref6.sr_tail_extract(lo6)
        assert np.array_equal(c6, rc6) and np.array_equal(np.sort(m6), np.sort(rm6))
        g6 = engs[1].sr_quantiles_merge(0.95, [c6], [m6], q6[2])
        assert g6[2] == 0 and np.array_equal(g6[0], q6[0], equal_nan=True) and np.array_equal(g6[1], q6[1], equal_nan=True)
        engs[0].set_snp_meta(r, uqe, POS, paint, g)

        # (ii) the model over ranks
        def run(world):
            from ldweaver_amd.dist import deal_blocks
            mine_of = deal_blocks(blocks, world)
            tg = dist_srp.ThreadGroup(world)
            res, errs = [None] * world, [None] * world

            def work(rk):
                try:
                    e = engs[rk]
                    e.reset_speculation()
                    mine = np.asarray(mine_of[rk], dtype=np.int64)
                    if len(mine):
                        e.mi_all_pairs(blocks[mine], **kw)
                    else:
                        e.links_begin(1)
                        e.links_end()
                    res[rk] = dist_srp.merge_n_sort_sr_links_dist(e, nclust, 20000.0, cut, POS, paint, g, mine, n_sr_blocks, run_aracne=True,
                                                                  order_links=True, comm=tg.comm(rk))
                except BaseException as ex:   # noqa: BLE001
                    errs[rk] = ex
                    tg.barrier.abort()

            th = [threading.Thread(target=work, args=(rk,)) for rk in range(world)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for ex in errs:
                if ex is not None and not isinstance(ex, threading.BrokenBarrierError):
                    raise ex
            return res

        one = run(1)[0]
        assert len(one[0]["MI"]) > 500 and 0 < one[1].sum() < len(one[1])
        for world in (2, 3):
            got = run(world)
            _same(one, got[0])
            sent = [sum(x[2]["bytes_sent"].values()) for x in got]
            rows = [x[2]["local_rows"] for x in got]
            cand = [x[2]["candidates"] for x in got]
            for rk in range(1, world):
                # what a rank sent against its MI column alone (the r04 gather's 8 B per row).  At this small shape a (cluster, len) group has ~10
                # rows per rank, so "the rows from the smallest local 95 % order statistic up" are a third of them; at C4's shape (375 rows per
                # group and rank) they are ~7 % (bench.py --sr-tail dist: docs/HISTORY.md 7b)
                # (the per-group tables — counts, bounds: 4 x 480 KB — are most of what travels here; they do not grow with the table)
                assert cand[rk] < 0.6 * rows[rk] and sent[rk] < 8 * cand[rk] + 3_000_000, (world, rk, sent, rows, cand)

        # (iii) against the one-table device model on one engine
        engs[0].reset_speculation()
        engs[0].mi_all_pairs(blocks, **kw)
        redd, flags, aux = srp_host.merge_n_sort_sr_links_device(engs[0], nclust, 20000.0, cut, POS, paint, g, run_aracne=True, order_links=True)
        for k in ("a", "b", "MI", "clust_c", "row"):
            assert np.array_equal(redd[k], one[0][k]), k
        np.testing.assert_allclose(one[0]["srp_max"], redd["srp_max"], rtol=1e-9)
        assert np.array_equal(flags, one[1])
        assert np.array_equal(aux["mean_dist"], one[2]["mean_dist"], equal_nan=True)     # (order statistics: exact)
        np.testing.assert_allclose(one[2]["shape"], aux["shape"], rtol=1e-10)
        # ... and with its sums taken per reference block too (what perform_MI_computation does on every route): the same bits
        redd, flags, aux = srp_host.merge_n_sort_sr_links_device(engs[0], nclust, 20000.0, cut, POS, paint, g, run_aracne=True, order_links=True,
                                                                 block_rows=n_sr_blocks)
        for k in ("a", "b", "MI", "clust_c", "row", "srp_max", "first_clust", "dup"):
            assert np.array_equal(redd[k], one[0][k]), k
        assert np.array_equal(flags, one[1]) and np.array_equal(aux["shape"], one[2]["shape"]) and np.array_equal(aux["stats"], one[2]["stats"])

        # (v) the same inside the library: ldw_mi_all_pairs_multi(.., LDW_MI_SR_ROWS_STAY) leaves every context its own short-range rows and
        # ldw_sr_len_quantiles_multi / _excess_stats_multi / _pvalues_multi run the protocol over the contexts (worker threads, host-staged
        # exchanges) behind the one-table signatures — what a single-process host (R) calls
        from ldweaver_amd.engine import EngineGroup
        lr_one = engs[0].links(1)
        for n in (1, 2, 3):
            for e in engs:
                e.reset_speculation()
            info = Engine.mi_all_pairs_multi(engs[:n], blocks, sr_rows_stay=True, **kw)
            for u, v in zip(lr_one, engs[0].links(1)):
                assert np.array_equal(u, v), n
            assert np.array_equal(engs[0].block_stats()["n_sr"], n_sr_blocks)
            if n > 1:
                assert engs[0].links_count(0) == int(n_sr_blocks[info["owner"] == 0].sum()) < int(n_sr_blocks.sum())   # its own share only
            grp = EngineGroup(engs[:n])
            redd, flags, aux = srp_host.merge_n_sort_sr_links_device(grp, nclust, 20000.0, cut, POS, paint, g, run_aracne=True, order_links=True)
            for k in ("a", "b", "MI", "clust_c", "srp_max", "first_clust", "dup"):
                assert np.array_equal(redd[k], one[0][k]), (n, k)
            assert np.array_equal(flags, one[1]) and np.array_equal(aux["shape"], one[2]["shape"]) and np.array_equal(aux["stats"], one[2]["stats"]), n
            assert np.array_equal(aux["mean_dist"], one[2]["mean_dist"], equal_nan=True)
            pa, pb, pm = grp.sr_pool()
            assert len(pm) == one[2]["n_pool"]
        # a group that does not belong to the last multi call is refused
        Engine.mi_all_pairs_multi(engs[:2], blocks, sr_rows_stay=True, **kw)
        with pytest.raises(Exception, match="short-range rows|contexts given"):
            EngineGroup([engs[0], engs[2]]).sr_len_quantiles(nclust, 20000.0)
        # perform_MI_computation(engines=[...], sr_tail="dist") == the single-engine job, both files byte for byte
        from ldweaver_amd.snpdat import CdsVar, SnpDat
        sd = SnpDat(states=syn["states"], POS=POS, g=g, uqe=uqe, r=r)
        cv = CdsVar(paint=paint, nclust=nclust)
        outs = {}
        for tag, kws in (("one", dict(engine=engs[0])), ("three", dict(engines=engs, sr_tail="dist"))):
            d = tmp_path / tag
            d.mkdir()
            outs[tag] = MIH.perform_MI_computation(sd, hdw, cv, lr_save_path=str(d / "lr.tsv"), sr_save_path=str(d / "sr.tsv"), plt_folder=str(d / "P"),
                                                   max_blk_sz=B, lr_retain_links=2e5, verbose=False, alignment_resident=True, srp_cutoff=cut, **kws)
        assert (tmp_path / "one" / "lr.tsv").read_bytes() == (tmp_path / "three" / "lr.tsv").read_bytes()
        assert (tmp_path / "one" / "sr.tsv").read_bytes() == (tmp_path / "three" / "sr.tsv").read_bytes() and len(outs["one"]) == len(outs["three"]) > 500
    finally:
        for e in engs:
            e.close()
