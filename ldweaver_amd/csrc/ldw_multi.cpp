// In-process multi-GPU layer below the C ABI (r05; SURVEY.md 8(b)(5): "mi_all_pairs(... block list, lr prob, device list -> sr table,
// lr table)", 8(e)).  The reference's loop over block pairs (R/computePairwiseMI.R:103-116) is serial; its block pairs are independent
// given the replicated state matrix and the long-range filter is per block (:352-358), so they are dealt over the caller's contexts —
// one per GPU, all holding the same alignment, weights and SNP meta data — each context runs its share on a worker thread of this
// process, and the link tables are assembled in ctx[0] in make_blocks order by peer-to-peer copies over xGMI (no collective: the only
// exchange is this one variable-length gather, as in ldweaver_amd/dist.py, which does the same across processes with RCCL).
// A host that cannot start one process per GPU — R through .Call is the case this is for — reaches all GPUs of a node this way.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "ldw_internal.h"

namespace {

// relative GPU time of a block pair (ldweaver_amd/dist.py::block_cost, measured r03: a diagonal pair — half the pairs, but the dense
// short-range band with its exact GEMM and whole-unit fp64 kernel — costs 1.65 off-diagonal ones, i.e. 3.3 times its own pair count)
inline int64_t block_cost(const int32_t *b) {
    const int64_t nf = (int64_t)b[1] - b[0] + 1, nt = (int64_t)b[3] - b[2] + 1;
    const bool diag = b[0] == b[2] && b[1] == b[3];
    return diag ? (int64_t)((double)(nf * (nf - 1) / 2) * 3.3) : nf * nt;
}

struct Worker {
    std::vector<int32_t> blocks;   // this context's share, make_blocks order
    std::vector<int64_t> ids;      // their indices in the caller's list
    int rc = LDW_OK;
    std::string err;
    double ms = 0;
};

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// dst (device of c0) <- src (device of ck); the copy is queued on the SOURCE context's stream: it is ordered behind that context's own
// kernels without an event, and the sources' copies run concurrently, each over its own xGMI link to the destination GPU
inline hipError_t copy_rows(void *dst, const ldw_ctx *c0, const void *src, const ldw_ctx *ck, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if (ck->device == c0->device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ck->stream);
    return hipMemcpyPeerAsync(dst, c0->device, src, ck->device, bytes, ck->stream);
}

}  // namespace

extern "C" {

int ldw_deal_blocks(const int32_t *blocks, int64_t nblocks, int n_ranks, int32_t *owner_out) {
    LDW_REQUIRE(blocks && owner_out && nblocks > 0 && n_ranks >= 1, LDW_ERR_ARG, "ldw_deal_blocks: bad argument");
    std::vector<int64_t> cost((size_t)nblocks), order((size_t)nblocks), load((size_t)n_ranks, 0);
    for (int64_t b = 0; b < nblocks; ++b) {
        const int32_t *q = blocks + b * 4;
        LDW_REQUIRE(q[0] >= 1 && q[1] >= q[0] && q[2] >= 1 && q[3] >= q[2], LDW_ERR_ARG, "ldw_deal_blocks: block %lld = (%d,%d,%d,%d) is not a 1-based inclusive range pair",
                    (long long)b, q[0], q[1], q[2], q[3]);
        cost[(size_t)b] = block_cost(q);
    }
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return cost[(size_t)x] > cost[(size_t)y]; });   // longest processing time first
    for (int64_t bi : order) {
        const int rk = (int)(std::min_element(load.begin(), load.end()) - load.begin());   // (first of the least loaded: numpy's argmin)
        owner_out[bi] = rk;
        load[(size_t)rk] += cost[(size_t)bi];
    }
    return LDW_OK;
}

int ldw_mi_all_pairs_multi(ldw_ctx **ctx, int n_ctx, const int32_t *blocks, int64_t nblocks, const ldw_mi_params *p, int32_t *owner_out, double *ms_out) {
    if (ctx && n_ctx >= 1 && ctx[0]) ctx[0]->multi_owner.clear();   // whatever this call returns, the deal of an EARLIER call no longer describes ctx[0]
    LDW_REQUIRE(ctx && n_ctx >= 1 && n_ctx <= 64 && blocks && p && nblocks > 0, LDW_ERR_ARG, "ldw_mi_all_pairs_multi: bad argument");
    for (int k = 0; k < n_ctx; ++k) {
        LDW_REQUIRE(ctx[k], LDW_ERR_ARG, "ldw_mi_all_pairs_multi: context %d is null", k);
        for (int j = 0; j < k; ++j) LDW_REQUIRE(ctx[j] != ctx[k], LDW_ERR_ARG, "ldw_mi_all_pairs_multi: contexts %d and %d are the same", j, k);
        LDW_REQUIRE(ctx[k]->L == ctx[0]->L && ctx[k]->N == ctx[0]->N && ctx[k]->L > 0, LDW_ERR_STATE,
                    "ldw_mi_all_pairs_multi: context %d holds a %lld x %lld alignment, context 0 %lld x %lld (every context needs the same alignment, weights and SNP meta data)",
                    k, (long long)ctx[k]->L, (long long)ctx[k]->N, (long long)ctx[0]->L, (long long)ctx[0]->N);
        LDW_REQUIRE(ctx[k]->have_weights && ctx[k]->have_meta, LDW_ERR_STATE, "ldw_mi_all_pairs_multi: context %d has no weights or SNP meta data", k);
        LDW_REQUIRE(ctx[k]->neff == ctx[0]->neff && ctx[k]->g == ctx[0]->g && ctx[k]->h_POS == ctx[0]->h_POS, LDW_ERR_STATE,
                    "ldw_mi_all_pairs_multi: context %d differs from context 0 in its weights, genome length or positions", k);
    }
    const double t_0 = now_ms();
    ldw_ctx *c0 = ctx[0];
    // ---- 1) the deal: cost-weighted, every context keeps make_blocks order
    std::vector<int32_t> owner((size_t)nblocks);
    if (int rc = ldw_deal_blocks(blocks, nblocks, n_ctx, owner.data())) return rc;
    if (owner_out) memcpy(owner_out, owner.data(), (size_t)nblocks * 4);
    std::vector<Worker> W((size_t)n_ctx);
    for (int64_t b = 0; b < nblocks; ++b) {
        Worker &w = W[(size_t)owner[(size_t)b]];
        w.blocks.insert(w.blocks.end(), blocks + b * 4, blocks + b * 4 + 4);
        w.ids.push_back(b);
    }
    // ---- 2) every context runs its share on a thread of its own (a fresh std::thread each: nothing is re-executed, no process is started)
    auto run = [&](int k) {
        Worker &w = W[(size_t)k];
        const double t0 = now_ms();
        if (hipSetDevice(ctx[k]->device) != hipSuccess) {
            w.rc = LDW_ERR_HIP;
            w.err = "hipSetDevice failed";
            return;
        }
        if (!w.ids.empty()) {
            w.rc = ldw_mi_all_pairs(ctx[k], w.blocks.data(), (int64_t)w.ids.size(), p, 1);
            if (w.rc != LDW_OK) w.err = ldw_last_error();   // (the message is thread-local: taken here, reported by the caller's thread)
        } else {
            // a context the deal left without a block holds EMPTY tables afterwards, not those of an earlier call (tools/fuzz_sr_model.py: one block pair, two contexts —
            // the idle one still held its rows of the problem before, and the model over the contexts refused its "share")
            w.rc = ldw_links_begin(ctx[k], 1);
            if (w.rc == LDW_OK) w.rc = ldw_links_end(ctx[k]);
            if (w.rc != LDW_OK) w.err = ldw_last_error();
        }
        w.ms = now_ms() - t0;
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < n_ctx; ++k) th.emplace_back(run, k);
        run(0);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(c0->device);
    // ---- 3) error agreement: one context that failed fails the call (the others' tables are left as they are)
    for (int k = 0; k < n_ctx; ++k)
        if (W[(size_t)k].rc != LDW_OK) {
            ldw::set_error("ldw_mi_all_pairs_multi: context %d (device %d, %lld blocks): %s", k, ctx[k]->device, (long long)W[(size_t)k].ids.size(), W[(size_t)k].err.c_str());
            return W[(size_t)k].rc;
        }
    const double t_1 = now_ms();
    if (ms_out) {
        ms_out[0] = t_1 - t_0;   // deal + the slowest context's pass
        ms_out[1] = 0;           // the gather (below)
        for (int k = 0; k < n_ctx && k < 8; ++k) ms_out[2 + k] = W[(size_t)k].ms;
    }
    c0->multi_owner.clear();
    if (n_ctx == 1) return LDW_OK;
    const bool rows_stay = (p->flags & LDW_MI_SR_ROWS_STAY) != 0 && p->keep_sr;   // r05: only the long-range table is assembled; (7c) runs the model over the contexts
    // ---- 4) the gather: rows of block b lie at the running offset of its owner's table; their place in ctx[0]'s assembled table is the
    //         running offset over ALL blocks in the caller's (make_blocks) order
    std::vector<ldw::BlockStat> stats((size_t)nblocks);
    for (int k = 0; k < n_ctx; ++k) {
        const Worker &w = W[(size_t)k];
        LDW_REQUIRE(w.ids.empty() || ctx[k]->stats.size() == w.ids.size(), LDW_ERR_STATE, "ldw_mi_all_pairs_multi: context %d reports %lld blocks, ran %lld", k,
                    (long long)ctx[k]->stats.size(), (long long)w.ids.size());
        for (size_t i = 0; i < w.ids.size(); ++i) stats[(size_t)w.ids[i]] = ctx[k]->stats[i];
    }
    int64_t tot[2] = {0, 0};
    for (int64_t b = 0; b < nblocks; ++b) {
        tot[0] += stats[(size_t)b].n_sr;
        tot[1] += stats[(size_t)b].n_lr_kept;
    }
    for (int k = 0; k < n_ctx; ++k) {
        int64_t s = 0, l = 0;
        for (int64_t id : W[(size_t)k].ids) {
            s += stats[(size_t)id].n_sr;
            l += stats[(size_t)id].n_lr_kept;
        }
        if (W[(size_t)k].ids.empty()) continue;
        LDW_REQUIRE(s == ctx[k]->n_sr && l == ctx[k]->n_lr, LDW_ERR_STATE, "ldw_mi_all_pairs_multi: context %d holds %lld / %lld rows, its blocks report %lld / %lld", k,
                    (long long)ctx[k]->n_sr, (long long)ctx[k]->n_lr, (long long)s, (long long)l);
    }
    // peer access towards GPU 0 (a context on another device writes into ctx[0]'s memory); already-enabled is fine, unavailable falls back to
    // the runtime's staged copy
    for (int k = 1; k < n_ctx; ++k) {
        if (ctx[k]->device == c0->device) continue;
        int can = 0;
        if (hipSetDevice(ctx[k]->device) == hipSuccess && hipDeviceCanAccessPeer(&can, ctx[k]->device, c0->device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(c0->device, 0);
            if (e != hipSuccess) (void)hipGetLastError();
        }
    }
    LDW_HIP(hipSetDevice(c0->device));
    // The short-range rows are a pure function of the positions and the block geometry (R/computePairwiseMI.R:306-333), so their index columns
    // need not travel: ctx[0] rebuilds (a, b) with the band enumerator (ldw_sr_pairs_fill) and only the MI column (8 of 16 bytes per row) crosses
    // xGMI — what the RCCL gather of dist.py does.  The DEFAULT since the enumerator builds its intervals on the device (r05: 0.8 ms for C4's
    // 9e7 rows; its host loop took 12-19 ms, which made this the option, LDW_MULTI_SR_MI_ONLY, earlier in the round); LDW_MULTI_SR_FULL_ROWS=1 sends all
    // three columns (also what unsorted positions, a fractional genome length and SR-only passes do).
    const bool sr_mi_only = !rows_stay && c0->pos_sorted && c0->g == std::floor(c0->g) && !p->sr_only && p->keep_sr && getenv("LDW_MULTI_SR_FULL_ROWS") == nullptr;
    ldw::DevBuf nA[2], nB[2], nM[2];
    auto fail = [&](int rc) {
        for (int w = 0; w < 2; ++w) {
            nA[w].release();
            nB[w].release();
            nM[w].release();
        }
        return rc;
    };
    for (int w = rows_stay ? 1 : 0; w < 2; ++w) {
        const size_t n = (size_t)std::max<int64_t>(tot[w], 1);
        if (int rc = nA[w].reserve(n * 4)) return fail(rc);
        if (int rc = nB[w].reserve(n * 4)) return fail(rc);
        if (int rc = nM[w].reserve(n * 8)) return fail(rc);
    }
    if (hipStreamSynchronize(c0->stream) != hipSuccess) return fail(LDW_ERR_HIP);   // (the fresh buffers are visible to every queue before a peer writes into them)
    std::vector<int64_t> src_off((size_t)n_ctx * 2, 0);
    int64_t dst_off[2] = {0, 0};
    hipError_t he = hipSuccess;
    for (int64_t b = 0; b < nblocks && he == hipSuccess;) {
        // consecutive blocks of one owner are consecutive in its table too: one copy per run and column
        const int k = owner[(size_t)b];
        int64_t rows[2] = {0, 0}, e = b;
        while (e < nblocks && owner[(size_t)e] == k) {
            rows[0] += stats[(size_t)e].n_sr;
            rows[1] += stats[(size_t)e].n_lr_kept;
            ++e;
        }
        const ldw_ctx *ck = ctx[k];
        if (hipSetDevice(ck->device) != hipSuccess) {
            he = hipErrorInvalidDevice;
            break;
        }
        for (int w = rows_stay ? 1 : 0; w < 2 && he == hipSuccess; ++w) {
            const int64_t so = src_off[(size_t)k * 2 + w], d = dst_off[w], n = rows[w];
            const ldw::DevBuf &sa = w == 0 ? ck->sr_a : ck->lr_a, &sb = w == 0 ? ck->sr_b : ck->lr_b, &sm = w == 0 ? ck->sr_mi : ck->lr_mi;
            if (!(w == 0 && sr_mi_only)) {
                he = copy_rows(nA[w].as<int32_t>() + d, c0, sa.as<int32_t>() + so, ck, (size_t)n * 4);
                if (he == hipSuccess) he = copy_rows(nB[w].as<int32_t>() + d, c0, sb.as<int32_t>() + so, ck, (size_t)n * 4);
            }
            if (he == hipSuccess) he = copy_rows(nM[w].as<double>() + d, c0, sm.as<double>() + so, ck, (size_t)n * 8);
            src_off[(size_t)k * 2 + w] += n;
            dst_off[w] += n;
        }
        b = e;
    }
    (void)hipSetDevice(c0->device);
    int rc = he == hipSuccess ? LDW_OK : ldw::hip_fail(he, "peer copy of link rows", __FILE__, __LINE__);
    if (rc == LDW_OK && sr_mi_only && tot[0] > 0) {
        int64_t n_fill = 0;
        rc = ldw_sr_pairs_fill(c0, blocks, nblocks, p->sr_dist, nA[0].as<int32_t>(), nB[0].as<int32_t>(), tot[0], &n_fill);
        if (rc == LDW_OK && n_fill != tot[0]) {
            ldw::set_error("ldw_mi_all_pairs_multi: the band enumerator gives %lld short-range rows, the contexts' passes %lld", (long long)n_fill, (long long)tot[0]);
            rc = LDW_ERR_STATE;
        }
    }
    for (int k = 0; k < n_ctx; ++k) {   // every source's copies have landed
        if (hipSetDevice(ctx[k]->device) != hipSuccess || hipStreamSynchronize(ctx[k]->stream) != hipSuccess) {
            if (rc == LDW_OK) rc = ldw::hip_fail(hipGetLastError(), "waiting for the peer copies", __FILE__, __LINE__);
        }
    }
    (void)hipSetDevice(c0->device);
    if (rc != LDW_OK) return fail(rc);
    // ---- 5) ctx[0] adopts the assembled tables (its own share's buffers are released) and the per-block records of ALL blocks
    if (!rows_stay) {
        std::swap(c0->sr_a, nA[0]);
        std::swap(c0->sr_b, nB[0]);
        std::swap(c0->sr_mi, nM[0]);
    }
    std::swap(c0->lr_a, nA[1]);
    std::swap(c0->lr_b, nB[1]);
    std::swap(c0->lr_mi, nM[1]);
    fail(LDW_OK);
    if (!rows_stay) c0->n_sr = tot[0];   // (rows_stay: ctx[0] keeps the short-range rows of its own share, like every other context)
    c0->n_lr = tot[1];
    c0->n_red = c0->n_pool = 0;
    c0->stats = stats;                    // the records of ALL blocks, in the caller's order (ldw_block_stats)
    if (rows_stay) c0->multi_owner = owner;
    if (ms_out) ms_out[1] = now_ms() - t_1;
    return LDW_OK;
}

int ldw_hamming_weights_multi(ldw_ctx **ctx, int n_ctx, int32_t thresh, double *hdw_out) {
    LDW_REQUIRE(ctx && n_ctx >= 1 && n_ctx <= 64 && hdw_out, LDW_ERR_ARG, "ldw_hamming_weights_multi: bad argument");
    for (int k = 0; k < n_ctx; ++k)
        LDW_REQUIRE(ctx[k] && ctx[k]->L == ctx[0]->L && ctx[k]->N == ctx[0]->N && ctx[k]->N > 0, LDW_ERR_STATE, "ldw_hamming_weights_multi: context %d holds another alignment (or none)", k);
    const int64_t N = ctx[0]->N;
    // strips of 128-sequence row tiles of the symmetric comparison, cut at equal area of the triangle (dist.py::hamming_tile_strips)
    const int64_t ntiles = (N + 127) / 128;
    std::vector<double> cum((size_t)ntiles + 1, 0.0);
    for (int64_t t = 0; t < ntiles; ++t) cum[(size_t)t + 1] = cum[(size_t)t] + (double)(ntiles - t);
    std::vector<int64_t> cuts((size_t)n_ctx + 1);
    for (int k = 0; k <= n_ctx; ++k) cuts[(size_t)k] = std::lower_bound(cum.begin(), cum.end(), cum.back() * (double)k / (double)n_ctx) - cum.begin();
    cuts[0] = 0;
    cuts[(size_t)n_ctx] = ntiles;
    std::vector<std::vector<int64_t>> cnt((size_t)n_ctx, std::vector<int64_t>((size_t)N, 0));
    std::vector<int> rcs((size_t)n_ctx, LDW_OK);
    std::vector<std::string> errs((size_t)n_ctx);
    auto run = [&](int k) {
        const int64_t t0 = cuts[(size_t)k], t1 = std::max(cuts[(size_t)k], cuts[(size_t)k + 1]);
        if (t1 <= t0) return;
        if (hipSetDevice(ctx[k]->device) != hipSuccess) {
            rcs[(size_t)k] = LDW_ERR_HIP;
            errs[(size_t)k] = "hipSetDevice failed";
            return;
        }
        rcs[(size_t)k] = ldw_hamming_counts(ctx[k], thresh, (int32_t)t0, (int32_t)t1, cnt[(size_t)k].data());
        if (rcs[(size_t)k] != LDW_OK) errs[(size_t)k] = ldw_last_error();
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < n_ctx; ++k) th.emplace_back(run, k);
        run(0);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(ctx[0]->device);
    for (int k = 0; k < n_ctx; ++k)
        if (rcs[(size_t)k] != LDW_OK) {
            ldw::set_error("ldw_hamming_weights_multi: context %d (device %d): %s", k, ctx[k]->device, errs[(size_t)k].c_str());
            return rcs[(size_t)k];
        }
    for (int64_t j = 0; j < N; ++j) {   // integers in, so the weights do not depend on how many contexts shared the work
        int64_t n = 0;
        for (int k = 0; k < n_ctx; ++k) n += cnt[(size_t)k][(size_t)j];
        hdw_out[j] = 1.0 / ((double)n + 1.0);
    }
    return LDW_OK;
}

}  // extern "C"

// ---- (7c) the short-range model over the contexts of this process (r05) ---------------------------------------------------------------------
namespace {

// run f(k) for every context on a thread of its own (context 0 on the caller's); the first failure (lowest k) becomes the caller's error
template <class F>
int for_each_ctx(ldw_ctx **ctx, int n_ctx, const char *who, F f) {
    std::vector<int> rcs((size_t)n_ctx, LDW_OK);
    std::vector<std::string> errs((size_t)n_ctx);
    auto run = [&](int k) {
        if (hipSetDevice(ctx[k]->device) != hipSuccess) {
            rcs[(size_t)k] = LDW_ERR_HIP;
            errs[(size_t)k] = "hipSetDevice failed";
            return;
        }
        try {
            rcs[(size_t)k] = f(k);
            if (rcs[(size_t)k] != LDW_OK) errs[(size_t)k] = ldw_last_error();
        } catch (const std::exception &e) {
            rcs[(size_t)k] = LDW_ERR_STATE;
            errs[(size_t)k] = e.what();
        }
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < n_ctx; ++k) th.emplace_back(run, k);
        run(0);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(ctx[0]->device);
    for (int k = 0; k < n_ctx; ++k)
        if (rcs[(size_t)k] != LDW_OK) {
            ldw::set_error("%s: context %d (device %d): %s", who, k, ctx[k]->device, errs[(size_t)k].c_str());
            return rcs[(size_t)k];
        }
    return LDW_OK;
}

// the contexts hold the shares ldw_mi_all_pairs_multi(.., LDW_MI_SR_ROWS_STAY) left?  mine[k] = the blocks of context k in its table's order, rows[k] their row counts
int shares_of(ldw_ctx **ctx, int n_ctx, const char *who, bool &spread, std::vector<std::vector<int64_t>> &mine, std::vector<std::vector<int64_t>> &rows) {
    LDW_REQUIRE(ctx && n_ctx >= 1 && n_ctx <= 64 && ctx[0], LDW_ERR_ARG, "%s: bad argument", who);
    for (int k = 0; k < n_ctx; ++k) LDW_REQUIRE(ctx[k], LDW_ERR_ARG, "%s: context %d is null", who, k);
    const ldw_ctx *c0 = ctx[0];
    spread = n_ctx > 1 && !c0->multi_owner.empty();
    mine.assign((size_t)n_ctx, {});
    rows.assign((size_t)n_ctx, {});
    if (!spread) return LDW_OK;
    LDW_REQUIRE(c0->multi_owner.size() == c0->stats.size(), LDW_ERR_STATE, "%s: context 0 is not the context 0 of the last ldw_mi_all_pairs_multi call", who);
    for (size_t b = 0; b < c0->multi_owner.size(); ++b) {
        const int k = c0->multi_owner[b];
        LDW_REQUIRE(k >= 0 && k < n_ctx, LDW_ERR_STATE, "%s: block %lld belongs to context %d, %d contexts given", who, (long long)b, k, n_ctx);
        mine[(size_t)k].push_back((int64_t)b);
        rows[(size_t)k].push_back(c0->stats[b].n_sr);
    }
    for (int k = 0; k < n_ctx; ++k) {
        int64_t s = 0;
        for (int64_t r : rows[(size_t)k]) s += r;
        LDW_REQUIRE(s == ctx[k]->n_sr, LDW_ERR_STATE, "%s: context %d holds %lld short-range rows, its blocks of the last ldw_mi_all_pairs_multi call %lld", who, k,
                    (long long)ctx[k]->n_sr, (long long)s);
    }
    return LDW_OK;
}

}  // namespace

extern "C" {

int ldw_sr_len_quantiles_multi(ldw_ctx **ctx, int n_ctx, int nclust, double sr_dist, double prob, int32_t S, double *q_lo_out, double *q_hi_out, int64_t *n_out) {
    bool spread = false;
    std::vector<std::vector<int64_t>> mine, rows;
    if (int rc = shares_of(ctx, n_ctx, "ldw_sr_len_quantiles_multi", spread, mine, rows)) return rc;
    if (!spread) return ldw_sr_len_quantiles(ctx[0], nclust, sr_dist, prob, S, q_lo_out, q_hi_out, n_out);
    LDW_REQUIRE(q_lo_out && q_hi_out && n_out && nclust >= 1 && S >= 1, LDW_ERR_ARG, "ldw_sr_len_quantiles_multi: bad argument");
    const size_t G = (size_t)nclust * (size_t)S;
    // 1) every context's own counts and order statistics
    std::vector<std::vector<double>> qlo((size_t)n_ctx, std::vector<double>(G)), qhi((size_t)n_ctx, std::vector<double>(G));
    std::vector<std::vector<int64_t>> cnt((size_t)n_ctx, std::vector<int64_t>(G));
    if (int rc = for_each_ctx(ctx, n_ctx, "ldw_sr_len_quantiles_multi", [&](int k) {
            return ldw_sr_len_quantiles(ctx[k], nclust, sr_dist, prob, S, qlo[(size_t)k].data(), qhi[(size_t)k].data(), cnt[(size_t)k].data());
        }))
        return rc;
    // 2) the bound: the smallest local lower statistic of a group lies at or below the group's own (docs/HISTORY.md 7b)
    std::vector<double> lower(G, std::nan(""));
    std::vector<int64_t> total(G, 0);
    for (size_t i = 0; i < G; ++i)
        for (int k = 0; k < n_ctx; ++k) {
            total[i] += cnt[(size_t)k][i];
            if (cnt[(size_t)k][i] > 0 && !(lower[i] <= qlo[(size_t)k][i])) lower[i] = qlo[(size_t)k][i];   // (NaN-safe minimum)
        }
    // 3) every context's rows at or above it (values only, grouped len-major): extracted into a buffer on ITS device, then device to device into one buffer on
    //    context 0's (peer copies queued on the source's stream, as the link-table gather does) — at config 5 the candidates are 1.4 GB: through pageable host
    //    vectors (the first version) that was most of the call
    std::vector<std::vector<int64_t>> tcnt((size_t)n_ctx, std::vector<int64_t>(G));
    std::vector<int64_t> tn((size_t)n_ctx, 0);
    if (int rc = for_each_ctx(ctx, n_ctx, "ldw_sr_len_quantiles_multi", [&](int k) {
            int64_t n = 0;
            if (int r = ldw_sr_tail_extract(ctx[k], nclust, S, lower.data(), tcnt[(size_t)k].data(), nullptr, 0, 0, &n)) return r;
            tn[(size_t)k] = n;
            if (n == 0) return (int)LDW_OK;
            if (int r = ctx[k]->srd_out.reserve((size_t)n * 8)) return r;
            return ldw_sr_tail_extract(ctx[k], nclust, S, lower.data(), tcnt[(size_t)k].data(), ctx[k]->srd_out.as<double>(), n, 1, &n);
        }))
        return rc;
    ldw_ctx *c0 = ctx[0];
    int64_t others = 0;
    for (int k = 1; k < n_ctx; ++k) others += tn[(size_t)k];
    ldw::DevBuf stage;   // the other contexts' candidates on context 0's device
    struct Rel {
        ldw::DevBuf &b;
        ~Rel() { b.release(); }
    } rel{stage};
    LDW_HIP(hipSetDevice(c0->device));
    if (int rc = stage.reserve((size_t)std::max<int64_t>(others, 1) * 8)) return rc;
    LDW_HIP(hipStreamSynchronize(c0->stream));   // (the fresh buffer is visible to every queue before a peer writes into it)
    std::vector<const double *> pm((size_t)n_ctx);
    std::vector<const int64_t *> pc((size_t)n_ctx);
    {
        int64_t off = 0;
        hipError_t he = hipSuccess;
        pm[0] = c0->srd_out.as<double>();
        pc[0] = tcnt[0].data();
        for (int k = 1; k < n_ctx && he == hipSuccess; ++k) {
            pm[(size_t)k] = stage.as<double>() + off;
            pc[(size_t)k] = tcnt[(size_t)k].data();
            if (tn[(size_t)k] == 0) continue;
            if (hipSetDevice(ctx[k]->device) != hipSuccess) {
                he = hipErrorInvalidDevice;
                break;
            }
            he = copy_rows(stage.as<double>() + off, c0, ctx[k]->srd_out.p, ctx[k], (size_t)tn[(size_t)k] * 8);
            off += tn[(size_t)k];
        }
        for (int k = 1; k < n_ctx; ++k)   // every source's copy has landed
            if (hipSetDevice(ctx[k]->device) != hipSuccess || hipStreamSynchronize(ctx[k]->stream) != hipSuccess) {
                if (he == hipSuccess) he = hipGetLastError();
            }
        (void)hipSetDevice(c0->device);
        if (he != hipSuccess) return ldw::hip_fail(he, "peer copy of the quantile candidates", __FILE__, __LINE__);
    }
    // 4) context 0 selects by rank from the top
    if (int rc = ldw_sr_quantiles_merge(c0, nclust, S, prob, n_ctx, pm.data(), pc.data(), total.data(), 1, q_lo_out, q_hi_out, nullptr)) return rc;
    memcpy(n_out, total.data(), G * 8);
    return LDW_OK;
}

int ldw_sr_excess_stats_multi(ldw_ctx **ctx, int n_ctx, int nclust, int32_t S, const double *mean_dist, double *stats_out) {
    bool spread = false;
    std::vector<std::vector<int64_t>> mine, rows;
    if (int rc = shares_of(ctx, n_ctx, "ldw_sr_excess_stats_multi", spread, mine, rows)) return rc;
    LDW_REQUIRE(stats_out && nclust >= 1, LDW_ERR_ARG, "ldw_sr_excess_stats_multi: bad argument");
    const size_t per = (size_t)nclust * 5;
    if (!spread) {
        // one table: per block when its block structure is known (the rows of the last pass / gather), else strips of the whole table
        ldw_ctx *c = ctx[0];
        int64_t s = 0;
        std::vector<int64_t> r;
        for (const auto &st : c->stats) {
            r.push_back(st.n_sr);
            s += st.n_sr;
        }
        if (r.empty() || s != c->n_sr) return ldw_sr_excess_stats(c, nclust, S, mean_dist, stats_out);
        std::vector<double> part(r.size() * per);
        if (int rc = ldw_sr_excess_stats_blocks(c, nclust, S, mean_dist, (int64_t)r.size(), r.data(), part.data())) return rc;
        for (size_t t = 0; t < per; ++t) stats_out[t] = 0.0;
        for (size_t b = 0; b < r.size(); ++b)
            for (size_t t = 0; t < per; ++t) stats_out[t] += part[b * per + t];
        return LDW_OK;
    }
    const size_t nb = ctx[0]->stats.size();
    std::vector<std::vector<double>> part((size_t)n_ctx);
    if (int rc = for_each_ctx(ctx, n_ctx, "ldw_sr_excess_stats_multi", [&](int k) {
            part[(size_t)k].assign(std::max<size_t>(rows[(size_t)k].size(), 1) * per, 0.0);
            return ldw_sr_excess_stats_blocks(ctx[k], nclust, S, mean_dist, (int64_t)rows[(size_t)k].size(), rows[(size_t)k].data(), part[(size_t)k].data());
        }))
        return rc;
    std::vector<const double *> of(nb, nullptr);
    for (int k = 0; k < n_ctx; ++k)
        for (size_t i = 0; i < mine[(size_t)k].size(); ++i) of[(size_t)mine[(size_t)k][i]] = part[(size_t)k].data() + i * per;
    for (size_t t = 0; t < per; ++t) stats_out[t] = 0.0;
    for (size_t b = 0; b < nb; ++b)   // make_blocks order: the same sum for any deal
        for (size_t t = 0; t < per; ++t) stats_out[t] += of[b][t];
    return LDW_OK;
}

int ldw_sr_pvalues_multi(ldw_ctx **ctx, int n_ctx, int nclust, int32_t S, const double *mean_dist, const double *shape, double srp_cutoff, int64_t *n_red_out,
                         int64_t *n_pool_out, double *min_mi_out) {
    bool spread = false;
    std::vector<std::vector<int64_t>> mine, rows;
    if (int rc = shares_of(ctx, n_ctx, "ldw_sr_pvalues_multi", spread, mine, rows)) return rc;
    if (!spread) return ldw_sr_pvalues(ctx[0], nclust, S, mean_dist, shape, srp_cutoff, n_red_out, n_pool_out, min_mi_out);
    LDW_REQUIRE(n_red_out && n_pool_out, LDW_ERR_ARG, "ldw_sr_pvalues_multi: null output");
    // 1) p-values where the rows lie; the smallest kept MI over all contexts
    std::vector<int64_t> nr((size_t)n_ctx, 0), np((size_t)n_ctx, 0);
    std::vector<double> mn((size_t)n_ctx, std::nan(""));
    if (int rc = for_each_ctx(ctx, n_ctx, "ldw_sr_pvalues_multi",
                              [&](int k) { return ldw_sr_pvalues(ctx[k], nclust, S, mean_dist, shape, srp_cutoff, &nr[(size_t)k], nullptr, &mn[(size_t)k]); }))
        return rc;
    double gmin = std::nan("");
    for (int k = 0; k < n_ctx; ++k)
        if (nr[(size_t)k] > 0 && !(gmin <= mn[(size_t)k])) gmin = mn[(size_t)k];
    // 2) every context's pool for THAT minimum, its kept links and pool to the host
    struct Kept {
        std::vector<int64_t> row;
        std::vector<int32_t> a, b, cc, first;
        std::vector<uint8_t> dup;
        std::vector<double> mi, srp;
        std::vector<int32_t> pa, pb;
        std::vector<double> pmi;
    };
    std::vector<Kept> K((size_t)n_ctx);
    if (int rc = for_each_ctx(ctx, n_ctx, "ldw_sr_pvalues_multi", [&](int k) {
            Kept &q = K[(size_t)k];
            if (int r = ldw_sr_pool_build(ctx[k], gmin, &np[(size_t)k])) return r;
            const size_t n = (size_t)nr[(size_t)k], m = (size_t)np[(size_t)k];
            q.row.resize(n); q.a.resize(n); q.b.resize(n); q.cc.resize(n); q.first.resize(n); q.dup.resize(n); q.mi.resize(n); q.srp.resize(n);
            q.pa.resize(m); q.pb.resize(m); q.pmi.resize(m);
            if (n)
                if (int r = ldw_sr_reduced_fetch(ctx[k], (int64_t)n, q.row.data(), q.a.data(), q.b.data(), q.mi.data(), q.cc.data(), q.first.data(), q.dup.data(), q.srp.data())) return r;
            if (m)
                if (int r = ldw_sr_pool_fetch(ctx[k], (int64_t)m, q.pa.data(), q.pb.data(), q.pmi.data())) return r;
            return (int)LDW_OK;
        }))
        return rc;
    // 3) rows of a context's table -> rows of the job's table (make_blocks order); all kept links in that order
    const size_t nb = ctx[0]->stats.size();
    std::vector<int64_t> goff(nb + 1, 0);
    for (size_t b = 0; b < nb; ++b) goff[b + 1] = goff[b] + ctx[0]->stats[b].n_sr;
    int64_t n_red = 0, n_pool = 0;
    for (int k = 0; k < n_ctx; ++k) {
        n_red += nr[(size_t)k];
        n_pool += np[(size_t)k];
    }
    struct Ref {
        int64_t grow;
        int32_t k;
        int64_t i;
    };
    std::vector<Ref> order;
    order.reserve((size_t)n_red);
    for (int k = 0; k < n_ctx; ++k) {
        std::vector<int64_t> loff(rows[(size_t)k].size() + 1, 0);
        for (size_t i = 0; i < rows[(size_t)k].size(); ++i) loff[i + 1] = loff[i] + rows[(size_t)k][i];
        for (int64_t i = 0; i < nr[(size_t)k]; ++i) {
            const int64_t r = K[(size_t)k].row[(size_t)i];
            const size_t bi = (size_t)(std::upper_bound(loff.begin(), loff.end(), r) - loff.begin()) - 1;
            LDW_REQUIRE(bi < rows[(size_t)k].size(), LDW_ERR_STATE, "ldw_sr_pvalues_multi: context %d reports row %lld of %lld", k, (long long)r, (long long)loff.back());
            order.push_back(Ref{goff[(size_t)mine[(size_t)k][bi]] + (r - loff[bi]), k, i});
        }
    }
    std::sort(order.begin(), order.end(), [](const Ref &x, const Ref &y) { return x.grow < y.grow; });   // (rows are distinct)
    std::vector<int32_t> A((size_t)n_red), B((size_t)n_red), PA((size_t)n_pool), PB((size_t)n_pool);
    std::vector<double> M((size_t)n_red), SRP((size_t)n_red), PM((size_t)n_pool);
    std::vector<uint32_t> META((size_t)n_red);
    for (size_t j = 0; j < order.size(); ++j) {
        const Kept &q = K[(size_t)order[j].k];
        const size_t i = (size_t)order[j].i;
        A[j] = q.a[i];
        B[j] = q.b[i];
        M[j] = q.mi[i];
        SRP[j] = q.srp[i];
        META[j] = (uint32_t)q.cc[i] | ((uint32_t)q.first[i] << 8) | ((uint32_t)(q.dup[i] ? 1 : 0) << 16);
    }
    size_t o = 0;
    for (int k = 0; k < n_ctx; ++k) {
        const Kept &q = K[(size_t)k];
        std::copy(q.pa.begin(), q.pa.end(), PA.begin() + (std::ptrdiff_t)o);
        std::copy(q.pb.begin(), q.pb.end(), PB.begin() + (std::ptrdiff_t)o);
        std::copy(q.pmi.begin(), q.pmi.end(), PM.begin() + (std::ptrdiff_t)o);
        o += q.pmi.size();
    }
    // 4) context 0 adopts them: ldw_sr_reduced_fetch / ldw_sr_pool_fetch / ldw_aracne_device follow as after ldw_sr_pvalues
    LDW_HIP(hipSetDevice(ctx[0]->device));
    if (int rc = ldw::reduced_import_full(ctx[0], n_red, A.data(), B.data(), M.data(), META.data(), SRP.data(), n_pool, PA.data(), PB.data(), PM.data())) return rc;
    ctx[0]->multi_owner.clear();   // (its short-range table is the kept set now: the shares are gone)
    *n_red_out = n_red;
    *n_pool_out = n_pool;
    if (min_mi_out) *min_mi_out = gmin;
    return LDW_OK;
}

}  // extern "C"
