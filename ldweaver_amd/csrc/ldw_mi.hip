// MI epilogue, link selection and the block drivers (twins of perform_MI_computation_ACGTN,
// R/computePairwiseMI.R:167-386, and of the block loop of perform_MI_computation, :103-116).
//
// First block of a call sequence (no histogram-bucket guess yet), speculation misses, ldw_mi_block:
//   GEMM, 5 limbs (ldw_gemm_bits.hip) -> k_mi_epilogue_fast / _rest (ldw_mi_eval.inc): one thread per SNP pair turns the fixed-point joint sums into MI
//   (src/computeMI.cpp:19), writes the dense MI block, scatters the short-range links straight to their final rows and
//   histograms the long-range MI values in LDS -> k_pick_bucket: ranks of the two order statistics of quantile type 7
//   and the histogram bucket holding them -> k_lr_gather: every long-range pair at or above that bucket.
// Every other block (speculative: the guess is an earlier same-kind block's bucket minus an adaptive margin), DEFAULT path when the
// weights allow it (launch_block_apx, ldw_apx.hip): k_pack_panel -> k_zero4 -> k_build_packs -> gemm_apx_kernel (ONE dual-digit int8
//   pass into the int32 block G'; on off-diagonal blocks its epilogue applies the threshold table and neither stores nor flags for
//   screening the regions that pass) -> k_mi_screen<RM, true> / k_mi_screen_generic<true> (rigorous fp32 upper bound of MI from G';
//   long-range candidates go to PAIR lists, units with a short-range pair to the unit list) -> k_pair_sums / k_pair_mi (exact int64
//   sums of the listed pairs by class-wise popcounts, fp64 MI) + gemm_bits_kernel<5> on the band's tiles -> k_mi_units (short-range
//   units) -> k_pick_bucket.
// The limb paths (ldw_set_path(1), or weights the approximation cannot serve):
//   GEMM, 3 high limbs -> k_build_packs (per-block SNP constants in epilogue order) -> k_mi_screen / k_mi_screen_generic
//   (fp32 upper bound of MI per pair; lists the units = 64 from-side SNPs x 1 to-side SNP that hold a short-range pair
//   or a pair that may reach the guessed bucket) -> gemm_lo_units_kernel (the 2 low limbs of the listed units' joint sums)
//   -> k_mi_units<true|false> (fp64 MI of the listed units, short-range rows to their final rows, candidates >= the
//   guessed bucket appended and counted) -> k_pick_bucket (verifies the guess).
// Then, either way: k_sel_thresh / k_sel_mark / k_sel_scatter / k_sel_clear (radix select of the threshold, bitmap ranks: no sort), or,
// for blocks without a guess and very large candidate sets, two radix sorts (by MI, then by reference row order) with k_lr_thresh in
// between -> k_lr_append.
// The short-range test needs no arithmetic per pair: POS is ascending, so the partners of a to-side SNP within sr_dist
// (circularly) are at most three index intervals of the from-side list, found on the host by binary search (ColInfo).
// The fused alternative (GEMM + epilogue in one kernel, ldw_set_fused) lives in ldw_fused.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include "ldw_prim.h"

#include "ldw_internal.h"
#include "ldw_dev.h"
#ifndef LDW_SCREEN_V
#define LDW_SCREEN_V 2   // columns in flight per wave in the multi-cell screen (tuning: make CXXFLAGS+=-DLDW_SCREEN_V=4)
#endif
#ifndef LDW_SCREEN_WAVES
#define LDW_SCREEN_WAVES 6   // waves per SIMD the screen is compiled for (80 VGPRs; 5: 96)
#endif
#include "ldw_epi.h"
#include <condition_variable>
#include <mutex>
#include <thread>

#include "ldw_apx.h"

using namespace ldw;

namespace ldw {
#include "ldw_mi_screen.inc"
#include "ldw_mi_eval.inc"
#include "ldw_mi_select.inc"
}  // namespace ldw

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {
#include "ldw_mi_block.inc"
#include "ldw_mi_items.inc"
}  // namespace

namespace ldw {
int ensure_streams(ldw_ctx *c) {
    if (!c->copy_stream) {
        LDW_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < LDW_NSLOT; ++k) {
            LDW_HIP(hipEventCreateWithFlags(&c->ev_up[k], hipEventDisableTiming));
            LDW_HIP(hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
        }
        for (int k = 0; k < LDW_NSLOT; ++k) {
            LDW_HIP(hipEventCreateWithFlags(&c->ev_pick[k], hipEventDisableTiming));
            LDW_HIP(hipHostMalloc(&c->pin_pick[k], (size_t)LDW_SPAN_MAX * PICK_STRIDE + 64, hipHostMallocDefault));
        }
        LDW_HIP(hipEventCreateWithFlags(&c->ev_lrc, hipEventDisableTiming));
        LDW_HIP(hipHostMalloc(&c->pin_lrc, 64, hipHostMallocDefault));
        {   // the block-wide kernels (GEMM, screens) fill the chip; the tail of the previous block on the main stream is a chain of
            // small latency-bound kernels that should be dispatched as soon as they are ready: lowest priority for this stream
            int lo_p = 0, hi_p = 0;
            LDW_HIP(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
            static const bool prio = exp_env("LDW_NO_STREAM_PRIO") == nullptr;
            // LDW_CU_RESERVE=m (odd, experiment): the GEMM stream may not use every m-th CU, so that the main stream's chain of small
            // kernels always finds free CUs while a block-wide kernel runs (an odd modulus spreads the reserved CUs over the XCDs
            // whether the mask bits run XCD by XCD or interleave them)
            static const int cu_mod = [] { const char *e = exp_env("LDW_CU_RESERVE"); return e ? atoi(e) : 0; }();
            if (cu_mod >= 3 && (cu_mod & 1)) {
                int cus = 256;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
                std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
                for (int i = 0; i < cus; ++i)
                    if (i % cu_mod != cu_mod - 1) mask[(size_t)i / 32] |= 1u << (i % 32);
                LDW_HIP(hipExtStreamCreateWithCUMask(&c->gemm_stream, (uint32_t)mask.size(), mask.data()));
            } else if (prio) LDW_HIP(hipStreamCreateWithPriority(&c->gemm_stream, hipStreamNonBlocking, lo_p));
            else LDW_HIP(hipStreamCreateWithFlags(&c->gemm_stream, hipStreamNonBlocking));
        }
        for (int k = 0; k < LDW_NSLOT; ++k) LDW_HIP(hipEventCreateWithFlags(&c->ev_gemm[k], hipEventDisableTiming));
        // r05: what the streaming lr_links.tsv writer needs on the device side (ldw_lr_stream_begin): made here, not inside a job
        LDW_HIP(hipStreamCreateWithFlags(&c->lr_st, hipStreamNonBlocking));
        for (auto &e : c->lr_ev) LDW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        LDW_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->lr_counts), sizeof(int64_t) * 64, hipHostMallocDefault));
        c->lr_pin_cap = (size_t)32 << 20;   // (2 M rows: a batch of eight items of a C4 pass is 0.2-0.6 M)
        LDW_HIP(hipHostMalloc(&c->lr_pin, c->lr_pin_cap, hipHostMallocDefault));
    }
    return LDW_OK;
}

int join_prepare(ldw_ctx *c) {
    if (!c) return LDW_OK;
    int rc = LDW_OK;
    for (int which = 0; which < 2; ++which) {
        std::thread *&t = which == 0 ? c->prep_thread : c->prep_thread2;
        if (!t) continue;
        if (t->joinable()) t->join();
        delete t;
        t = nullptr;
        int &trc = which == 0 ? c->prep_rc : c->prep_rc2;
        if (trc != LDW_OK && rc == LDW_OK) {
            rc = trc;
            set_error("side thread of %s: %s", which == 0 ? "ldw_ctx_create" : "ldw_ctx_reserve", (which == 0 ? c->prep_err : c->prep_err2).c_str());
        }
        trc = LDW_OK;
    }
    return rc;
}

// ldw_ctx_reserve's side thread: the per-slot device buffers of a pass of blocks of `blk` SNPs (spans of up to nseg of them), from estimates
// of the row counts (1.25 rows per SNP + padding; a C4 block has 1.16) — whatever turns out too small grows where it is used, as before.
int reserve_slot_buffers(ldw_ctx *c, int64_t Npad, int64_t blk, int64_t nseg) {
    const int64_t KW = Npad / 64;
    const int64_t nt = blk * nseg;
    const size_t RF = (size_t)((blk * 5 / 4 + 512 + 127) / 128 * 128), RT = (size_t)((nt * 5 / 4 + 512 + 127) / 128 * 128);
    const size_t nf_tiles = (size_t)(blk / 64 + 6), nf_slots = nf_tiles * 64;
    const size_t n_units = nf_tiles * (size_t)nt;
    const size_t cap = pair_cap_for(blk, nt, (int)nseg);
    const size_t o_cph = ((size_t)nt * sizeof(ColMeta) + 255) / 256 * 256, o_rph = ((size_t)nf_slots * sizeof(RowPack) + 255) / 256 * 256;
    const size_t one = (size_t)blk * (size_t)blk;
    for (int s = 0; s < LDW_NSLOT; ++s) {
        if (int rc = c->panel[s][0].reserve(RF * (size_t)KW * 16)) return rc;   // (the scaled panel of k_pack_panel: 16 bytes per word)
        if (int rc = c->panel[s][1].reserve(RT * (size_t)KW * 16)) return rc;
        if (int rc = c->Gapx[s].reserve(RF * RT * 4)) return rc;
        if (int rc = c->apx_units[s].reserve(64 + 2 * n_units * 8 + 64)) return rc;
        if (int rc = c->apx_packs[s].reserve(2 * o_cph + 2 * o_rph + ((size_t)blk + (size_t)nt) * 4 + 1024)) return rc;
        if (int rc = c->apx_mini[s].reserve(((size_t)nt + nf_slots) * 32 + 512)) return rc;
        if (int rc = c->pairs[s].reserve(256 + (size_t)PAIR_PATHS * PAIR_SHARDS * cap * sizeof(PairEnt) + (size_t)maybe_cap_for((int64_t)RT, (int64_t)RF) * sizeof(ApxMaybe) + 64)) return rc;
        if (int rc = c->apx_bins[s].reserve(2 * (RT + RF) + (size_t)nt + nf_slots + 256 + (RT / 128) * (RF / 64) * 4)) return rc;
        if (int rc = c->apx_clean[s].reserve((RT / 32) * (RF / 64) + 64)) return rc;
        if (int rc = c->hist[s].reserve((size_t)nseg * NBINS * 8)) return rc;
        const size_t cand = std::max<size_t>(one, (size_t)nseg * std::min<size_t>(one, (size_t)PAIR_PATHS * PAIR_SHARDS * cap));
        if (int rc = c->cand_key[s].reserve(cand * 8)) return rc;
        if (int rc = c->cand_val[s].reserve(cand * 8)) return rc;
        if (int rc = gx(c, s).reserve(RF * RF * 8)) return rc;   // the exact 5-limb sums of a single block's band tiles
    }
    if (int rc = c->pair_sums.reserve((size_t)PAIR_PATHS * PAIR_SHARDS * cap * 16 * 8)) return rc;
    return LDW_OK;
}

void warm_mi() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_zero4));
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_mi_screen<1, true>));
    (void)hipGetLastError();
}
}  // namespace ldw

extern "C" {

int ldw_mi_block(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, int quirk_mode,
                 double *MI_out, int on_device) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(from_idx && to_idx && MI_out, LDW_ERR_ARG, "ldw_mi_block: null argument");
    LDW_REQUIRE(quirk_mode == LDW_QUIRK_REFERENCE || quirk_mode == LDW_QUIRK_INTENDED, LDW_ERR_ARG, "bad quirk mode");
    EmitArgs E;
    memset(&E, 0, sizeof(E));
    if (int rc = run_block_mi(c, from_idx, nf, to_idx, nt, quirk_mode, E)) return rc;
    LDW_HIP(hipMemcpyAsync(MI_out, c->MIblk.p, (size_t)nf * nt * 8, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    float t01 = 0, t12 = 0;
    LDW_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
    LDW_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
    c->last_ms[0] = t01;
    c->last_ms[1] = t12;
    c->last_ms[2] = 0;
    c->last_ms[3] = t01 + t12;
    return LDW_OK;
}

int ldw_joint_tables(ldw_ctx *c, const int32_t *pair_a, const int32_t *pair_b, int64_t np, int64_t *counts_out,
                     int64_t *fixed_out, int *frac_bits_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(pair_a && pair_b && np > 0, LDW_ERR_ARG, "ldw_joint_tables: bad argument");
    if (int rc = join_prepare(c)) return rc;   // (ldw_ctx_reserve's side thread sizes c->G's neighbours gx(c, s) — no entry point touches them beside it)
    if (int rc = ensure_rows(c)) return rc;
    if (frac_bits_out) *frac_bits_out = c->frac_bits;
    const int64_t CH = 1024;
    std::vector<int8_t> ones((size_t)c->Npad, 1);
    ldw::DevBuf d_ones, d_out, d_cmarg;
    int rc = LDW_OK;
    auto cleanup = [&]() { d_ones.release(); d_out.release(); d_cmarg.release(); };
    if ((rc = d_ones.reserve((size_t)c->Npad)) || (rc = d_out.reserve((size_t)CH * 25 * 8)) ||
        (rc = d_cmarg.reserve((size_t)c->L * 40))) {
        cleanup();
        return rc;
    }
    hipError_t he = hipMemcpyAsync(d_ones.p, ones.data(), ones.size(), hipMemcpyHostToDevice, c->stream);
    if (he != hipSuccess) { cleanup(); return ldw::hip_fail(he, "memcpy ones", __FILE__, __LINE__); }
    hipLaunchKernelGGL(k_slot_counts, dim3((unsigned)((c->L + 255) / 256)), dim3(256), 0, c->stream,
                       c->counts.as<int32_t>(), c->slot_meta.as<uint32_t>(), c->L, d_cmarg.as<int64_t>());
    for (int64_t p0 = 0; p0 < np && rc == LDW_OK; p0 += CH) {
        const int64_t n = std::min(CH, np - p0);
        SideLists SF, ST;
        if ((rc = build_side(c, pair_a + p0, n, SF))) break;
        if ((rc = build_side(c, pair_b + p0, n, ST))) break;
        const int RFpad = SF.Rpad, RTpad = ST.Rpad;
        if ((rc = upload_i32(c, c->rowlist_f, SF.rowlist)) || (rc = upload_i32(c, c->rowlist_t, ST.rowlist)) ||
            (rc = upload_i32(c, c->lrow_f, SF.lrow)) || (rc = upload_i32(c, c->lrow_t, ST.lrow)))
            break;
        std::vector<int32_t> vf(pair_a + p0, pair_a + p0 + n), vt(pair_b + p0, pair_b + p0 + n);
        if ((rc = upload_i32(c, c->idx_f, vf)) || (rc = upload_i32(c, c->idx_t, vt))) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = LDW_ERR_HIP; break; }
        if ((rc = c->G.reserve((size_t)RFpad * RTpad * 8))) break;
        for (int pass = 0; pass < 2 && rc == LDW_OK; ++pass) {
            int64_t *host_out = pass == 0 ? counts_out : fixed_out;
            if (!host_out) continue;
            rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, c->rowlist_t.as<int32_t>(), RTpad, c->rowlist_f.as<int32_t>(), RFpad, c->G.as<int64_t>(),
                                  pass == 0 ? 1 : c->nlimbs, pass == 0 ? d_ones.as<int8_t>() : c->digits.as<int8_t>(), 0);
            if (rc) break;
            hipLaunchKernelGGL(k_tables, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, c->G.as<int64_t>(), RFpad,
                               c->idx_f.as<int32_t>(), c->lrow_f.as<int32_t>(), c->idx_t.as<int32_t>(),
                               c->lrow_t.as<int32_t>(), (int)n, c->slot_meta.as<uint32_t>(),
                               pass == 0 ? d_cmarg.as<int64_t>() : c->slot_pfix.as<int64_t>(), d_out.as<int64_t>());
            he = hipMemcpyAsync(host_out + p0 * 25, d_out.p, (size_t)n * 25 * 8, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) rc = ldw::hip_fail(he, "joint tables copy", __FILE__, __LINE__);
        }
    }
    cleanup();
    return rc;
}

int ldw_links_begin(ldw_ctx *c, int64_t nblocks_capacity) {
    if (int rc = check_gpu(c)) return rc;
    if (int rc = join_prepare(c)) return rc;
    LDW_REQUIRE(nblocks_capacity > 0, LDW_ERR_ARG, "ldw_links_begin: capacity must be positive");
    if (c->blk_capacity != 0) {
        // the previous pass never reached ldw_links_end (an error return in the middle of ldw_mi_all_pairs / ldw_mi_block_links): kernels of
        // the aborted pass may still be queued on the three streams and read the index lists, staging images and per-slot buffers
        // that this pass is about to overwrite
        if (c->gemm_stream) LDW_HIP(hipStreamSynchronize(c->gemm_stream));
        if (c->copy_stream) LDW_HIP(hipStreamSynchronize(c->copy_stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        c->blk_capacity = 0;
    }
    if (int rc = ensure_rows(c)) return rc;  // uses ctx->small for staging; link bookkeeping takes it over below
    const size_t need = 64 + (size_t)LDW_NSLOT * LDW_SPAN_MAX * PICK_STRIDE + (size_t)nblocks_capacity * 32 + 64;
    if (int rc = c->small.reserve(need)) return rc;
    LDW_HIP(hipMemsetAsync(c->small.p, 0, need, c->stream));
    if (int rc = ensure_streams(c)) return rc;
    // everything queued on the main stream so far (row map, weights) must be visible to the GEMM stream
    LDW_HIP(hipEventRecord(c->ev_up[0], c->stream));
    LDW_HIP(hipStreamWaitEvent(c->gemm_stream, c->ev_up[0], 0));
    while ((int64_t)c->ev_pool.size() < nblocks_capacity * EVB) {
        hipEvent_t e;
        LDW_HIP(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    for (int k = 0; k < LDW_NSLOT; ++k) c->done_recorded[k] = false;
    c->early_sr = false;
    c->ev_valid.assign((size_t)nblocks_capacity, 1);
    c->lrc_recorded = false;
    c->n_sr = 0;
    c->n_lr = 0;
    c->stats.clear();
    c->multi_owner.clear();   // (a deal of ldw_mi_all_pairs_multi(.., LDW_MI_SR_ROWS_STAY) describes the tables of THAT pass only: ADVICE r05)
    c->trace.clear();
    c->blk_capacity = nblocks_capacity;
    c->blk_cursor = 0;
    for (int i = 0; i < 4; ++i) c->last_ms[i] = 0;
    return LDW_OK;
}

static int links_check(ldw_ctx *c, const ldw_mi_params *p) {
    LDW_REQUIRE(c->blk_capacity > 0 && c->blk_cursor < c->blk_capacity, LDW_ERR_STATE,
                "call ldw_links_begin with enough capacity first");
    LDW_REQUIRE(p->sr_only || p->lr_links_approx > 0, LDW_ERR_ARG, "lr_links_approx must be positive");
    LDW_REQUIRE(p->quirk_mode == LDW_QUIRK_REFERENCE || p->quirk_mode == LDW_QUIRK_INTENDED, LDW_ERR_ARG, "bad quirk mode");
    LDW_REQUIRE(c->have_meta, LDW_ERR_STATE, "SNP meta data (POS, g) not set");
    return LDW_OK;
}

int ldw_mi_block_links(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt,
                       const ldw_mi_params *p) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(from_idx && to_idx && p, LDW_ERR_ARG, "ldw_mi_block_links: null argument");
    if (int rc = links_check(c, p)) return rc;
    SmallLayout sl;
    links_layout(c, sl);
    HostBlock hb;
    if (int rc = prep_block(c, from_idx, nf, to_idx, nt, p, (int)(c->blk_cursor % LDW_NSLOT), c->blk_cursor, hb)) return rc;
    if (int rc = submit_a(c, hb, p, sl)) return rc;
    if (int rc = submit_b(c, hb, p, sl)) return rc;
    if (int rc = finish_block(c, hb, p, sl)) return rc;
    ++c->blk_cursor;
    return LDW_OK;
}

int ldw_links_end(ldw_ctx *c) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(c->blk_capacity > 0, LDW_ERR_STATE, "ldw_links_end without ldw_links_begin");
    SmallLayout sl;
    links_layout(c, sl);
    const int64_t nb = c->blk_cursor;
    int64_t h_lr = 0;
    std::vector<int64_t> si((size_t)nb * 3 + 1);
    std::vector<double> sd((size_t)nb + 1);
    int64_t h_viol = 0, h_apx[3] = {0, 0, 0};
    LDW_HIP(hipMemcpyAsync(&h_lr, sl.lr_count, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(&h_viol, sl.lr_count + 1, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(h_apx, sl.lr_count + 2, 24, hipMemcpyDeviceToHost, c->stream));
    unsigned long long h_skip = 0;
    if (c->apx_skip.p) {   // wave tiles the approximate GEMMs pruned: not executed work (ldw_gemm_stats)
        LDW_HIP(hipMemcpyAsync(&h_skip, c->apx_skip.p, 8, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipMemsetAsync(c->apx_skip.p, 0, 8, c->stream));
    }
    if (nb > 0) {
        LDW_HIP(hipMemcpyAsync(si.data(), sl.stats_i, (size_t)nb * 24, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipMemcpyAsync(sd.data(), sl.stats_d, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
    }
    LDW_HIP(hipStreamSynchronize(c->stream));
    c->n_lr = h_lr;
    c->screen_violations += h_viol;
    c->apx_units_listed += h_apx[0];
    c->apx_pairs_listed += h_apx[1];
    c->maybe_entries += h_apx[2];
    c->apx_waves_skipped += (int64_t)h_skip;
    c->gemm_stat[1] -= (double)h_skip * c->apx_ops_per_wave;
    c->stats.resize((size_t)nb);
    for (int64_t b = 0; b < nb; ++b) {
        c->stats[b].n_lr_total = si[b * 3 + 0];
        c->stats[b].n_lr_kept = si[b * 3 + 1];
        c->stats[b].n_sr = si[b * 3 + 2];
        c->stats[b].disc_thresh = sd[b];
        float t01 = 0, t12 = 0, t23 = 0;
        hipEvent_t *ev = &c->ev_pool[(size_t)b * EVB];
        if (b < (int64_t)c->ev_valid.size() && !c->ev_valid[(size_t)b]) continue;   // a later segment of a span: its time is in the span's first block
        LDW_HIP(hipEventElapsedTime(&t01, ev[0], ev[1]));
        LDW_HIP(hipEventElapsedTime(&t12, ev[c->engine == LDW_ENGINE_MFMA ? 4 : 1], ev[2]));
        if (c->engine == LDW_ENGINE_MFMA && !c->fused) {   // the screens of the approximate path run behind the GEMM on its stream
            float t15 = 0;
            if (hipEventElapsedTime(&t15, ev[1], ev[5]) == hipSuccess) t12 += t15;
            else (void)hipGetLastError();   // (an event this block never recorded: the failed query must not surface at the next launch check)
        }
        LDW_HIP(hipEventElapsedTime(&t23, ev[2], ev[3]));
        c->last_ms[0] += t01;
        c->last_ms[1] += t12;
        c->last_ms[2] += t23;
        c->last_ms[3] += t01 + t12 + t23;
        static const bool trace_on = getenv("LDW_BLOCK_TRACE") != nullptr;
        if (trace_on && b < (int64_t)c->trace.size()) {
            const ldw::BlockTrace &tr = c->trace[(size_t)b];
            float span = 0;
            int64_t pb = b - 1;
            while (pb >= 0 && pb < (int64_t)c->ev_valid.size() && !c->ev_valid[(size_t)pb]) --pb;   // (the previous block that recorded events: a span's first)
            if (pb >= 0 && hipEventElapsedTime(&span, c->ev_pool[(size_t)pb * EVB + 3], ev[3]) != hipSuccess) (void)hipGetLastError();   // selection end of the item before -> selection end of b
            fprintf(stderr, "[ldw block %3lld] %s path %d guess %4d true %4d%s cand %8lld kept %7lld  gemm %.3f epi %.3f sel %.3f  span %.3f ms\n", (long long)b,
                    tr.diag ? "diag" : "off ", tr.path, tr.guess, tr.B_true, tr.missed ? " MISS" : "", tr.n_cand, (long long)c->stats[b].n_lr_kept, t01, t12, t23, span);
        }
    }
    c->blk_capacity = 0;
    return LDW_OK;
}

// ---- spans (r04): which blocks of the list may share a launch sequence ----
// A block can be part of a span when its to side lies strictly AFTER its from side (make_blocks order: i < j), it is square, far enough
// from its from side that no pair is short-range (POS ascends over the alignment: the test of build_cols on the four end positions), and
// neither side holds a SNP the fused table test cannot place (no indicator row, or unflagged slots: h_span_bad).
// Returns 0 (no), 1 (long-range-only) or 2 (a CORNER block: its few short-range pairs — the facing ends of two neighbouring blocks, or the two
// ends of the circle — go to an SR sub-pass, the rest joins the span).
static int span_candidate(const ldw_ctx *c, const int32_t *b, const ldw_mi_params *p) {
    const int64_t fs = b[0], fe = b[1], ts = b[2], te = b[3];
    if (!(fs >= 1 && fe >= fs && ts > fe && te >= ts && te <= c->L)) return 0;
    const int64_t nf = fe - fs + 1, nt = te - ts + 1;
    if (nf != nt || nf < 2048) return 0;
    if ((int64_t)c->h_span_bad.size() != c->L + 1) return 0;
    if (c->h_span_bad[(size_t)fe] - c->h_span_bad[(size_t)fs - 1] != 0 || c->h_span_bad[(size_t)te] - c->h_span_bad[(size_t)ts - 1] != 0) return 0;
    if (!(2 * p->sr_dist < c->g)) return 0;
    const std::vector<int32_t> &P = c->h_POS;
    const double pf_min = P[(size_t)fs - 1], pf_max = P[(size_t)fe - 1], pt_min = P[(size_t)ts - 1], pt_max = P[(size_t)te - 1];
    if (pt_min - pf_max > p->sr_dist && pf_min + c->g - pt_max > p->sr_dist) return 1;
    // Measured (C4, same box, 20 cold steps): spans of long-range-only blocks 36.0 ms per pass; with the corner blocks inside them 39.8 ms although the
    // serialized kernel time fell from 36.6 to 35.2 ms — 20 large items instead of 28 alternate "diagonal block, span of eight" and the two queues no
    // longer fill each other's gaps (profiles/r04_timeline_corner_spans.txt) — so corner blocks stay items of their own unless LDW_SPAN_CORNERS=1 /
    // ldw_set_span(.., corners) asks for them (kept, tested: test_spans_equal_block_by_block runs both)
    static const bool corners_env = exp_env("LDW_SPAN_CORNERS") != nullptr;
    if (!corners_env && !c->span_corners) return 0;
    // short-range pairs of the block: POS ascends, so they sit where the two ranges face each other (directly, or across the origin)
    auto count_le = [&](int64_t lo, int64_t hi, double v) { return (int64_t)(std::upper_bound(P.begin() + (lo - 1), P.begin() + hi, (int32_t)std::floor(v)) - (P.begin() + (lo - 1))); };
    const int64_t f_tail = nf - count_le(fs, fe, pt_min - p->sr_dist - 1.0), t_head = count_le(ts, te, pf_max + p->sr_dist);      // from SNPs within sr_dist of the to side's first / to SNPs of the from side's last
    const int64_t f_head = count_le(fs, fe, pt_max + p->sr_dist - c->g), t_tail = nt - count_le(ts, te, pf_min - p->sr_dist + c->g - 1.0);
    const int64_t est = f_tail * t_head + f_head * t_tail;   // (an upper bound of the pair count; a quarter of a block's side at most per corner)
    return (est > 0 && est <= 4000000 && f_tail <= nf / 4 && t_head <= nt / 4 && f_head <= nf / 4 && t_tail <= nt / 4) ? 2 : 0;
}
// what the whole pass must offer (checked once, after the cold-start probes: a positive guess for off-diagonal blocks exists)
static bool spans_possible(const ldw_ctx *c, const ldw_mi_params *p) {
    static const bool env_off = getenv("LDW_NO_SPAN") != nullptr;
    return c->span_on && !env_off && c->span_max >= 2 && c->prune && c->engine == LDW_ENGINE_MFMA && c->apx_ok && !c->fused && c->path_mode != 1 && c->screen == 1 &&
           !p->sr_only && speculation_pays(c, p) && c->pos_sorted && c->spec_B_next[0] > 0 && c->tab11_on &&
           (p->quirk_mode != LDW_QUIRK_REFERENCE || c->r_min >= 2.0);
}
// r04c: a DIAGONAL block can run as SR sub-pass (list order: band GEMM + whole units) + long-range pass with its rows ordered by weight like any
// long-range-only block (tile pruning, clean regions), the short-range pairs kept out of its candidates: LDW_DIAG_SPLIT=1 / ldw_set_span(on | 4)
static bool diag_split_ok(const ldw_ctx *c, const int32_t *b) {
    static const bool env_on = exp_env("LDW_DIAG_SPLIT") != nullptr;
    if (!env_on && !c->diag_split) return false;
    const int64_t fs = b[0], fe = b[1];
    if (!(b[2] == fs && b[3] == fe) || fe - fs + 1 < 2048) return false;
    if ((int64_t)c->h_span_bad.size() != c->L + 1) return false;
    return c->h_span_bad[(size_t)fe] - c->h_span_bad[(size_t)fs - 1] == 0;
}
struct WorkItem {
    int64_t b0;
    int nseg;           // 1: an ordinary block
    uint32_t sr_mask;   // segments of a span that are corner blocks (SR sub-pass)
};

int ldw_mi_all_pairs(ldw_ctx *c, const int32_t *blocks, int64_t nblocks, const ldw_mi_params *p, int reset) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(blocks && p && nblocks > 0, LDW_ERR_ARG, "ldw_mi_all_pairs: bad argument");
    LDW_REQUIRE(reset, LDW_ERR_ARG, "ldw_mi_all_pairs: appending to earlier calls is not supported (reset must be 1)");
    static const bool host_timing0 = getenv("LDW_HOST_TIMING") != nullptr;
    auto now0 = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = now0();
    if (int rc = join_prepare(c)) return rc;
    const double t_joined = now0();
    if (int rc = ldw_links_begin(c, nblocks)) return rc;
    if (int rc = links_check(c, p)) return rc;
    const double t_begun = now0();
    SmallLayout sl;
    links_layout(c, sl);
    std::vector<int32_t> fi, ti;
    auto fill = [&](int64_t b) -> int {
        const int32_t fs = blocks[b * 4 + 0], fe = blocks[b * 4 + 1], ts = blocks[b * 4 + 2], te = blocks[b * 4 + 3];
        LDW_REQUIRE(fs >= 1 && fe >= fs && fe <= c->L && ts >= 1 && te >= ts && te <= c->L, LDW_ERR_ARG,
                    "block %lld = (%d,%d,%d,%d) outside 1..%lld", (long long)b, fs, fe, ts, te, (long long)c->L);
        fi.resize((size_t)(fe - fs + 1));
        ti.resize((size_t)(te - ts + 1));
        for (int32_t k = fs; k <= fe; ++k) fi[k - fs] = k - 1;
        for (int32_t k = ts; k <= te; ++k) ti[k - ts] = k - 1;
        return LDW_OK;
    };
    if (p->keep_sr && c->pos_sorted && 2 * p->sr_dist < c->g) {
        // Size the short-range table ONCE: geometric growth re-copies up to a gigabyte and synchronises both streams every time (ten times in
        // the first pass of C4), and would double-buffer tens of GB at C5.  The blocks of a pass hold every unordered SNP pair at most once,
        // so the number of pairs within sr_dist on the circle bounds the rows: POS ascends — a two-pointer walk, O(L) (r03 ran build_cols
        // over every block for this, 0.2 ms each, and only for passes of more than 200 blocks).
        const int64_t Ls = c->L;
        const std::vector<int32_t> &P = c->h_POS;
        int64_t total_sr = 0, hi = 0, wlo = 0;
        if (c->sr_total_dist == p->sr_dist && c->sr_total >= 0) total_sr = c->sr_total;   // (the walk is 1 ms at L = 100k: once per positions and sr_dist)
        else {
        for (int64_t a = 0; a < Ls; ++a) {
            if (hi < a + 1) hi = a + 1;
            while (hi < Ls && (double)P[(size_t)hi] - (double)P[(size_t)a] <= p->sr_dist) ++hi;
            total_sr += hi - a - 1;                                   // partners ahead of a, directly
        }
        for (int64_t a = 0; a < Ls; ++a) {   // partners across the origin: b before a with P[b] + g - P[a] <= sr_dist (the limit ascends with a)
            const double lim = p->sr_dist - c->g + (double)P[(size_t)a];
            if (lim < (double)P[0]) continue;
            while (wlo < Ls && (double)P[(size_t)wlo] <= lim) ++wlo;
            total_sr += std::min<int64_t>(wlo, a);
        }
        c->sr_total = total_sr;
        c->sr_total_dist = p->sr_dist;
        }
        // r05: a SHARE of the block list (a rank or context of a multi-GPU job: under three quarters of the pair space) is sized for the rows of ITS blocks — the band
        // enumerator counts them on the device in well under a millisecond (ldw_sr_pairs_fill without outputs; cached per block list) — instead of the whole alignment's:
        // at config 5 that is 4.5 instead of 36 GB per rank of eight.  Whatever this under-estimates still grows where it is used (ensure_links_capacity per item).
        int64_t want_sr = total_sr;
        {
            double share_pairs = 0;
            uint64_t key = 1469598103934665603ull ^ (uint64_t)(int64_t)(p->sr_dist * 16.0);
            for (int64_t b = 0; b < nblocks; ++b) {
                share_pairs += (double)(blocks[b * 4 + 1] - blocks[b * 4 + 0] + 1) * (double)(blocks[b * 4 + 3] - blocks[b * 4 + 2] + 1);
                for (int q = 0; q < 4; ++q) key = (key ^ (uint64_t)(uint32_t)blocks[b * 4 + q]) * 1099511628211ull;
            }
            if (share_pairs < 0.375 * (double)Ls * (double)Ls && c->g == std::floor(c->g)) {
                if (c->sr_share_rows >= 0 && c->sr_share_key == key) want_sr = std::min(total_sr, c->sr_share_rows);
                else {
                    int64_t n_share = 0;
                    if (ldw_sr_pairs_fill(c, blocks, nblocks, p->sr_dist, nullptr, nullptr, 0, &n_share) == LDW_OK && n_share <= total_sr) {
                        c->sr_share_key = key;
                        c->sr_share_rows = n_share;
                        want_sr = n_share;
                    }
                }
            }
        }
        if (int rc = ensure_links_capacity(c, want_sr + 1024, 0)) return rc;
    }
    if (!p->sr_only) {
        // r05: the long-range table sized ONCE as well.  The per-block filter keeps about lr_retain_links rows in all (R/computePairwiseMI.R:347-358:
        // prob = 1 - lr_retain_links / lr_links_approx), and what a selection needs beyond the rows kept so far is its own candidate count (<= 2^20 on
        // the sort-free path): 1.25 lr_retain_links + 4 M rows, at most the pass's pairs.  A fresh context grew the table nine times in its first
        // pass — each time a stream synchronisation, and with lr_links.tsv streaming (ldw_lr_stream_begin) a wait for the writer thread on top
        // (tools/lr_stream_probe.py: first pass 72-88 ms against 41).  Whatever this under-estimates still grows where it is used.
        double tot_pairs = 0;
        for (int64_t b = 0; b < nblocks; ++b) tot_pairs += (double)(blocks[b * 4 + 1] - blocks[b * 4 + 0] + 1) * (double)(blocks[b * 4 + 3] - blocks[b * 4 + 2] + 1);
        const double want = std::min(tot_pairs, std::max(0.0, p->lr_retain_links) * 1.25 + (double)(4 << 20));
        if (want < 4e9)
            if (int rc = ensure_links_capacity(c, c->n_sr, (int64_t)want)) return rc;
    }
    // Software pipeline, three ITEMS deep on the host (an item = one block, or a span of consecutive long-range-only blocks of one block
    // row: r04): item i's epilogue chain is submitted (main stream), then at once the block-wide pass of item i+1 (GEMM stream; prepared
    // earlier), and only then the host waits for item i's pick(s).
    // r03: the lists of an item are built by a HELPER THREAD that runs ahead of the submitting thread (prep_block is pure host work into the
    // slot's pinned staging buffer: 0.45 ms per 10k x 10k block — once the GPU side of a block had come down to 0.5 ms it was the loop's
    // critical path).  Hand-over through counters under one mutex: item k may be prepared once item k - RING is finished (its ring entry
    // is free), item k - LDW_NSLOT has been submitted (the slot's staging buffer then belongs to an upload the helper waits for: ev_up) and
    // the plan covers it (n_planned: the items behind the leading blocks are only known after the cold-start probes — whether spans may
    // form depends on a guess existing); the GEMM stream runs up to LDW_NSLOT - 1 = 2 items ahead of the item the main stream evaluates.
    constexpr int RING = 8;
    HostBlock hb[RING];
    static const bool host_timing = getenv("LDW_HOST_TIMING") != nullptr;
    double th[5] = {0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // the plan: items in block order.  Before the probes only the leading run of blocks that can never be part of a span is planned.
    std::vector<WorkItem> items;
    std::vector<uint8_t> cand((size_t)nblocks, 0);
    for (int64_t b = 0; b < nblocks; ++b) cand[(size_t)b] = (uint8_t)span_candidate(c, blocks + b * 4, p);
    int64_t lead = 0;
    while (lead < nblocks && !cand[(size_t)lead] && !diag_split_ok(c, blocks + lead * 4)) ++lead;   // (a block that may be split is planned after the probes)
    for (int64_t b = 0; b < lead; ++b) items.push_back(WorkItem{b, 1, 0u});
    struct Shared {
        std::mutex m;
        std::condition_variable cv;
        int64_t n_sub = 0, n_done = 0, n_planned = 0, next = 0;   // next: the item the next free helper takes
        std::vector<uint8_t> prepped;                              // per item (helpers finish out of order)
        bool plan_final = false;
        int rc = LDW_OK;
        int64_t bad = INT64_MAX;             // the first item whose preparation failed (items before it are still good: r05)
        bool stop = false, probing = true;   // probing: the cold-start probes (calling thread) still use the last slot's staging buffer
        std::string err;
    } sh;
    sh.n_planned = (int64_t)items.size();
    sh.plan_final = lead == nblocks;
    sh.prepped.assign((size_t)nblocks + 1, 0);
    // r04: TWO helpers (a span of eight blocks with a corner segment takes ~3 ms of list building, a diagonal block in front of it gives the
    // GPU 1.3 ms of work: one helper left the GEMM stream waiting).  Each takes the next item; items k, k+1, k+2 use different slots.
    auto worker = [&]() {
        (void)hipSetDevice(c->device);
        std::vector<int32_t> wfi, wti;   // this helper's own index lists
        for (;;) {
            WorkItem it{0, 0, 0u};
            int64_t k = 0;
            {
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] {
                    if (sh.stop) return true;
                    if (sh.next >= sh.n_planned) return sh.plan_final;   // (plan complete and nothing left: leave)
                    return sh.next < sh.n_done + RING && sh.next < sh.n_sub + LDW_NSLOT - (sh.probing ? 1 : 0);
                });
                if (sh.stop || sh.next >= sh.n_planned) return;
                k = sh.next++;
                it = items[(size_t)k];
            }
            int rc = LDW_OK;
            try {   // (prep_block allocates a dozen std::vectors: an exception on this thread must come back as an error code, not std::terminate)
                const int32_t fs = blocks[it.b0 * 4 + 0], fe = blocks[it.b0 * 4 + 1];
                bool ok = fs >= 1 && fe >= fs && fe <= c->L;
                SpanPlan spn;
                wti.clear();
                for (int q = 0; q < it.nseg && ok; ++q) {
                    const int32_t ts = blocks[(it.b0 + q) * 4 + 2], te = blocks[(it.b0 + q) * 4 + 3];
                    ok = ts >= 1 && te >= ts && te <= c->L;
                    if (!ok) break;
                    spn.start[q] = (int32_t)wti.size();
                    spn.nt[q] = te - ts + 1;
                    for (int32_t x = ts; x <= te; ++x) wti.push_back(x - 1);
                }
                spn.nseg = it.nseg;
                if (!ok) {
                    set_error("block %lld = (%d,%d,%d,%d) outside 1..%lld", (long long)it.b0, fs, fe, blocks[it.b0 * 4 + 2], blocks[it.b0 * 4 + 3], (long long)c->L);
                    rc = LDW_ERR_ARG;
                } else {
                    wfi.resize((size_t)(fe - fs + 1));
                    for (int32_t q = fs; q <= fe; ++q) wfi[q - fs] = q - 1;
                    const int slot = (int)(k % LDW_NSLOT);
                    if (c->up_recorded[slot] && hipEventSynchronize(c->ev_up[slot]) != hipSuccess) {   // the staging buffer of this slot has been uploaded
                        set_error("prep: hipEventSynchronize failed");
                        rc = LDW_ERR_HIP;
                    }
                    HostBlock &h = hb[k % RING];
                    const bool split1 = it.nseg == 1 && (it.sr_mask & 1u);   // a diagonal block: SR sub-pass + weight-ordered long-range pass
                    if (rc == LDW_OK) rc = prep_block(c, wfi.data(), (int64_t)wfi.size(), wti.data(), (int64_t)wti.size(), p, slot, it.b0, h, it.nseg > 1 ? &spn : nullptr, -1, false, 0, split1);
                    if (rc == LDW_OK && split1) {
                        if (h.n_sr_blk > 0 && !h.lo.band_full && !h.generic && h.lo.ordered) {
                            const size_t base = (h.total + 255) / 256 * 256;
                            HostBlock sub;
                            rc = prep_block(c, wfi.data(), (int64_t)wfi.size(), wti.data(), (int64_t)wti.size(), p, slot, it.b0, sub, nullptr, -1, true, base);
                            if (rc == LDW_OK) {
                                h.stage_total = (base + sub.total + 255) / 256 * 256;
                                h.subs.push_back(std::move(sub));
                                h.subs_seg.push_back(0);
                            }
                        } else {   // nothing to split after all: the ordinary form of the block
                            rc = prep_block(c, wfi.data(), (int64_t)wfi.size(), wti.data(), (int64_t)wti.size(), p, slot, it.b0, h);
                        }
                    }
                    if (rc == LDW_OK && it.nseg > 1) {
                        h.span_from = wfi;
                        h.span_to = wti;
                        // the SR sub-passes of its corner segments: the block alone, in list order, behind the span's own image
                        size_t base = (h.total + 255) / 256 * 256;
                        for (int q = 0; q < it.nseg && rc == LDW_OK; ++q) {
                            if (!((it.sr_mask >> q) & 1u)) continue;
                            HostBlock sub;
                            rc = prep_block(c, wfi.data(), (int64_t)wfi.size(), wti.data() + spn.start[q], spn.nt[q], p, slot, it.b0 + q, sub, nullptr, -1, true, base);
                            if (rc != LDW_OK) break;
                            if (sub.n_sr_blk <= 0) continue;   // (no pair within sr_dist after all)
                            if (sub.lo.band_full || sub.generic) {
                                set_error("corner block %lld cannot take an SR sub-pass", (long long)(it.b0 + q));
                                rc = LDW_ERR_STATE;
                                break;
                            }
                            h.seg_n_sr[q] = sub.n_sr_blk;
                            h.seg_lr_total[q] -= sub.n_sr_blk;
                            base = (base + sub.total + 255) / 256 * 256;
                            h.subs.push_back(std::move(sub));
                            h.subs_seg.push_back(q);
                        }
                        h.stage_total = base;
                    }
                }
            } catch (const std::exception &e) {
                set_error("preparing block %lld: %s", (long long)it.b0, e.what());
                rc = LDW_ERR_HIP;
            } catch (...) {
                set_error("preparing block %lld: unknown exception", (long long)it.b0);
                rc = LDW_ERR_HIP;
            }
            std::lock_guard<std::mutex> lk(sh.m);
            if (rc != LDW_OK) {
                if (sh.rc == LDW_OK || k < sh.bad) {
                    sh.rc = rc;
                    sh.err = ldw_last_error();
                }
                sh.bad = std::min(sh.bad, k);
                sh.stop = true;
            } else {
                sh.prepped[(size_t)k] = 1;
            }
            sh.cv.notify_all();
            if (rc != LDW_OK) return;
        }
    };
    static const int n_helpers = [] { const char *e = exp_env("LDW_HELPERS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 3 ? 3 : v); }();
    std::thread helpers[3];
    for (int i = 0; i < n_helpers; ++i) helpers[i] = std::thread(worker);
    struct Joiner {   // every way out of this function stops and joins the helpers
        Shared &sh;
        std::thread *t;
        int n;
        ~Joiner() {
            {
                std::lock_guard<std::mutex> lk(sh.m);
                sh.stop = true;
            }
            sh.cv.notify_all();
            for (int i = 0; i < n; ++i)
                if (t[i].joinable()) t[i].join();
        }
    } joiner{sh, helpers, n_helpers};
    const double t_sized = now0();
    // (the helper is already building the first blocks' lists while the probes run; it stays out of the last slot until they are done)
    // cold start: a sampled guess for each block kind that has none yet (probe_kind_guess), taken from the first block of the kind
    static const bool probe_on = getenv("LDW_NO_PROBE") == nullptr;
    if (probe_on && !p->sr_only && c->engine != LDW_ENGINE_HIST_STATES && !c->fused && c->pos_sorted && speculation_pays(c, p)) {
        bool done_kind[2] = {false, false};
        Probe probes[2];
        int n_probe = 0;
        size_t pin_base = 0;
        for (int64_t b = 0; b < nblocks && !(done_kind[0] && done_kind[1]); ++b) {
            const bool diag = blocks[b * 4 + 0] == blocks[b * 4 + 2] && blocks[b * 4 + 1] == blocks[b * 4 + 3];
            const int kind = diag ? 1 : 0;
            if (done_kind[kind]) continue;
            done_kind[kind] = true;
            if (c->spec_B_next[kind] >= 0) continue;
            if (int rc = fill(b)) return rc;
            const int64_t npairs = diag ? (int64_t)fi.size() * ((int64_t)fi.size() - 1) / 2 : (int64_t)fi.size() * (int64_t)ti.size();
            if (npairs < PROBE_MIN_PAIRS) continue;
            // (a staging buffer that has to grow for the second sample is reallocated: the first sample's upload must have left it)
            if (n_probe > 0 && c->pin_cap[LDW_NSLOT - 1] < pin_base + 2 * probes[0].hb.total + 65536) LDW_HIP(hipStreamSynchronize(c->stream));
            if (int rc = probe_enqueue(c, fi.data(), (int64_t)fi.size(), ti.data(), (int64_t)ti.size(), p, sl, kind, n_probe, pin_base, probes[n_probe])) return rc;
            pin_base = (pin_base + probes[n_probe].hb.total + 255) / 256 * 256;
            ++n_probe;
        }
        if (n_probe > 0) {
            LDW_HIP(hipStreamSynchronize(c->stream));
            for (int k = 0; k < n_probe; ++k) probe_collect(c, probes[k]);
        }
    }
    const double t_probed = now0();
    {   // the rest of the plan: consecutive candidates of one block row, to sides ascending, form a span (at most span_max blocks,
        // nf x nt below the 32-bit index limit of the unit lists, the int32 block below ~6 GB)
        const bool spans = spans_possible(c, p);
        std::vector<WorkItem> rest;
        for (int64_t b = lead; b < nblocks;) {
            int n = 1;
            if (spans && cand[(size_t)b]) {
                const int64_t nf = blocks[b * 4 + 1] - blocks[b * 4 + 0] + 1;
                int64_t nt_tot = blocks[b * 4 + 3] - blocks[b * 4 + 2] + 1;
                static const int env_max = [] { const char *e = getenv("LDW_SPAN_MAX"); return e ? atoi(e) : 0; }();   // (A/B measurements)
                const int nmax = std::min<int>(env_max >= 1 ? env_max : c->span_max, LDW_SPAN_MAX);
                while (n < nmax && b + n < nblocks && cand[(size_t)(b + n)] && blocks[(b + n) * 4 + 0] == blocks[b * 4 + 0] && blocks[(b + n) * 4 + 1] == blocks[b * 4 + 1] &&
                       blocks[(b + n) * 4 + 2] > blocks[(b + n - 1) * 4 + 3]) {
                    const int64_t nt_k = blocks[(b + n) * 4 + 3] - blocks[(b + n) * 4 + 2] + 1;
                    if (nf * (nt_tot + nt_k) >= 1500000000LL || (nt_tot + nt_k) > 900000) break;
                    nt_tot += nt_k;
                    ++n;
                }
            }
            uint32_t srm = 0;
            for (int k = 0; k < n; ++k)
                if (n > 1 && cand[(size_t)(b + k)] == 2) srm |= 1u << k;
            if (n == 1 && spans && diag_split_ok(c, blocks + b * 4)) srm = 1u;
            rest.push_back(WorkItem{b, n, srm});
            b += n;
        }
        c->early_sr = spans;
        std::lock_guard<std::mutex> lk(sh.m);
        items.insert(items.end(), rest.begin(), rest.end());
        sh.n_planned = (int64_t)items.size();
        sh.plan_final = true;
        sh.probing = false;
    }
    sh.cv.notify_all();
    const int64_t nitems = (int64_t)items.size();
    const double t_planned = now0();
    if (host_timing0)
        fprintf(stderr, "[ldw host us] waiting for the side threads %.0f  links_begin (row map if stale, bookkeeping) %.0f  table sizing + helper start %.0f  cold-start probes %.0f  plan %.0f\n",
                t_joined - t_enter, t_begun - t_joined, t_sized - t_begun, t_probed - t_sized, t_planned - t_probed);
    // blocks until item k is prepared (true) — or, with wait = false, says whether it is
    auto prepped = [&](int64_t k, bool wait, int &rc) -> bool {
        std::unique_lock<std::mutex> lk(sh.m);
        // (a failure of a LATER item does not concern this one: the helpers take the items in order, so every item before the failed one is
        // prepared or being prepared — the pass runs on to the failed item itself, like the reference's loop: R/computePairwiseMI.R:103-116)
        if (wait) sh.cv.wait(lk, [&] { return (sh.rc != LDW_OK && k >= sh.bad) || sh.prepped[(size_t)k] != 0; });
        rc = (sh.rc != LDW_OK && k >= sh.bad) ? sh.rc : LDW_OK;
        if (rc != LDW_OK) set_error("%s", sh.err.c_str());
        return sh.prepped[(size_t)k] != 0;
    };
    int64_t n_sub = 0;   // items [0, n_sub) have been submitted to the GEMM stream
    auto submit_next = [&]() -> int {
        if (int rc = submit_a(c, hb[n_sub % RING], p, sl)) return rc;
        ++n_sub;
        {
            std::lock_guard<std::mutex> lk(sh.m);
            sh.n_sub = n_sub;
        }
        sh.cv.notify_all();
        return LDW_OK;
    };
    const int64_t ahead = c->overlap ? LDW_NSLOT - 1 : 1;
    {
        int rc = LDW_OK;
        prepped(0, true, rc);
        if (rc) return rc;
        if ((rc = submit_next())) return rc;
    }
    int fail_rc = LDW_OK;
    std::string fail_msg;
    int64_t b_fail = -1;          // the item the loop was at when it failed
    bool b_fail_submitted = false, b_fail_finished = false;
    auto failed = [&](int rc, int64_t b, bool sub_b, bool fin) {
        fail_rc = rc;
        fail_msg = ldw_last_error();
        b_fail = b;
        b_fail_submitted = sub_b;
        b_fail_finished = fin;
    };
    for (int64_t b = 0; b < nitems && fail_rc == LDW_OK; ++b) {
        HostBlock &cur = hb[b % RING];
        double t0 = now();
        if (int rc = submit_b(c, cur, p, sl)) { failed(rc, b, false, false); break; }   // unfused: epilogue + pick of item b (main stream)
        th[0] += now() - t0;
        t0 = now();
        // items b+1 .. b+ahead (GEMM stream) run beside them: b+1 is waited for, the ones after it are taken if they are ready
        while (n_sub < nitems && n_sub <= b + ahead) {
            int rc = LDW_OK;
            const double tw = now();
            const bool ready = prepped(n_sub, n_sub == b + 1, rc);
            th[1] += now() - tw;
            if (rc) { failed(rc, b, true, false); break; }
            if (!ready || !can_submit_early(c, hb[n_sub % RING], p)) break;
            if ((rc = submit_next())) { failed(rc, b, true, false); break; }
        }
        if (fail_rc != LDW_OK) break;
        th[2] += now() - t0;
        t0 = now();
        // (tried: the second phase of block b+1 queued HERE, before the host waits for block b's pick, so that the main stream has work during the
        // round trip — block b's selection then runs behind it, its slot is released later, and the pass got slower: 43.0 against 40.2 ms)
        if (int rc = finish_block(c, cur, p, sl)) { failed(rc, b, true, false); break; }   // round trip + selection of item b
        th[3] += now() - t0;
        {
            std::lock_guard<std::mutex> lk(sh.m);
            sh.n_done = b + 1;
        }
        sh.cv.notify_all();
        c->blk_cursor += items[(size_t)b].nseg;
        lr_stream_push(c, c->stream, sl.lr_count, c->blk_cursor);   // (no-op without ldw_lr_stream_begin: the rows of this item may go to lr_links.tsv now)
        if (n_sub <= b + 1 && b + 1 < nitems) {                          // overlap off / no guess yet: one item after the other
            int rc = LDW_OK;
            prepped(b + 1, true, rc);
            if (rc) { failed(rc, b, true, true); break; }
            if ((rc = submit_next())) { failed(rc, b, true, true); break; }
        }
    }
    if (fail_rc != LDW_OK) {
        // r05: with lr_links.tsv streaming, the items that were submitted before the failure are run to their end, so that the file holds
        // the rows of EVERY block in front of the failed one — what the reference's loop has appended when it stops at a block (:362).
        // Best effort: a second failure in here is ignored, the first one is reported.
        if (c->lr_stream && b_fail >= 0) {
            for (int64_t bb = b_fail; bb < n_sub; ++bb) {
                HostBlock &h = hb[bb % RING];
                if (bb == b_fail && b_fail_finished) continue;
                if (!(bb == b_fail && b_fail_submitted))
                    if (submit_b(c, h, p, sl) != LDW_OK) break;
                if (finish_block(c, h, p, sl) != LDW_OK) break;
                c->blk_cursor += items[(size_t)bb].nseg;
                lr_stream_push(c, c->stream, sl.lr_count, c->blk_cursor);
            }
        }
        set_error("%s", fail_msg.c_str());
        return fail_rc;
    }
    const double t_loop = now0();
    const int rc_end = ldw_links_end(c);
    if (host_timing)
        fprintf(stderr, "[ldw host us] item loop %.0f  links_end %.0f  whole call %.0f\n", t_loop - t_planned, now0() - t_loop, now0() - t_enter);
    if (host_timing)
        fprintf(stderr, "[ldw host us/item] submit_b %.1f  wait for the helper's prep %.1f  submit_a (incl. that wait) %.1f  finish (incl. wait) %.1f  items %lld (blocks %lld)\n", th[0] / nitems, th[1] / nitems,
                th[2] / nitems, th[3] / nitems, (long long)nitems, (long long)nblocks);
    return rc_end;
}

int ldw_sr_pairs_fill(ldw_ctx *c, const int32_t *blocks, int64_t nblocks, double sr_dist, int32_t *a_out, int32_t *b_out, int64_t capacity, int64_t *n_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(blocks && nblocks > 0 && n_out && sr_dist >= 0, LDW_ERR_ARG, "ldw_sr_pairs_fill: bad argument");
    LDW_REQUIRE(c->have_meta && c->pos_sorted, LDW_ERR_STATE, "ldw_sr_pairs_fill: needs SNP positions in ascending order (ldw_set_snp_meta)");
    *n_out = 0;
    // r05: everything on the device (k_cols_dev ... k_sr_fill_all) — the host loop below built the intervals of C4's 20 band blocks in 17 ms, which a
    // multi-GPU job's rank 0 paid after every gather.  Integral genome length (every len is then an exact integer on host and device alike); a column
    // whose interval check fails (none on ascending positions) sends the call down the host path, as does LDW_SR_PAIRS_HOST=1 (A/B, tests).
    if (c->g == std::floor(c->g) && getenv("LDW_SR_PAIRS_HOST") == nullptr) {
        std::vector<SrBlkDev> hb;
        int64_t ncols = 0;
        const std::vector<int32_t> &Ph = c->h_POS;
        for (int64_t b = 0; b < nblocks; ++b) {
            const int32_t fs = blocks[b * 4 + 0], fe = blocks[b * 4 + 1], ts = blocks[b * 4 + 2], te = blocks[b * 4 + 3];
            LDW_REQUIRE(fs >= 1 && fe >= fs && fe <= c->L && ts >= 1 && te >= ts && te <= c->L, LDW_ERR_ARG, "block %lld = (%d,%d,%d,%d) outside 1..%lld", (long long)b, fs, fe,
                        ts, te, (long long)c->L);
            if (ts > fe && 2 * sr_dist < c->g) {   // (as below: ranges further apart than sr_dist directly and across the origin)
                const double pf_min = Ph[(size_t)fs - 1], pf_max = Ph[(size_t)fe - 1], pt_min = Ph[(size_t)ts - 1], pt_max = Ph[(size_t)te - 1];
                if (pt_min - pf_max > sr_dist && pf_min + c->g - pt_max > sr_dist) continue;
            }
            hb.push_back(SrBlkDev{fs - 1, fe - fs + 1, ts - 1, te - ts + 1, (fs == ts && fe == te) ? 1 : 0, 0, ncols});
            ncols += te - ts + 1;
        }
        if (hb.empty()) return LDW_OK;
        const size_t nb = hb.size();
        const size_t o_cu = nb * sizeof(SrBlkDev), o_cl = o_cu + (size_t)ncols * 4, o_rows = (o_cl + (size_t)ncols * 4 + 7) / 8 * 8, o_base = o_rows + nb * 8,
                     o_bad = o_base + (nb + 1) * 8;
        if (int rc = c->srd_seg.reserve(o_bad + 8)) return rc;
        if (int rc = c->srd_out.reserve((size_t)ncols * sizeof(ColInfo))) return rc;
        char *aux = c->srd_seg.as<char>();
        SrBlkDev *d_blk = reinterpret_cast<SrBlkDev *>(aux);
        int32_t *d_cu = reinterpret_cast<int32_t *>(aux + o_cu), *d_cl = reinterpret_cast<int32_t *>(aux + o_cl);
        int64_t *d_rows = reinterpret_cast<int64_t *>(aux + o_rows), *d_base = reinterpret_cast<int64_t *>(aux + o_base);
        int *d_bad = reinterpret_cast<int *>(aux + o_bad);
        ColInfo *d_cols = c->srd_out.as<ColInfo>();
        LDW_HIP(hipMemcpyAsync(d_blk, hb.data(), nb * sizeof(SrBlkDev), hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemsetAsync(d_bad, 0, 8, c->stream));
        hipLaunchKernelGGL(k_cols_dev, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, c->stream, c->POS.as<int32_t>(), c->g, sr_dist, d_blk, (int)nb, ncols, d_cols, d_cu,
                           d_cl, d_bad);
        hipLaunchKernelGGL(k_cols_scan, dim3((unsigned)nb), dim3(256), 0, c->stream, d_blk, d_cols, d_cu, d_cl, d_rows);
        hipLaunchKernelGGL(k_blk_bases, dim3(1), dim3(64), 0, c->stream, d_rows, (int)nb, d_base);
        LDW_HIP(hipGetLastError());
        int64_t total = 0;
        int bad = 0;
        LDW_HIP(hipMemcpyAsync(&total, d_base + nb, 8, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        if (!bad) {
            *n_out = total;
            if (total > 0 && a_out && b_out) {
                LDW_REQUIRE(total <= capacity, LDW_ERR_SIZE, "ldw_sr_pairs_fill: capacity %lld < %lld rows", (long long)capacity, (long long)total);
                hipLaunchKernelGGL(k_sr_fill_all, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, c->stream, d_cols, d_blk, (int)nb, ncols, d_base, a_out, b_out);
                LDW_HIP(hipGetLastError());
                LDW_HIP(hipStreamSynchronize(c->stream));
            }
            return LDW_OK;
        }
    }
    std::vector<int32_t> fi, ti;
    std::vector<ColInfo> cols2[2];
    int64_t base = 0;
    // r05: two record buffers in turn (the upload of block b + 1 no longer waits for the kernel of block b: one stream synchronisation per block
    // made this 12 ms for C4's 55 blocks), and the block pairs whose ranges lie further apart than sr_dist on the circle — 35 of the 55 — are
    // recognised from their four end positions (POS ascends) without building their columns.
    ldw::DevBuf dcols[2];
    hipEvent_t done[2] = {nullptr, nullptr};
    struct Rel {
        ldw::DevBuf *b;
        hipEvent_t *e;
        ~Rel() {
            for (int k = 0; k < 2; ++k) {
                b[k].release();
                if (e[k]) (void)hipEventDestroy(e[k]);
            }
        }
    } rel{dcols, done};
    const std::vector<int32_t> &P = c->h_POS;
    int64_t n_filled = 0;
    for (int64_t b = 0; b < nblocks; ++b) {
        const int32_t fs = blocks[b * 4 + 0], fe = blocks[b * 4 + 1], ts = blocks[b * 4 + 2], te = blocks[b * 4 + 3];
        LDW_REQUIRE(fs >= 1 && fe >= fs && fe <= c->L && ts >= 1 && te >= ts && te <= c->L, LDW_ERR_ARG, "block %lld = (%d,%d,%d,%d) outside 1..%lld", (long long)b, fs, fe,
                    ts, te, (long long)c->L);
        if (ts > fe && 2 * sr_dist < c->g) {   // disjoint ranges, to side after the from side: no pair within sr_dist directly or across the origin?
            const double pf_min = P[(size_t)fs - 1], pf_max = P[(size_t)fe - 1], pt_min = P[(size_t)ts - 1], pt_max = P[(size_t)te - 1];
            if (pt_min - pf_max > sr_dist && pf_min + c->g - pt_max > sr_dist) continue;
        }
        const int64_t nf = fe - fs + 1, nt = te - ts + 1;
        fi.resize((size_t)nf);
        ti.resize((size_t)nt);
        for (int64_t k = 0; k < nf; ++k) fi[(size_t)k] = fs - 1 + (int32_t)k;
        for (int64_t k = 0; k < nt; ++k) ti[(size_t)k] = ts - 1 + (int32_t)k;
        const bool diag = fs == ts && fe == te;
        int64_t n_blk = 0;
        const int k = (int)(n_filled & 1);
        if (done[k]) LDW_HIP(hipEventSynchronize(done[k]));   // (the upload and the kernel that used this pair of buffers two blocks ago)
        std::vector<ColInfo> &cols = cols2[k];
        if (int rc = build_cols(c, fi.data(), nf, ti.data(), nt, diag, sr_dist, cols, n_blk)) return rc;
        if (n_blk > 0 && a_out && b_out) {
            LDW_REQUIRE(base + n_blk <= capacity, LDW_ERR_SIZE, "ldw_sr_pairs_fill: capacity %lld < %lld rows", (long long)capacity, (long long)(base + n_blk));
            if (!done[k]) LDW_HIP(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            if (int rc = dcols[k].reserve(cols.size() * sizeof(ColInfo))) return rc;
            LDW_HIP(hipMemcpyAsync(dcols[k].p, cols.data(), cols.size() * sizeof(ColInfo), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_sr_fill, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, c->stream, dcols[k].as<ColInfo>(), (int)nf, (int)nt, fs - 1, ts - 1, diag ? 1 : 0, base,
                               a_out, b_out);
            LDW_HIP(hipGetLastError());
            LDW_HIP(hipEventRecord(done[k], c->stream));
            ++n_filled;
        }
        base += n_blk;
    }
    LDW_HIP(hipStreamSynchronize(c->stream));
    *n_out = base;
    return LDW_OK;
}

// inspection only: the first 16 pairs verify mode counted as violations since the library was loaded — (from SNP, to SNP, exact MI, the level they were judged against) —
// and how many there were; resets the record.  out: 1 + 64 doubles.
int ldw_debug_violations(ldw_ctx *c, double *out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(out, LDW_ERR_ARG, "ldw_debug_violations: null output");
    LDW_HIP(hipDeviceSynchronize());
    unsigned long long n = 0, zero = 0;
    LDW_HIP(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_viol_n), 8));
    LDW_HIP(hipMemcpyFromSymbol(out + 1, HIP_SYMBOL(g_viol_rec), 64 * 8));
    LDW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_viol_n), &zero, 8));
    out[0] = (double)n;
    return LDW_OK;
}

int ldw_debug_tab11(ldw_ctx *c, double W, double lo, double delta, double eta, double sprime, int32_t *out, double *cbin_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(out && cbin_out && W > 0 && lo > 0 && delta >= 0 && eta >= 0 && sprime > 0, LDW_ERR_ARG, "ldw_debug_tab11: bad argument");
    constexpr int NB = 64;
    ldw::DevBuf t;
    if (int rc = t.reserve((size_t)NB * NB * 8)) return rc;
    const float cbin = (float)(NB / std::sqrt(W + 1.0));
    hipLaunchKernelGGL(ldw::k_build_tab11, dim3(NB * NB * 32 / 256), dim3(256), 0, c->stream, W, lo, delta, eta, sprime, NB, cbin, t.as<int2>());
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpyAsync(out, t.p, (size_t)NB * NB * 8, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    t.release();
    if (he != hipSuccess) return ldw::hip_fail(he, "ldw_debug_tab11", __FILE__, __LINE__);
    *cbin_out = (double)cbin;
    return LDW_OK;
}

int ldw_set_span(ldw_ctx *c, int on, int max_blocks) {
    LDW_REQUIRE(c && (max_blocks == 0 || (max_blocks >= 2 && max_blocks <= LDW_SPAN_MAX)), LDW_ERR_ARG, "ldw_set_span: max_blocks must be 0 or 2..%d", LDW_SPAN_MAX);
    LDW_REQUIRE(!(on & 6) || LDW_HAS_EXPERIMENTS, LDW_ERR_STATE, "ldw_set_span: corner spans / split diagonal blocks (measured slower, r04) are only in the LDW_EXPERIMENTS build");
    c->diag_split = (on & 4) != 0;     // bit 2: diagonal blocks as SR sub-pass + weight-ordered long-range pass
    c->span_corners = (on & 2) != 0;   // bit 1: corner blocks (few short-range pairs) join the spans, their short-range pairs go to SR sub-passes (off by default: slower)
    c->span_on = on != 0;
    if (max_blocks) c->span_max = max_blocks;
    return LDW_OK;
}

int ldw_overflow_report(ldw_ctx *c, int64_t out[4]) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_overflow_report: null argument");
    out[0] = c->pair_list_overflows;
    out[1] = c->maybe_overflows;
    out[2] = c->maybe_off ? 1 : 0;
    out[3] = c->maybe_entries;
    return LDW_OK;
}

int ldw_span_report(ldw_ctx *c, int64_t out[4]) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_span_report: null argument");
    out[0] = c->span_items;
    out[1] = c->span_blocks;
    out[2] = c->span_fallbacks;
    out[3] = c->span_on ? 1 : 0;
    return LDW_OK;
}

int ldw_set_pair_cap(uint32_t cap) {
    LDW_REQUIRE(cap == 0 || (cap >= 16 && cap <= (1u << 22)), LDW_ERR_ARG, "ldw_set_pair_cap: 0 or 16..2^22");
    g_pair_cap_override.store(cap);
    return LDW_OK;
}

int ldw_set_overlap(ldw_ctx *c, int on) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    c->overlap = on != 0;
    return LDW_OK;
}

int ldw_set_screen(ldw_ctx *c, int mode) {
    LDW_REQUIRE(c && mode >= 0 && mode <= 2, LDW_ERR_ARG, "ldw_set_screen: mode must be 0, 1 or 2");
    c->screen = mode;
    return LDW_OK;
}

int ldw_set_mixed(ldw_ctx *c, int on) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    c->mixed = on != 0;
    return LDW_OK;
}

int ldw_set_fused(ldw_ctx *c, int on) {
    LDW_REQUIRE(!on || LDW_HAS_EXPERIMENTS, LDW_ERR_STATE, "ldw_set_fused(1): the fused GEMM + epilogue kernel (measured slower since r01) is only in the LDW_EXPERIMENTS build");
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    c->fused = on != 0;
    return LDW_OK;
}

int ldw_links_count(ldw_ctx *c, int which, int64_t *n_out) {
    LDW_REQUIRE(c && n_out && (which == 0 || which == 1), LDW_ERR_ARG, "ldw_links_count: bad argument");
    *n_out = which == 0 ? c->n_sr : c->n_lr;
    return LDW_OK;
}

int ldw_links_device_ptrs(ldw_ctx *c, int which, const int32_t **a_out, const int32_t **b_out, const double **MI_out, int64_t *n_out) {
    if (int rc = check_gpu(c)) return rc;
    if (int rc = join_prepare(c)) return rc;
    LDW_REQUIRE((which == 0 || which == 1) && a_out && b_out && MI_out && n_out, LDW_ERR_ARG, "ldw_links_device_ptrs: bad argument");
    LDW_REQUIRE(c->blk_capacity == 0, LDW_ERR_STATE, "ldw_links_device_ptrs: a link pass is still open (ldw_links_end)");
    LDW_HIP(hipStreamSynchronize(c->stream));
    *a_out = (which == 0 ? c->sr_a : c->lr_a).as<int32_t>();
    *b_out = (which == 0 ? c->sr_b : c->lr_b).as<int32_t>();
    *MI_out = (which == 0 ? c->sr_mi : c->lr_mi).as<double>();
    *n_out = which == 0 ? c->n_sr : c->n_lr;
    return LDW_OK;
}

#ifdef LDW_SCREEN_STATS
int ldw_debug_screen_stats(unsigned long long *out16) {   // (measurement build only; not in the header)
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ldw::g_scr_stats), 16 * 8) != hipSuccess) return LDW_ERR_HIP;
    return LDW_OK;
}
#endif
int ldw_links_fetch(ldw_ctx *c, int which, int32_t *a_out, int32_t *b_out, double *MI_out, int64_t capacity,
                    int on_device) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(which == 0 || which == 1, LDW_ERR_ARG, "ldw_links_fetch: which must be 0 (sr) or 1 (lr)");
    const int64_t n = which == 0 ? c->n_sr : c->n_lr;
    LDW_REQUIRE(capacity >= n, LDW_ERR_SIZE, "ldw_links_fetch: capacity %lld < %lld rows", (long long)capacity, (long long)n);
    if (n == 0) return LDW_OK;
    LDW_REQUIRE(a_out && b_out && MI_out, LDW_ERR_ARG, "ldw_links_fetch: null output");
    const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    ldw::DevBuf &A = which == 0 ? c->sr_a : c->lr_a, &B = which == 0 ? c->sr_b : c->lr_b, &M = which == 0 ? c->sr_mi : c->lr_mi;
    LDW_HIP(hipMemcpyAsync(a_out, A.p, (size_t)n * 4, k, c->stream));
    LDW_HIP(hipMemcpyAsync(b_out, B.p, (size_t)n * 4, k, c->stream));
    LDW_HIP(hipMemcpyAsync(MI_out, M.p, (size_t)n * 8, k, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_links_import(ldw_ctx *c, int which, const int32_t *a, const int32_t *b, const double *MI, int64_t n, int on_device) {
    if (int rc = check_gpu(c)) return rc;
    if (int rc = join_prepare(c)) return rc;
    LDW_REQUIRE(which == 0 || which == 1, LDW_ERR_ARG, "ldw_links_import: which must be 0 (sr) or 1 (lr)");
    LDW_REQUIRE(n >= 0 && (n == 0 || (a && b && MI)), LDW_ERR_ARG, "ldw_links_import: bad argument");
    LDW_REQUIRE(c->blk_capacity == 0, LDW_ERR_STATE, "ldw_links_import: a link pass is still open (ldw_links_end)");
    if (c->gemm_stream) LDW_HIP(hipStreamSynchronize(c->gemm_stream));
    ldw::DevBuf &A = which == 0 ? c->sr_a : c->lr_a, &B = which == 0 ? c->sr_b : c->lr_b, &M = which == 0 ? c->sr_mi : c->lr_mi;
    if (int rc = A.reserve((size_t)std::max<int64_t>(n, 1) * 4)) return rc;
    if (int rc = B.reserve((size_t)std::max<int64_t>(n, 1) * 4)) return rc;
    if (int rc = M.reserve((size_t)std::max<int64_t>(n, 1) * 8)) return rc;
    if (n > 0) {
        const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        LDW_HIP(hipMemcpyAsync(A.p, a, (size_t)n * 4, k, c->stream));
        LDW_HIP(hipMemcpyAsync(B.p, b, (size_t)n * 4, k, c->stream));
        LDW_HIP(hipMemcpyAsync(M.p, MI, (size_t)n * 8, k, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
    }
    (which == 0 ? c->n_sr : c->n_lr) = n;
    c->n_red = c->n_pool = 0;   // whatever was derived from the old table is stale:
    c->stats.clear();           // the per-block records of the pass that made it (ldw_block_stats answers LDW_ERR_ARG until the next pass) ...
    c->multi_owner.clear();     // ... and the deal of an in-process pass whose shares these rows replace
    return LDW_OK;
}

int ldw_block_stats(ldw_ctx *c, int64_t nblocks, int64_t *n_lr_total, int64_t *n_lr_kept, int64_t *n_sr,
                    double *disc_thresh) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    LDW_REQUIRE((int64_t)c->stats.size() == nblocks, LDW_ERR_ARG, "ldw_block_stats: last call processed %lld blocks, not %lld",
                (long long)c->stats.size(), (long long)nblocks);
    for (int64_t b = 0; b < nblocks; ++b) {
        if (n_lr_total) n_lr_total[b] = c->stats[b].n_lr_total;
        if (n_lr_kept) n_lr_kept[b] = c->stats[b].n_lr_kept;
        if (n_sr) n_sr[b] = c->stats[b].n_sr;
        if (disc_thresh) disc_thresh[b] = c->stats[b].disc_thresh;
    }
    return LDW_OK;
}

}  // extern "C"
