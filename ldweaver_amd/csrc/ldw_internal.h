// Internal declarations shared by the translation units of libldweaver_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ldweaver_amd.h"

// Pipeline slots of the all-pairs loop: the block-wide kernels (GEMM stream) may run LDW_NSLOT - 1 blocks ahead of the block whose
// lists the main stream is evaluating; every per-block buffer exists once per slot (block b uses slot b % LDW_NSLOT).
#define LDW_NSLOT 3

// LDW_EXPERIMENTS (make EXPERIMENTS=1 -> libldweaver_amd_exp.so): the measured-slower variants kept as the record of what was tried — the
// pipelined and LDS-shared approximate GEMMs, the fused GEMM + epilogue kernel, the byte-state histogram kernel, corner spans, the split
// diagonal blocks, the list-driven screen — and the environment switches that select them.  The DEFAULT library holds none of it: the
// entry points that would select a variant return LDW_ERR_STATE, the switches are not read (ldw::exp_env).
#ifdef LDW_EXPERIMENTS
#define LDW_HAS_EXPERIMENTS 1
#else
#define LDW_HAS_EXPERIMENTS 0
#endif

namespace ldw {

// an environment switch of the experiments build (null in the default library, whatever the environment says)
inline const char *exp_env(const char *name) {
#ifdef LDW_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define LDW_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return ldw::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define LDW_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            ldw::set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // returns LDW_OK / error; contents NOT preserved on growth
    int reserve_keep(size_t bytes, size_t used, hipStream_t s);  // preserves the first `used` bytes
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

constexpr int TILE = 128;     // GEMM block tile (rows on both sides)
constexpr int KSTEP = 128;    // bytes of K (sequences) per pipeline stage
constexpr int NBINS = 4096;   // level-1 histogram bins of the lr quantile search

struct BlockStat {
    int64_t n_lr_total = 0, n_lr_kept = 0, n_sr = 0;
    double disc_thresh = 0;
};
// what the host knew / learnt about a block (LDW_BLOCK_TRACE=1 prints one line per block at ldw_links_end)
struct BlockTrace {
    int diag = 0, guess = -1, B_true = -1, margin = 0, path = 0 /* 0 plain, 1 mixed, 2 apx, 3 fused */, missed = 0, probed = 0;
    long long n_cand = 0;
};

}  // namespace ldw

struct ldw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double last_ms[4] = {0, 0, 0, 0};
    int engine = LDW_ENGINE_MFMA;

    // ---- alignment ----
    int64_t L = 0, N = 0, Npad = 0;  // Npad: N rounded up to a multiple of KSTEP (128)
    int64_t KW = 0;                  // 64-bit words per bit row = Npad / 64
    ldw::DevBuf states;              // uint8 [L][N]
    ldw::DevBuf chars;               // raw alignment characters [cN][cL] kept by ldw_alignment_scan
    int64_t cN = 0, cL = 0;

    // ---- weights ----
    bool have_weights = false;
    int nlimbs = 5;
    int frac_bits = 0;
    double neff = 0;            // sum(hdw) like R's sum()
    int64_t total_fixed = 0;    // sum of quantised weights
    ldw::DevBuf digits;         // int8 [nlimbs][Npad] balanced base-256 digits of V_s
    ldw::DevBuf vfixed;         // int64 [Npad] quantised weights V_s (0 in the padding)
    std::vector<int64_t> h_vfixed;
    // mixed-precision path (nlimbs == 5): V = V_hi * 2^16 + V_lo, V_hi = limbs 2..4
    std::vector<int64_t> h_vfixed_hi;
    int64_t total_fixed_hi = 0;
    double lo_abs_sum = 0;      // sum_s |V_lo,s| * 2^-F: bound of the low limbs' contribution to any joint sum
    int mixed = 1;              // 1: high-limb GEMM + gathered low-limb GEMM for the listed units in speculative blocks

    // ---- sequence order of the bit rows: position p of a row = sequence seq_perm[p]; positions ascend in weight, so
    // ---- sequences of equal weight (a weight CLASS) are contiguous.  digits / dig_a / dig_b are indexed by position.
    std::vector<int32_t> h_seq_perm;   // [Npad], -1 in the padding
    ldw::DevBuf seq_perm;
    // ---- approximate-GEMM path (ldw_apx.hip): V_p ~ a_p * b_p * 2^e(macro step of p), one int8 MFMA pass with both
    // ---- operands masked by a digit; exact joint sums of the listed units by class-wise popcounts
    int path_mode = 0;                 // ldw_set_path: 0 auto, 1 mixed/plain path, 2 force the approximate path
    int select_mode = 0;               // ldw_set_select: 0 auto (sort-free selection where it applies), 1 always the two radix sorts
    bool apx_ok = false;               // the weights allow the approximate path (precision and class structure)
    bool apx_fine = false;             // block exponents per MFMA k-step (32 positions) instead of per macro step (128): weights with a large dynamic range
    std::string apx_gate = "weights not set";   // "ok" or the gate that keeps the path off (ldw_path_report)
    ldw::DevBuf dig_a, dig_b;          // uint8 [Npad]
    ldw::DevBuf apx_shift;             // int32 [2 KW]: right shift of the accumulators before MFMA k-step k (32 positions)
    int apx_e_last = 0;                // accumulators end in units of 2^apx_e_last (fixed-point units of V)
    int apx_transitions = 0;           // k-steps with a shift: each loses < 1 unit (of ITS exponent) of a joint sum
    double apx_lost_units = 0;         // what that adds up to in units of 2^apx_e_last (< 2: the exponents ascend)
    double apx_delta = 0;              // max_p |V'_p - V_p| / V_p
    std::vector<int64_t> h_vapx;       // [Npad] by SEQUENCE: V'_s = a b 2^e (exact integer)
    ldw::DevBuf slot_papx;             // int64 [L][5] by slot: floor(marginal of V' / 2^apx_e_last)
    ldw::DevBuf pop_segs, pop_wbeg;    // popcount segments (PopSeg) and first segment of every 32-bit word (+1)
    ldw::DevBuf pop_vpos;              // r05: int64 [Npad] fixed-point weight by POSITION (k_pair_sums_bits: weightings with many classes)
    bool pair_bits_attr = false;       // the > 64 KB dynamic-LDS attribute of k_pair_sums_bits has been set (this context's device)
    int n_pop_segs = 0, n_classes = 0;
    ldw::DevBuf panel[LDW_NSLOT][2];           // [slot][from, to]: packed bit panels of a block's row lists, [KW/2][Rpad][2] u64
    ldw::DevBuf Gapx[LDW_NSLOT];               // int32 [RTpad][RFpad] approximate joint sums, one per pipeline slot
    ldw::DevBuf tab11[2];              // threshold tables of the biallelic pairs (k_build_tab11), int2 [nb][nb]: [off-diagonal, diagonal] blocks
    double tab11_lo[2] = {0, 0};       // MI level each was built for (0: none)
    int tab11_cur[2] = {0, 0};         // r04: each kind's buffer holds a RING of 4 tables; this one is current (a rebuild takes the next: items in flight keep theirs)
    int64_t tab11_builds = 0;
    float tab11_c = 0;
    int tab11_nb = 0;
    bool tab11_on = true;              // LDW_NO_TAB11 switches the table off (A/B measurements)
    ldw::DevBuf pair_sums;             // exact joint sums of the listed pairs (16 per pair)
    ldw::DevBuf pairs[LDW_NSLOT];              // per pipeline slot: pair lists of the approximate screen (counters + PAIR_PATHS x PAIR_SHARDS lists)
    ldw::DevBuf sub_units[LDW_NSLOT], sub_packs[LDW_NSLOT], sub_bins[LDW_NSLOT], sub_live[LDW_NSLOT];   // the same four for an item's SR sub-pass
    ldw::DevBuf scr_live[LDW_NSLOT];           // per slot: counter, per-tile summaries and the list of the (tile, column group) combinations the screen has work for
    ldw::DevBuf apx_bins[LDW_NSLOT], apx_clean[LDW_NSLOT];    // per slot: threshold-table bin of every row of the two row lists; clean-region flags of the GEMM epilogue
    ldw::DevBuf apx_mini[LDW_NSLOT];           // per slot: the 32-byte per-SNP extracts k_screen_maybe reads (MiniCol [nt], MiniRow [64 * from-tiles])
    ldw::DevBuf apx_units[LDW_NSLOT], apx_packs[LDW_NSLOT];   // per slot: per-(tile, class) unit lists + counters; per-block SNP constants
    int64_t apx_blocks = 0, apx_units_listed = 0, apx_pairs_listed = 0, probe_blocks = 0, generic_blocks = 0;
    // Tile pruning (docs/HISTORY.md 5.1d): in blocks without a short-range pair the one-row SNPs are ordered by the weight of their minor
    // state, so that a wave tile of the approximate GEMM spans few bins of the threshold table; a tile whose whole bin rectangle is
    // unconditionally below the level (no joint count can lift such a pair to it) is flagged clean without being computed.
    bool prune = true;                 // LDW_NO_PRUNE switches ordering and skipping off (A/B measurements)
    ldw::DevBuf apx_skip;              // uint64: wave tiles the GEMM skipped since the counter was last read
    double apx_ops_per_wave = 0;       // executed-operation accounting of the skipped tiles (ldw_gemm_stats)
    int64_t apx_waves_skipped = 0, apx_waves_total = 0;
    // Spans (r04, docs/HISTORY.md 6b): consecutive long-range-only blocks of one block row run as ONE launch sequence over their concatenated to side
    bool span_on = true;               // ldw_set_span / LDW_NO_SPAN
    bool diag_split = false;           // ldw_set_span(on | 4) / LDW_DIAG_SPLIT: diagonal blocks as SR sub-pass + weight-ordered long-range pass
    bool span_corners = false;         // ldw_set_span(on | 2) / LDW_SPAN_CORNERS: corner blocks join the spans (SR sub-passes); measured slower, off by default
    int span_max = 8;                  // most reference blocks per span (LDW_SPAN_MAX env, <= ldw::LDW_SPAN_MAX)
    int64_t maybe_entries = 0;           // entries handed to the maybe list since the context was created (ldw_overflow_report out[3])
    int64_t pair_list_overflows = 0, maybe_overflows = 0;   // blocks / segments redone because a pair list / the maybe list overflowed (ldw_overflow_report)
    bool maybe_off = false;              // the maybe list overflowed in this pass: off until ldw_reset_speculation / new weights
    int64_t span_items = 0, span_blocks = 0, span_fallbacks = 0;   // spans run, reference blocks they covered, segments redone non-speculatively
    int64_t span_sr_subs = 0;          // SR sub-passes run for corner segments of spans
    bool early_sr = false;             // this pass assigns an item's short-range rows when the item is SUBMITTED (submit order = block order), not in its second phase
    std::atomic<int64_t> sorted_blocks{0};   // (prep_block runs on the helper thread and, for the cold-start probes, on the calling thread at once)
    std::mutex order_mtx;                    // guards order_cache
    // Per-SNP bound behind the pruning of the wider tables (k_snp_sup): snp_sup[a * 4 + 2 * m + (k - 2)] = the largest MI SNP a (2 or
    // 3 states, all flagged, r = its number of states) can reach with ANY partner of k = 2 or 3 flagged states and r = k, in the
    // intended (m = 0) and the reference (m = 1: RXY at its floor r_min^2 / 4) reading of RXY; +inf for other SNPs.
    ldw::DevBuf snp_sup;
    double r_min = 0;                  // smallest r of the alignment (ldw_set_snp_meta)
    struct OrderCache { std::vector<int32_t> idx, order; };
    std::vector<std::unique_ptr<OrderCache>> order_cache;   // per SNP list: its one-row SNPs in ascending order of the minor state's weight

    // ---- per-SNP meta ----
    bool have_meta = false;
    double g = 0;
    ldw::DevBuf r, uqe, POS, paint;  // double[L], uint8[L][5], int32[L], int32[L]
    std::vector<double> h_r;
    std::vector<int32_t> h_POS, h_paint;
    int32_t paint_min = 0, paint_max = 0;
    bool pos_sorted = true;      // POS ascends over the whole alignment (the reference's parser emits it so; any order is accepted)
    double sr_total_dist = -1;   // number of SNP pairs within sr_total_dist on the circle (sizes the short-range table once; reset with the meta data)
    int64_t sr_total = -1;
    uint64_t sr_share_key = 0;   // r05: the last SHARE of the block list a pass was sized for (hash of blocks + sr_dist) and its exact short-range row count
    int64_t sr_share_rows = -1;

    // ---- row map (built lazily from alignment + weights + meta) ----
    bool rows_ready = false;
    int64_t R = 0;               // number of indicator rows over all SNPs
    ldw::DevBuf Mbits;           // uint64 [R + TILE][KW]: bit s set where sequence s carries the row's state
    ldw::DevBuf row0;            // int32 [L+1]: first row of each SNP
    ldw::DevBuf slot_meta;       // uint32 [L]: nrows (3 bits) | uq-by-slot (5 bits <<3) | slot states (5 x 3 bits << 8)
    ldw::DevBuf slot_pfix;       // int64 [L][5]: fixed-point marginal of the state in each slot
    ldw::DevBuf slot_pfix_hi;    // the same for the high-limb weights V_hi (nlimbs == 5), built on demand
    bool hi_ready = false;
    ldw::DevBuf glo;             // int32: low-limb joint sums of the listed units (gathered GEMM)
    ldw::DevBuf lo_rows;         // int32 row lists of the gathered GEMM's workgroups
    ldw::DevBuf packs;           // per-block SNP constants in epilogue order (ColMeta / RowPack arrays, k_build_packs)
    ldw::DevBuf counts;          // int32 [L][5] per-SNP state counts
    ldw::DevBuf pfix_state;      // int64 [L][5]: fixed-point marginal of each state (histogram engine)
    std::vector<int32_t> h_row0;
    std::vector<uint32_t> h_slot_meta;
    std::vector<int32_t> h_counts;
    std::vector<int64_t> h_minor_w;   // [L]: fixed-point weight of slot 0 of a biallelic r = 2 SNP with one row (else INT64_MAX)
    std::vector<int32_t> h_span_bad;  // [L + 1] prefix count of the SNPs a span cannot hold: no indicator row, or 1-2 rows with an unflagged slot (ensure_rows)

    // ---- per-block workspaces ----
    ldw::DevBuf G, G2, G3;       // int64 [RTpad][RFpad] fixed-point joint sums, one per pipeline slot (gx())
    ldw::DevBuf MIblk;           // double [nf*nt]
    ldw::DevBuf rowlist_f, rowlist_t, idx_f, idx_t, lrow_f, lrow_t, perm_f, perm_t;
    ldw::DevBuf epi_rest;        // plain path's epilogue: counter (64-B slot) + list of the units k_mi_epilogue_fast left to k_mi_epilogue_rest
    ldw::DevBuf scr_units;       // uint32 count (64-B slot) + list of the block's units the fp32 screen wants evaluated exactly
    ldw::DevBuf hist[LDW_NSLOT], cand_key[LDW_NSLOT], cand_val[LDW_NSLOT];   // per pipeline slot: histogram of the lr candidates, candidate list
    ldw::DevBuf colcnt, cand_key2, cand_val2, scratch, small;
    ldw::DevBuf miss_key, miss_val;   // candidate list of a span's segment that is redone on its own (the slot's lists hold its neighbours')
    std::vector<uint8_t> ev_valid;    // per block of the pass: its stage events were recorded (the segments of a span share the first one's)
    ldw::DevBuf sel_bitmap, sel_chunks, sel_prefix;   // fast selection (k_sel_thresh): bitmap, chunk and super-chunk (sel_prefix) counters, all-zero between blocks

    // ---- link tables (device resident) ----
    ldw::DevBuf sr_a, sr_b, sr_mi, lr_a, lr_b, lr_mi;
    int64_t n_sr = 0, n_lr = 0;
    int64_t blk_capacity = 0, blk_cursor = 0;  // ldw_links_begin / ldw_mi_block_links / ldw_links_end

    // ---- short-range model and ARACNE on the device-resident sr table (ldw_srp.hip) ----
    void *lr_stream = nullptr;   // r05: lr_links.tsv appended while the pass runs (ldw_tsv.cpp: LrStream), if ldw_lr_stream_begin opened one
    // its device-side resources, made once with the context's other streams (hipStreamCreate is a 12-ms call: not inside a job's pass)
    hipStream_t lr_st = nullptr;
    hipEvent_t lr_ev[64] = {};
    int64_t *lr_counts = nullptr;   // pinned [64]
    void *lr_pin = nullptr;         // pinned staging of a batch of rows (a, b, MI); grows on demand
    size_t lr_pin_cap = 0;
    void *tsv_async = nullptr;   // r04: the asynchronous link-table writer, if one is running (ldw_tsv.cpp: TsvAsync)
    void *pin_fetch = nullptr;   // r04: pinned host arena the tsv writer fetches a link table into (see ldw_write_links_tsv)
    size_t pin_fetch_cap = 0;
    ldw::DevBuf srm_pack, srm_key, srm_pack2, srm_key2, srm_pay, srm_pay2, srm_off, srm_q, srm_n, srm_md, srm_part, srm_shape, srm_cnt, srm_tmp;
    ldw::DevBuf srd_lower, srd_cur, srd_out, srd_seg;   // r05, the model over ranks (ldw_sr_tail_extract ...): per-group bounds, cursors, the extracted rows, block segments
    double ham_stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ldw_hamming_stats: columns, padded K, stage times and algorithmic bytes of the last ldw_hamming_weights
    double gemm_stat[6] = {0, 0, 0, 0, 0, 0};   // ldw_gemm_stats: launches and executed int8 ops of the block-wide GEMMs
    ldw::DevBuf red_row, red_meta, red_srp, pool_a, pool_b, pool_mi, ar_key, ar_val, ar_key2, ar_val2, ar_off, ar_flags;
    int64_t n_red = 0, n_pool = 0;
    bool red_from_lr = false;    // red_row indexes the long-range table (ldw_lr_tukey) instead of the short-range one
    int srm_S = 0, srm_nclust = 0;   // geometry of the last ldw_sr_len_quantiles call
    double srm_sr_dist = 0;

    // ---- pipelined block staging: host prep of block i+1 overlaps the GPU work of block i ----
    hipStream_t copy_stream = nullptr, gemm_stream = nullptr;
    hipEvent_t ev_gemm[LDW_NSLOT] = {};
    bool overlap = true;                 // GEMM of block b+1 on its own stream beside the epilogue/selection of block b
    void *pin[LDW_NSLOT + 1] = {};   // pinned host staging, one packed buffer per slot (+ 1: a span segment that is redone on its own)
    size_t pin_cap[LDW_NSLOT + 1] = {};
    ldw::DevBuf dstage[LDW_NSLOT + 1];               // device image of the packed buffer
    hipEvent_t ev_up[LDW_NSLOT] = {}, ev_done[LDW_NSLOT] = {};
    bool done_recorded[LDW_NSLOT] = {};
    bool up_recorded[LDW_NSLOT] = {};    // ev_up[slot] has been recorded at least once
    void *pin_pick[LDW_NSLOT] = {};   // pinned landing zone of the per-block PickOut, one per slot
    hipEvent_t ev_pick[LDW_NSLOT] = {};
    void *pin_lrc = nullptr;             // pinned copy of the running long-range row count
    hipEvent_t ev_lrc = nullptr;
    bool lrc_recorded = false;
    bool fused = false;                  // GEMM + epilogue in one kernel whenever a bucket guess exists (ldw_fused.hip); off:
                                         // GEMM -> k_mi_screen -> k_mi_units, which measures 7 % faster on C4 (docs/HISTORY.md 5.2)
    bool spec_seen[2] = {false, false};  // a block of this kind (off-diagonal, diagonal) has set its own guess
    bool spec_probed[2] = {false, false};   // the kind's current guess came from a cold-start probe of the kind itself
    std::vector<hipEvent_t> ev_pool;     // 4 timing events per block
    int spec_hist[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};   // the last true buckets per kind (adaptive margin of the guess)
    int spec_hist_n[2] = {0, 0};
    bool spec_small[2] = {false, false};   // the kind's blocks keep few rows (< 5000): noisier thresholds, wider margins
    int spec_B_next[2] = {-1, -1};       // bucket guess for the speculative long-range gather: [off-diagonal, diagonal]
    int64_t spec_misses = 0, fused_blocks = 0, unfused_blocks = 0, screen_violations = 0, mixed_blocks = 0;
    int screen = 1;                      // fp32 screen in front of the fp64 MI evaluation (0 off, 1 on, 2 verify)
    std::vector<ldw::BlockStat> stats;
    std::vector<int32_t> multi_owner;    // r05: ctx[0] of ldw_mi_all_pairs_multi(.., LDW_MI_SR_ROWS_STAY): the deal (owner of every block); empty otherwise
    std::vector<ldw::BlockTrace> trace;
    // ldw_ctx_reserve (r04): what a job's FIRST pass otherwise pays inside its timed loop — two hipStreamCreate (12 ms each on this box), the
    // pinned staging buffers (hipHostMalloc: 0.2 ms per MB), the lazy load of the code objects of the pass's kernels — done by a side thread
    // while the caller uploads the alignment and runs the Hamming GEMM.  Joined by every entry point that uses what it prepares.
    std::thread *prep_thread = nullptr, *prep_thread2 = nullptr;   // started by ldw_ctx_create (streams, code objects) / ldw_ctx_reserve (staging buffers)
    int prep_rc = 0, prep_rc2 = 0;
    std::string prep_err, prep_err2;
};

namespace ldw {
// launchers implemented in the .hip files (all asynchronous on ctx->stream)
int ensure_rows(ldw_ctx *ctx);
int launch_hist(ldw_ctx *ctx, const int32_t *idx_f, int nf, const int32_t *idx_t, int nt, const int64_t *pfix_state,
                int quirk, int lower_only, double *MI);
// G[t][f] = sum_k [row t has bit k][row f has bit k] * sum_j digits[j][k] 256^j over the bit matrix Mbits[rows][KW words]
int launch_gemm_bits(ldw_ctx *ctx, const uint64_t *Mbits, int64_t KW, const int32_t *rowlist_t, int RTpad, const int32_t *rowlist_f,
                     int RFpad, int64_t *G, int nlimbs, const int8_t *digits, int lower_only, hipStream_t stream = nullptr, int by0 = 0,
                     int by1 = -1,    // by0..by1: strip of 128-row to-side tiles to compute (default: all)
                     const uint8_t *tile_mask = nullptr,    // [RTpad / 128][RFpad / 64]: 0 = skip the tile (default: all tiles)
                     const uint32_t *tile_list = nullptr, int n_tile_list = 0);   // r06: the same tiles as a list (by << 16 | bx): a 1-D grid of exactly those
// the same exact sums (all limbs) by class-wise popcounts over the bit rows (ldw_hist.hip, LDW_ENGINE_HIST)
int launch_cooc_popc(ldw_ctx *ctx, const int32_t *rowlist_t, int RTpad, const int32_t *rowlist_f, int RFpad, int64_t *G, int lower_only,
                     hipStream_t stream = nullptr);
int fill_rows_bits(ldw_ctx *ctx, const int32_t *d_rowinfo, int64_t R);
int prepare_apx_weights(ldw_ctx *ctx);   // ldw_apx.hip: dual digits, exponents, popcount segments from h_vfixed / h_seq_perm
int check_gpu(ldw_ctx *ctx);
int ensure_hi_marginals(ldw_ctx *ctx);   // slot_pfix_hi on demand (mixed-precision path)
void lr_stream_push(ldw_ctx *ctx, hipStream_t s, const int64_t *d_lr_count, int64_t blocks_done);   // ldw_tsv.cpp: no-ops without an open stream
void lr_stream_drain(ldw_ctx *ctx);
int join_prepare(ldw_ctx *ctx);      // waits for the side thread of ldw_ctx_reserve (no-op without one); its error, if any, becomes the caller's
int reserve_slot_buffers(ldw_ctx *ctx, int64_t Npad, int64_t blk, int64_t nseg);   // ldw_mi.hip: per-slot device buffers from the block geometry
int ensure_streams(ldw_ctx *ctx);    // the copy / GEMM streams, per-slot events and pinned pick records of the all-pairs loop (once per context)
void warm_mi();                      // lazy code-object loads of the translation units whose kernels a pass launches (hipFuncGetAttributes)
void warm_apx();
void warm_gemm_bits();
void ctx_count(int d);               // ldw_api.hip: live contexts of the process (the last one to go trims the device free list to its idle cap)
struct DrainedScope { DrainedScope(); ~DrainedScope(); };   // releases inside: the caller has drained every stream that could touch the blocks
int launch_state_counts(ldw_ctx *ctx);   // ldw_api.hip: per-SNP state counts into ctx->counts on the context's stream (no copy, no synchronisation)
size_t device_pool_trim();          // ldw_api.hip: give the released device blocks kept for re-use back to the runtime (ldw_host_trim); bytes
void warm_srp();
// ldw_srp.hip: ldw_sr_reduced_import with the kept links' meta words (clust_c | first << 8 | dup << 16) and srp values (both may be null)
int reduced_import_full(ldw_ctx *ctx, int64_t n_red, const int32_t *a, const int32_t *b, const double *MI, const uint32_t *meta, const double *srp, int64_t n_pool,
                        const int32_t *pool_a, const int32_t *pool_b, const double *pool_MI);
void warm_post();
}  // namespace ldw
