/*
 * ldweaver_amd.h — C ABI of the MI355X-native all-pairs weighted-MI engine.
 *
 * This is the drop-in boundary for the ONE hot path of Sudaraka88/LDWeaver:
 *   perform_MI_computation()            R/computePairwiseMI.R:46-145
 *   estimate_Hamming_distance_weights() R/performPopulationStuctureCorrection.R:20-81
 *   .ACGTN2num()                        src/ACGTN2num_parallel.cpp:10-43
 * and the native helpers they call through `.Call` (src/RcppExports.cpp:154-167).
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes (no R, Rcpp or
 * torch types), returns an int status (0 = LDW_OK) and leaves a thread-local
 * message retrievable with ldw_last_error().  The caller allocates every output
 * (variable-length link tables use a two-call size query).  The R-side binding a
 * maintainer would add is shown in INTEGRATION.md and kept as source in r_shim/.
 *
 * Conventions
 *   states  uint8 [L][N] row-major, values 0..4 = A,C,G,T,N: the dense equivalent of the five
 *           one-hot sparse matrices of `snp.dat` (R/extractSNPs.R:138-141), encoded by the rule of
 *           src/getACGTNsites.cpp:229-265.
 *   SNP indices in block descriptors are 1-based inclusive like make_blocks()
 *           (R/computePairwiseMI.R:147-165); index arrays are 0-based.
 *   MI blocks are column-major nf x nt doubles, element (a,b) at a + b*nf, exactly the R matrix
 *           `MI` of perform_MI_computation_ACGTN (R/computePairwiseMI.R:268).
 */
#ifndef LDWEAVER_AMD_H
#define LDWEAVER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDW_OK 0
#define LDW_ERR_ARG 1      /* bad argument (shape, range, null pointer)            */
#define LDW_ERR_HIP 2      /* a HIP runtime call or kernel launch failed            */
#define LDW_ERR_STATE 3    /* call order violated (e.g. MI before weights are set)  */
#define LDW_ERR_NOGPU 4    /* no usable gfx950 device: there is NO CPU fallback     */
#define LDW_ERR_SIZE 5     /* caller buffer too small (see the size query)          */

/* quirk modes of the block kernel */
#define LDW_QUIRK_REFERENCE 0 /* reproduce Q1: RXY read by linear index of the nt x nf matrix (R/computePairwiseMI.R:261 + src/computeMI.cpp:19) */
#define LDW_QUIRK_INTENDED 1  /* RXY = 0.25*r_a*r_b */

#define LDW_ENGINE_MFMA 0 /* i8 MFMA fixed-point co-occurrence GEMM + fp64 epilogue (default) */
#define LDW_ENGINE_HIST 1 /* joint histograms on bit planes: LDS-tiled class-wise popcounts (VALU), exact int64 sums, same fp64 epilogue and results */
#define LDW_ENGINE_HIST_STATES 2 /* the first histogram kernel: byte states in LDS, 25 sums per pair updated sequence by sequence (independent cross-check; ~200x slower) */

typedef struct ldw_ctx ldw_ctx;

/* ---- library / device ------------------------------------------------------------------ */
int ldw_version(void);
/* bit 0: built with -DLDW_EXPERIMENTS (make EXPERIMENTS=1): the measured-slower kernel variants and their environment switches are in the
 * library (the fused GEMM + epilogue kernel, LDW_ENGINE_HIST_STATES, corner spans / split diagonal blocks of ldw_set_span, the alternative
 * approximate GEMMs, the list-driven screen).  The default library returns 0, holds none of them and answers LDW_ERR_STATE where one is asked for. */
int ldw_build_info(void);
const char *ldw_last_error(void);
/* number of visible HIP devices (0 when none); never initialises a context */
int ldw_device_count(void);

int ldw_ctx_create(int device, ldw_ctx **out);
int ldw_ctx_destroy(ldw_ctx *ctx);
/* run everything on an externally owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int ldw_ctx_set_stream(ldw_ctx *ctx, void *hip_stream);
int ldw_ctx_sync(ldw_ctx *ctx);
/* elapsed ms of the kernels of the last ldw_mi_block / ldw_mi_all_pairs call, by stage, measured with
 * HIP events on the context's stream: [0] gemm, [1] epilogue, [2] selection, [3] total */
int ldw_ctx_last_timing(ldw_ctx *ctx, double ms_out[4]);

/* diagnostics since the context was created: out[0] = blocks whose speculative long-range gather had to fall back to
 * the dense pass, out[1] = blocks run by the fused GEMM + epilogue kernel, out[2] = blocks run by the two-kernel path,
 * out[3] = pairs the fp32 screen would have lost (counted in ldw_set_screen mode 2 only; must stay 0) */
int ldw_ctx_counters(ldw_ctx *ctx, int64_t out[4]);
/* the same four, then out[4] = blocks run in the mixed-precision path (ldw_set_mixed), out[5] = blocks run in the
 * approximate-GEMM path (ldw_set_path), out[6] = units its screen listed (they hold a short-range pair), out[7] = long-range candidate
 * pairs its screen listed */
int ldw_ctx_counters2(ldw_ctx *ctx, int64_t out[8]);
/* Work the block-wide GEMMs of this context EXECUTED since the last reset (for the roofline: executed int8 operations /
 * kernel time / peak): out[0] launches and out[1] int8 operations (2 x rows x rows x positions of the wave tiles that do not exit
 * at once) of the approximate GEMM (gemm_apx_kernel), out[2] / out[3] the same for the unmasked limb GEMM (gemm_bits_kernel<J>,
 * all J limbs), out[4] launches of the band-masked limb GEMM, out[5] launches of the approximate GEMM that applied the threshold
 * table in their epilogue (long-range-only blocks).  reset != 0 clears the counts. */
int ldw_gemm_stats(ldw_ctx *ctx, double out[6], int reset);

/* ---- (1) .ACGTN2num  — src/ACGTN2num_parallel.cpp:10-43, R/RcppExports.R:4-6 ------------ */
/* nv: 5 x L doubles, column-major, mutated IN PLACE (host memory, as R hands it over);
 * ref: L bytes = first character of each element of `cv`; ncores is accepted and ignored. */
int ldw_acgtn2num(ldw_ctx *ctx, double *nv, const char *ref, int64_t L, int ncores);
/* same on device-resident buffers (no copies) */
int ldw_acgtn2num_dev(ldw_ctx *ctx, double *nv_dev, const char *ref_dev, int64_t L);

/* ---- (3) .fastHadamard — src/computeMI.cpp:11-21, R/RcppExports.R:8-10 ------------------- */
/* element-wise twin over the linear index c < n; MI updated in place. on_device != 0: all pointers
 * are device pointers. */
int ldw_fast_hadamard(ldw_ctx *ctx, double *MI, const double *den, const double *uq, const double *pxy,
                      const double *pxpy, const double *RXY, const double *pXrX, const double *pYrY,
                      int64_t n, int on_device);

/* ---- alignment residency ------------------------------------------------------------------ */
/* Upload (on_device == 0) or adopt a copy of (on_device != 0) the L x N state matrix. */
int ldw_set_alignment(ldw_ctx *ctx, const uint8_t *states, int64_t L, int64_t N, int on_device);
/* 5-state encoder of src/getACGTNsites.cpp:229-265 on the device: chars [N][L_total] (sequence-major,
 * as a FASTA holds them) -> states [n_pos][N] for the 1-based retained columns pos[n_pos]; the result
 * becomes the context's alignment.  Also returns the 5 x n_pos ACGTN_table (may be NULL). */
int ldw_encode_alignment(ldw_ctx *ctx, const char *chars, int64_t N, int64_t L_total, const int32_t *pos,
                         int64_t n_pos, int32_t *acgtn_table_out);
/* First half of `extractAlnParam` (src/getACGTNsites.cpp:47-90): upload the raw alignment chars [N][L_total]
 * (they stay resident: a following ldw_encode_alignment may pass chars = NULL) and count A/a, C/c, G/g, T/t and
 * "everything else" per column: allele_counts_out is 5 x L_total int32, column-major like `allele_counts`.
 * The SNP filter itself (:104-166) is O(L_total) host logic on these counts. */
int ldw_alignment_scan(ldw_ctx *ctx, const char *chars, int64_t N, int64_t L_total, int32_t *allele_counts_out);
/* per-SNP state counts (5 x L, column-major like ACGTN_table) of the resident alignment */
int ldw_state_counts(ldw_ctx *ctx, int32_t *counts_out);
/* copy the resident states back (tests) */
int ldw_get_alignment(ldw_ctx *ctx, uint8_t *states_out);

/* ---- (2) estimate_Hamming_distance_weights — R/performPopulationStuctureCorrection.R:20-81 */
/* hdw_out[j] = 1 / (#{i : L - shared[i][j] < thresh} + 1), thresh = as.integer(L*threshold) computed by
 * the caller.  shared_out (N x N int32, may be NULL) receives the exact shared-state counts. */
int ldw_hamming_weights(ldw_ctx *ctx, int32_t thresh, double *hdw_out, int32_t *shared_out);
/* Sharded form: the N x N comparison is symmetric, so only pairs (t, f), t <= f, are computed; this entry
 * point does the strip of 128-sequence row tiles [tile0, tile1) (tile1 <= ceil(N / 128) rounded to the padding) and
 * returns counts_out[j] = the strip's contribution to #{i : L - shared[i][j] < thresh} for EVERY j in 0..N-1.  The strips
 * of all ranks add up to the full count n_j (self included), hdw[j] = 1 / (n_j + 1): exact integers, so every rank
 * derives bit-identical weights after one all-reduce of N counts. */
int ldw_hamming_counts(ldw_ctx *ctx, int32_t thresh, int32_t tile0, int32_t tile1, int64_t *counts_out);
/* r06 — what the last ldw_hamming_weights of this context did, for its roofline (bench.py `roofline_hamming`): out[0] = bit columns K (one per minor state + one
 * "not the major state" column per multi-allelic SNP: ~1.3 L), [1] = K padded to the GEMM's word pairs, [2] = ms of the kernels in front of the GEMM (column bits, bit
 * transpose, per-sequence counts; HIP events), [3] = ms of the lower-triangular int8 GEMM, [4] = ms of the N x N neighbour count, [5] / [6] = algorithmic bytes
 * of the kernels in front of / behind the GEMM, [7] = wall ms of the whole call on the host (allocations, state counts, column list, uploads included).
 * The GEMM's executed int8 operations are in ldw_gemm_stats (bits_ops). */
int ldw_hamming_stats(ldw_ctx *ctx, double out[8]);

/* ---- MI set-up ------------------------------------------------------------------------------ */
/* Per-sequence weights hdw[N] (R/computePairwiseMI.R:77,89).  The engine uses v_s = fl(sqrt(hdw_s))^2
 * like the reference's sqrt-scaled one-hots, quantised to nlimbs*8-bit fixed point (nlimbs in 1..6,
 * 0 = default 5). */
int ldw_set_weights(ldw_ctx *ctx, const double *hdw, int64_t N, int nlimbs);
/* r[L] (snp.dat$r), uqe[L][5] row-major 0/1 (snp.dat$uqe), POS[L] (snp.dat$POS: any order, like the reference — its own parser emits
 * ascending positions, and blocks whose lists ascend take the fast paths; a block in another order runs the plain path and a
 * predicate-based pair list), paint[L] (cds_var$paint), g genome length (snp.dat$g). */
int ldw_set_snp_meta(ldw_ctx *ctx, const double *r, const uint8_t *uqe, const int32_t *POS,
                     const int32_t *paint, double g);
int ldw_set_engine(ldw_ctx *ctx, int engine);

/* ---- (4) one block: perform_MI_computation_ACGTN + computeMI_Sprase + fastHadamard fused --- */
/* from_idx[nf], to_idx[nt]: 0-based SNP indices (contiguous ranges for ordinary blocks, arbitrary
 * subsets in SR-only mode, R/computePairwiseMI.R:179-189).  MI_out: nf*nt doubles column-major, host
 * (on_device == 0) or device.  All nf*nt entries are produced, like the reference's MI matrix. */
int ldw_mi_block(ldw_ctx *ctx, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt,
                 int quirk_mode, double *MI_out, int on_device);
/* exact weighted joint tables in fixed point and plain integer joint counts for a list of SNP pairs:
 * counts_out[p][25] (row X of SNP a, column Y of SNP b), fixed_out[p][25] (sum of quantised weights,
 * value = fixed * 2^-frac_bits).  Either output may be NULL. */
int ldw_joint_tables(ldw_ctx *ctx, const int32_t *pair_a, const int32_t *pair_b, int64_t npairs,
                     int64_t *counts_out, int64_t *fixed_out, int *frac_bits_out);

/* ---- (5) the a-5 loop: blocks -> sr / lr link tables --------------------------------------- */
typedef struct ldw_mi_params {
    double sr_dist;          /* R/computePairwiseMI.R:47  default 20000 */
    double lr_retain_links;  /* default 1e6 */
    double lr_links_approx;  /* R/computePairwiseMI.R:94-97, computed by the host (R RNG) */
    int32_t sr_only;         /* perform_SR_analysis_only: lr part skipped */
    int32_t quirk_mode;      /* LDW_QUIRK_* */
    int32_t keep_sr;         /* 0: do not materialise sr links (throughput measurement of lr only) */
    int32_t flags;           /* LDW_MI_* bits (0: none) */
} ldw_mi_params;
/* r05, ldw_mi_all_pairs_multi only: the short-range rows STAY on the contexts that computed them — only the long-range table is assembled in
 * ctx[0] — and the short-range model runs over the contexts (ldw_sr_len_quantiles_multi, ldw_sr_excess_stats_multi, ldw_sr_pvalues_multi below) */
#define LDW_MI_SR_ROWS_STAY 1

/* blocks[nblocks][4] = (from_s, from_e, to_s, to_e), 1-based inclusive, e.g. this rank's share of
 * make_blocks().  Links are appended to the context's device-resident tables in block order and, within
 * a block, in the reference's row order (R/computePairwiseMI.R:306-310).  reset != 0 clears the tables. */
int ldw_mi_all_pairs(ldw_ctx *ctx, const int32_t *blocks, int64_t nblocks, const ldw_mi_params *p,
                     int reset);
/* r05 — the same loop over SEVERAL contexts of this process, one per GPU ("block list, lr prob, device list -> sr table,
 * lr table"; the loop it shards is R/computePairwiseMI.R:103-116).  Every context must hold the same alignment, weights and SNP meta data
 * (ldw_set_alignment / ldw_set_weights / ldw_set_snp_meta on each; ldw_hamming_weights_multi shares the weights' own computation).  The
 * block pairs are dealt over the contexts by cost (ldw_deal_blocks: a diagonal pair counts 3.3 times its pairs — its dense short-range
 * band —, longest first to the least loaded, every context keeps make_blocks order; the same deal as ldweaver_amd/dist.py), each context
 * runs ldw_mi_all_pairs on its share on a worker thread of its own, and the link tables are assembled in ctx[0] in the caller's block order —
 * the order the reference appends in — by peer-to-peer copies (each source's rows over its own xGMI link).  Afterwards ctx[0] is exactly in the
 * state ldw_mi_all_pairs(ctx[0], all blocks) would have left it in — tables, ldw_block_stats over all nblocks — so the short-range model,
 * ARACNE, the post-processing and the tsv writers run on it unchanged; the other contexts keep their own shares.  The long-range filter is
 * per block (:352-358), so the retained set does not depend on n_ctx.  One failing context fails the call (its message is reported).
 * owner_out (nblocks, may be NULL) receives the deal; ms_out (10 doubles, may be NULL): [0] deal + slowest pass, [1] gather, [2..9] the
 * pass of contexts 0..7.  n_ctx = 1 is ldw_mi_all_pairs(ctx[0], ..., reset = 1).  p->flags & LDW_MI_SR_ROWS_STAY: see (7c). */
int ldw_mi_all_pairs_multi(ldw_ctx **ctx, int n_ctx, const int32_t *blocks, int64_t nblocks, const ldw_mi_params *p,
                           int32_t *owner_out, double *ms_out);
/* the deal alone (host only, no context): owner_out[b] = rank of block b */
int ldw_deal_blocks(const int32_t *blocks, int64_t nblocks, int n_ranks, int32_t *owner_out);
/* estimate_Hamming_distance_weights over several contexts holding the same alignment: the symmetric sequence x sequence comparison is
 * cut into strips of 128-sequence row tiles of equal area, one per context (ldw_hamming_counts), the integer neighbour counts are added
 * on the host: hdw_out[N] is bit-identical to ldw_hamming_weights on one context. */
int ldw_hamming_weights_multi(ldw_ctx **ctx, int n_ctx, int32_t thresh, double *hdw_out);
/* The same loop opened up for blocks that are not contiguous index ranges (SR-only mode drops SNPs
 * without a short-range partner before each block, R/computePairwiseMI.R:179-189):
 * ldw_links_begin(capacity in blocks) ; ldw_mi_block_links(...) per block ; ldw_links_end(). */
int ldw_links_begin(ldw_ctx *ctx, int64_t nblocks_capacity);
int ldw_mi_block_links(ldw_ctx *ctx, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt,
                       const ldw_mi_params *p);
int ldw_links_end(ldw_ctx *ctx);
/* which: 0 = short-range, 1 = long-range (after the per-block quantile filter). */
/* on (default): the co-occurrence GEMM of block b+1 runs on a second stream beside the epilogue and link selection
 * of block b (~5 % faster end to end).  off: all kernels of all blocks run back to back on the context's stream, so
 * that the per-stage times of ldw_ctx_last_timing are exclusive kernel times (what bench.py's roofline uses).
 * Results are identical either way. */
int ldw_set_overlap(ldw_ctx *ctx, int on);
/* on: blocks for which a histogram-bucket guess exists (every block but the first of a call sequence) run the
 * co-occurrence GEMM and the MI epilogue as ONE kernel: the joint sums stay in LDS.  off (default): GEMM -> G in HBM ->
 * k_mi_screen -> k_mi_units; measured faster on C4 (122 vs 132 ms per step) because the epilogue hides its latencies
 * with occupancy the fused kernel cannot have.  Link tables are identical either way up to the rounding of MI (<= 1e-15:
 * on diagonal blocks the fused kernel may meet a pair in mirrored roles). */
int ldw_set_fused(ldw_ctx *ctx, int on);
/* Mixed precision (default on; 5 weight limbs, two-kernel path, speculative blocks): the block-wide co-occurrence GEMM runs
 * with the 3 HIGH limbs of the fixed-point weights only — all the fp32 screen needs; its margin is widened by a rigorous
 * bound of what the low limbs can add — and the exact joint sums of the units the screen lists (3-4 % of an off-diagonal
 * block, the short-range band of a diagonal one) get their 2 low limbs from a gathered GEMM over just those rows:
 * sum = (high << 16) + low, the same integers as the 5-limb GEMM.  Every MI that is emitted is computed from exact sums. */
int ldw_set_mixed(ldw_ctx *ctx, int on);
/* Which block-wide pass feeds the screen of the speculative blocks (every block but the first of a call sequence):
 * 0 (default) = the approximate-GEMM path when the weights allow it — ONE int8 MFMA pass with dual-digit block-floating-
 *     point weights (V ~ a b 2^e, rigorous relative error bound in the screen's margin), exact joint sums of the listed
 *     units by class-wise popcounts over the weight classes (sequences of equal weight are contiguous in the bit rows),
 *     exact re-screen, fp64 — else the limb paths; 1 = the limb paths of ldw_set_mixed only; 2 = the approximate path or
 *     LDW_ERR_STATE at block time when the weights do not allow it (too many distinct weights, > 30k sequences).
 * Every MI that is emitted is computed from the exact fixed-point sums either way: the link tables do not depend on it. */
int ldw_set_path(ldw_ctx *ctx, int mode);
/* Long-range selection of the speculative blocks: 0 (default) = without a sort where it applies (radix select of the threshold,
 * bitmap ranks over the row-order key space: ldw_mi.hip k_sel_*), 1 = always the general path (two radix sorts).  Same tables. */
int ldw_set_select(ldw_ctx *ctx, int mode);
/* Forget what earlier passes of this context learnt about the workload — the per-kind histogram-bucket guesses of the long-range
 * threshold, their spread history and the biallelic threshold table — without touching the alignment, the weights or any
 * buffer.  The next ldw_mi_all_pairs then runs as the FIRST pass of a job does (the reference visits every block pair once,
 * R/computePairwiseMI.R:103-116); bench.py calls it before every timed step.  Results never depend on this state. */
int ldw_reset_speculation(ldw_ctx *ctx);
/* Which execution path the blocks of this context took since it was created (a real data set may fail a gate silently):
 * out[0] blocks through the approximate-GEMM path, out[1] through the mixed-precision limb path, out[2] through the plain path
 * (5-limb GEMM + fp64 MI of every pair: blocks without a bucket guess and every block when neither fast path applies),
 * out[3] through the fused kernel, out[4] speculation misses (blocks redone non-speculatively), out[5] blocks whose guess came from
 * the sampled probe of the block itself (cold starts), out[6] pairs listed for exact evaluation, out[7] units listed.
 * gate (capacity bytes, may be NULL) receives a short text: "ok" ("ok (block exponents per 32 positions)" when the weights'
 * dynamic range needs the finer exponents) or which gate keeps the approximate path off
 * ("delta 5.1e-03 > 4e-03", "Npad 40960 > 30720", "popcount segment tables 70000 B > 60000 B of LDS", "weights not set"). */
int ldw_path_report(ldw_ctx *ctx, int64_t out[8], char *gate, int capacity);
/* Tile pruning of the approximate path (default on; LDW_NO_PRUNE in the environment = off).  In a block pair without a short-range
 * pair the order of the rows within a slot class is free, so the biallelic SNPs are ordered by the weight of their minor state;
 * a 128 x 64 wave tile of the approximate GEMM whose rectangle of threshold-table bins holds only unconditional entries — no joint
 * count can lift a pair of such marginals to the block's level; real alignments are full of near-singleton sites — is then
 * flagged clean without being computed or screened.  The link tables do not depend on it (the same table entries dismiss the
 * same pairs either way; verify mode, ldw_set_screen 2, checks the pruned tiles' pairs in fp64 like every other dismissal).
 * ldw_prune_report: out[0] blocks whose rows were ordered, out[1] wave tiles pruned, out[2] wave tiles of the GEMMs that could
 * prune (both since the context was created; pruned tiles are not counted as executed work by ldw_gemm_stats), out[3] = on. */
int ldw_set_prune(ldw_ctx *ctx, int on);
int ldw_prune_report(ldw_ctx *ctx, int64_t out[4]);
/* r04 — spans.  The reference's loop visits the block pairs of a block row one at a time (R/computePairwiseMI.R:103-116); consecutive
 * LONG-RANGE-ONLY block pairs of one row (same from range, to ranges ascending, no pair within sr_dist) are run as ONE launch sequence
 * over their concatenated to side, every reference block keeping its own histogram, threshold, candidate list and place in the append
 * order (the lr filter is per block: :352-358).  Results never depend on it.  ldw_set_span: on != 0 (default; on = 3 also lets CORNER
 * block pairs — the neighbouring pair of the row, the pair that closes the circle: a few short-range pairs — join the spans, their
 * short-range pairs evaluated by an SR sub-pass: correct, measured slower on MI355X, so not the default), at most max_blocks
 * (2..8; 0 keeps the current value) reference blocks per span.  ldw_span_report: out[0] spans run, out[1] reference blocks they covered,
 * out[2] segments redone on their own after a wrong guess, out[3] on.
 * ldw_set_pair_cap (tests only): a fixed capacity for the pair lists of the approximate path (0: automatic) — a list that overflows makes
 * its block fall back like a wrong guess; process-wide. */
/* r04 — off the critical path.  ldw_ctx_create makes the two extra streams of the all-pairs loop itself (12 ms each on MI355X: part of
 * creating a context, not of a job's pass) and starts a side thread that loads the code objects of the pass's kernels; ldw_ctx_reserve starts a second one for the pinned staging buffers and the
 * per-slot device buffers (sized from L, N and max_blk_sz: call it right AFTER uploading the alignment — beside the upload it slowed the H2D
 * copy from 9.5 to 17 ms — and it hides behind the Hamming GEMM and the set-up calls).  Both are
 * optional (everything is also made lazily: LDW_NO_PREPARE=1 switches them off); the entry points that use what they prepare wait for them. */
int ldw_ctx_reserve(ldw_ctx *ctx, int64_t L, int64_t N, int64_t max_blk_sz);
/* r04 — the index columns of the short-range table from positions alone.  The short-range rows a pass emits for a block pair of contiguous SNP
 * ranges are a pure function of POS, g, sr_dist and the block geometry (R/computePairwiseMI.R:306-333: upper-triangle rows column by
 * column, then the lower ones; pos1 = POS_t[col], pos2 = POS_f[row]) — so a multi-GPU gather sends only their MI column and rank 0 rebuilds
 * (a, b) here.  blocks: nblocks x 4 (from_s, from_e, to_s, to_e), 1-based inclusive, in the order of the table; a_out / b_out: DEVICE int32 arrays
 * of `capacity` rows (both null: count only); *n_out = rows.  Needs POS ascending (ldw_set_snp_meta). */
int ldw_sr_pairs_fill(ldw_ctx *ctx, const int32_t *blocks, int64_t nblocks, double sr_dist, int32_t *a_out, int32_t *b_out, int64_t capacity, int64_t *n_out);
int ldw_set_span(ldw_ctx *ctx, int on, int max_blocks);
int ldw_span_report(ldw_ctx *ctx, int64_t out[4]);
/* r05 — list overflows.  The default path lists its candidates in fixed-capacity device lists; a list that overflows makes its block (or
 * its segment of a span) be redone on the plain path (counted in spec_misses like a wrong bucket guess), so results never depend on a
 * capacity.  out[0] blocks / segments redone because a PAIR list overflowed, out[1] because the MAYBE list of the approximate GEMM's
 * epilogue did (sized for the worst case since r05: non-zero only under the test override LDW_MAYBE_CAP), out[2] = 1 while the maybe list
 * is switched off for the rest of the pass after such an overflow (ldw_reset_speculation switches it on again), out[3] entries handed to the
 * maybe list since the context was created. */
int ldw_overflow_report(ldw_ctx *ctx, int64_t out[4]);
int ldw_set_pair_cap(uint32_t cap);
/* inspection only: the per-SNP bounds behind the pruning of the 2 x 3 / 3 x 3 tables.  out[a * 4 + 2 * m + (k - 2)] = the largest MI
 * SNP a (2 or 3 states, all flagged in uqe, r = its number of states) can reach with ANY partner that has k = 2 or 3 flagged states
 * and r = k — the maximum of the MI over the joint tables with a's marginals, which is convex there and sits at a vertex: every
 * state of a sends all its weight to one state of the partner — under the intended (m = 0) and the reference (m = 1: RXY at its
 * floor min(r)^2 / 4) reading of RXY; 1e300 for other SNPs and for SNPs with a sizeable minor state (not evaluated).  Needs the
 * alignment, the weights and the SNP meta data; capacity in doubles (>= 4 L). */
int ldw_snp_bounds(ldw_ctx *ctx, double *out, int64_t capacity);
/* inspection only (r05): out[0] = pairs that verify mode (ldw_set_screen 2) counted as "the screen would have lost this one" since the last call, out[1 + 4 k ..] = (from SNP,
 * to SNP, exact MI, level) of the first 16 of them.  65 doubles. */
int ldw_debug_violations(ldw_ctx *ctx, double *out);
/* (test hook) the threshold table of the biallelic pairs (k_build_tab11) for a total weight W, an MI level lo, the approximate sums' relative
 * error delta, their absolute slack eta and the unit sprime of the int32 sums: out[64 * 64 * 2] = (Lq, Hq) of entry [bin of the to side][bin of the
 * from side], bin = min(63, floor(sqrtf(p) * cbin)); a sum n' with Lq < n' < Hq is dismissed.  tests/test_gpu_parity.py checks the table against
 * the MI formula on a grid of joint tables. */
int ldw_debug_tab11(ldw_ctx *ctx, double W, double lo, double delta, double eta, double sprime, int32_t *out, double *cbin_out);
/* (test hooks, r06 — BOUNDS.md; tests/test_bounds.py brute-forces every bound of the default path through them; ldweaver_amd/csrc/ldw_debug.hip)
 * ldw_debug_apx_params: the constants the approximate screen's bound is built from for the CURRENT weights, as the engine derives them:
 *   out[0] F (fraction bits of the fixed-point weights), [1] e_last, [2] delta = max |V'/V - 1|, [3] lost units of a GEMM entry, [4] sum of the fixed-point weights,
 *   [5] neff, [6] apx_EG, [7] apx_dfac, [8] apx_s1, [9] apx_c1, [10] apx_W, [11] apx_unit = 2^(e_last - F), [12] scr_scale of the approximate screen,
 *   [13] scr_shift and [14] scr_scale of the exact-limb screen, [15] bit 0: the path is usable, bit 1: block exponents per 32 positions,
 *   [16] lo_abs_sum = sum |V_lo| 2^-F and [17] lo_bound, the margin the mixed-precision screen adds for the two low limbs, [18] apx_MU (units a floor marginal can be low: 1, or 0 at e_last = 0), [19] limbs;
 *   vfixed_out / vapx_out (may be NULL; capacity >= N): the exact fixed-point weight V_s and its dual-digit approximation V'_s = a b 2^e of every SEQUENCE.
 * ldw_debug_rows: row0_out[L + 1] = first indicator row of every SNP, slot_meta_out[L] = rows (3 bits) | uqe flag of slot i << (3 + i) | state of slot i << (8 + 3 i).
 * ldw_debug_apx_gemm: gemm_apx_kernel over the given indicator rows (indices 0..R; R = the all-zero padding row): out[nrt][nrf] = the int32 sums G' in units of 2^e_last.
 * ldw_debug_screen_bound: the engine's own device functions on n caller-made joint tables (arrays by case: g[16] = sums of the indicator rows, g[j * 4 + i] = slot i of the
 *   from-side SNP x slot j of the to-side SNP; pa / pb[5] integer marginals by slot; pX / pY[5] weighted marginals; rr[3] = r_a, r_b, RXY; masks[2] = slot meta of both SNPs,
 *   kinds 1 / 3 only; params = the 20 numbers of ldw_debug_apx_params, which the caller may alter).  kind 0: full_cells_screen<na, nb, APX> — the approximate path's upper
 *   bound of MI; 1: pair_screen_generic<APX>; 2: full_cells_screen<na, nb> on exact sums (an fp32 MI); 3: pair_screen_generic on exact sums; 4: full_cells_mi<na, nb>, the
 *   fp64 value the engine emits (out64).  na, nb in {1, 2} for kinds 0 / 2 / 4. */
int ldw_debug_apx_params(ldw_ctx *ctx, double out[20], int64_t *vfixed_out, int64_t *vapx_out, int64_t capacity);
int ldw_debug_rows(ldw_ctx *ctx, int32_t *row0_out, uint32_t *slot_meta_out, int64_t capacity);
int ldw_debug_apx_gemm(ldw_ctx *ctx, const int32_t *rows_t, int nrt, const int32_t *rows_f, int nrf, int32_t *out);
int ldw_debug_screen_bound(ldw_ctx *ctx, int kind, int na, int nb, int64_t n, const int64_t *g, const int64_t *pa, const int64_t *pb, const float *pX, const float *pY,
                           const double *rr, const uint32_t *masks, const double params[20], float *out, double *out64);
/* diagnostics of the approximate path after ldw_set_weights: out[0] = usable (0/1), out[1] = max relative error delta of the
 * dual-digit weights, out[2] = weight classes, out[3] = popcount segments, out[4] = exponent transitions, out[5] = e_last */
int ldw_apx_info(ldw_ctx *ctx, double out[6]);
/* fp32 screen in front of the fp64 MI evaluation, in blocks that run the speculative selection: a long-range pair
 * only matters if its MI reaches the guessed histogram bucket, so MI is first bounded in fp32 (v_log_f32, proven
 * error < 1.3e-5 nats, margin 2e-4) and the exact value is computed for the waves that hold a pair which may pass, or a
 * short-range pair.  Every MI that is emitted is the exact one, so the link tables do not depend on the mode.
 * 0 = off, 1 = on (default), 2 = verify: evaluate everything both ways and count lost pairs in ldw_ctx_counters[3]. */
int ldw_set_screen(ldw_ctx *ctx, int mode);
int ldw_links_count(ldw_ctx *ctx, int which, int64_t *n_out);
/* a_out/b_out: 0-based SNP index of the from-side (pos2) and to-side (pos1) SNP; MI_out. capacity in
 * rows; on_device selects the destination space.  block_row_offsets_out[nblocks+1] (host, may be NULL)
 * gives each processed block's first row. */
int ldw_links_fetch(ldw_ctx *ctx, int which, int32_t *a_out, int32_t *b_out, double *MI_out,
                    int64_t capacity, int on_device);
/* The table itself, without a copy: DEVICE pointers to the context's own (a, b, MI) columns and the row count.  Valid until the
 * next call that changes the table (ldw_mi_all_pairs, ldw_links_begin, ldw_links_import, ldw_ctx_destroy); read-only. */
int ldw_links_device_ptrs(ldw_ctx *ctx, int which, const int32_t **a_out, const int32_t **b_out, const double **MI_out,
                          int64_t *n_out);
/* Replace the context's short-range (which = 0) or long-range (1) table by caller data (host or device memory): how the
 * rank that received the other ranks' tables in the multi-GPU gather hands the assembled table to the short-range model,
 * ARACNE and the post-processing entry points below, which work on the context's tables. */
int ldw_links_import(ldw_ctx *ctx, int which, const int32_t *a, const int32_t *b, const double *MI, int64_t n, int on_device);
/* per-block diagnostics of the last ldw_mi_all_pairs call: n_lr_total, n_lr_kept, n_sr, and the
 * quantile threshold (NaN when no lr links); arrays of length nblocks (may be NULL). */
int ldw_block_stats(ldw_ctx *ctx, int64_t nblocks, int64_t *n_lr_total, int64_t *n_lr_kept,
                    int64_t *n_sr, double *disc_thresh);

/* ---- (6) ARACNE — R/io_functions.R:101-164 + src/fintersect.cpp, src/computeMI.cpp:44-77 ---- */
/* flags_out[i] = 1 unless some common neighbour Y of (X,Z) = (chk_pos1[i], chk_pos2[i]) in the full link
 * set has MI(X,Z) < MI(X,Y) and MI(X,Z) < MI(Z,Y).  Positions are compared as exact values. */
int ldw_aracne(ldw_ctx *ctx, const double *chk_pos1, const double *chk_pos2, const double *chk_MI,
               int64_t n_chk, const double *full_pos1, const double *full_pos2, const double *full_MI,
               int64_t n_full, uint8_t *flags_out);

/* ---- (7) short-range model on the device-resident sr table — mergeNsort_sr_links,
 *          R/computePairwiseMI.R:400-495 — and ARACNE on its result.  The table is the one left by
 *          ldw_mi_all_pairs / ldw_links_end; cluster ids come from the paint of ldw_set_snp_meta
 *          (1..nclust, nclust <= 255); len must be integral (integer genome length).  The caller keeps the two
 *          O(1)-sized numerical steps (the log-log OLS fit :428 and the beta MLE :452) and calls, in order: ---- */
/* (:417-424) per cluster c (0-based) and integer len l = 1..S, S = ceil(sr_dist)-1: n_out[c*S + l-1] = number
 * of links with that len touching cluster c+1 (0 < len < sr_dist), and the two order statistics that
 * quantile(type 7, prob) interpolates: ranks floor(h) and ceil(h) of the ascending MI values, h = (n-1)*prob
 * (NaN where n = 0). */
int ldw_sr_len_quantiles(ldw_ctx *ctx, int nclust, double sr_dist, double prob, int32_t S,
                         double *q_lo_out, double *q_hi_out, int64_t *n_out);
/* (:444-452) mean_dist[c*S + l-1] = fitted decay of cluster c+1 looked up BY THE VALUE of len (quirk Q5:
 * NaN beyond the number of distinct lens).  stats_out[c*5 + k] over the links with diff = MI - mean_dist > 0:
 * n, sum diff, sum diff^2, sum log(diff), sum log(1-diff) — reduced in a fixed order. */
int ldw_sr_excess_stats(ldw_ctx *ctx, int nclust, int32_t S, const double *mean_dist, double *stats_out);
/* (:453, :475-490) shape[c*3 + {0,1,2}] = beta shape1, shape2, log B(shape1, shape2).  srp = -log P(X > diff)
 * per link and cluster, maximum over the link's clusters (ties: smaller cluster id), links with
 * srp > srp_cutoff are kept on the device (n_red), and the ARACNE pool = links with a positive excess in
 * some cluster and MI >= min MI kept (n_pool; n_pool_out = NULL: no pool is built, see ldw_sr_pool_build). */
int ldw_sr_pvalues(ldw_ctx *ctx, int nclust, int32_t S, const double *mean_dist, const double *shape,
                   double srp_cutoff, int64_t *n_red_out, int64_t *n_pool_out, double *min_mi_out);
/* kept links in no particular order: row in the sr table (ldw_links_fetch order) and that row's (a, b, MI),
 * clust_c, the first cluster (ascending id) in which the link has a positive excess, whether
 * clust1 != clust2, srp_max. */
int ldw_sr_reduced_fetch(ldw_ctx *ctx, int64_t capacity, int64_t *row_out, int32_t *a_out, int32_t *b_out,
                         double *MI_out, int32_t *clust_c_out, int32_t *first_clust_out, uint8_t *dup_out,
                         double *srp_out);
int ldw_sr_pool_fetch(ldw_ctx *ctx, int64_t capacity, int32_t *a_out, int32_t *b_out, double *MI_out);
/* runARACNE (R/io_functions.R:101-164) for the kept links against the pool, both device resident;
 * flags_out[i] belongs to row_out[i] of ldw_sr_reduced_fetch (after ldw_sr_pvalues) or ldw_lr_reduced_fetch (after
 * ldw_lr_tukey). */
int ldw_aracne_device(ldw_ctx *ctx, int64_t capacity, uint8_t *flags_out);

/* ---- (7b) r05 — the same model with the short-range table LEFT on the GPUs that computed it (multi-GPU jobs: SURVEY.md 8(e); the
 *          reference's mergeNsort_sr_links, R/computePairwiseMI.R:400-495, sees one table).  Every rank calls (7) on its own rows; what
 *          travels between ranks is per-group bounds and counts, ~7 % of the MI column, five sums per block and cluster, the kept links and
 *          the ARACNE pool — not the table (host side: ldweaver_amd/dist_srp.py; protocol and proof of the bound: BOUNDS.md 10, docs/HISTORY.md 7b). ---- */
/* Rows of the context's table at or above a per-(cluster, len) bound.  lower[c*S + l-1] (host; NaN: send nothing, -inf: every member).  A group's
 * rows are the table rows with that len whose pos1 or pos2 lies in cluster c+1 (a row of two clusters is a member of both, :411-414).
 * cnt_out[(l-1)*nclust + c] (host, len-major) = rows passing, *n_out their sum; mi_out (host or device, `capacity` doubles; NULL: count
 * only) = their MI values grouped in that order, in no particular order within a group.  After ldw_sr_len_quantiles (same nclust, S). */
int ldw_sr_tail_extract(ldw_ctx *ctx, int nclust, int32_t S, const double *lower, int64_t *cnt_out, double *mi_out, int64_t capacity,
                        int on_device, int64_t *n_out);
/* The order statistics of quantile(type 7, prob) per (cluster, len) from the candidates of n_src ranks: mi_src[r] / cnt_src[r] = what
 * ldw_sr_tail_extract gave on rank r (values: host or device by on_device; counts: host), n_total[c*S + l-1] = the group's size over all
 * ranks.  Every row that is NOT among the candidates must lie below every candidate of its group (bounds from ldw_sr_len_quantiles of the
 * ranks: the smallest local lower order statistic is one).  q_lo_out / q_hi_out as ldw_sr_len_quantiles.  *violations_out = groups whose
 * order statistic does not lie among the candidates (NaN there; NULL: such a group is an error).  Needs no table, alignment or meta data. */
int ldw_sr_quantiles_merge(ldw_ctx *ctx, int nclust, int32_t S, double prob, int n_src, const double *const *mi_src,
                           const int64_t *const *cnt_src, const int64_t *n_total, int on_device, double *q_lo_out, double *q_hi_out,
                           int64_t *violations_out);
/* ldw_sr_excess_stats per reference block: the table's rows are those of `nblocks` blocks in order, rows_per_block[b] each (sum = the
 * table's rows); stats_out[(b*nclust + c)*5 + k].  A block's sums depend on its own rows only (64 strips, fixed order), so the sum over
 * all blocks in make_blocks order is bit-identical however the blocks were dealt over ranks. */
int ldw_sr_excess_stats_blocks(ldw_ctx *ctx, int nclust, int32_t S, const double *mean_dist, int64_t nblocks, const int64_t *rows_per_block,
                               double *stats_out);
/* The ARACNE pool of this context's rows for a minimum taken over all ranks (ldw_sr_pvalues with n_pool_out = NULL builds none): rows with a
 * positive excess in some cluster and MI >= min_mi (:489-490); NaN: empty.  Then ldw_sr_pool_fetch. */
int ldw_sr_pool_build(ldw_ctx *ctx, double min_mi, int64_t *n_pool_out);
/* Rank 0: adopt the kept links and the pool of all ranks (host arrays, 0-based from-side / to-side SNP index and MI).  The kept links REPLACE
 * the context's short-range table (rows 0..n_red-1); ldw_aracne_device then answers for them in that order. */
int ldw_sr_reduced_import(ldw_ctx *ctx, int64_t n_red, const int32_t *a, const int32_t *b, const double *MI, int64_t n_pool,
                          const int32_t *pool_a, const int32_t *pool_b, const double *pool_MI);

/* ---- (7c) r05 — (7) over the contexts of ONE process after ldw_mi_all_pairs_multi(.., flags = LDW_MI_SR_ROWS_STAY): the protocol of (7b) run by the
 *          library itself (a worker thread per context, exchanges staged through host memory), behind the signatures of (7) — so a single-process
 *          host (R: r_shim/) keeps its three calls and its own fit / optimiser between them.  ctx[0] of that call must be ctx[0] here.  With one
 *          context, or tables that were gathered (no flag), they are (7) on ctx[0] — with the excess sums taken per block when the table's block
 *          structure is known, so that one context and several give the same bits. ---- */
int ldw_sr_len_quantiles_multi(ldw_ctx **ctx, int n_ctx, int nclust, double sr_dist, double prob, int32_t S, double *q_lo_out, double *q_hi_out,
                               int64_t *n_out);
int ldw_sr_excess_stats_multi(ldw_ctx **ctx, int n_ctx, int nclust, int32_t S, const double *mean_dist, double *stats_out);
/* ... and the kept links of ALL contexts end up in ctx[0] (they replace its short-range table, ordered as in the job's table: make_blocks order), with the
 * pool of all contexts: ldw_sr_reduced_fetch / ldw_sr_pool_fetch / ldw_aracne_device on ctx[0] follow as after ldw_sr_pvalues. */
int ldw_sr_pvalues_multi(ldw_ctx **ctx, int n_ctx, int nclust, int32_t S, const double *mean_dist, const double *shape, double srp_cutoff,
                         int64_t *n_red_out, int64_t *n_pool_out, double *min_mi_out);

/* ---- (8) consumers of the link tables, on the device-resident tables ------------------ */
/* Numeric core of analyse_long_range_links (R/lr_analyser.R:72-111): q13_out = quantile(MI, c(.25,.75)) (type 7) of the
 * long-range table, thresholds_out = q3 + (1.5, 3) IQR — or, when fewer than min_links (reference: 5000) links exceed
 * min(thresholds) although the table has that many rows, quantile(MI, 1 - c(4000, 5000)/n) (*fallback_out = 1, the
 * reference's warning).  sr_a / sr_b / sr_mi (host, n_sr_rows >= 0 rows: 0-based from-side / to-side SNP index and MI) is the
 * short-range table the reference reads back from sr_links.tsv (R/lr_analyser.R:67), i.e. the REDUCED set
 * perform_MI_computation returned (srp_max > srp_cutoff, R/computePairwiseMI.R:122,140) — not the raw short-range table.
 * Leaves on the device: the outlier links lr[MI > min(thresholds)] (n_red, table order) and the ARACNE pool
 * rbind(lr, sr)[MI > min(thresholds)] (n_pool, R/lr_analyser.R:106-109).  Then ldw_lr_reduced_fetch / ldw_aracne_device. */
int ldw_lr_tukey(ldw_ctx *ctx, int64_t min_links, const int32_t *sr_a, const int32_t *sr_b, const double *sr_mi,
                 int64_t n_sr_rows, double q13_out[2], double thresholds_out[2], int *fallback_out,
                 int64_t *n_red_out, int64_t *n_pool_out);
/* the outlier links: row in the lr table (ldw_links_fetch order) and that row's (a, b, MI) */
int ldw_lr_reduced_fetch(ldw_ctx *ctx, int64_t capacity, int64_t *row_out, int32_t *a_out, int32_t *b_out, double *MI_out);
/* Numeric core of genomewide_LDMap (R/LDSummaryPlot.R:55-106): pos_vec = sorted unique positions of all links (kept
 * if from < pos < to when a window is given; from = to = 0: genome-wide), symmetric sparse MI matrix over their ranks,
 * block sums with the kernel of .mat(n, reducer) (:176-178), / reducer^2, log10(. + 1e-5), rescaled to [0, 1] (:157-163).
 * reducer = 0: round(length(pos_vec) / 1e3) like the reference.  htm_out: B x B doubles, B = n_pos / reducer (integer
 * division), symmetric; htm_out = NULL only returns the sizes.  reducer <= 1 (the reference's unreduced dense plot) is
 * refused with LDW_ERR_ARG. */
int ldw_ldmap(ldw_ctx *ctx, int32_t reducer, int32_t from, int32_t to, int64_t *n_pos_out, int32_t *reducer_out, int32_t *B_out,
              double *htm_out, int64_t capacity);

/* ---- (9) the tsv files — write.table(x, file, append = T, quote = F, row.names = F, col.names = F, sep = '\t'),
 *          R/computePairwiseMI.R:140 (sr_links.tsv) and :362 (lr_links.tsv); readers R/io_functions.R:32-66 ------------ */
#define LDW_COL_INT32 0
#define LDW_COL_INT64 1
#define LDW_COL_DOUBLE 2
/* One double as write.table prints it (R's formatReal, digits = 15: fewest significant digits that reproduce the 15-digit
 * value; fixed notation unless wider than scientific, so 100000 -> "1e+05").  out: >= 48 bytes, NUL-terminated. */
int ldw_format_number(double x, char *out, int capacity);
/* R's `set.seed(seed); sample(n, size)` (1-based, without replacement; Mersenne-Twister, rejection sampling: R >= 3.6 defaults) — the
 * draw behind the 10 % SNP subset of R/computePairwiseMI.R:94-97 (`lr_links_approx`).  n > 1e7 with size <= n / 2 takes R's hashing
 * variant (`sample.int(useHash = TRUE)`, do_sample2: elements re-drawn while they repeat an earlier one), as R itself does.  Host only. */
int ldw_r_sample(uint32_t seed, int64_t n, int64_t size, int64_t *out);
/* nrows x ncols numeric table (host columns of kind LDW_COL_*), tab-separated, no header, appended (append != 0) or
 * truncating; rows are formatted by nthreads host threads (0 = all cores) and written in order.  bytes_out may be NULL. */
int ldw_write_table_tsv(const char *path, int append, int64_t nrows, int ncols, const int32_t *col_kind, const void *const *cols,
                        int nthreads, int64_t *bytes_out);
/* The context's short-range (which = 0) or long-range (1) link table as the reference's MI_df rows `pos1 pos2 clust1 clust2 len MI`
 * (R/computePairwiseMI.R:319-331: pos1 = POS of the to-side SNP, integer columns; clust, len, MI doubles), fetched from the
 * device and formatted by host threads.  An empty table writes nothing, like the reference (:360). */
int ldw_write_links_tsv(ldw_ctx *ctx, int which, const char *path, int append, int nthreads, int64_t *rows_out, int64_t *bytes_out);
/* r04 — the same table written BESIDE the caller's next calls: _begin fetches the table from the device (synchronously: the table may be
 * replaced afterwards) and returns while host threads derive, format and write it; _end waits for them and reports rows, bytes and the
 * writer's status.  lr_links.tsv (R/computePairwiseMI.R:362) does not depend on the short-range model that follows it (:119-126), so a job
 * hides the 15 ms of its million rows behind that model.  One asynchronous table per context at a time (_begin, the synchronous call,
 * ldw_set_snp_meta and ldw_ctx_destroy finish a pending one first); _end without _begin returns 0 rows. */
int ldw_write_links_tsv_begin(ldw_ctx *ctx, int which, const char *path, int append, int nthreads);
int ldw_write_links_tsv_end(ldw_ctx *ctx, int64_t *rows_out, int64_t *bytes_out);
int ldw_tsv_join(ldw_ctx *ctx);
/* Host memory the library keeps between calls — the tsv writers' pooled buffers (process-wide, ~100 MB after a C4 job) and the context's pinned
 * fetch arena (16 B per row of the largest table written, <= 2.25 GB) — is released here (ctx may be NULL: the pool only).  Call it BETWEEN jobs:
 * on this driver stack giving large host regions back next to GPU work stalls the process's next GPU call (docs/HISTORY.md 8).  bytes_out: released.
 * r05: also the DEVICE blocks (>= 64 MB each, <= LDW_DEVPOOL_GB = 48 GB in all) that released buffers and destroyed contexts leave for the next taker — fetching
 * device memory from the driver costs up to 40 ms per GB on this stack, a second for every context created after another one was destroyed. */
int ldw_host_trim(ldw_ctx *ctx, int64_t *bytes_out);
/* SINGLE WRITER PER PATH: the tsv writers position their workers' writes by offsets computed from the file's size at the start of the call
 * (pwrite), so two writers appending to one path at the same time — two contexts or ranks, or a synchronous call beside a pending
 * ldw_write_links_tsv_begin / ldw_lr_stream_begin on the same file — overwrite each other.  One process appends to lr_links.tsv, as in the
 * reference's serial loop (R/computePairwiseMI.R:103-116). */
/* r05 — lr_links.tsv appended WHILE the pass runs, as the reference appends it block by block (R/computePairwiseMI.R:362).  _begin (before
 * ldw_mi_all_pairs; append = 0 truncates the file first) opens a writer thread on the context; after every finished item of the pass (a
 * block, or a span of blocks) the rows it added to the long-range table — final: the filter is per block (:352-358) — are fetched on a
 * stream of their own, formatted like ldw_write_links_tsv and appended.  The file is at all times a prefix of the complete file in whole
 * blocks; a pass that fails at block k leaves the rows of blocks 0..k-1 (the items already submitted are run to their end first), a killed
 * process those of the items written so far.  _end waits for the writer and reports rows, bytes and the number of blocks whose rows are in
 * the file (any output may be NULL); without _begin it returns zeros.  Single context, single pass: the multi-context gather reorders rows
 * and keeps the table-at-once writer. */
int ldw_lr_stream_begin(ldw_ctx *ctx, const char *path, int append, int nthreads);
int ldw_lr_stream_end(ldw_ctx *ctx, int64_t *rows_out, int64_t *bytes_out, int64_t *blocks_out);   /* waits for a pending asynchronous table, discarding its counts (status returned) */

/* ---- small native helpers kept for finest-grain A/B parity (host memory) -------------------- */
/* .compareToRow src/computeMI.cpp:25-41: ret[j] = any(x[j,] in y); x is nr x nc column-major */
int ldw_compare_to_row(const double *x, int64_t nr, int64_t nc, const double *y, int64_t ny, uint8_t *ret);
/* .vecPosMatch src/computeMI.cpp:44-59: 1-based first position of x[i] in y, 0 if absent */
int ldw_vec_pos_match(const double *x, int64_t nx, const double *y, int64_t ny, double *ret);
/* .compareTriplet src/computeMI.cpp:63-77 */
int ldw_compare_triplet(const double *MI0X, const double *MI0Z, int64_t n, double MI0, int *ret);
/* .fast_intersect src/fintersect.cpp:6-32; out capacity >= min(na, nb); *n_out = result length */
int ldw_fast_intersect(const int32_t *A, int64_t na, const int32_t *B, int64_t nb, int32_t *out,
                       int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* LDWEAVER_AMD_H */
