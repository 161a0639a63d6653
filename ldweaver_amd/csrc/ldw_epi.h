// Device-side pieces of the MI epilogue shared by k_mi_epilogue (ldw_mi.hip) and the fused GEMM + epilogue kernel
// (ldw_fused.hip): histogram buckets, short-range interval tests, the fp64 log helpers, the per-pair MI
// (src/computeMI.cpp:19 cell by cell) and the emission of one finished pair.
#pragma once
#include <cmath>
#include "ldw_internal.h"
#include "ldw_dev.h"
#include "ldw_log.h"

namespace ldw {

// ------------------------------------------------------------------------------------------------
// helpers shared by host and device
// ------------------------------------------------------------------------------------------------
// Histogram bucket of an MI value: monotone non-decreasing in mi, 128 buckets per octave (0.5 % wide) from 2^-20
// up, taken straight from the IEEE-754 bits (exponent + 7 mantissa bits) — no floating-point arithmetic.  Values
// below 2^-20 (and negatives, which quirk Q1 can produce) fall in bucket 0.
constexpr int BUCKET_OFF = (1023 - 20) << 7;
__host__ __device__ __forceinline__ int mi_bucket(double mi) {
    long long u;
    memcpy(&u, &mi, 8);
    if (u <= 0) return 0;  // -x, -0, +0
    const int b = (int)(u >> 45) - BUCKET_OFF;
    return b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
}
// lower edge of bucket B (B >= 1)
__host__ __device__ __forceinline__ double bucket_lo(int B) {
    const long long u = (long long)(B + BUCKET_OFF) << 45;
    double v;
    memcpy(&v, &u, 8);
    return v;
}

// Short-range partners of one to-side SNP: up to three disjoint, ascending index intervals [s,e) of the
// from-side list, plus the first row of its upper (a_loc < b_loc) and lower (a_loc > b_loc) segment in
// the short-range table (relative to the block's base row).
struct ColInfo {
    int32_t s[3], e[3];
    int32_t pad[2];
    int64_t off_u, off_l;
};

// (bitwise on purpose: with && / || the compiler branches per interval under a divergent exec mask — three s_and_saveexec chains per pair in the screens of
// the blocks with short-range pairs)
__host__ __device__ __forceinline__ bool col_is_sr(const ColInfo &c, int a) {
    return (((a >= c.s[0]) & (a < c.e[0])) | ((a >= c.s[1]) & (a < c.e[1])) | ((a >= c.s[2]) & (a < c.e[2]))) != 0;
}
// number of short-range partners in [lo, hi)
__host__ __device__ __forceinline__ int col_count(const ColInfo &c, int lo, int hi) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = c.s[k] > lo ? c.s[k] : lo, b = c.e[k] < hi ? c.e[k] : hi;
        n += b > a ? b - a : 0;
    }
    return n;
}

// which segment a pair belongs to: 0 = upper (a<b, off-diagonal blocks only), 1 = lower (a>b), -1 = not a pair
__host__ __device__ __forceinline__ int pair_seg(int a_loc, int b_loc, int lower_only) {
    if (a_loc == b_loc) return -1;
    if (a_loc > b_loc) return 1;
    return lower_only ? -1 : 0;
}

// ------------------------------------------------------------------------------------------------
// fp64 helpers of the epilogue: no IEEE division, no libm call.  Accuracy ~2e-15; the
// epilogue is not bit-matched to the reference (tolerance 1e-6 on MI), see DESIGN.md.
// ------------------------------------------------------------------------------------------------
// v_rcp_f64 is good to 4.5e-8 (measured on gfx950); one Newton step brings it to 2e-15
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// HALF of log(N / D): ldw_log.h (one source for the device and for the host check tests/test_log_host.py), with v_rcp_f64 as the estimate.
// Measured and NOT shipped (profiles/r06_epi_split.txt): one reciprocal shared by the four cells of a table (1 / (b0 b1 b2 b3) and nine
// multiplications instead of four quarter-rate v_rcp_f64 + Newton steps) saves 9 of ~115 issue slots per 2 x 2 table on paper and costs 1.6 of
// 33.4 ms: the four cells' operands stay live together, k_mi_epilogue_fast spills 68 registers and k_mi_units<true> falls from 4 to 3 waves.
struct DeviceRcpEstimate {
    __device__ __forceinline__ double operator()(double x) const { return __builtin_amdgcn_rcp(x); }
};
__device__ __forceinline__ double fast_half_log_ratio(double N, double D) { return half_log_ratio_core(N, D, DeviceRcpEstimate{}); }
// pxy of a fixed-point joint sum 0 <= n < 2^52: n scale + 1/2 (scale = 2^-F, a power of two) as ONE fma on the bit pattern of 2^52 + n —
// exact, the same bits as the conversion followed by fma(n, scale, 0.5)
__device__ __forceinline__ double pxy_of(int64_t n, double scale, double half_m) {   // half_m = 0.5 - 2^52 scale (exact: 0.5 - an integer, F <= 52)
    return fma(__longlong_as_double(n | 0x4330000000000000LL), scale, half_m);
}

// ------------------------------------------------------------------------------------------------
// what happens to one finished pair: dense store, short-range scatter, long-range histogram
// ------------------------------------------------------------------------------------------------
struct EmitArgs {
    double *MI;            // dense block, column-major nf x nt
    const ColInfo *cols;   // null: dense store only (ldw_mi_block)
    int nf, lower_only, keep_sr, do_lr;
    int write_dense;       // store the dense MI block (needed by k_lr_gather; off when the gather is speculative)
    int spec_B;            // >= 0: append long-range pairs with bucket >= spec_B to the candidate list right here,
                           //       and histogram ONLY those (the pairs below are counted analytically)
    int any_sr;            // 0: no pair of this block is short-range (skip the interval tests)
    double spec_lo;        // lower edge of bucket spec_B minus a guard: cheap reject before the bucket arithmetic
    int64_t sr_base;
    int32_t *sr_a, *sr_b;
    double *sr_mi;
    unsigned long long *n_cand;
    uint64_t *ckey, *cval;
    // fp32 screen of the epilogue (speculative mode only): 0 off, 1 on, 2 verify
    int scr_mode, scr_shift;             // fixed-point sums are cut to their top 31 bits: n >> scr_shift
    float scr_scale;                     // 2^scr_shift * 2^-frac_bits
    float scr_eps;                       // margin of the screen: SCREEN_EPS (+ the bound of the low limbs in the mixed path)
    unsigned long long *scr_viol;        // verify mode: pairs the screen would have lost
    // approximate-GEMM path (ldw_apx.h): the joint sums the screen reads are int32 sums of the approximate weights
    // V' = a b 2^e in units of 2^e_last, each within a relative apx_delta of the exact sum plus a few truncated units
    int apx;                             // 1: G is int32 [RTpad][RFpad]
    float apx_EG;                        // units of 2^e_last a GEMM entry can have lost at the exponent transitions of the K loop
    float apx_dfac;                      // 1.01 delta / (1 - delta)
    float apx_s1;                        // 2^(e_last - F) (2 ln(neff + 12.5) + 2.1) / (1 - delta): what one lost unit can add to den * MI
    // r03: the first-order term of the bound splits into sum ln(x den / d) (x - x') and sum (x - x'); the second sum is not an error
    // term at all but the difference of two totals — every sequence lies in exactly one cell, so sum x = W (the fixed-point total)
    // and sum x' = (sum of the from-side SNP's approximate marginals) 2^(e_last - F), clamped cells only add to it — known per SNP
    // (full_cells_screen).  apx_c1 = 0.02 (the c of the bound; 1.02 restores the r02 form, LDW_SCREEN_R02_BOUND), apx_W = W (0: r02 form),
    // apx_unit = 2^(e_last - F)
    float apx_c1;
    double apx_W, apx_unit;
    // r06: units a MARGINAL of the approximate weights is below its sum: slot_papx = floor(sum V' / 2^e_last) loses less than one — and nothing when e_last = 0
    // (unit weights: V' = V = 1, a unit is a whole sequence and one lost unit per derived cell made the bound useless there: tests/test_bounds.py)
    float apx_MU;
};

// The constants of the approximate path's screen bound (full_cells_screen<.., APX>, pair_screen_generic<APX>) from the context's weights: the
// screen reads int32 sums of the approximate weights V' in units of 2^e_last; its bound of the exact MI carries the relative error delta of the
// weights and the units lost to truncation.  One place for the engine (make_emit_args, ldw_mi_items.inc) and the test hook (ldw_debug_apx_params):
// BOUNDS.md 3 states what each constant has to cover.
inline void apx_screen_params(const ldw_ctx *c, EmitArgs &E, bool r02_bound = false) {
    const double den = c->neff > 1.0 ? c->neff : 1.0;
    E.apx = 1;
    E.apx_EG = (float)(c->apx_lost_units * 1.001);
    E.apx_dfac = (float)(1.01 * c->apx_delta / (1.0 - c->apx_delta));
    E.apx_unit = std::ldexp(1.0, c->apx_e_last - c->frac_bits);
    E.apx_s1 = (float)(E.apx_unit * (2.0 * std::log(den + 12.5) + (r02_bound ? 3.1 : 2.1)) / (1.0 - c->apx_delta) * 1.01);
    E.apx_c1 = r02_bound ? 1.02f : 0.02f;
    E.apx_W = r02_bound ? 0.0 : std::ldexp((double)c->total_fixed, -c->frac_bits);
    E.scr_shift = 0;
    E.scr_scale = (float)std::ldexp(1.0, c->apx_e_last - c->frac_bits);
    E.scr_eps = 2e-4f;   // SCREEN_EPS (below)
    E.apx_MU = c->apx_e_last == 0 ? 0.0f : 1.0f;
}

// H16: the LDS histogram holds two 16-bit counters per word (bucket b in half b & 1 of word b >> 1) — for a workgroup that sees at most 65 535
// pairs (k_mi_epilogue_fast: 64 x EPI_COLS), so that its LDS footprint lets five workgroups share a CU
template <bool H16>
__device__ __forceinline__ void hist_bump(unsigned int *sh_hist, int bk) {
    if (H16) atomicAdd(&sh_hist[bk >> 1], 1u << ((bk & 1) << 4));
    else atomicAdd(&sh_hist[bk], 1u);
}
template <bool H16 = false>
__device__ __forceinline__ void emit_pair(const EmitArgs &E, const ColInfo &c, int a_loc, int b_loc, int sa, int sb,
                                          double mi, unsigned int *sh_hist) {
    if (E.write_dense) E.MI[(int64_t)a_loc + (int64_t)b_loc * E.nf] = mi;
    if (!E.cols) return;
    const int seg = pair_seg(a_loc, b_loc, E.lower_only);
    if (seg < 0) return;
    if (E.any_sr && col_is_sr(c, a_loc)) {
        if (E.keep_sr) {
            const int64_t dst = E.sr_base + (seg == 0 ? c.off_u + col_count(c, 0, a_loc) : c.off_l + col_count(c, b_loc + 1, a_loc));
            E.sr_a[dst] = sa;
            E.sr_b[dst] = sb;
            E.sr_mi[dst] = mi;
        }
    } else if (E.do_lr) {
        if (E.spec_B >= 0) {
            // speculative mode: only the (rare) pairs at or above the guessed bucket are histogrammed and appended
            if (mi >= E.spec_lo) {
                const int bk = mi_bucket(mi);
                if (bk >= E.spec_B) {
                    hist_bump<H16>(sh_hist, bk);
                    const unsigned long long p = atomicAdd(E.n_cand, 1ull);
                    E.ckey[p] = f64_key(mi);
                    E.cval[p] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)E.nf);
                }
            }
        } else {
            hist_bump<H16>(sh_hist, mi_bucket(mi));
        }
    }
}

// emission in the speculative selection mode without a dense block or an LDS histogram (fused kernel, k_mi_units):
// candidates are rare, their bucket counts go straight to the global histogram
__device__ __forceinline__ void emit_pair_spec(const EmitArgs &E, const ColInfo &c, int a_loc, int b_loc, int sa, int sb, double mi,
                                           unsigned long long *__restrict__ ghist) {
    const int seg = pair_seg(a_loc, b_loc, E.lower_only);
    if (seg < 0) return;
    if (E.any_sr && col_is_sr(c, a_loc)) {
        if (E.keep_sr) {
            const int64_t dst = E.sr_base + (seg == 0 ? c.off_u + col_count(c, 0, a_loc) : c.off_l + col_count(c, b_loc + 1, a_loc));
            E.sr_a[dst] = sa;
            E.sr_b[dst] = sb;
            E.sr_mi[dst] = mi;
        }
    } else if (E.do_lr) {
        if (mi >= E.spec_lo) {
            const int bk = mi_bucket(mi);
            if (bk >= E.spec_B) {
                atomicAdd(&ghist[bk], 1ull);
                const unsigned long long p = atomicAdd(E.n_cand, 1ull);
                E.ckey[p] = f64_key(mi);
                E.cval[p] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)E.nf);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MI epilogue: one thread per SNP pair; a wave = 64 consecutive from-side SNPs at one to-side SNP and walks
// EPI_COLS/4 consecutive to-side SNPs, so everything indexed by the to-side SNP is wave-uniform.
// ------------------------------------------------------------------------------------------------
constexpr int EPI_COLS = 128;  // to-side SNPs per workgroup (4 waves x 32)

// Mixed-precision path: geometry of the per-tile unit lists and of the low-limb joint sums of one block.
// A from-TILE is the 64 from-side SNPs of one epilogue wave (perm_f order); the to-side SNPs are split by row-slot class
// lc = 0, 1, 2 (1, 2, 4 indicator-row slots).  Units listed by k_mi_screen go to tl[tile * nt + uoff[lc] + k]; the
// gathered GEMM then gives unit k of (tile, lc) the low-limb rows rowbase[lc] + k * c .. + c of that tile, each
// 64 * cmax_f[tile] ints long (from-side SNP `lane`, slot i at lane * cmax + i), starting at glo + tile_base[tile].
struct LoGeom {
    int32_t n_lc[3], uoff[3], rowbase[3];
    int32_t RTlo, ntiles, on;          // low-limb rows per tile (sum of the classes' rows padded to 128); on: path active
    const int32_t *cmax_f;             // [ntiles] widest row-slot class among the tile's SNPs
    const int64_t *tile_base;          // [ntiles] offset of the tile's low-limb block in glo (ints)
    unsigned int *cnt;                 // [ntiles * 3] listed units per (tile, class)
    uint32_t *tl;                      // [ntiles * nt] per-tile lists of column slots q
    int32_t *glo;                      // low-limb joint sums
    const int64_t *slot_pfix_hi;       // [L][5] marginals of the high-limb weights by slot
    int hi_shift;                      // G holds sums of V_hi = (V - V_lo) / 2^hi_shift
};
__host__ __device__ __forceinline__ int lo_class(int nrows) { return nrows <= 1 ? 0 : (nrows == 2 ? 1 : 2); }

// r04: a SPAN = several reference blocks that share their from side (one block row of make_blocks: R/computePairwiseMI.R:147-165), run as ONE
// launch sequence over the concatenated to side.  Everything per-SNP and per-row is as for a single block; what belongs to the REFERENCE
// block stays per segment: a column's local index (ColMeta::bl, counted from its segment's first SNP: pair identity a_loc != b_loc — quirk Q3 —,
// the upper / lower triangle, the row-order key a + b nf of the selection), RXY as the reference's linear index reads it (Q1), the histogram,
// candidate list and pick of the long-range filter (per block: R/computePairwiseMI.R:352-358).  A column carries its segment in
// ColMeta::ci.pad[0] and its segment's first to-side index in ci.pad[1] (k_build_packs).
constexpr int LDW_SPAN_MAX = 8;
struct SpanSeg {
    unsigned long long *n_cand;     // candidate counter of the segment (PickOut::n_cand)
    uint64_t *ckey, *cval;          // its candidate list
    unsigned long long *ghist;      // its NBINS histogram counters
};

// r05: what k_screen_maybe needs of a biallelic SNP, in 32 bytes instead of the ~200-byte ColMeta / RowPack (its entries arrive from the GEMM's
// epilogue in region order, ~17 M per span on data without rare states, and the kernel was bound by the gathers of the two full records:
// 1.32 ms per span).  Built by k_build_packs from the _hi packs (marginals of the APPROXIMATE weights: < 2^31 units); 0 / -1 for other SNPs.
struct MiniCol {
    int32_t pb0;        // approximate minor marginal (units of 2^e_last)
    float pY0, pY1;     // weighted marginals of the two states (ColMeta::pYf)
    float rb, rq;       // r of the SNP; Q1 on square blocks / spans: r[idx_f[b_loc]]
    int32_t bl, seg0;   // local index in its reference block; first to-side index of its segment (a span's ColInfo::pad[1])
    int32_t sb;
};
struct MiniRow {
    int32_t pa0, pa1;
    float pX0, pX1;
    float ra, rta;
    int32_t a_loc, sa;
};

struct PairEnt;
struct EpiArgs {
    const int64_t *G;
    int RFpad, RTpad;
    LoGeom lo;
    const int32_t *idx_f, *lrow_f, *idx_t, *lrow_t;
    int nf, nt;
    const uint32_t *slot_meta;
    const int64_t *slot_pfix;
    const double *r;
    double neff, scale;
    int quirk;
    // Both epilogue orders end with the SNPs that have >= 3 minor states or none: from-tiles >= gen_t0 and column slots
    // >= gen_q0 are the domain of k_mi_screen_generic, the rest that of k_mi_screen
    int gen_t0, gen_q0;
    // Ready-made per-block SNP constants (k_build_packs): what stage_cols / load_col / load_row_side otherwise derive through
    // a chain of 3-4 dependent loads (permutation -> SNP index -> slot meta / marginals / r / intervals).  null: derive.
    // The _hi copies carry the marginals of the high-limb weights in pb / pa (screen of the mixed-precision path).
    const struct ColMeta *colpack, *colpack_hi;   // [nt], epilogue order perm_t
    const struct RowPack *rowpack, *rowpack_hi;   // [64 * from-tiles], epilogue order perm_f (padded)
    const float *rloc_f, *rloc_t;                 // r of the from- / to-side SNPs by LOCAL index (quirk Q1 on ragged blocks)
    // approximate-GEMM path: long-range candidates of units WITHOUT a short-range pair are listed pair by pair (one in a
    // thousand pairs passes the screen: evaluating the whole unit of 64 for it wastes 98 %): PAIR_PATHS x PAIR_SHARDS lists of
    // pl_cap entries (from-slot index << 32 | column slot), list = path * PAIR_SHARDS + (workgroup & 7)
    struct PairEnt *pl_pairs;
    unsigned int *pl_n;
    uint32_t pl_cap;
    const int32_t *row0;      // first bit row of every SNP (pair entries carry the rows, so that k_pair_sums starts loading at once)
    // threshold table of the biallelic x biallelic pairs (k_build_tab11): entry [bin of the to-side minor marginal][bin of the
    // from-side one] = (Lq, Hq): a joint sum n' with Lq < n' < Hq cannot reach tab_lo whatever the rest of the pair looks like
    const int2 *tab11;
    int tab_nb;               // bins per side (sqrt scale: bin = min(tab_nb - 1, floor(sqrt(p) tab_c)))
    float tab_c;
    // regions of 32 column slots x one from-tile that the GEMM's epilogue found clean (every pair inside its table thresholds):
    // clean[(q / 32) * clean_stride + tile]; null: no such information
    const uint8_t *clean;
    int clean_stride;
    // Pruning flags of the block's SNPs in epilogue order (k_build_packs; null: none): PF_KIND = 2 or 3 flagged states with r to match
    // (0: anything else), PF_DEAD2 / PF_DEAD3 = the SNP cannot reach the block's level with ANY partner of kind 2 / 3 (k_snp_sup).
    // sflag_f[64 * tile + lane], sflag_t[column slot]
    const uint8_t *sflag_f, *sflag_t;
    const double *snp_sup;    // [L][4] (k_snp_sup)
    int sr_excl;              // 1: the block's short-range pairs are evaluated elsewhere (an SR sub-pass over the same block in list order): the
                              // screens only keep them out of the long-range candidates and never list a unit for them (E.any_sr is 0 then)
    const MiniCol *mini_c;    // [nt] / [64 * from-tiles] (k_screen_maybe); null: not built
    const MiniRow *mini_r;
    int span;                 // > 0: the to side is the concatenation of `span` reference blocks (segments), nt of each = nf
    SpanSeg sseg[LDW_SPAN_MAX];
    EmitArgs E;
};
constexpr unsigned PF_KIND = 3u, PF_DEAD2 = 4u, PF_DEAD3 = 8u, PF_PAD = 0x80u;
// may the pair of two SNPs with these flags be dropped unseen?  (either one is dead versus the other's kind)
__host__ __device__ __forceinline__ bool pf_pair_dead(unsigned fa, unsigned fb) {
    const unsigned ka = fa & PF_KIND, kb = fb & PF_KIND;
    if (ka < 2u || kb < 2u) return false;
    return (fa & (kb == 2u ? PF_DEAD2 : PF_DEAD3)) != 0u || (fb & (ka == 2u ? PF_DEAD2 : PF_DEAD3)) != 0u;
}
__host__ __device__ __forceinline__ int tab_bin(float p, float c, int nb) {
    const int b = (int)(sqrtf(p > 0.0f ? p : 0.0f) * c);
    return b < nb - 1 ? b : nb - 1;
}
// a listed candidate pair: from-slot index (64 * tile + lane), column slot, first bit row | row count << 29 of both SNPs
struct PairEnt {
    uint32_t t, q, ra, rb;
};
constexpr int PAIR_PATHS = 5;    // (NA, NB) = (1,1) (2,1) (1,2) (2,2) straight-line code, 4 = predicated
constexpr int PAIR_SHARDS = 8;

// arguments of the fused GEMM + epilogue kernel (ldw_fused.hip); A.G is unused there
struct FusedArgs {
    const uint64_t *Mbits;
    int64_t KW, Kpad;
    const int32_t *rowlist_t, *rowlist_f;
    const int8_t *digits;
    const int32_t *pos_f, *pos_t;   // local SNP index starting at each row position of the padded row lists, -1 elsewhere
    const uint8_t *cls_f, *cls_t;   // slot-count class (1, 2, 4) of every 32-row group of the row lists
    unsigned long long *ghist;      // NBINS counters of the long-range candidates
    EpiArgs A;
};
int launch_fused(ldw_ctx *ctx, const FusedArgs &F, int RFpad, int RTpad, int nlimbs, hipStream_t stream);
// arguments of the gathered low-limb GEMM (ldw_gemm_bits.hip)
struct LoGemmArgs {
    const uint64_t *Mbits;
    int64_t KW, Kpad;
    const int8_t *digits_lo;   // limbs 0 and 1
    const int32_t *perm_f, *idx_f, *perm_t, *idx_t, *row0;
    int32_t zero_row, nf, nt;
    const int32_t *tf_list;    // (tile, fs) pairs: the 64-slot from-side sub-tiles that exist (fs < cmax_f[tile])
    LoGeom lo;
};

int launch_gemm_lo_units(ldw_ctx *ctx, const LoGemmArgs &P, int n_tf, hipStream_t stream);

// everything the epilogue needs about one to-side SNP, staged in LDS once per workgroup so that the
// per-pair loop has no dependent global loads except its G entries
struct ColMeta {
    int32_t sb;
    uint32_t mb;
    int32_t rb0, bl;   // first row position in the to-side row list; local index of the SNP in the to-side list
    double rb;      // r of the to-side SNP
    double rq;      // Q1 on square blocks: r[idx_f[b_loc]]
    double pYd[5];
    int64_t pb[5];
    float pYf[5];
    int32_t pad2;
#ifdef LDW_COLMETA_PAD   // measurement only: how sensitive are the screens to the size of the staged column?
    char padx[LDW_COLMETA_PAD];
#endif
    ColInfo ci;
};

// per-lane constants of the from-side SNP
struct RowSide {
    int sa, na;
    uint32_t ma;
    int64_t ra0;
    double ra, rta;  // rta: Q1 on square blocks, r[idx_t[a_loc]]
    int64_t pa[5];
    double pXd[5];
    float pXf[5];
};

struct RowPack {
    RowSide R;
    int32_t a_loc;   // local index in the from-side list, -1: padding slot of the tile
    int32_t pad;
};

// MI of one pair.  NAM / NB bound the unrolled slot loops (na <= NAM for every lane of the wave, nb <= NB);
// the run-time slot counts still mask the individual cells.
// Where the joint sums of one pair live.  g points at G(slot 0 of a, slot 0 of b); slot i of the from-side SNP and slot j of
// the to-side SNP are at g[i * si + j * sj] (global G block: si = 1, sj = RFpad, or transposed on the mirrored half of a
// diagonal block; LDS tile of the fused kernel: si = 1, sj = its padded row stride).  In the mixed-precision path the
// block-wide GEMM only carries the HIGH limbs of the weights (g, to be shifted left by `shift` bits) and the low limbs of
// the listed units come from the gathered GEMM (l, int32, own strides): sum = (g << shift) + l.
struct GAcc {
    const int64_t *g;
    int64_t si, sj;
    const int32_t *l;
    int64_t li, lj;
    int shift;
    int w32;   // g addresses int32 entries (approximate-GEMM path)
    __device__ __forceinline__ int64_t at(int i, int j) const {
        if (w32) return (int64_t)reinterpret_cast<const int32_t *>(g)[i * si + j * sj];
        const int64_t v = g[i * si + j * sj];
        return l ? (v << shift) + (int64_t)l[i * li + j * lj] : v;
    }
};
__device__ __forceinline__ GAcc gacc_plain(const int64_t *g, int64_t si, int64_t sj) { return GAcc{g, si, sj, nullptr, 0, 0, 0, 0}; }

template <int NAM, int NB>
__device__ __forceinline__ double pair_mi(const EpiArgs &A, const RowSide &R, const ColMeta &M, int a_loc, int b_loc,
                                          bool square, const GAcc &Ga) {
    const int na = R.na, nb = M.mb & 7;
    const uint32_t ma = R.ma, mb = M.mb;
    // joint sums of the row slots (from G), their row / column sums
    int64_t g[NAM][NB], rs[NAM], cs[NB];
#pragma unroll
    for (int i = 0; i < NAM; ++i) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) cs[j] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NAM; ++i) {
            int64_t v = 0;
            if (i < na && j < nb) v = Ga.at(i, j);
            g[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int64_t pa_drop = 0;
#pragma unroll
    for (int i = 0; i <= NAM; ++i)
        if (i == na) pa_drop = R.pa[i];
    int64_t dd = pa_drop;
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (j < nb) dd -= M.pb[j] - cs[j];

    const double ra = R.ra, rb = M.rb;
    const double den = A.neff + (ra * rb) * 0.5;  // R/computePairwiseMI.R:260
    double RXY;
    if (A.quirk == LDW_QUIRK_REFERENCE) {
        // rft is nt x nf but read by the linear index c = a + b*nf of the nf x nt matrix (Q1):
        // 0.25 * rf[c / nt] * rt[c % nt]; on square blocks c / nt = b_loc and c % nt = a_loc
        if (A.span) {   // (a span's segments are square blocks; b_loc is the column's index in ITS block, rt that block's to side)
            RXY = (M.rq * (double)A.rloc_t[M.ci.pad[1] + a_loc]) * 0.25;
        } else if (square) {
            RXY = (M.rq * R.rta) * 0.25;
        } else {
            const uint32_t c = (uint32_t)a_loc + (uint32_t)b_loc * (uint32_t)A.nf;
            const uint32_t q = c / (uint32_t)A.nt;
            RXY = (A.r[A.idx_f[q]] * A.r[A.idx_t[c - q * (uint32_t)A.nt]]) * 0.25;
        }
    } else {
        RXY = (ra * rb) * 0.25;
    }
    const double rX = 0.5 * ra, rY = 0.5 * rb;

    // sum over cells of pxy * log(pxy / (pX pY + RXY + pX rX + pY rY) * den), divided by den at the end
    double acc = 0.0;
    const double half_m = fma(-4503599627370496.0, A.scale, 0.5);
#pragma unroll
    for (int i = 0; i <= NAM; ++i) {
        if (i <= na && ((ma >> (3 + i)) & 1)) {
            const double pX = R.pXd[i];
            const double pXr = fma(pX, rX, RXY);
#pragma unroll
            for (int j = 0; j <= NB; ++j) {
                if (j <= nb && ((mb >> (3 + j)) & 1)) {
                    int64_t nfix;
                    if (i < NAM && j < NB && i < na && j < nb) nfix = g[i < NAM ? i : 0][j < NB ? j : 0];
                    else if (i < NAM && i < na) nfix = R.pa[i] - rs[i < NAM ? i : 0];   // j == nb
                    else if (j < NB && j < nb) nfix = M.pb[j] - cs[j < NB ? j : 0];     // i == na
                    else nfix = dd;
                    const double pY = M.pYd[j];
                    const double pxy = pxy_of(nfix, A.scale, half_m);
                    const double d = fma(pY, rY, fma(pX, pY, pXr));
                    acc = fma(pxy, fast_half_log_ratio(pxy * den, d), acc);
                }
            }
        }
    }
    return acc * fast_rcp(0.5 * den);   // (the cells summed HALF logarithms)
}

// Straight-line variant for the common case: every lane has exactly NA row slots, the column has exactly NB,
// and every slot of both SNPs is flagged in uqe — no per-cell predication, every index static.
// The joint table in fixed point: rows = slots of the from-side SNP (the last one is the dropped state), columns = slots
// of the to-side SNP; the cells without an indicator row follow from the marginals by exact integer subtraction.
template <int NA, int NB>
struct FullCells {
    int64_t n[NA + 1][NB + 1];
};

// the int32 table of the approximate path from entries already in registers (g[j][i]: slot i of the from-side SNP, slot j of the to-side SNP):
// the same cells as full_cells computes (the approximate marginals and their differences fit 31 bits: they are sums of the int32 block)
template <int NA, int NB>
struct FullCells32 {
    int n[NA + 1][NB + 1];
};
template <int NA, int NB>
__device__ __forceinline__ void full_cells32(const RowSide &R, const ColMeta &M, const int (&g)[NB][NA], FullCells32<NA, NB> &C) {
    int rs[NA], cs[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) cs[j] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = g[j][i];
            C.n[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int dd = (int)R.pa[NA];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        C.n[NA][j] = (int)M.pb[j] - cs[j];
        dd -= C.n[NA][j];
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) C.n[i][NB] = (int)R.pa[i] - rs[i];
    C.n[NA][NB] = dd;
}

template <int NA, int NB>
__device__ __forceinline__ void full_cells(const RowSide &R, const ColMeta &M, const GAcc &Ga, FullCells<NA, NB> &C) {
    int64_t rs[NA > 0 ? NA : 1], cs[NB > 0 ? NB : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) cs[j] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int64_t v = Ga.at(i, j);
            C.n[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int64_t dd = R.pa[NA];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        C.n[NA][j] = M.pb[j] - cs[j];
        dd -= C.n[NA][j];
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) C.n[i][NB] = R.pa[i] - rs[i];
    C.n[NA][NB] = dd;
}

// RXY of src/computeMI.cpp:19 as the reference reads it (quirk Q1) or as intended
__device__ __forceinline__ double pair_rxy(const EpiArgs &A, const RowSide &R, const ColMeta &M, int a_loc, int b_loc, bool square) {
    if (A.quirk == LDW_QUIRK_REFERENCE) {
        if (A.span) return (M.rq * (double)A.rloc_t[M.ci.pad[1] + a_loc]) * 0.25;
        if (square) return (M.rq * R.rta) * 0.25;
        const uint32_t c = (uint32_t)a_loc + (uint32_t)b_loc * (uint32_t)A.nf;
        const uint32_t q = c / (uint32_t)A.nt;
        return (A.r[A.idx_f[q]] * A.r[A.idx_t[c - q * (uint32_t)A.nt]]) * 0.25;
    }
    return (R.ra * M.rb) * 0.25;
}

// The sum over the cells of a fully flagged (NA + 1) x (NB + 1) table from the cells' pxy (src/computeMI.cpp:19): the ONE fma chain every
// straight-line kernel runs, whatever way it came by the pxy (full_cells_mi: from the integer cells; k_mi_epilogue_fast: exact fp64 sums) and
// by den, rX, rY, hrcp = 1 / (den / 2) (per pair, or once per slot-count class where r is the class's) — equal tables give equal bits.
template <int NA, int NB>
__device__ __forceinline__ double cells_mi_x(const double (&x)[NA + 1][NB + 1], const double *pX, const double *pY, double RXY, double den, double rX,
                                             double rY, double hrcp) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i <= NA; ++i) {
        const double pXr = fma(pX[i], rX, RXY);
#pragma unroll
        for (int j = 0; j <= NB; ++j) {
            const double d = fma(pY[j], rY, fma(pX[i], pY[j], pXr));
            acc = fma(x[i][j], fast_half_log_ratio(x[i][j] * den, d), acc);
        }
    }
    return acc * hrcp;   // (the cells summed HALF logarithms)
}

template <int NA, int NB>
__device__ __forceinline__ double full_cells_mi(const EpiArgs &A, const RowSide &R, const ColMeta &M, double RXY,
                                                const FullCells<NA, NB> &C) {
    const double ra = R.ra, rb = M.rb;
    const double den = A.neff + (ra * rb) * 0.5;
    const double half_m = fma(-4503599627370496.0, A.scale, 0.5);
    double x[NA + 1][NB + 1];
#pragma unroll
    for (int i = 0; i <= NA; ++i)
#pragma unroll
        for (int j = 0; j <= NB; ++j) x[i][j] = pxy_of(C.n[i][j], A.scale, half_m);
    return cells_mi_x<NA, NB>(x, R.pXd, M.pYd, RXY, den, 0.5 * ra, 0.5 * rb, fast_rcp(0.5 * den));
}

// The same sum in fp32 with v_log_f32, an order of magnitude cheaper than the fp64 evaluation: a SCREEN.  In the
// speculative selection mode a long-range pair matters only if its MI reaches the guessed bucket, which one pair in
// a thousand does; the screen decides that for the rest within SCREEN_EPS and the exact evaluation runs only for
// waves that hold a pair which passes (or a short-range pair).  Error budget, in nats: every cell contributes
// (pxy/den) * (log2 u - log2 v) * ln2 with sum(pxy/den) = 1; v_log_f32 is good to 1 ulp of a result below 64 in
// magnitude (2^-18 = 3.8e-6), the operands carry <= 3 roundings of 2^-24 each and the top 31 bits of the fixed-point
// sum (<= 1e-7 on log2): |error| < 1.3e-5 ; SCREEN_EPS = 2e-4 leaves a factor of 15.  ldw_set_screen(ctx, 2) evaluates
// every pair both ways and counts the pairs the screen would have lost (tests assert zero).
constexpr float SCREEN_EPS = 2e-4f;

// APX (approximate-GEMM path): the cells are int32 sums of the approximate weights V' (units of 2^e_last) and the value
// returned is an UPPER BOUND of the MI of the exact sums.  With x = pxy of the exact sums and x' the one evaluated here:
// |x - x'| <= (delta x' + eta) / (1 - delta), eta = lost units * 2^(e_last - F) (truncations at the exponent transitions,
// floor of the marginals; every cell derived by subtraction inherits those of its terms), and f(x) = x ln(x den / d) has
// f' = ln(x den / d) + 1, so  den MI <= sum f(x') + delta/(1-delta) sum x' (|ln(x' den / d)| + 1 + c) + eta/(1-delta) sum (|ln| + 1 + c)
// with c = 2 (delta + 2 eta) / (1 - delta) < 0.02 and |ln| + 1 <= 2 ln(neff + 12.5) + 3 in the last sum (lo_bound, ldw_mi.hip).
// r03 (fully flagged tables, i.e. this function): f is convex, so f(x) <= f(x') + f'(x) (x - x') with f'(x) = ln(x den / d) + 1, and
//     sum_cells f'(x) (x - x') = sum ln(x den / d) (x - x')  +  sum (x - x').
// The first sum is bounded as before WITHOUT the "+ 1" (|ln(x den / d)| <= |ln(x' den / d)| + c).  The second is exact: every sequence
// lies in exactly one cell, so sum x = W + cells / 2, and the derived cells telescope — row i of the table sums to the approximate
// marginal pa'_i whatever the GEMM entries lost — so sum x' = (sum_i pa'_i) 2^(e_last - F) + cells / 2 (more where a negative derived
// cell was clamped): sum (x - x') <= W - (sum_i pa'_i) 2^(e_last - F), a per-SNP constant of a few 1e-2 (the weights are rounded to
// the NEAREST dual-digit product: the errors of the 57 weight classes of C4 largely cancel) instead of delta sum x' = 0.27.  The
// margin this removes is delta = 1.5e-3 nats: 2.77e6 -> 2.17e6 listed pairs per C4 pass (LDW_SCREEN_R02_BOUND restores the old form).
template <int NA, int NB, bool APX = false, typename FC = FullCells<NA, NB>>
__device__ __forceinline__ float full_cells_screen(const EpiArgs &A, const RowSide &R, const ColMeta &M, double RXY,
                                                   const FC &C) {
    const float ra = (float)R.ra, rb = (float)M.rb;
    const float den = (float)A.neff + (ra * rb) * 0.5f;
    const float rX = 0.5f * ra, rY = 0.5f * rb, rxy = (float)RXY;
    float acc = 0.0f, acc_abs = 0.0f, xsum = 0.0f;
#pragma unroll
    for (int i = 0; i <= NA; ++i) {
        const float pX = R.pXf[i];
        const float pXr = fmaf(pX, rX, rxy);
#pragma unroll
        for (int j = 0; j <= NB; ++j) {
            const float pY = M.pYf[j];
            float pxy;
            if (APX) {
                const int n = (int)C.n[i][j];
                pxy = fmaf((float)(n < 0 ? 0 : n), A.E.scr_scale, 0.5f);
            } else {
                pxy = fmaf((float)(uint32_t)(C.n[i][j] >> A.E.scr_shift), A.E.scr_scale, 0.5f);
            }
            const float d = fmaf(pY, rY, fmaf(pX, pY, pXr));
#ifdef LDW_ABLATE_SCREEN_LOGS   // timing ablation only (wrong results): what the two v_log_f32 per cell cost (tools/r03_ablate.sh)
            const float lt = (pxy * den - d) * 1e-30f;   // (tiny: nothing passes, the arithmetic around the logarithms stays)
#else
            const float lt = __builtin_amdgcn_logf(pxy * den) - __builtin_amdgcn_logf(d);
#endif
            acc = fmaf(pxy, lt, acc);
            if (APX) {
                acc_abs = fmaf(pxy, fabsf(lt), acc_abs);
                xsum += pxy;
            }
        }
    }
    if (APX) {
        // units (2^e_last) by which the cells of this table can be off, summed over the cells.  A GEMM entry n' is at most EG units
        // below its sum (truncation at the exponent transitions), a marginal less than 1 unit below (floor), and the errors of a
        // derived cell have OPPOSITE signs: row-derived (i, NB) = pa_i - sum_j n_ij is off by (-NB EG, MU), column-derived likewise,
        // the corner pa_NA - sum_j pb_j + sum_ij n_ij by (-NB MU, MU + NA NB EG); MU = 1, or 0 where the marginals are exact (EmitArgs::apx_MU).
        const float EG = A.E.apx_EG, MU = A.E.apx_MU;
        const float lost_units = (float)(NA * NB) * EG + (float)NA * fmaxf(MU, (float)NB * EG) + (float)NB * fmaxf(MU, (float)NA * EG) +
                                 fmaxf((float)NB * MU, MU + (float)(NA * NB) * EG);
        float dW = 0.0f;
        if (A.E.apx_W > 0.0) {   // sum (x - x') <= W - (sum of the from-side SNP's approximate marginals), in weight units (+ rounding room)
            int64_t ta = 0;
#pragma unroll
            for (int i = 0; i <= NA; ++i) ta += R.pa[i];
            dW = (float)(A.E.apx_W - (double)ta * A.E.apx_unit) + 4e-6f * (float)A.E.apx_W;
        }
        const float extra = A.E.apx_dfac * fmaf(acc_abs, 0.6931471805599453f, A.E.apx_c1 * xsum) + lost_units * A.E.apx_s1 + dW;
        return fmaf(acc, 0.6931471805599453f, extra) * __builtin_amdgcn_rcpf(den);
    }
    return acc * (0.6931471805599453f * __builtin_amdgcn_rcpf(den));
}

// The screen for ANY slot counts (SNPs with >= 3 minor states, or none, on either side): the predicated cell loop of
// pair_mi<4, 4> in fp32.  Same error budget as full_cells_screen (up to 25 cells instead of 9: the log terms are still
// weighted by pxy / den, which sum to 1).
template <bool APX = false>
__device__ __forceinline__ float pair_screen_generic(const EpiArgs &A, const RowSide &R, const ColMeta &M, double RXY, const GAcc &Ga) {
    const int na = R.na, nb = M.mb & 7;
    const uint32_t ma = R.ma, mb = M.mb;
    int64_t g[4][4], rs[4], cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t v = 0;
            if (i < na && j < nb) v = Ga.at(i, j);
            g[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int64_t pa_drop = 0;
#pragma unroll
    for (int i = 0; i <= 4; ++i)
        if (i == na) pa_drop = R.pa[i];
    int64_t dd = pa_drop;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nb) dd -= M.pb[j] - cs[j];
    const float ra = (float)R.ra, rb = (float)M.rb;
    const float den = (float)A.neff + (ra * rb) * 0.5f;
    const float rX = 0.5f * ra, rY = 0.5f * rb, rxy = (float)RXY;
    float acc = 0.0f, acc_abs = 0.0f, xsum = 0.0f;
#pragma unroll
    for (int i = 0; i <= 4; ++i) {
        if (i <= na && ((ma >> (3 + i)) & 1)) {
            const float pX = R.pXf[i];
            const float pXr = fmaf(pX, rX, rxy);
#pragma unroll
            for (int j = 0; j <= 4; ++j) {
                if (j <= nb && ((mb >> (3 + j)) & 1)) {
                    int64_t nfix;
                    if (i < 4 && j < 4 && i < na && j < nb) nfix = g[i < 4 ? i : 0][j < 4 ? j : 0];
                    else if (i < 4 && i < na) nfix = R.pa[i] - rs[i < 4 ? i : 0];
                    else if (j < 4 && j < nb) nfix = M.pb[j] - cs[j < 4 ? j : 0];
                    else nfix = dd;
                    const float pY = M.pYf[j];
                    float pxy;
                    if (APX) pxy = fmaf((float)(nfix < 0 ? 0 : (int)nfix), A.E.scr_scale, 0.5f);
                    else pxy = fmaf((float)(uint32_t)(nfix >> A.E.scr_shift), A.E.scr_scale, 0.5f);
                    const float d = fmaf(pY, rY, fmaf(pX, pY, pXr));
                    const float lt = __builtin_amdgcn_logf(pxy * den) - __builtin_amdgcn_logf(d);
                    acc = fmaf(pxy, lt, acc);
                    if (APX) {
                        acc_abs = fmaf(pxy, fabsf(lt), acc_abs);
                        xsum += pxy;
                    }
                }
            }
        }
    }
    if (APX) {   // see full_cells_screen: the units the cells can be off by, for na x nb indicator rows
        const float EG = A.E.apx_EG, MU = A.E.apx_MU, fa = (float)na, fb = (float)nb;
        const float lost_units = fa * fb * EG + fa * fmaxf(MU, fb * EG) + fb * fmaxf(MU, fa * EG) + fmaxf(fb * MU, MU + fa * fb * EG);
        // (tables with unflagged cells: the totals argument of full_cells_screen does not apply; the r02 form, apx_s1 + one unit of |ln| + 1)
        const float extra = A.E.apx_dfac * fmaf(acc_abs, 0.6931471805599453f, 1.02f * xsum) + lost_units * (A.E.apx_s1 + (float)A.E.apx_unit * 1.02f);
        return fmaf(acc, 0.6931471805599453f, extra) * __builtin_amdgcn_rcpf(den);
    }
    return acc * (0.6931471805599453f * __builtin_amdgcn_rcpf(den));
}

template <int NA, int NB>
__device__ __forceinline__ double pair_mi_full(const EpiArgs &A, const RowSide &R, const ColMeta &M, int a_loc, int b_loc,
                                               bool square, const GAcc &Ga) {
    FullCells<NA, NB> C;
    full_cells<NA, NB>(R, M, Ga, C);
    return full_cells_mi<NA, NB>(A, R, M, pair_rxy(A, R, M, a_loc, b_loc, square), C);
}

// ------------------------------------------------------------------------------------------------
// pieces shared by the screen / epilogue / unit kernels (ldw_mi.hip, ldw_apx.hip): all of them walk a block in the same
// units, one unit = the 64 from-side SNPs of a wave (perm_f order) x one to-side SNP (perm_t order); both orders group
// equal slot counts.
// ------------------------------------------------------------------------------------------------
// hi_cells (mixed-precision screen): the integer marginals pb / pa that the joint-table cells are derived from are those of
// the high-limb weights, consistent with the high-limb G; the floating-point marginals stay the exact ones.
__device__ __forceinline__ void stage_cols(const EpiArgs &A, const int32_t *__restrict__ perm_t, bool square, ColMeta *cm,
                                           bool hi_cells = false, int cgy = -1) {   // cgy: column group (default: blockIdx.y)
    if (threadIdx.x < EPI_COLS) {
        const int q = (cgy >= 0 ? cgy : (int)blockIdx.y) * EPI_COLS + threadIdx.x;
        if (q < A.nt && A.colpack) {
            cm[threadIdx.x] = (hi_cells ? A.colpack_hi : A.colpack)[q];
        } else if (q < A.nt) {
            const int b_loc = perm_t[q];
            ColMeta m;
            m.sb = A.idx_t[b_loc];
            m.mb = A.slot_meta[m.sb];
            m.rb0 = A.lrow_t[b_loc];
            m.bl = b_loc;
            m.rb = A.r[m.sb];
            m.rq = square ? A.r[A.idx_f[b_loc]] : 0.0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                m.pb[j] = A.slot_pfix[(int64_t)m.sb * 5 + j];
                m.pYd[j] = (double)m.pb[j] * A.scale;
                m.pYf[j] = (float)m.pYd[j];
                if (hi_cells) m.pb[j] = A.lo.slot_pfix_hi[(int64_t)m.sb * 5 + j];
            }
            m.pad2 = A.tab11 ? tab_bin(m.pYf[0], A.tab_c, A.tab_nb) : 0;
            if (A.E.cols) m.ci = A.E.cols[b_loc];
            cm[threadIdx.x] = m;
        }
    }
}

// the same for ONE column slot q, executed by every lane of a wave with uniform addresses (k_mi_units)
__device__ __forceinline__ void load_col(const EpiArgs &A, const int32_t *__restrict__ perm_t, bool square, int q, ColMeta &m,
                                         bool hi_cells = false) {
    if (A.colpack) {
        m = (hi_cells ? A.colpack_hi : A.colpack)[q];
        return;
    }
    const int b_loc = perm_t[q];
    m.sb = A.idx_t[b_loc];
    m.mb = A.slot_meta[m.sb];
    m.rb0 = A.lrow_t[b_loc];
    m.bl = b_loc;
    m.rb = A.r[m.sb];
    m.rq = square ? A.r[A.idx_f[b_loc]] : 0.0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        m.pb[j] = A.slot_pfix[(int64_t)m.sb * 5 + j];
        m.pYd[j] = (double)m.pb[j] * A.scale;
        m.pYf[j] = (float)m.pYd[j];
        if (hi_cells) m.pb[j] = A.lo.slot_pfix_hi[(int64_t)m.sb * 5 + j];
    }
    m.pad2 = A.tab11 ? tab_bin(m.pYf[0], A.tab_c, A.tab_nb) : 0;   // bin of the minor-state marginal (threshold table)
    if (A.E.cols) m.ci = A.E.cols[b_loc];   // (null: dense store only, ldw_mi_block — emit_pair does not look at the intervals then)
    else m.ci = ColInfo{};
}

// per-lane constants of the from-side SNP; returns whether the lane holds one
__device__ __forceinline__ bool load_row_side_at(const EpiArgs &A, const int32_t *__restrict__ perm_f, bool square, int t, RowSide &R,
                                                 int &a_loc, bool hi_cells) {
    // perm_f is padded with -1 so that every tile of 64 holds SNPs of ONE slot-count class (build_perm_tiles)
    const int pf = perm_f[t];
    const bool a_ok = pf >= 0;
    a_loc = a_ok ? pf : 0;
    R.sa = A.idx_f[a_loc];
    R.ma = A.slot_meta[R.sa];
    R.na = a_ok ? (int)(R.ma & 7) : 0;
    R.ra0 = A.lrow_f[a_loc];
    R.ra = A.r[R.sa];
    R.rta = (square && a_ok) ? A.r[A.idx_t[a_loc]] : 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        R.pa[i] = A.slot_pfix[(int64_t)R.sa * 5 + i];
        R.pXd[i] = (double)R.pa[i] * A.scale;
        R.pXf[i] = (float)R.pXd[i];
        if (hi_cells) R.pa[i] = A.lo.slot_pfix_hi[(int64_t)R.sa * 5 + i];
    }
    return a_ok;
}
__device__ __forceinline__ bool load_row_side(const EpiArgs &A, const int32_t *__restrict__ perm_f, bool square, int tile, RowSide &R,
                                              int &a_loc, bool hi_cells = false) {
    const int t = tile * 64 + (threadIdx.x & 63);
    if (A.rowpack) {
        const RowPack &P = (hi_cells ? A.rowpack_hi : A.rowpack)[t];
        R = P.R;
        a_loc = P.a_loc < 0 ? 0 : P.a_loc;
        return P.a_loc >= 0;
    }
    return load_row_side_at(A, perm_f, square, t, R, a_loc, hi_cells);
}

// every SNP of the wave's tile has the same slot count na0 (1 or 2) and all of its slots flagged in uqe.  The padding lanes
// of a partly filled tile (the last one of a class) do not count: they run the same straight-line code on the constants of
// SNP 0 and every use of their result is masked by a_ok — excluding such tiles left ALL of their units unscreened.
__device__ __forceinline__ bool wave_is_full(const RowSide &R, bool a_ok, int &na0) {
    const unsigned long long okm = __ballot(a_ok);
    if (okm == 0ull) {
        na0 = 0;
        return false;
    }
    na0 = __builtin_amdgcn_readlane(R.na, __builtin_ctzll(okm));
    const bool a_full = !a_ok || (R.na == na0 && (((R.ma >> 3) & ((2u << na0) - 1u)) == ((2u << na0) - 1u)));
    return (na0 == 1 || na0 == 2) && __ballot(!a_full) == 0ull;
}
__device__ __forceinline__ bool col_is_fast(uint32_t mb) {
    const int nb = (int)(mb & 7);
    return (nb == 1 || nb == 2) && (((mb >> 3) & ((2u << nb) - 1u)) == ((2u << nb) - 1u));
}

// Row lists are ordered by slot-count class, not by SNP index, so on a diagonal block (symmetric G, tiles above
// the diagonal of ROW positions skipped by the GEMM) the entry of a pair may only exist transposed.
__device__ __forceinline__ GAcc g_entry(const EpiArgs &A, const RowSide &R, const ColMeta &M) {
    const bool tr = A.E.lower_only && R.ra0 < (int64_t)M.rb0;
    const int64_t off = tr ? R.ra0 * A.RFpad + M.rb0 : (int64_t)M.rb0 * A.RFpad + R.ra0;
    if (A.E.apx) {
        GAcc g = gacc_plain(reinterpret_cast<const int64_t *>(reinterpret_cast<const int32_t *>(A.G) + off), tr ? (int64_t)A.RFpad : 1,
                            tr ? 1 : (int64_t)A.RFpad);
        g.w32 = 1;
        return g;
    }
    return gacc_plain(A.G + off, tr ? (int64_t)A.RFpad : 1, tr ? 1 : (int64_t)A.RFpad);
}

// RXY as the screens need it.  mode 0: intended (r_a r_b); 1: reference quirk Q1 on a square block (r[from[b_loc]] r[to[a_loc]],
// both staged per SNP); 3: the same on a span (r[to[a_loc]] of the column's segment: one look-up in rloc_t); 2: Q1 on a ragged block — the linear index c = a_loc + b_loc nf of the nf x nt matrix read as
// nt x nf: r[from[c / nt]] r[to[c % nt]], looked up in the per-block local-order tables.
__device__ __forceinline__ double screen_rxy(const EpiArgs &A, const RowSide &R, const ColMeta &M, int a_loc, int b_loc, int mode) {
    if (mode == 2) {
        const uint32_t c = (uint32_t)a_loc + (uint32_t)b_loc * (uint32_t)A.nf;
        const uint32_t q = c / (uint32_t)A.nt;
        return (double)(A.rloc_f[q] * A.rloc_t[c - q * (uint32_t)A.nt]) * 0.25;
    }
    if (mode == 3) return (M.rq * (double)A.rloc_t[M.ci.pad[1] + a_loc]) * 0.25;   // a span: Q1 of the column's own (square) block
    return (mode == 1 ? M.rq * R.rta : R.ra * M.rb) * 0.25;
}

// unit list entry (64 bits): from-tile * nt + column slot (bits 0-30), index k of the unit in its (tile, class) list (bits
// 31-50) and the class (bits 51-52) in the mixed-precision path; bit 63: verify mode only, a unit the screen dismissed
constexpr uint64_t UNIT_DISMISSED = 0x8000000000000000ull;


}  // namespace ldw
