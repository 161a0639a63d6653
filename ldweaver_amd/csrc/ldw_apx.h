// Approximate-GEMM path of the speculative blocks (ldw_apx.hip), shared declarations.
//
// The block-wide co-occurrence pass only has to feed the fp32 SCREEN (which pairs can reach the long-range threshold at
// all?), so it may use approximate weights as long as the error is bounded rigorously:
//     V_p  ~  V'_p = a_p * b_p * 2^e(m),      a_p, b_p in 0..127, m = p / 128 (macro step of position p)
// Positions ascend in weight (ldw_set_weights), so the exponent e(m) is non-decreasing and changes at a handful of macro
// steps; the accumulators are shifted right by e(m) - e(m-1) there and end in units of 2^e_last.  With BOTH operands
// masked by a digit (to side: a, from side: b) ONE int8 MFMA pass carries ~12 bits of every weight instead of the 8 of a
// single limb: the pass that took 3 (mixed precision) or 5 limbs takes 1.
//
// What the screen lists gets EXACT joint sums.  Long-range candidates (one pair in a few thousand) are listed PAIR by pair and
// need no GEMM at all: sum_s V_s x_s y_s = sum over 32-bit words and weight classes of V_class popcount(x & y & class mask), one
// wave per pair, lane = word (k_pair_sums), then the fp64 MI (k_pair_mi).  Units that hold a short-range pair are listed
// whole — the short-range band is dense — and evaluated from the exact 5-limb GEMM of just the tiles the band touches.
#pragma once
#include "ldw_internal.h"
#include "ldw_epi.h"

namespace ldw {

// one (32-bit word, weight class) intersection of the position axis
struct PopSeg {
    uint32_t mask;    // bits of the word that belong to the class
    uint32_t flush;   // 1: last segment of its class: fold the class count into the 64-bit sums with weight V
    int64_t V;        // fixed-point weight of the class
};

constexpr int APX_TW = 128;      // rows per side of one wave tile of the approximate GEMM (4 x 4 MFMA tiles of 32 x 32)
struct ApxGemmArgs {
    const uint64_t *panel_t, *panel_f;   // [M2][Rpad][2]: words 2m, 2m+1 of row list position r — for gemm_apx_kernel as the SCALED pieces of k_pack_panel (16 bytes per word)
    int RTpad, RFpad, M2;
    const uint8_t *dig_a, *dig_b;        // [128 * M2] by position
    const int32_t *shift;                // [4 M2]: right shift before k-step k (32 positions)
    int32_t *G;                          // [RTpad][RFpad]
    int lower_only;
    int fine;                            // exponents per k-step (panels interleaved by k_pack_panel) instead of per macro step
    // Threshold-table test in the epilogue (long-range-only blocks): bin_t / bin_f give every row of the two row lists its bin of
    // the 64 x 64 table `tab` (255: not a biallelic r = 2 SNP's row).  A region of 32 to-rows x 64 from-rows in which every
    // entry n' satisfies Lq < n' < Hq — what the screen would find for every one of its 2048 pairs — is flagged in
    // clean[(trow / 32) * (RFpad / 64) + fcol / 64] and NOT stored: the screen skips it without reading anything.
    int fuse;
    const uint8_t *bin_t, *bin_f;
    const int2 *tab;
    int tab_nb;
    uint8_t *clean;
    const uint8_t *sr_mask;              // optional [RTpad / 128][RFpad / 64]: tiles that hold a short-range pair are never clean
    // Tile pruning (P.fuse only): a wave tile whose rows all carry table bins and whose bin rectangle [min, max] x [min, max] holds
    // only unconditional entries (-1, INT_MAX) — no n' can make such a pair fail the test — flags its regions clean and leaves
    // before the K loop; lane 0 counts it in *skip_ctr.  Null: every tile is computed.
    unsigned long long *skip_ctr;
    // with skip_ctr: k_apx_live_tiles writes the numbers of the wave tiles that have to be computed to tile_list (ty * ntx + tx, ntx =
    // RFpad / 64) and their count to *n_live (zeroed by the caller); the GEMM then runs over that list, four tiles per workgroup
    uint32_t *tile_list;
    unsigned int *n_live;
    // the same for the wider tables: pruning flags (PF_*, ldw_epi.h) of every ROW of the two row lists, 0 for padding rows.  A tile
    // whose to-rows are all of one kind and dead versus the (one) kind of its from-rows, or the other way round, is not computed.
    const uint8_t *rflag_t, *rflag_f;
    // r04 — failing pairs straight from the accumulators (long-range blocks with pair lists, not the diagonal ones): a table-eligible region
    // with at most APX_MAYBE_MAX entries outside their thresholds appends those entries {to row, from row, n'} to `maybe` (capacity maybe_cap;
    // *maybe_n counts, and may exceed the capacity: the consumer, k_screen_maybe, then forces the block's overflow path) and is flagged CLEAN:
    // it is neither stored nor screened.  Null: such a region is stored and screened whole, as before.
    struct ApxMaybe *maybe;
    unsigned int *maybe_n;
    unsigned int maybe_cap;
};
struct ApxMaybe {
    uint32_t trow, fcol;   // row-list positions = column slot / from slot of the epilogue orders (biallelic rows: position == slot)
    int32_t n;             // the approximate joint sum n'
};
#ifndef LDW_MAYBE_MAX
#define LDW_MAYBE_MAX 96
#endif
constexpr int APX_MAYBE_MAX = LDW_MAYBE_MAX;   // more failing entries than this in a region of 2048: storing the region is the cheaper path

// (rowlist2 / Rpad2 / panel2: a second panel in the same launch)
int launch_pack_panel(ldw_ctx *c, const int32_t *rowlist, int Rpad, uint64_t *panel, hipStream_t st, const int32_t *rowlist2 = nullptr, int Rpad2 = 0,
                      uint64_t *panel2 = nullptr);
int launch_apx_live_tiles(ldw_ctx *c, const ApxGemmArgs &P, hipStream_t st);   // P.skip_ctr set: before launch_gemm_apx, same stream
int launch_gemm_apx(ldw_ctx *c, const ApxGemmArgs &P, hipStream_t st);
// the pair lists of A (filled by the approximate screen): exact sums, fp64 MI, emission
int launch_pairs_exact(ldw_ctx *c, const EpiArgs &A, unsigned long long *ghist, int64_t *sums, hipStream_t st);

}  // namespace ldw
