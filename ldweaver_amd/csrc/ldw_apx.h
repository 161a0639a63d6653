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
// The units the screen lists (a few per cent) get their EXACT joint sums without any GEMM: sequences of equal weight
// are contiguous, so sum_s V_s x_s y_s = sum_classes V_c popcount(x & y over the class) — 32 sequences per v_and +
// v_bcnt, the class weights folded in with one 64-bit multiply-add per class (k_units_pop).  The exact sums then pass the
// fp32 screen once more (margin SCREEN_EPS only) before the fp64 evaluation (k_mi_units).
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
// Entry of a per-(from-tile, to-class) unit list A.lo.tl: bits 0-28 the column slot q.  The approximate screen sets bit 31 in
// verify mode for a unit it would dismiss; k_units_pop rewrites the entry with the verdict of the exact re-screen:
constexpr uint32_t UNIT_TL_DISMISSED = 0x80000000u;   // verify mode: evaluate anyway, count a violation if it would emit
constexpr uint32_t UNIT_TL_DROPPED = 0x40000000u;     // ruled out by the exact sums: skip
constexpr uint32_t UNIT_TL_GENERIC = 0x20000000u;     // needs the predicated fp64 code (k_mi_units_tl<false>)
constexpr uint32_t UNIT_TL_Q = 0x1FFFFFFFu;
// exact joint sums of unit k of list (tile, lc): cs + cs_base[tile * 3 + lc] + k * (64 * CF * CT), slot (i, j) of lane l at
// + (j * CF + i) * 64 + l, CF = cmax_f[tile], CT = 1 << lc

struct ApxGemmArgs {
    const uint64_t *panel_t, *panel_f;   // [M2][Rpad][2]: words 2m, 2m+1 of row list position r
    int RTpad, RFpad, M2;
    const uint8_t *dig_a, *dig_b;        // [128 * M2] by position
    const int32_t *shift;                // [M2]
    int32_t *G;                          // [RTpad][RFpad]
    int lower_only;
};

struct PopArgs {
    const uint64_t *Mbits;       // row-major bit rows [R + TILE][KW] (to side: wave-uniform loads)
    int64_t KW;
    const uint64_t *panel_f;     // packed from-side panel [M2][RFpad][2]
    int RFpad, M2;
    const PopSeg *segs;
    const int32_t *wbeg;         // [4 * M2 + 1] first segment of every 32-bit word
    const int32_t *perm_t, *idx_t, *row0, *cmax_f;
    int32_t zero_row;
    int nseg;                    // segment records
    int debug;                   // LDW_POP_DEBUG: timing experiments (1: one macro step only, 2: no epilogue)
    int x_shift;                 // fp32 screen of the EXACT sums: n >> x_shift, times x_scale (EmitArgs::scr_shift / scr_scale of the limb paths)
    float x_scale;
    EpiArgs A;                   // rowpack / colpack, lists (A.lo.cnt, A.lo.tl, A.lo.uoff), emission constants
    int64_t *cs;                 // exact joint sums of the kept units
    const int64_t *cs_base;      // [ntiles * 3] start of each list's sums (int64 units)
};

int launch_pack_panel(ldw_ctx *c, const int32_t *rowlist, int Rpad, uint64_t *panel, hipStream_t st);
int launch_gemm_apx(ldw_ctx *c, const ApxGemmArgs &P, hipStream_t st);
// launches every (CF, CT) variant that can have work: n_tiles_cf[k] from-tiles of class 1, 2, 4; A.lo.n_lc[k] to-side SNPs
// the pair lists of A (filled by the approximate screen): exact sums, fp64 MI, emission
int launch_pairs_exact(ldw_ctx *c, const EpiArgs &A, unsigned long long *ghist, int64_t *sums, hipStream_t st);
int launch_units_pop(ldw_ctx *c, const PopArgs &P, int nf_tiles, const int n_tiles_cf[3], hipStream_t st);

}  // namespace ldw
