// MI epilogue, link selection and the block drivers (twins of perform_MI_computation_ACGTN,
// R/computePairwiseMI.R:167-386, and of the block loop of perform_MI_computation, :103-116).
//
// First block of a call sequence (no histogram-bucket guess yet), speculation misses, ldw_mi_block:
//   GEMM, 5 limbs (ldw_gemm_bits.hip) -> k_mi_epilogue: one thread per SNP pair turns the fixed-point joint sums into MI
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

// the exact-sum block of pipeline slot s
static inline ldw::DevBuf &gx(ldw_ctx *c, int s) { return s == 0 ? c->G : (s == 1 ? c->G2 : c->G3); }

// Per-block SNP constants in epilogue order, once per block instead of once per workgroup (k_mi_screen) or per unit
// (k_mi_units): blockIdx.y = 0: thread i builds column slot i (i < nt); blockIdx.y = 1: from-side slot i (i < 64 * tiles) — two independent chains
// of dependent loads, side by side instead of one after the other in the same thread (24 -> 13 us per C4 block); A.colpack / A.rowpack are null here.
__global__ __launch_bounds__(256) void k_build_packs(EpiArgs A, const int32_t *__restrict__ perm_f, const int32_t *__restrict__ perm_t,
                                                     int nf_slots, int with_hi, ColMeta *__restrict__ cp, ColMeta *__restrict__ cp_hi,
                                                     RowPack *__restrict__ rp, RowPack *__restrict__ rp_hi, float *__restrict__ rloc_f,
                                                     float *__restrict__ rloc_t, uint8_t *__restrict__ bin_t = nullptr,
                                                     uint8_t *__restrict__ bin_f = nullptr, int RTpad = 0, int RFpad = 0,
                                                     uint8_t *__restrict__ sflag_f = nullptr, uint8_t *__restrict__ sflag_t = nullptr,
                                                     uint8_t *__restrict__ rflag_f = nullptr, uint8_t *__restrict__ rflag_t = nullptr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool square = A.nf == A.nt;
    // pruning flags (PF_*): kind of the SNP and whether its bound (k_snp_sup) stays below the block's level for partners of kind 2 / 3;
    // by epilogue slot for the screen, by row of the row lists (zeroed beforehand: padding rows) for the approximate GEMM
    const double lvl = A.E.spec_lo - (double)A.E.scr_eps;
    auto prune_flags = [&](int snp, uint32_t meta, double r) -> unsigned {
        const int n = (int)(meta & 7);
        const uint32_t full = (2u << n) - 1u;
        if (!(n == 1 || n == 2) || ((meta >> 3) & full) != full || r != (double)(n + 1)) return 0u;
        const double *sp = A.snp_sup + (int64_t)snp * 4 + (A.quirk == LDW_QUIRK_REFERENCE ? 2 : 0);
        return (unsigned)(n + 1) | (sp[0] < lvl ? PF_DEAD2 : 0u) | (sp[1] < lvl ? PF_DEAD3 : 0u);
    };
    // bins of the threshold table by ROW of the two row lists for the GEMM's epilogue test (only used when the host has checked
    // that a one-row SNP's position in its row list equals its slot here: no SNP without a row in the block)
    int my_bt = 255, my_bf = 255;
    const bool to_side = blockIdx.y == 0;
    if (!to_side && i < A.nf) rloc_f[i] = (float)A.r[A.idx_f[i]];
    if (to_side && i < A.nt) rloc_t[i] = (float)A.r[A.idx_t[i]];
    if (to_side && i < A.nt) {
        ColMeta m;
        load_col(A, perm_t, square, i, m, false);
        if (A.span) {   // a column of a span: its index within ITS reference block, and the r of quirk Q1 on that (square) block
            m.bl -= m.ci.pad[1];
            m.rq = A.r[A.idx_f[m.bl]];
        }
        cp[i] = m;
        if ((m.mb & 7) == 1 && col_is_fast(m.mb) && m.rb == 2.0) my_bt = m.pad2 & 63;
        if (sflag_t) {
            const unsigned f = prune_flags(m.sb, m.mb, m.rb);
            sflag_t[i] = (uint8_t)f;
            for (int j = 0; j < (int)(m.mb & 7); ++j) rflag_t[m.rb0 + j] = (uint8_t)f;
        }
        if (with_hi) {
#pragma unroll
            for (int j = 0; j < 5; ++j) m.pb[j] = A.lo.slot_pfix_hi[(int64_t)m.sb * 5 + j];
            cp_hi[i] = m;
        }
    }
    if (!to_side && i < nf_slots) {
        RowPack P;
        int a_loc;
        const bool ok = load_row_side_at(A, perm_f, square, i, P.R, a_loc, false);
        P.a_loc = ok ? a_loc : -1;
        P.pad = A.tab11 ? tab_bin(P.R.pXf[0], A.tab_c, A.tab_nb) : 0;   // bin of the minor-state marginal (threshold table)
        rp[i] = P;
        if (ok && P.R.na == 1 && ((P.R.ma >> 3) & 3u) == 3u && P.R.ra == 2.0) my_bf = P.pad & 63;
        if (sflag_f) {
            const unsigned f = ok ? prune_flags(P.R.sa, P.R.ma, P.R.ra) : PF_PAD;
            sflag_f[i] = (uint8_t)f;
            if (ok)
                for (int j = 0; j < P.R.na; ++j) rflag_f[P.R.ra0 + j] = (uint8_t)f;
        }
        if (with_hi) {
#pragma unroll
            for (int k = 0; k < 5; ++k) P.R.pa[k] = A.lo.slot_pfix_hi[(int64_t)P.R.sa * 5 + k];
            rp_hi[i] = P;
        }
    }
    if (to_side && bin_t && i < RTpad) bin_t[i] = (uint8_t)my_bt;
    if (!to_side && bin_f && i < RFpad) bin_f[i] = (uint8_t)my_bf;
}

// ------------------------------------------------------------------------------------------------
// k_mi_screen: the fp32 screen of the two-kernel path (speculative selection mode).  A long-range pair only matters if
// its MI reaches the guessed histogram bucket, which about one pair in a thousand does; this kernel bounds MI in fp32
// (full_cells_screen, ldw_epi.h) for every unit whose SNPs have 1 or 2 fully flagged slots and writes ONE byte per
// unit: does any of its 64 pairs need the exact value (short-range pair, or upper bound >= the bucket's lower edge)?
// k_mi_epilogue then evaluates only the flagged units (and the units this kernel does not handle).  The kernel is kept
// free of the fp64 path on purpose: ~70 VGPRs instead of 145, so 6-7 waves per SIMD hide the latency of the G loads
// that bound the one-kernel epilogue (one 512-B load in flight per wave, 3 waves per SIMD: 1.3 TB/s), and U columns per
// iteration put U independent loads in flight per wave.
// ------------------------------------------------------------------------------------------------
// long-range candidates of one column of one wave -> pair list `path` (approximate-GEMM path)
__device__ __forceinline__ void append_pairs(const EpiArgs &A, int path, unsigned long long m, bool mine, uint32_t t, uint32_t q, int sa, int sb) {
    const int lane = threadIdx.x & 63;
    const int sub = path * PAIR_SHARDS + (int)((t >> 6) & (PAIR_SHARDS - 1));   // (t = 64 * from-tile + lane: sharded by tile)
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(A.pl_n + sub, (unsigned int)__popcll(m));
    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
    if (mine) {
        const unsigned int pos = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < A.pl_cap) {   // overflow: k_pick_bucket sees the counter
            const int32_t r0a = A.row0[sa], r0b = A.row0[sb];
            PairEnt e;
            e.t = t;
            e.q = q;
            e.ra = (uint32_t)r0a | ((uint32_t)(A.row0[sa + 1] - r0a) << 29);
            e.rb = (uint32_t)r0b | ((uint32_t)(A.row0[sb + 1] - r0b) << 29);
            A.pl_pairs[(int64_t)sub * A.pl_cap + pos] = e;
        }
    }
}

template <int NA, int NB, int U, int RM, bool APX>
__device__ __forceinline__ unsigned int screen_cols(const EpiArgs &A, const RowSide &R, const ColMeta *cmu, int a_loc, bool a_ok, float lo, int q0, int tile) {
    const bool test_sr = A.E.any_sr != 0 || A.sr_excl != 0, keep_sr = A.E.keep_sr != 0 && !A.sr_excl, do_lr = A.E.do_lr != 0;
    FullCells<NA, NB> C[U];
    if (do_lr) {   // an SR-only pass needs no MI here at all: a unit is wanted iff it holds a short-range pair
#pragma unroll
        for (int u = 0; u < U; ++u) full_cells<NA, NB>(R, cmu[u], g_entry(A, R, cmu[u]), C[u]);
    }
    unsigned int bits = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const ColMeta &M = cmu[u];
        const int b_loc = M.bl;
        const bool act = a_ok && (A.E.lower_only ? a_loc > b_loc : a_loc != b_loc);
        float ms = 0.0f;
        if (do_lr) ms = full_cells_screen<NA, NB, APX>(A, R, M, screen_rxy(A, R, M, a_loc, b_loc, RM), C[u]);
        const bool is_sr = test_sr && col_is_sr(M.ci, a_loc);
        if (APX && A.pl_pairs) {
            // a unit with a short-range pair is evaluated whole (its band is dense); otherwise only the candidates themselves
            const bool need_lr = act && !is_sr && do_lr && ms >= lo;
            if (__ballot(act && is_sr && keep_sr) != 0ull) {
                bits |= 1u << u;
            } else {
                const unsigned long long m = __ballot(need_lr);
                if (m != 0ull) append_pairs(A, (NA - 1) + 2 * (NB - 1), m, need_lr, (uint32_t)(tile * 64 + (threadIdx.x & 63)), (uint32_t)(q0 + u), R.sa, M.sb);
            }
        } else {
            const bool need = act && (is_sr ? keep_sr : (do_lr && ms >= lo));
            if (__ballot(need) != 0ull) bits |= 1u << u;
        }
    }
    return bits;
}

#ifdef LDW_SCREEN_STATS
__device__ unsigned long long g_scr_stats[16];
#endif
// r04: the same columns on the approximate path with pair lists, for blocks whose entries are never read transposed (all but the diagonal
// ones).  Measured (LDW_SCREEN_EXP, profiles/r04_screen_breakdown.txt): 73 % of a span's screen is this multi-cell evaluation, and its waves
// spend 73 % of their cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) with the VALU at 45 %: two columns' entries in flight per wave were too
// few.  Here the entries of all U columns (and the r look-ups of quirk Q1 on a span) are requested up front — to-side row base in SGPRs, the
// lane's from-side row as a 32-bit offset — and the tables are built in int32 from registers, one column at a time.
template <int NA, int NB, int U, int RM>
__device__ __forceinline__ unsigned int screen_cols_apx(const EpiArgs &A, const RowSide &R, const ColMeta *cmu, int a_loc, bool a_ok, float lo, int q0,
                                                        int tile) {
    const bool test_sr = A.E.any_sr != 0 || A.sr_excl != 0, keep_sr = A.E.keep_sr != 0 && !A.sr_excl, do_lr = A.E.do_lr != 0;
    const int32_t *G32 = reinterpret_cast<const int32_t *>(A.G);
    int raw[U][NB][NA];
    float rl[U];
    if (do_lr) {   // (an SR-only pass needs no MI here at all: a unit is wanted iff it holds a short-range pair)
        if (!A.E.lower_only) {
            const uint32_t ra0 = (uint32_t)R.ra0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rb0 = __builtin_amdgcn_readfirstlane(cmu[u].rb0);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int32_t *row = G32 + (int64_t)(rb0 + j) * A.RFpad;
#pragma unroll
                    for (int i = 0; i < NA; ++i) raw[u][j][i] = row[ra0 + (uint32_t)i];
                }
            }
        } else {
            // a diagonal block: the GEMM skips the tiles above the diagonal of ROW positions, so the entry of a pair may only exist transposed
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rb0 = cmu[u].rb0;
                const bool tr = R.ra0 < rb0;
                const int32_t *g = G32 + (tr ? R.ra0 * A.RFpad + rb0 : rb0 * A.RFpad + R.ra0);
                const int64_t si = tr ? (int64_t)A.RFpad : 1, sj = tr ? 1 : (int64_t)A.RFpad;
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int i = 0; i < NA; ++i) raw[u][j][i] = g[i * si + j * sj];
            }
        }
        if (RM == 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) rl[u] = A.rloc_t[__builtin_amdgcn_readfirstlane(cmu[u].ci.pad[1]) + (a_loc < 0 ? 0 : a_loc)];
        }
    }
    unsigned int bits = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const ColMeta &M = cmu[u];
        const int b_loc = M.bl;
        float ms = 0.0f;
        if (do_lr) {
            FullCells32<NA, NB> C;
            full_cells32<NA, NB>(R, M, raw[u], C);
            // (r is a small integer: the products are exact in fp32, the value is the one screen_rxy returns)
            const float rxy = RM == 3 ? ((float)M.rq * rl[u]) * 0.25f : (float)screen_rxy(A, R, M, a_loc, b_loc, RM);
#ifdef LDW_ABLATE_SCREEN_MATH   // timing ablation only (wrong results): the loads and the control flow without the bound's arithmetic
            ms = (float)(C.n[NA][NB] ^ C.n[0][0]) * 1e-30f + rxy * 1e-30f;
#else
            ms = full_cells_screen<NA, NB, true>(A, R, M, (double)rxy, C);
#endif
#ifdef LDW_SCREEN_STATS   // measurement build only: how often would a cheaper first-level bound let a pair / a (wave, column) through?
            {
                const float ra = (float)R.ra, rb = (float)M.rb, den = (float)A.neff + (ra * rb) * 0.5f, Lden = __builtin_amdgcn_logf(den);
                float S = 0.0f, SA = 0.0f, xs = 0.0f, SC = 0.0f;
#pragma unroll
                for (int i = 0; i <= NA; ++i)
#pragma unroll
                    for (int j = 0; j <= NB; ++j) {
                        const int nn = C.n[i][j];
                        const float x = fmaf((float)(nn < 0 ? 0 : nn), A.E.scr_scale, 0.5f);
                        const float dl = (R.pXf[i] + 1.0f) * (M.pYf[j] + 1.0f);
                        const float l = __builtin_amdgcn_logf(x) + Lden - __builtin_amdgcn_logf(dl);
                        S = fmaf(x, l, S);
                        SA = fmaf(x, fabsf(l), SA);
                        xs += x;
                        SC += x * x * den * __builtin_amdgcn_rcpf(dl) - x;
                    }
                const float EG = A.E.apx_EG;
                const float lost_units = (float)(NA * NB) * EG + (float)NA * fmaxf(1.0f, (float)NB * EG) + (float)NB * fmaxf(1.0f, (float)NA * EG) +
                                         fmaxf((float)NB, 1.0f + (float)(NA * NB) * EG);
                int64_t ta = 0;
#pragma unroll
                for (int i = 0; i <= NA; ++i) ta += R.pa[i];
                const float dW = (float)(A.E.apx_W - (double)ta * A.E.apx_unit) + 4e-6f * (float)A.E.apx_W;
                const float ex = A.E.apx_dfac * fmaf(SA + 2.65f * xs, 0.6931471805599453f, A.E.apx_c1 * xs) + lost_units * A.E.apx_s1 + dW;
                const float U1 = fmaf(S, 0.6931471805599453f, ex) * __builtin_amdgcn_rcpf(den), U2 = (SC + ex) * __builtin_amdgcn_rcpf(den);
                const bool actl = a_ok && a_loc != M.bl;
                const unsigned long long b0 = __ballot(actl), b1 = __ballot(actl && ms >= lo), b2 = __ballot(actl && U1 >= lo), b3 = __ballot(actl && U2 >= lo),
                                         v1 = __ballot(actl && U1 < ms), v2 = __ballot(actl && U2 < ms);
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&g_scr_stats[0], (unsigned long long)__popcll(b0));
                    atomicAdd(&g_scr_stats[1], (unsigned long long)__popcll(b1));
                    atomicAdd(&g_scr_stats[2], (unsigned long long)__popcll(b2));
                    atomicAdd(&g_scr_stats[3], (unsigned long long)__popcll(b3));
                    atomicAdd(&g_scr_stats[4], 1ull);
                    atomicAdd(&g_scr_stats[5], b1 ? 1ull : 0ull);
                    atomicAdd(&g_scr_stats[6], b2 ? 1ull : 0ull);
                    atomicAdd(&g_scr_stats[7], b3 ? 1ull : 0ull);
                    atomicAdd(&g_scr_stats[8], (unsigned long long)__popcll(v1));
                    atomicAdd(&g_scr_stats[9], (unsigned long long)__popcll(v2));
                }
            }
#endif
        }
        const bool act = a_ok && (A.E.lower_only ? a_loc > b_loc : a_loc != b_loc);
        const bool is_sr = test_sr && col_is_sr(M.ci, a_loc);
        if (A.pl_pairs) {
            // a unit with a short-range pair is evaluated whole (its band is dense); otherwise only the candidates themselves
            const bool need_lr = act && !is_sr && do_lr && ms >= lo;
            if (test_sr && __ballot(act && is_sr && keep_sr) != 0ull) {
                bits |= 1u << u;
            } else {
                const unsigned long long m = __ballot(need_lr);
                if (m != 0ull) append_pairs(A, (NA - 1) + 2 * (NB - 1), m, need_lr, (uint32_t)(tile * 64 + (threadIdx.x & 63)), (uint32_t)(q0 + u), R.sa, M.sb);
            }
        } else {   // (verify mode: whole units)
            const bool need = act && (is_sr ? keep_sr : (do_lr && ms >= lo));
            if (__ballot(need) != 0ull) bits |= 1u << u;
        }
    }
    return bits;
}

// ------------------------------------------------------------------------------------------------
// Threshold table of the biallelic x biallelic pairs (both SNPs: one indicator row, r = 2).  With the marginals fixed the
// joint table has ONE free number, x = pxy of (minor, minor), and MI is convex in it with its minimum at independence: MI
// reaches a level lo only for x <= L or x >= H.  L and H depend on the two minor marginals alone (RXY >= 1 with equality in the
// intended mode, and MI falls as RXY grows: the table is built for RXY = 1), so they are tabulated once per weighting over
// bins of the two marginals — conservatively: the largest L and the smallest H over the bin (sampled on a 3 x 3 grid, L and H
// ascend with both marginals) — and shifted by everything the approximate sum n' may be off: H (1 - delta) - eta, L (1 + delta)
// + eta, in units of the int32 block.  The screen then dismisses a pair with Lq < n' < Hq on two integer compares and runs the
// 25-cell log evaluation only for the columns in which some lane fails that test (one pair in a few thousand does).
// Every dismissal is still checked in verify mode (ldw_set_screen 2).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mi11_exact(double x, double pa, double pb, double W, double den) {
    const double A0 = pa + 1.0, A1 = W - pa + 1.0, B0 = pb + 1.0, B1 = W - pb + 1.0;
    const double x01 = A0 - x, x10 = B0 - x, x11 = den - A0 - B0 + x;
    if (!(x > 0.0 && x01 > 0.0 && x10 > 0.0 && x11 > 0.0)) return 1e30;
    return (x * log(x * den / (A0 * B0)) + x01 * log(x01 * den / (A0 * B1)) + x10 * log(x10 * den / (A1 * B0)) + x11 * log(x11 * den / (A1 * B1))) / den;
}

// 32 lanes per bin pair: lanes 0..8 bisect H for one point of the 3 x 3 sample grid each, lanes 16..24 bisect L; xor-shuffles
// combine them.  A bisection is a chain of dependent fp64 logarithms — latency, not throughput — so the table is built wide:
// 512 workgroups, 20 steps per lane (always ending on the safe side: fewer steps only loosen a threshold by interval / 2^20),
// cheap enough to rebuild whenever a block's level has drifted (launch_block_apx).
__global__ __launch_bounds__(256) void k_build_tab11(double W, double lo, double delta, double eta, double sprime, int NB, float cbin, int2 *__restrict__ tab) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int id = gid >> 5, role = gid & 31;
    const int smp = role & 15;
    const bool wantL = role >= 16;
    const bool live = id < NB * NB && smp < 9;
    const int idc = id < NB * NB ? id : NB * NB - 1;
    const int jb = idc / NB, ia = idc % NB;   // [bin of the to side][bin of the from side]
    const double den = W + 2.0;
    double Lmax = -1e30, Hmin = 1e30;
    bool reach = false;   // this lane's sample point can reach the level at all on the H side (x above independence)
    if (live) {
        const int sa = smp / 3, sb = smp % 3;
        // bin k covers sqrt(p) * cbin in [k, k + 1); the last bin is open-ended: sample it up to the total weight
        const double ua = (ia + 0.5 * sa) / cbin, ub = (jb + 0.5 * sb) / cbin;
        double pa = ua * ua, pb = ub * ub;
        if (ia == NB - 1 && sa > 0) pa = sa == 1 ? 0.5 * (pa + W) : W;
        if (jb == NB - 1 && sb > 0) pb = sb == 1 ? 0.5 * (pb + W) : W;
        pa = pa > W ? W : pa;
        pb = pb > W ? W : pb;
        const double E = (pa + 1.0) * (pb + 1.0) / den;
        double xlo = 0.5, xhi = (pa < pb ? pa : pb) + 0.5;
        const double xmin_feas = (pa + 1.0) + (pb + 1.0) - den + 0.5;
        if (xmin_feas > xlo) xlo = xmin_feas;
        if (!wantL) {
            if (E < xhi && mi11_exact(xhi, pa, pb, W, den) >= lo) {
                double a = E > xlo ? E : xlo, b = xhi;
                for (int it = 0; it < 20; ++it) {
                    const double m = 0.5 * (a + b);
                    if (mi11_exact(m, pa, pb, W, den) < lo) a = m; else b = m;
                }
                Hmin = a;   // MI(a) < lo: everything below a (and above E) is safe
                reach = true;
            } else {
                // r04 (found by the adversarial alignment: a rare state that co-occurs PERFECTLY with a common one — x = xhi, the largest
                // feasible value — was dismissed).  This sample point cannot reach the level even at its largest feasible x = xhi, but a
                // point with larger marginals in the same bin may: H ascends with both marginals, so over the bin it is smallest where the
                // level just becomes reachable, and there H = that point's own xhi >= this sample's xhi.  An unreachable sample therefore
                // bounds H from below by ITS xhi (r03 let it contribute "no bound", and the minimum over the samples then skipped the
                // stretch between the last unreachable and the first reachable sample).  A bin none of whose samples reaches the level is
                // unreachable as a whole (the perfect-association MI is largest in a corner of the bin): unconditional, below.
                Hmin = xhi;
            }
        } else if (E > xlo && mi11_exact(xlo, pa, pb, W, den) >= lo) {
            double a = xlo, b = E < xhi ? E : xhi;
            for (int it = 0; it < 20; ++it) {
                const double m = 0.5 * (a + b);
                if (mi11_exact(m, pa, pb, W, den) < lo) b = m; else a = m;
            }
            Lmax = b;
        }
    }
    int any_reach = reach ? 1 : 0;
#pragma unroll
    for (int off = 1; off <= 16; off <<= 1) {
        const double l2 = __shfl_xor(Lmax, off), h2 = __shfl_xor(Hmin, off);
        Lmax = l2 > Lmax ? l2 : Lmax;
        Hmin = h2 < Hmin ? h2 : Hmin;
        any_reach |= __shfl_xor(any_reach, off);
    }
    if (!any_reach) Hmin = 1e30;
    if (role != 0 || id >= NB * NB) return;
    // n' is at most eta / sprime units below the sum of the approximate weights, which is within delta of the exact sum
    int2 e;
    const double hq = (Hmin * (1.0 - delta) - eta - 0.5) / sprime, lq = (Lmax * (1.0 + delta) + eta - 0.5) / sprime;
    e.y = Hmin > 1e29 ? 2147483647 : (hq < 0.0 ? 0 : (hq > 2147483000.0 ? 2147483647 : (int)floor(hq) - 1));
    e.x = Lmax < -1e29 ? -1 : (lq < -1.0 ? -1 : (lq > 2147483000.0 ? 2147483646 : (int)ceil(lq) + 1));
    tab[id] = e;
}

// biallelic x biallelic columns of the approximate screen through the threshold table
template <int U, int RM>
__device__ __forceinline__ unsigned int screen_cols_tab(const EpiArgs &A, const RowSide &R, int binA, const ColMeta *cmu, int a_loc, bool a_ok, float lo,
                                                        int q0, int tile) {
    const bool test_sr = A.E.any_sr != 0 || A.sr_excl != 0, keep_sr = A.E.keep_sr != 0 && !A.sr_excl;
    int n[U];
    int2 th[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        n[u] = (int)g_entry(A, R, cmu[u]).at(0, 0);
        th[u] = A.tab11[cmu[u].pad2 * A.tab_nb + binA];
    }
    unsigned int bits = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const ColMeta &M = cmu[u];
        const int b_loc = M.bl;
        const bool act = a_ok && (A.E.lower_only ? a_loc > b_loc : a_loc != b_loc);
        const bool is_sr = test_sr && col_is_sr(M.ci, a_loc);
        if (test_sr && __ballot(act && is_sr && keep_sr) != 0ull) {   // a unit with a short-range pair is evaluated whole
            bits |= 1u << u;
            continue;
        }
        const bool maybe = act && !is_sr && !(n[u] > th[u].x && n[u] < th[u].y);
        if (__ballot(maybe) == 0ull) continue;
        // some lane is beyond its thresholds: the full bound for this column
        FullCells<1, 1> C;
        const int64_t v = n[u];
        C.n[0][0] = v;
        C.n[1][0] = M.pb[0] - v;
        C.n[0][1] = R.pa[0] - v;
        C.n[1][1] = R.pa[1] - C.n[1][0];
        const float ms = full_cells_screen<1, 1, true>(A, R, M, screen_rxy(A, R, M, a_loc, b_loc, RM), C);
        const bool need_lr = maybe && ms >= lo;
        const unsigned long long m = __ballot(need_lr);
        if (m == 0ull) continue;
        if (A.pl_pairs) append_pairs(A, 0, m, need_lr, (uint32_t)(tile * 64 + (threadIdx.x & 63)), (uint32_t)(q0 + u), R.sa, M.sb);
        else bits |= 1u << u;   // verify mode: whole units (the dismissed ones are evaluated too and must not produce anything)
    }
    return bits;
}

// U columns at once for biallelic x biallelic units (4 cells each, the bulk of the work); wider tables go two (or one)
// at a time, which keeps the kernel near 64 VGPRs
template <int NA, int U, int RM, bool APX>
__device__ __forceinline__ unsigned int screen_cols_nb(int nb, const EpiArgs &A, const RowSide &R, const ColMeta *cmu, int a_loc,
                                                       bool a_ok, float lo, int q0, int tile) {
    static_assert(U <= 4, "screen_cols_apx keeps U columns' entries in registers");
    if (APX) return nb == 1 ? screen_cols_apx<NA, 1, U, RM>(A, R, cmu, a_loc, a_ok, lo, q0, tile) : screen_cols_apx<NA, 2, U, RM>(A, R, cmu, a_loc, a_ok, lo, q0, tile);
    if (NA == 1 && nb == 1) return screen_cols<NA, 1, U, RM, APX>(A, R, cmu, a_loc, a_ok, lo, q0, tile);   // (the table path is taken by the caller)
    constexpr int V = LDW_SCREEN_V < U ? LDW_SCREEN_V : U;   // columns in flight for the multi-cell tables (2: 387 -> 372 us per C4 block in r02)
    unsigned int bits = 0;
    for (int u = 0; u < U; u += V) {
        const unsigned int b = nb == 1 ? screen_cols<NA, 1, V, RM, APX>(A, R, cmu + u, a_loc, a_ok, lo, q0 + u, tile)
                                       : screen_cols<NA, 2, V, RM, APX>(A, R, cmu + u, a_loc, a_ok, lo, q0 + u, tile);
        bits |= b << u;
    }
    return bits;
}

// Append the wanted units of a wave (bit k = column slot q_base + k of from-tile `tile`) to the flat list and, in the
// mixed-precision path, to the list of their (tile, row-slot class), which is what the gathered low-limb GEMM walks; the
// index k there tells k_mi_units where the low limbs of the unit's joint sums are.  In verify mode every unit of `all` is
// listed, the unwanted ones marked.
// There are two flat lists: units whose SNPs have 1 or 2 fully flagged slots on both sides (bit set in `fastmask`) go to
// list 0, which the straight-line kernel k_mi_units<true> walks; the others to list 1 (k_mi_units<false>, predicated code).
// units / n_units point at list 0 / its counter; list 1 follows at units + list_stride / n_units + 1.
__device__ __forceinline__ void list_wave_units(const EpiArgs &A, const ColMeta *cm, int c_first, int q_base, int n_it, unsigned int all,
                                                unsigned int wanted, unsigned int fastmask, uint64_t *__restrict__ units,
                                                unsigned int *__restrict__ n_units, int64_t list_stride, int tile = -1) {
    if (tile < 0) tile = (int)blockIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned int listed = A.E.scr_mode == 2 ? all : wanted;
    if (listed == 0) return;
    const bool mine = lane < 32 && ((listed >> lane) & 1u);
    uint64_t kfield = 0;
    if (A.lo.on) {
        const int my_lc = lane < n_it ? lo_class((int)(cm[c_first + (lane < 32 ? lane : 0)].mb & 7)) : 0;
#pragma unroll
        for (int lc = 0; lc < 3; ++lc) {
            const unsigned int m = (unsigned int)__ballot(mine && my_lc == lc);
            if (m == 0) continue;
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(&A.lo.cnt[tile * 3 + lc], (unsigned int)__popc(m));
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (mine && my_lc == lc) {
                const unsigned int k = base + __popc(m & ((1u << lane) - 1u));
                // bit 31 (verify mode only): a unit the screen dismissed
                A.lo.tl[(int64_t)tile * A.nt + A.lo.uoff[lc] + k] = (uint32_t)(q_base + lane) | (((wanted >> lane) & 1u) ? 0u : 0x80000000u);
                kfield = ((uint64_t)k << 31) | ((uint64_t)lc << 51);
            }
        }
    }
    if (!units) return;   // approximate path: only the per-(tile, class) lists are read (k_units_pop builds the final flat lists)
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const unsigned int m = listed & (which == 0 ? fastmask : ~fastmask);
        if (m == 0) continue;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(n_units + which, (unsigned int)__popc(m));
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < 32 && ((m >> lane) & 1u)) {
            const uint64_t u = (uint64_t)((uint32_t)tile * (uint32_t)A.nt + (uint32_t)(q_base + lane)) | kfield;
            units[which * list_stride + base + __popc(m & ((1u << lane) - 1u))] = ((wanted >> lane) & 1u) ? u : (u | UNIT_DISMISSED);
        }
    }
}

// RM: how RXY is read (screen_rxy) — a template parameter so that the common square-block code carries neither the
// division nor the table look-ups of the ragged case
// One workgroup's share of the screen: from-tile `tile` (64 SNPs, perm_f order) x column group `cgy` (EPI_COLS column slots, perm_t order).
template <int RM, bool APX>
__device__ __forceinline__ void screen_wg(const EpiArgs &A, const int32_t *__restrict__ perm_f, const int32_t *__restrict__ perm_t,
                                          uint64_t *__restrict__ units, unsigned int *__restrict__ n_units, int64_t list_stride, ColMeta *cm, int tile,
                                          int cgy) {
    const bool square = A.nf == A.nt;
    const bool mixed = A.lo.on != 0 || APX;   // cells derived from the marginals of the weights the block-wide sums were taken with
    if (APX && A.clean && A.E.scr_mode != 2 && tile < A.clean_stride && tile < A.gen_t0) {
        // all four 32-column regions of this workgroup found clean by the GEMM's epilogue: nothing to stage, nothing to list
        bool all_clean = true;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int qb = cgy * EPI_COLS + w * (EPI_COLS / 4);
            const bool whole = qb + EPI_COLS / 4 <= A.nt && qb + EPI_COLS / 4 <= A.gen_q0;
            all_clean = all_clean && whole && A.clean[(int64_t)(qb / 32) * A.clean_stride + tile] != 0;
        }
        if (all_clean) return;
    }
    const int wave = threadIdx.x >> 6;
    const int c_first = wave * (EPI_COLS / 4);
    const int q_base = cgy * EPI_COLS + c_first;
    int n_it = A.nt - q_base;
    n_it = n_it > EPI_COLS / 4 ? EPI_COLS / 4 : n_it;
    // pruning by kind (k_snp_sup): columns whose pairs with EVERY SNP of the tile are dead — the tile's SNPs all dead versus the
    // column's kind, or all of one kind the column's SNP is dead against — are dismissed unread (the approximate GEMM has not
    // even computed most of their sums: apx_tile_prunable).  Not in verify mode, which has to see what is dismissed.  From the
    // flags alone, so that a workgroup with nothing else to do leaves before it stages anything.
    unsigned int dead_cols = 0, tri_cols = 0;
    // Diagonal blocks (lower_only: a pair exists once, with a_loc > b_loc): both epilogue orders keep the list order within a class,
    // so about half of the (tile, column) combinations hold no pair at all — columns whose SNP comes after every SNP of the tile.
    // They used to run the whole bound and mask the result.
    if (A.E.lower_only) {
        int amax = perm_f[tile * 64 + (threadIdx.x & 63)];   // (-1: padding slot)
        const int pb = (int)(threadIdx.x & 63) < n_it ? perm_t[q_base + (threadIdx.x & 63)] : 0x7FFFFFFF;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_xor(amax, off);
            amax = o > amax ? o : amax;
        }
        dead_cols = (unsigned int)__ballot((int)(threadIdx.x & 63) < n_it && pb >= amax);
        tri_cols = dead_cols;
        if (__syncthreads_and((n_it <= 0 || (n_it == EPI_COLS / 4 && dead_cols == 0xFFFFFFFFu)) ? 1 : 0)) return;
    }
    unsigned int prune_cols = 0;   // (in a block with short-range pairs these columns are dropped only where the unit holds none: below)
    if (APX && A.sflag_f && A.E.do_lr) {
        const unsigned fa = A.sflag_f[tile * 64 + (threadIdx.x & 63)];
        const bool real = (fa & PF_PAD) == 0u;   // (a padding slot of the tile: no SNP)
        const unsigned ka = fa & PF_KIND;
        const bool all_k2 = __ballot(real && ka != 2u) == 0ull, all_k3 = __ballot(real && ka != 3u) == 0ull;
        if ((all_k2 || all_k3) && __ballot(real) != 0ull) {
            const bool all_dead2 = __ballot(real && !(fa & PF_DEAD2)) == 0ull, all_dead3 = __ballot(real && !(fa & PF_DEAD3)) == 0ull;
            const unsigned fb = (int)(threadIdx.x & 63) < n_it ? (unsigned)A.sflag_t[q_base + (threadIdx.x & 63)] : 0u;
            const unsigned kb = fb & PF_KIND;
            const bool dead = kb >= 2u && ((kb == 2u ? all_dead2 : all_dead3) || (fb & (all_k2 ? PF_DEAD2 : PF_DEAD3)) != 0u);
            prune_cols = (unsigned int)__ballot(dead);
        }
        if (!A.E.any_sr && A.E.scr_mode != 2) {   // (verify mode lists the pruned units as dismissed: their pairs are checked in fp64 like any other)
            dead_cols |= prune_cols;
            prune_cols = 0;
            const bool wave_done = n_it <= 0 || (n_it == EPI_COLS / 4 && dead_cols == 0xFFFFFFFFu);
            if (__syncthreads_and(wave_done ? 1 : 0)) return;
        }
    }
    stage_cols(A, perm_t, square, cm, mixed, cgy);
    __syncthreads();
    RowSide R;
    int a_loc, na0;
    const bool a_ok = load_row_side(A, perm_f, square, tile, R, a_loc, mixed);
    const bool wave_full = wave_is_full(R, a_ok, na0);
    if (n_it <= 0) return;
    const float lo = (float)A.E.spec_lo - A.E.scr_eps;
    // threshold-table path: every SNP of the tile biallelic with r = 2, long-range pass
    int binA = 0;
    bool use_tab = false;
    if (APX && A.tab11 && A.rowpack && A.E.do_lr && wave_full && na0 == 1) {
        binA = A.rowpack[tile * 64 + (threadIdx.x & 63)].pad;
        use_tab = __ballot(a_ok && R.ra != 2.0) == 0ull;
    }
    // wanted: units that need the fp64 kernel; handled: units this kernel could judge (the others are wanted by default)
    unsigned int wanted = 0, handled = 0;
    constexpr int U = 4;
    // from-tiles >= gen_t0 and column slots >= gen_q0 (SNPs with >= 3 minor states, or none) belong to k_mi_screen_generic
    if (tile >= A.gen_t0 || q_base >= A.gen_q0) return;
    if (q_base + n_it > A.gen_q0) n_it = A.gen_q0 - q_base;
    // a region the GEMM's epilogue found clean (every one of its 64 x 32 pairs inside the table thresholds — the very test the
    // table path below would make) is dismissed without a single load
    const bool region_clean = APX && A.clean && n_it == EPI_COLS / 4 && tile < A.clean_stride &&
                              A.clean[(int64_t)(q_base / 32) * A.clean_stride + tile] != 0;
    if (region_clean) {
        handled = 0xFFFFFFFFu;
    } else if (wave_full) {
        for (int it = 0; it < n_it; it += U) {
            if (((dead_cols >> it) & ((1u << U) - 1u)) == ((1u << U) - 1u) && it + U <= n_it && A.E.scr_mode != 2) {
                handled |= ((1u << U) - 1u) << it;
                continue;
            }
            const ColMeta *cmu = &cm[c_first + it];
            if (((((A.E.scr_mode != 2 ? dead_cols : 0u) | prune_cols) >> it) & ((1u << U) - 1u)) == ((1u << U) - 1u) && it + U <= n_it) {
                // prunable columns of a block with short-range pairs: dropped unless the unit holds one (its tile is in the band then)
                bool has_sr = false;
#pragma unroll
                for (int u = 0; u < U; ++u) has_sr = has_sr || col_is_sr(cmu[u].ci, a_loc);
                if (__ballot(a_ok && has_sr) == 0ull) {
                    handled |= ((1u << U) - 1u) << it;
                    continue;
                }
            }
            // the U columns of a group share one code path if they have the same slot count (the rule away from class borders)
            bool same = it + U <= n_it;
            const uint32_t mb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmu[0].mb);
            if (same) {
#pragma unroll
                for (int u = 1; u < U; ++u) {
                    const uint32_t mbu = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmu[u].mb);
                    same = same && (mbu & 7) == (mb0 & 7) && col_is_fast(mbu);
                }
                same = same && col_is_fast(mb0);
            }
            if (same) {
                unsigned int b;
                bool tab_ok = false;
                if (APX && use_tab && (mb0 & 7) == 1) {   // biallelic x biallelic with r = 2 on both sides: the threshold table
                    tab_ok = true;
#pragma unroll
                    for (int u = 0; u < U; ++u) tab_ok = tab_ok && cmu[u].rb == 2.0;
                }
#ifdef LDW_SCREEN_EXP   // timing ablation only (wrong results; profiles/r04_screen_breakdown.txt): bit 0 skips the table path, bit 1 the multi-cell path
                if ((tab_ok && ((LDW_SCREEN_EXP) & 1)) || (!tab_ok && ((LDW_SCREEN_EXP) & 2))) b = 0;
                else
#endif
                if (tab_ok) b = screen_cols_tab<U, RM>(A, R, binA, cmu, a_loc, a_ok, lo, q_base + it, tile);
                else
                    b = na0 == 1 ? screen_cols_nb<1, U, RM, APX>((int)(mb0 & 7), A, R, cmu, a_loc, a_ok, lo, q_base + it, tile)
                                 : screen_cols_nb<2, U, RM, APX>((int)(mb0 & 7), A, R, cmu, a_loc, a_ok, lo, q_base + it, tile);
                wanted |= b << it;
                handled |= ((1u << U) - 1u) << it;
            } else {
                for (int u = 0; u < U && it + u < n_it; ++u) {
                    const uint32_t mbu = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmu[u].mb);
                    if (!col_is_fast(mbu)) continue;
                    const unsigned int b = na0 == 1 ? screen_cols_nb<1, 1, RM, APX>((int)(mbu & 7), A, R, cmu + u, a_loc, a_ok, lo, q_base + it + u, tile)
                                                    : screen_cols_nb<2, 1, RM, APX>((int)(mbu & 7), A, R, cmu + u, a_loc, a_ok, lo, q_base + it + u, tile);
                    wanted |= b << (it + u);
                    handled |= 1u << (it + u);
                }
            }
        }
    }
    const unsigned int all = n_it >= 32 ? 0xFFFFFFFFu : ((1u << n_it) - 1u);
    if (A.E.scr_mode != 2) handled |= tri_cols;   // (columns without a pair need nobody's attention, whatever the tile looks like; verify mode lists
                                                  // every unit and reads `handled` as "straight-line code applies": left alone there)
    wanted = (wanted | ~handled) & all;
    list_wave_units(A, cm, c_first, q_base, n_it, all, wanted, handled, units, n_units, list_stride, tile);
}


// r04: the entries the GEMM's epilogue found outside their table thresholds (ApxGemmArgs::maybe): one lane per entry evaluates the full
// bound of that ONE pair — what screen_cols_tab does for the columns of a region in which some lane fails the table test — and lists the
// pair if it may reach the block's level.  Both SNPs are biallelic with r = 2 and fully flagged (their rows carry table bins), so a row-list
// position is the epilogue slot: trow = column slot, fcol = 64 * from-tile + lane.  Not used for diagonal blocks or in verify mode.
template <int RM>
__global__ __launch_bounds__(256) void k_screen_maybe(EpiArgs A, const ApxMaybe *__restrict__ maybe, const unsigned int *__restrict__ maybe_n,
                                                      unsigned int maybe_cap) {
    const unsigned int n_all = *maybe_n;
    if (n_all > maybe_cap) {   // the list overflowed: the block takes the pair lists' overflow path — through a word of its own behind the 40 list
        // counters, which k_pick_bucket reads with them (bumping a real counter would make its consumers read entries that were never written)
        if (blockIdx.x == 0 && threadIdx.x == 0) A.pl_n[PAIR_PATHS * PAIR_SHARDS] = 0xFFFFFFFFu;
        return;
    }
    const float lo = (float)A.E.spec_lo - A.E.scr_eps;
    const bool test_sr = A.E.any_sr != 0 || A.sr_excl != 0;
    for (unsigned int k = blockIdx.x * 256u + threadIdx.x; k < n_all; k += gridDim.x * 256u) {
        const ApxMaybe e = maybe[k];
        // (the _hi packs: integer marginals of the APPROXIMATE weights, the ones n' was summed with — what the screen's cells are derived from)
        const ColMeta &M = A.colpack_hi[e.trow];
        const RowPack &P = A.rowpack_hi[e.fcol];
        const RowSide &R = P.R;
        const int a_loc = P.a_loc, b_loc = M.bl;
        if (a_loc < 0 || a_loc == b_loc) continue;                       // (padding slot; quirk Q3: an off-diagonal block drops its own diagonal)
        if (test_sr && col_is_sr(M.ci, a_loc)) continue;                 // (cannot happen: tiles with a short-range pair are never table-tested)
        FullCells32<1, 1> C;
        const int v = e.n;
        C.n[0][0] = v;
        C.n[1][0] = (int)M.pb[0] - v;
        C.n[0][1] = (int)R.pa[0] - v;
        C.n[1][1] = (int)R.pa[1] - C.n[1][0];
        const float ms = full_cells_screen<1, 1, true>(A, R, M, screen_rxy(A, R, M, a_loc, b_loc, RM), C);
        if (!(ms >= lo)) continue;
        const int sub = (int)((e.fcol >> 6) & (PAIR_SHARDS - 1));        // path 0 (1 x 1 indicator rows), sharded by from-tile like append_pairs
        const unsigned int pos = atomicAdd(A.pl_n + sub, 1u);
        if (pos < A.pl_cap) {
            const int32_t r0a = A.row0[R.sa], r0b = A.row0[M.sb];
            PairEnt pe;
            pe.t = e.fcol;
            pe.q = e.trow;
            pe.ra = (uint32_t)r0a | ((uint32_t)(A.row0[R.sa + 1] - r0a) << 29);
            pe.rb = (uint32_t)r0b | ((uint32_t)(A.row0[M.sb + 1] - r0b) << 29);
            A.pl_pairs[(int64_t)sub * A.pl_cap + pos] = pe;
        }
    }
}

// the whole grid: one workgroup per (from-tile, column group)
template <int RM, bool APX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LDW_SCREEN_WAVES, 8))) void k_mi_screen(EpiArgs A, const int32_t *__restrict__ perm_f,
                                                                                            const int32_t *__restrict__ perm_t,
                                                                                            uint64_t *__restrict__ units,
                                                                                            unsigned int *__restrict__ n_units,
                                                                                            int64_t list_stride) {
    __shared__ ColMeta cm[EPI_COLS];
    screen_wg<RM, APX>(A, perm_f, perm_t, units, n_units, list_stride, cm, (int)blockIdx.x, (int)blockIdx.y);
}

#ifdef LDW_EXPERIMENTS
// r04: LIST-DRIVEN.  Three quarters of the (tile, column group) combinations of a long-range block have nothing to screen — all four of their
// regions flagged clean by the GEMM's epilogue or pruned with their tile, or every column dead against the tile's kind — and each of them still
// cost a workgroup dispatch (~4 ns: the floor of the full-grid kernel was 47 us per 10k x 10k block, 0.33 ms per span of seven, with every
// column dropped).  k_screen_live lists the combinations that are left; a fixed grid of workgroups strides over the list.
template <int RM, bool APX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_mi_screen_list(EpiArgs A, const int32_t *__restrict__ perm_f,
                                                                                                 const int32_t *__restrict__ perm_t,
                                                                                                 uint64_t *__restrict__ units,
                                                                                                 unsigned int *__restrict__ n_units, int64_t list_stride,
                                                                                                 const uint32_t *__restrict__ live,
                                                                                                 const unsigned int *__restrict__ n_live) {
    __shared__ ColMeta cm[EPI_COLS];
    const unsigned int n = *n_live;
    for (unsigned int i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t e = live[i];
        screen_wg<RM, APX>(A, perm_f, perm_t, units, n_units, list_stride, cm, (int)(e & 0xFFFFu), (int)(e >> 16));
        __syncthreads();   // (cm is staged again for the next entry)
    }
}

// Which (from-tile, column group) combinations does k_mi_screen have something to do for?  Its own workgroup-wide exits, from flags alone:
// outside its domain (tiles / columns of the generic screen); all four 32-column regions clean; every column dead against the tile's kind
// (long-range-only blocks); on a diagonal block every column's SNP behind every SNP of the tile.  One workgroup per column group, its
// threads stride over the tiles; entries = tile | column group << 16, appended per wave.
struct TileState {
    int amax;            // largest list index among the tile's SNPs (-1: none)
    unsigned kind;       // bit 0: all real SNPs of kind 2, bit 1: all of kind 3, bit 2: all dead versus kind 2, bit 3: all dead versus kind 3, bit 4: some real SNP
};
__global__ __launch_bounds__(64) void k_screen_tiles(EpiArgs A, const int32_t *__restrict__ perm_f, int ntiles, TileState *__restrict__ ts,
                                                     unsigned int *__restrict__ n_live) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    int a = perm_f[tile * 64 + lane];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_xor(a, off);
        a = o > a ? o : a;
    }
    unsigned kind = 0;
    if (A.sflag_f) {
        const unsigned fa = A.sflag_f[tile * 64 + lane];
        const bool real = (fa & PF_PAD) == 0u;
        const unsigned ka = fa & PF_KIND;
        kind = (__ballot(real && ka != 2u) == 0ull ? 1u : 0u) | (__ballot(real && ka != 3u) == 0ull ? 2u : 0u) | (__ballot(real && !(fa & PF_DEAD2)) == 0ull ? 4u : 0u) |
               (__ballot(real && !(fa & PF_DEAD3)) == 0ull ? 8u : 0u) | (__ballot(real) != 0ull ? 16u : 0u);
    }
    if (lane == 0) ts[tile] = TileState{a, kind};
    if (n_live && tile == 0 && lane == 0) *n_live = 0u;   // (k_screen_live, next on the stream, counts from zero)
}
template <bool APX>
__global__ __launch_bounds__(256) void k_screen_live(EpiArgs A, const int32_t *__restrict__ perm_t, const TileState *__restrict__ ts, int ntiles,
                                                     uint32_t *__restrict__ live, unsigned int *__restrict__ n_live) {
    __shared__ unsigned int s_dead[8];   // [tile state: (k2 | k3) x all_dead2 x all_dead3]: every column of the group dead for such a tile
    __shared__ int s_minb;
    const int cgy = blockIdx.x, t = threadIdx.x;
    const int q0 = cgy * EPI_COLS;
    if (t < 8) s_dead[t] = 1u;
    if (t == 0) s_minb = 0x7FFFFFFF;
    __syncthreads();
    const bool prune_on = APX && A.sflag_f && A.E.do_lr && !A.E.any_sr && A.E.scr_mode != 2;
    if (t < EPI_COLS) {
        const int q = q0 + t;
        if (q < A.nt) {
            if (A.E.lower_only) atomicMin(&s_minb, perm_t[q]);
            if (prune_on) {
                const unsigned fb = A.sflag_t[q], kb = fb & PF_KIND;
                for (int s = 0; s < 8; ++s) {
                    const bool all_k2 = (s & 4) == 0, ad2 = (s & 1) != 0, ad3 = (s & 2) != 0;   // s: bit 2 = tile of kind 3 (else kind 2)
                    const bool dead = kb >= 2u && ((kb == 2u ? ad2 : ad3) || (fb & (all_k2 ? PF_DEAD2 : PF_DEAD3)) != 0u);
                    if (!dead) s_dead[s] = 0u;   // (benign race: every writer stores 0)
                }
            }
        }
    }
    __syncthreads();
    const bool whole = q0 + EPI_COLS <= A.nt;
    for (int tile0 = 0; tile0 < ntiles; tile0 += 256) {
        const int tile = tile0 + t;
        bool keep = tile < ntiles && tile < A.gen_t0 && q0 < A.gen_q0;
        if (keep && APX && A.clean && A.E.scr_mode != 2 && tile < A.clean_stride && whole && q0 + EPI_COLS <= A.gen_q0) {
            bool all_clean = true;
#pragma unroll
            for (int w = 0; w < 4; ++w) all_clean = all_clean && A.clean[(int64_t)((q0 + w * (EPI_COLS / 4)) / 32) * A.clean_stride + tile] != 0;
            keep = !all_clean;
        }
        if (keep) {
            const TileState T = ts[tile];
            if (A.E.lower_only && whole && A.E.scr_mode != 2 && s_minb >= T.amax) keep = false;   // no pair: every column's SNP behind every SNP of the tile
            if (keep && prune_on && whole && (T.kind & 16u) && (T.kind & 3u)) {
                const int s = ((T.kind & 1u) ? 0 : 4) | ((T.kind >> 2) & 3u);
                if (s_dead[s]) keep = false;
            }
        }
        const unsigned long long mk = __ballot(keep);
        if (mk != 0ull) {
            unsigned int base = 0;
            const int lane = t & 63;
            if (lane == __builtin_ctzll(mk)) base = atomicAdd(n_live, (unsigned int)__popcll(mk));
            base = (unsigned int)__shfl((int)base, __builtin_ctzll(mk));
            if (keep) live[base + (unsigned int)__popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t)tile | ((uint32_t)cgy << 16);
        }
    }
}

#endif   // LDW_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// k_mi_screen_generic: the fp32 screen for the units k_mi_screen leaves out — a from-tile or a column whose SNPs have >= 3
// minor states (or none).  Those are 1-2 % of the units, but unscreened they were a quarter of the fp64 kernel's list,
// its slowest entries (predicated code), and a third of the gathered low-limb GEMM.  Launched over the generic
// from-tiles x all columns and over the other tiles x the generic columns.
// ------------------------------------------------------------------------------------------------
struct GenRegions {      // k_mi_screen_generic's two rectangles: a = from-tiles [tile0_a, tile0_a + nt_a) x column groups [0, ncg_a);
    int tile0_a, nt_a, ncg_a;   // b = from-tiles [0, nt_b) x ncg_b column groups from column slot q0_b
    int nt_b, q0_b, ncg_b;
};
constexpr int GEN_COLS = 16;   // column slots per workgroup of k_mi_screen_generic: 4 per wave — the kernel is a chain of
                               // dependent loads per column with nothing else to hide them, so the chains are kept short
template <bool APX>
__global__ __launch_bounds__(256) void k_mi_screen_generic(EpiArgs A, const int32_t *__restrict__ perm_f, const int32_t *__restrict__ perm_t,
                                                           uint64_t *__restrict__ units, unsigned int *__restrict__ n_units,
                                                           int64_t list_stride, GenRegions Rg) {
    __shared__ ColMeta cm[GEN_COLS];
    const bool square = A.nf == A.nt;
    const bool mixed = A.lo.on != 0 || APX;
    // both regions in ONE launch (1-D grid): the generic from-tiles x all columns, then the other tiles x the generic columns
    int bid = (int)blockIdx.x, tile, qb;
    if (bid < Rg.nt_a * Rg.ncg_a) {
        tile = Rg.tile0_a + bid % Rg.nt_a;
        qb = (bid / Rg.nt_a) * GEN_COLS;
    } else {
        bid -= Rg.nt_a * Rg.ncg_a;
        tile = bid % Rg.nt_b;
        qb = Rg.q0_b + (bid / Rg.nt_b) * GEN_COLS;
    }
    if (threadIdx.x < GEN_COLS) {
        const int q = qb + (int)threadIdx.x;
        if (q < A.nt) {
            ColMeta m;
            load_col(A, perm_t, square, q, m, mixed);
            cm[threadIdx.x] = m;
        }
    }
    __syncthreads();
    RowSide R;
    int a_loc;
    const bool a_ok = load_row_side(A, perm_f, square, tile, R, a_loc, mixed);
    const int wave = threadIdx.x >> 6;
    const int c_first = wave * (GEN_COLS / 4);
    const int q_base = qb + c_first;
    int n_it = A.nt - q_base;
    n_it = n_it > GEN_COLS / 4 ? GEN_COLS / 4 : n_it;
    if (n_it <= 0) return;
    const int rxy_mode = A.quirk == LDW_QUIRK_REFERENCE ? (A.span ? 3 : (square ? 1 : 2)) : 0;
    const float lo = (float)A.E.spec_lo - A.E.scr_eps;
    const bool test_sr = A.E.any_sr != 0 || A.sr_excl != 0, keep_sr = A.E.keep_sr != 0 && !A.sr_excl, do_lr = A.E.do_lr != 0;
    unsigned int wanted = 0, mine = 0;   // mine: the units of this wave that belong to this kernel
    for (int it = 0; it < n_it; ++it) {
        // the same split as k_mi_screen's: its domain is tile < gen_t0 and q < gen_q0, where it takes the fast units and
        // lists the others unscreened (unflagged slots); everything outside that domain is screened here
        if (tile < A.gen_t0 && q_base + it < A.gen_q0) continue;
        const ColMeta &M = cm[c_first + it];
        mine |= 1u << it;
        const int b_loc = M.bl;
        const bool act = a_ok && (A.E.lower_only ? a_loc > b_loc : a_loc != b_loc);
        const double rxy = screen_rxy(A, R, M, a_loc, b_loc, rxy_mode);
        const float ms = do_lr ? pair_screen_generic<APX>(A, R, M, rxy, g_entry(A, R, M)) : 0.0f;
        const bool is_sr = test_sr && col_is_sr(M.ci, a_loc);
        if (APX && A.pl_pairs) {
            const bool need_lr = act && !is_sr && do_lr && ms >= lo;
            if (__ballot(act && is_sr && keep_sr) != 0ull) {
                wanted |= 1u << it;
            } else {
                const unsigned long long m = __ballot(need_lr);
                if (m != 0ull) append_pairs(A, 4, m, need_lr, (uint32_t)(tile * 64 + (threadIdx.x & 63)), (uint32_t)(q_base + it), R.sa, M.sb);
            }
        } else {
            const bool need = act && (is_sr ? keep_sr : (do_lr && ms >= lo));
            if (__ballot(need) != 0ull) wanted |= 1u << it;
        }
    }
    list_wave_units(A, cm, c_first, q_base, n_it, mine, wanted & mine, 0u, units, n_units, list_stride, tile);
}

// would this pair leave a trace (short-range row or long-range candidate)?  Verify mode of the screen only.
__device__ __forceinline__ bool would_emit(const EmitArgs &E, const ColInfo &c, int a_loc, int b_loc, double mi) {
    if (pair_seg(a_loc, b_loc, E.lower_only) < 0) return false;
    if (E.any_sr && col_is_sr(c, a_loc)) return E.keep_sr != 0;
    return E.do_lr && mi >= E.spec_lo && mi_bucket(mi) >= E.spec_B;
}

// MI of one pair by the variant that fits the slot counts (wave-uniform choice)
__device__ __forceinline__ double unit_pair_mi(const EpiArgs &A, const RowSide &R, const ColMeta &M, int a_loc, int b_loc, bool square,
                                               bool fast, int na0, int na_max, int nb, const GAcc &Ga) {
    if (fast) {
        if (na0 == 1) return nb == 1 ? pair_mi_full<1, 1>(A, R, M, a_loc, b_loc, square, Ga) : pair_mi_full<1, 2>(A, R, M, a_loc, b_loc, square, Ga);
        return nb == 1 ? pair_mi_full<2, 1>(A, R, M, a_loc, b_loc, square, Ga) : pair_mi_full<2, 2>(A, R, M, a_loc, b_loc, square, Ga);
    }
    if (na_max == 1) {
        if (nb <= 1) return pair_mi<1, 1>(A, R, M, a_loc, b_loc, square, Ga);
        if (nb == 2) return pair_mi<1, 2>(A, R, M, a_loc, b_loc, square, Ga);
        return pair_mi<1, 4>(A, R, M, a_loc, b_loc, square, Ga);
    }
    if (na_max == 2) {
        if (nb <= 1) return pair_mi<2, 1>(A, R, M, a_loc, b_loc, square, Ga);
        if (nb == 2) return pair_mi<2, 2>(A, R, M, a_loc, b_loc, square, Ga);
        return pair_mi<2, 4>(A, R, M, a_loc, b_loc, square, Ga);
    }
    if (nb <= 1) return pair_mi<4, 1>(A, R, M, a_loc, b_loc, square, Ga);
    if (nb == 2) return pair_mi<4, 2>(A, R, M, a_loc, b_loc, square, Ga);
    return pair_mi<4, 4>(A, R, M, a_loc, b_loc, square, Ga);
}

// ------------------------------------------------------------------------------------------------
// k_mi_epilogue: fp64 MI of every pair of the block, emission of the pairs (dense MI block, short-range rows, LDS
// histogram / speculative candidates).  The path of blocks without a bucket guess and of ldw_mi_block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mi_epilogue(EpiArgs A, const int32_t *__restrict__ perm_f, const int32_t *__restrict__ perm_t,
                                                     unsigned long long *__restrict__ ghist) {
    __shared__ unsigned int sh_hist[NBINS];
    __shared__ ColMeta cm[EPI_COLS];
    const bool use_hist = A.E.cols && A.E.do_lr;
    if (use_hist)
        for (int i = threadIdx.x; i < NBINS; i += 256) sh_hist[i] = 0;
    const bool square = A.nf == A.nt;
    stage_cols(A, perm_t, square, cm);
    __syncthreads();

    // lanes walk the from-side SNPs in an order that groups equal slot counts, so that a wave runs the same cells
    const int wave = threadIdx.x >> 6;
    RowSide R;
    int a_loc, na0;
    const bool a_ok = load_row_side(A, perm_f, square, blockIdx.x, R, a_loc);
    const int na_max = (__ballot(R.na > 2) != 0ull) ? 4 : ((__ballot(R.na > 1) != 0ull) ? 2 : 1);
    const bool wave_full = wave_is_full(R, a_ok, na0);

    const int c_first = wave * (EPI_COLS / 4);
    const int q_base = blockIdx.y * EPI_COLS + c_first;
    int n_it = A.nt - q_base;
    n_it = n_it > EPI_COLS / 4 ? EPI_COLS / 4 : n_it;
    for (int it = 0; it < n_it; ++it) {
        const ColMeta &M = cm[c_first + it];
        const uint32_t mbu = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.mb);
        const int b_loc = M.bl;
        if (!a_ok) continue;
        if (A.E.lower_only && a_loc <= b_loc) continue;
        const double mi = unit_pair_mi(A, R, M, a_loc, b_loc, square, wave_full && col_is_fast(mbu), na0, na_max, (int)(mbu & 7), g_entry(A, R, M));
        emit_pair(A.E, M.ci, a_loc, b_loc, R.sa, M.sb, mi, sh_hist);
    }
    if (use_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < NBINS; i += 256)
            if (sh_hist[i]) atomicAdd(&ghist[i], (unsigned long long)sh_hist[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// k_mi_units: the fp64 evaluation of the units listed by k_mi_screen (speculative selection mode): one wave per unit,
// grid-stride over the list.  A few per cent of the block's units are listed, so what counts is latency (the SNP
// constants of both sides are fetched per unit) — hidden by the number of waves in flight — not throughput.
// ------------------------------------------------------------------------------------------------
// FAST: list 0, straight-line variants only — half the registers of the predicated code, so that its waves fit on a CU beside
// the two workgroups of the next block's GEMM (which runs on the other stream) instead of waiting for them to drain.
// Where a launch of k_mi_units finds its units: up to 4 flat lists (blockIdx.y)
struct UnitLists {
    const uint64_t *units[4];
    const unsigned int *n[4];
};

template <bool FAST>
__global__ __launch_bounds__(256) void k_mi_units(EpiArgs A, const int32_t *__restrict__ perm_f, const int32_t *__restrict__ perm_t,
                                                  UnitLists UL, unsigned long long *__restrict__ ghist) {
    const bool square = A.nf == A.nt;
    const int y = blockIdx.y;
    const uint64_t *__restrict__ units = UL.units[y];
    const unsigned int n = *UL.n[y];
    const unsigned int stride = gridDim.x * 4u;
    int cur_tile = -1, a_loc = 0, na0 = 0, na_max = 1;
    bool a_ok = false, wave_full = false;
    RowSide R;
    for (unsigned int i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n; i += stride) {
        const uint64_t u = units[i];
        const uint32_t ulo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
        const uint32_t uhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
        const bool dismissed = (uhi & 0x80000000u) != 0;
        const uint32_t v = ulo & 0x7FFFFFFFu;
        const int tile = (int)(v / (uint32_t)A.nt), q = (int)(v - (uint32_t)tile * (uint32_t)A.nt);
        if (tile != cur_tile) {
            a_ok = load_row_side(A, perm_f, square, tile, R, a_loc);
            na_max = (__ballot(R.na > 2) != 0ull) ? 4 : ((__ballot(R.na > 1) != 0ull) ? 2 : 1);
            wave_full = wave_is_full(R, a_ok, na0);
            cur_tile = tile;
        }
        ColMeta M;
        load_col(A, perm_t, square, q, M);
        const uint32_t mbu = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.mb);
        const int b_loc = M.bl;
        if (!a_ok) continue;
        if (A.E.lower_only && a_loc <= b_loc) continue;
        GAcc Ga = g_entry(A, R, M);
        if (A.lo.on) {   // G holds the high limbs only: the low limbs of this unit's joint sums come from the gathered GEMM
            const int k = (int)(((ulo >> 31) | (uhi << 1)) & 0xFFFFFu), lc = (int)((uhi >> 19) & 3u);
            const int cmax = A.lo.cmax_f[tile];
            const int64_t ld = 64 * (int64_t)cmax;
            Ga.l = A.lo.glo + A.lo.tile_base[tile] + ((int64_t)A.lo.rowbase[lc] + ((int64_t)k << lc)) * ld + (int64_t)(threadIdx.x & 63) * cmax;
            Ga.li = 1;
            Ga.lj = ld;
            Ga.shift = A.lo.hi_shift;
        }
        double mi;
        if (FAST) {
            const int nb = (int)(mbu & 7);
            if (na0 == 1) mi = nb == 1 ? pair_mi_full<1, 1>(A, R, M, a_loc, b_loc, square, Ga) : pair_mi_full<1, 2>(A, R, M, a_loc, b_loc, square, Ga);
            else mi = nb == 1 ? pair_mi_full<2, 1>(A, R, M, a_loc, b_loc, square, Ga) : pair_mi_full<2, 2>(A, R, M, a_loc, b_loc, square, Ga);
        } else {
            mi = unit_pair_mi(A, R, M, a_loc, b_loc, square, wave_full && col_is_fast(mbu), na0, na_max, (int)(mbu & 7), Ga);
        }
        if (dismissed) {   // verify mode: a pair of a dismissed unit that would have been emitted was lost by the screen
            if (would_emit(A.E, M.ci, a_loc, b_loc, mi)) atomicAdd(A.E.scr_viol, 1ull);
            continue;
        }
        emit_pair_spec(A.E, M.ci, a_loc, b_loc, R.sa, M.sb, mi, ghist);
    }
}

// one launch instead of four or five hipMemsetAsync per block (each a 4-5 us kernel of its own plus a dispatch gap): zeroes up to
// four small buffers, sizes in 16-byte pieces
struct ZeroArgs {
    uint4 *p[6];
    unsigned int n16[6];
};
__global__ __launch_bounds__(256) void k_zero4(ZeroArgs Z) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 6; ++k)
        for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < Z.n16[k]; i += gridDim.x * 256u) Z.p[k][i] = z;
}

// totals of the approximate path's lists (diagnostics: ldw_ctx_counters2): units listed (they hold a short-range pair),
// long-range candidate pairs listed
__global__ void k_apx_stats(const unsigned int *__restrict__ n_units, const unsigned int *__restrict__ pl_n, unsigned int pl_cap,
                            unsigned long long *__restrict__ acc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    acc[0] += (unsigned long long)n_units[0] + n_units[1];
    unsigned long long k = 0;
    if (pl_n)
        for (int i = 0; i < PAIR_PATHS * PAIR_SHARDS; ++i) k += pl_n[i] > pl_cap ? pl_cap : pl_n[i];
    acc[1] += k;
}

// the same emission for an MI block produced elsewhere (LDW_ENGINE_HIST)
__global__ __launch_bounds__(256) void k_post_mi(EmitArgs E, const int32_t *__restrict__ idx_f,
                                                 const int32_t *__restrict__ idx_t, int nt,
                                                 unsigned long long *__restrict__ ghist) {
    __shared__ unsigned int sh_hist[NBINS];
    for (int i = threadIdx.x; i < NBINS; i += 256) sh_hist[i] = 0;
    __syncthreads();
    const int a_loc = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = threadIdx.x >> 6;
    const int b_first = blockIdx.y * EPI_COLS + wave * (EPI_COLS / 4);
    if (a_loc < E.nf) {
        const int sa = idx_f[a_loc];
        for (int it = 0; it < EPI_COLS / 4; ++it) {
            const int b_loc = b_first + it;
            if (b_loc >= nt) break;
            if (E.lower_only && a_loc <= b_loc) continue;
            const ColInfo c = E.cols[b_loc];
            emit_pair(E, c, a_loc, b_loc, sa, idx_t[b_loc], E.MI[(int64_t)a_loc + (int64_t)b_loc * E.nf], sh_hist);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += 256)
        if (sh_hist[i]) atomicAdd(&ghist[i], (unsigned long long)sh_hist[i]);
}

// ------------------------------------------------------------------------------------------------
// Blocks whose SNP lists are NOT ascending in POS (the reference imposes no order on snp.dat$POS: R/computePairwiseMI.R:176-177,
// :306-333 work on whatever order the lists have).  The short-range partners of a column are then no index interval, so the
// interval machinery (ColInfo, the band masks, the screens) does not apply: such a block runs the plain path — exact GEMM + fp64 MI
// of every pair into the dense block — and these three kernels do the reference's pair list on it with the predicate itself
// (len = circ_len(pos1, pos2) <= sr_dist per pair): counts per column, short-range rows to their final place (all upper rows
// column-major, then all lower rows: R/computePairwiseMI.R:306-310), long-range histogram and candidate gather.  One wave per column.
// ------------------------------------------------------------------------------------------------
struct GenArgs {
    const double *MI;            // dense block, column-major nf x nt
    int nf, nt, lower_only, keep_sr, do_lr;
    const int32_t *idx_f, *idx_t, *POS;
    double g, sr_dist;
};

__device__ __forceinline__ bool gen_is_sr(const GenArgs &S, int a_loc, double pos1) {
    return circ_len(pos1, (double)S.POS[S.idx_f[a_loc]], S.g) <= S.sr_dist;
}

__global__ __launch_bounds__(256) void k_gen_count(GenArgs S, int32_t *__restrict__ cnt_u, int32_t *__restrict__ cnt_l,
                                                   unsigned long long *__restrict__ ghist) {
    __shared__ unsigned int sh_hist[NBINS];
    for (int i = threadIdx.x; i < NBINS; i += 256) sh_hist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, b_loc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b_loc < S.nt) {
        const double pos1 = (double)S.POS[S.idx_t[b_loc]];
        const double *col = S.MI + (int64_t)b_loc * S.nf;
        int cu = 0, cl = 0;
        for (int a_loc = lane; a_loc < S.nf; a_loc += 64) {
            const int seg = pair_seg(a_loc, b_loc, S.lower_only);
            if (seg < 0) continue;
            if (gen_is_sr(S, a_loc, pos1)) {
                cu += seg == 0;
                cl += seg == 1;
            } else if (S.do_lr) {
                atomicAdd(&sh_hist[mi_bucket(col[a_loc])], 1u);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cu += __shfl_xor(cu, off);
            cl += __shfl_xor(cl, off);
        }
        if (lane == 0) {
            cnt_u[b_loc] = cu;
            cnt_l[b_loc] = cl;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += 256)
        if (sh_hist[i]) atomicAdd(&ghist[i], (unsigned long long)sh_hist[i]);
}

// off_u / off_l: first row of the column's upper / lower short-range rows, relative to sr_base
__global__ __launch_bounds__(256) void k_gen_emit_sr(GenArgs S, const int64_t *__restrict__ off_u, const int64_t *__restrict__ off_l, int64_t sr_base,
                                                     int32_t *__restrict__ sr_a, int32_t *__restrict__ sr_b, double *__restrict__ sr_mi) {
    const int lane = threadIdx.x & 63, b_loc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b_loc >= S.nt) return;
    const int sb = S.idx_t[b_loc];
    const double pos1 = (double)S.POS[sb];
    const double *col = S.MI + (int64_t)b_loc * S.nf;
    int64_t ru = sr_base + off_u[b_loc], rl = sr_base + off_l[b_loc];
    for (int a0 = 0; a0 < S.nf; a0 += 64) {
        const int a_loc = a0 + lane;
        int seg = -1;
        bool sr = false;
        if (a_loc < S.nf) {
            seg = pair_seg(a_loc, b_loc, S.lower_only);
            sr = seg >= 0 && gen_is_sr(S, a_loc, pos1);
        }
        const unsigned long long mu = __ballot(sr && seg == 0), ml = __ballot(sr && seg == 1);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (sr) {
            const int64_t dst = seg == 0 ? ru + __popcll(mu & below) : rl + __popcll(ml & below);
            sr_a[dst] = S.idx_f[a_loc];
            sr_b[dst] = sb;
            sr_mi[dst] = col[a_loc];
        }
        ru += __popcll(mu);
        rl += __popcll(ml);
    }
}

// ------------------------------------------------------------------------------------------------
// joint tables for explicit pairs (test / inspection API): table[p][X][Y] from G entry (p, p)
// ------------------------------------------------------------------------------------------------
__global__ void k_tables(const int64_t *__restrict__ G, int RFpad, const int32_t *idx_f, const int32_t *lrow_f,
                         const int32_t *idx_t, const int32_t *lrow_t, int np, const uint32_t *slot_meta,
                         const int64_t *marg /* [L][5] by slot */, int64_t *out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    const int sa = idx_f[p], sb = idx_t[p];
    const uint32_t ma = slot_meta[sa], mb = slot_meta[sb];
    const int na = ma & 7, nb = mb & 7;
    int64_t cell[5][5];
    for (int i = 0; i <= 4; ++i)
        for (int j = 0; j <= 4; ++j) cell[i][j] = 0;
    int64_t rs[5] = {0, 0, 0, 0, 0}, cs[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            const int64_t v = G[((int64_t)lrow_t[p] + j) * RFpad + lrow_f[p] + i];
            cell[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int64_t dd = marg[(int64_t)sa * 5 + na];
    for (int i = 0; i < na; ++i) cell[i][nb] = marg[(int64_t)sa * 5 + i] - rs[i];
    for (int j = 0; j < nb; ++j) {
        cell[na][j] = marg[(int64_t)sb * 5 + j] - cs[j];
        dd -= cell[na][j];
    }
    cell[na][nb] = dd;
    int64_t *o = out + (int64_t)p * 25;
    for (int k = 0; k < 25; ++k) o[k] = 0;
    for (int i = 0; i <= na; ++i)
        for (int j = 0; j <= nb; ++j) {
            const int X = (ma >> (8 + 3 * i)) & 7, Y = (mb >> (8 + 3 * j)) & 7;
            o[X * 5 + Y] = cell[i][j];
        }
}

// unit-weight marginals by slot = state counts reordered
__global__ void k_slot_counts(const int32_t *counts, const uint32_t *slot_meta, int64_t L, int64_t *out) {
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a >= L) return;
    const uint32_t m = slot_meta[a];
    const int n = m & 7;
    for (int i = 0; i < 5; ++i) out[a * 5 + i] = (i <= n) ? counts[a * 5 + ((m >> (8 + 3 * i)) & 7)] : 0;
}

// ------------------------------------------------------------------------------------------------
// long-range selection
// ------------------------------------------------------------------------------------------------
struct PickOut {
    long long n;        // number of long-range pairs in the block
    long long lo, hi;   // 1-based ranks of the two order statistics of quantile type 7
    long long n_below;  // pairs in buckets below B
    double index, prob;
    int B;              // first bucket gathered
    int spec_ok;        // the speculative candidate list of the epilogue covers bucket B
    int B_true, over;   // bucket that holds rank lo (before the speculative override); over: bit 0 a pair list overflowed, bit 1 the maybe list did
    unsigned long long n_cand;  // filled by k_lr_gather
    long long n_kept;           // filled by k_lr_thresh
    double disc_thresh;
    long long kstart;
    long long n_below_true;     // pairs in buckets below B_true
};

// prob and quantile ranks of R/computePairwiseMI.R:352-354 (stats::quantile type 7), then the bucket
// holding rank lo.  One workgroup: chunked prefix sum over the NBINS counters.
// In speculative mode (spec_B >= 0) the histogram only holds buckets >= spec_B; n_total is the block's long-range
// pair count known to the host, and everything below spec_B is lumped into one virtual bucket.
__device__ __forceinline__ void pick_bucket_body(const unsigned long long *__restrict__ hist, double lr_retain, double lr_approx, int spec_B,
                                                 long long n_total, PickOut *__restrict__ out, const unsigned int *__restrict__ pl_n,
                                                 unsigned int pl_cap) {
    __shared__ long long part[256];
    __shared__ long long s_lo;
    constexpr int PER = NBINS / 256;
    const int t = threadIdx.x;
    long long loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        loc[k] = (long long)hist[t * PER + k];
        sum += loc[k];
    }
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        long long above = 0;
        for (int i = 0; i < 256; ++i) above += part[i];
        // speculative mode: pairs below spec_B were not histogrammed
        long long run = spec_B >= 0 ? n_total - above : 0;
        for (int i = 0; i < 256; ++i) {
            const long long v = part[i];
            part[i] = run;
            run += v;
        }
        PickOut o;
        memset(&o, 0, sizeof(o));
        o.n_cand = out->n_cand;  // speculative candidates appended by the epilogue
        o.n = run;
        o.B = NBINS;
        o.B_true = NBINS;
        o.disc_thresh = nan("");
        if (run > 0) {
            const double dn = (double)run;
            // every operation rounded on its own, in the reference's order (ldw_dev.h: lr_prob, q7_index)
            double prob = lr_prob(lr_retain, dn, lr_approx);
            if (!(prob > 0.0)) prob = 0.0;
            o.prob = prob;
            o.index = q7_index(dn - 1.0, prob);
            o.lo = (long long)floor(o.index);
            o.hi = (long long)ceil(o.index);
        }
        *out = o;
        s_lo = o.lo;
    }
    __syncthreads();
    const long long lo = s_lo;
    if (lo <= 0) return;
    long long cum = part[t];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (cum < lo && cum + loc[k] >= lo) {  // exactly one (thread, k) satisfies this
            out->B = t * PER + k;
            out->B_true = t * PER + k;
            out->n_below = cum;
            out->n_below_true = cum;
        }
        cum += loc[k];
    }
    __syncthreads();
    if (spec_B < 0) return;
    // speculative list = every long-range pair with bucket >= spec_B: usable iff rank lo lies at or above spec_B,
    // i.e. iff some bucket >= spec_B was found to hold it (the unhistogrammed mass below spec_B precedes part[0])
    if (lo > part[0]) {
        cum = part[t];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (t * PER + k == spec_B) {
                out->B = spec_B;
                out->n_below = cum;
                // a pair list of the approximate path that overflowed lost candidates: treat like a guess that was too high
                int over = 0;
                if (pl_n) {
                    for (int i = 0; i < PAIR_PATHS * PAIR_SHARDS; ++i) over |= pl_n[i] > pl_cap ? 1 : 0;
                    over |= pl_n[PAIR_PATHS * PAIR_SHARDS] > pl_cap ? 2 : 0;   // (word 40: the maybe list's overflow, k_screen_maybe)
                }
                out->over = over;
                out->spec_ok = over ? 0 : 1;
            }
            cum += loc[k];
        }
    } else if (t == 0) {
        out->n_cand = 0;  // guess too high: the fallback gather starts from an empty list
    }
}

__global__ __launch_bounds__(256) void k_pick_bucket(const unsigned long long *__restrict__ hist, double lr_retain,
                                                     double lr_approx, int spec_B, long long n_total,
                                                     PickOut *__restrict__ out, const unsigned int *__restrict__ pl_n, unsigned int pl_cap) {
    pick_bucket_body(hist, lr_retain, lr_approx, spec_B, n_total, out, pl_n, pl_cap);
}

// the same for every reference block of a span at once (blockIdx.x = segment): its own histogram, long-range pair count and pick
// record; the pair lists (and their overflow test) are the span's
struct PickSpanArgs {
    long long n_total[LDW_SPAN_MAX];
};
__global__ __launch_bounds__(256) void k_pick_bucket_span(const unsigned long long *__restrict__ hist, double lr_retain, double lr_approx, int spec_B,
                                                          PickSpanArgs S, char *__restrict__ picks, size_t pick_stride,
                                                          const unsigned int *__restrict__ pl_n, unsigned int pl_cap) {
    const int k = (int)blockIdx.x;
    pick_bucket_body(hist + (size_t)k * NBINS, lr_retain, lr_approx, spec_B, S.n_total[k], reinterpret_cast<PickOut *>(picks + (size_t)k * pick_stride), pl_n,
                     pl_cap);
}

struct GatherArgs {
    const double *MI;
    const ColInfo *cols;
    int nf, nt, lower_only;
};

// gather every long-range pair whose bucket is >= B: (MI key, order key)
__global__ __launch_bounds__(256) void k_lr_gather(GatherArgs S, PickOut *__restrict__ pick, uint64_t *__restrict__ ckey,
                                                   uint64_t *__restrict__ cval) {
    const int B = pick->B;
    if (B >= NBINS) return;
    const double lo_val = B > 0 ? bucket_lo(B) : 0.0;
    const int b0 = blockIdx.x * 16;
    for (int bb = 0; bb < 16; ++bb) {
        const int b_loc = b0 + bb;
        if (b_loc >= S.nt) break;
        const ColInfo c = S.cols[b_loc];
        const double *col = S.MI + (int64_t)b_loc * S.nf;
        for (int a_loc = threadIdx.x; a_loc < S.nf; a_loc += 256) {
            const int seg = pair_seg(a_loc, b_loc, S.lower_only);
            if (seg < 0) continue;  // also skips the never-written part of a diagonal block
            const double mi = col[a_loc];
            if (B > 0 && mi < lo_val) continue;  // cheap reject; the bucket test below decides
            if (col_is_sr(c, a_loc)) continue;
            if (mi_bucket(mi) < B) continue;
            const unsigned long long p = atomicAdd(&pick->n_cand, 1ull);
            ckey[p] = f64_key(mi);
            cval[p] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)S.nf);
        }
    }
}

// the same gather for a block in generic order (k_gen_count / k_gen_emit_sr above): the predicate instead of the intervals
__global__ __launch_bounds__(256) void k_gen_gather(GenArgs S, PickOut *__restrict__ pick, uint64_t *__restrict__ ckey, uint64_t *__restrict__ cval) {
    const int B = pick->B;
    if (B >= NBINS) return;
    const int lane = threadIdx.x & 63, b_loc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b_loc >= S.nt) return;
    const double pos1 = (double)S.POS[S.idx_t[b_loc]];
    const double *col = S.MI + (int64_t)b_loc * S.nf;
    for (int a_loc = lane; a_loc < S.nf; a_loc += 64) {
        const int seg = pair_seg(a_loc, b_loc, S.lower_only);
        if (seg < 0) continue;
        const double mi = col[a_loc];
        if (mi_bucket(mi) < B) continue;
        if (gen_is_sr(S, a_loc, pos1)) continue;
        const unsigned long long p = atomicAdd(&pick->n_cand, 1ull);
        ckey[p] = f64_key(mi);
        cval[p] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)S.nf);
    }
}


// candidates sorted ascending by MI: quantile type 7, then first kept index
__global__ void k_lr_thresh(const uint64_t *__restrict__ skey, PickOut *__restrict__ pick) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PickOut o = *pick;
    const long long m = (long long)o.n_cand;
    if (o.n <= 0 || m <= 0) {
        pick->n_kept = 0;
        pick->kstart = 0;
        return;
    }
    const long long off = o.n - m;  // ranks below the candidate set
    const double xlo = key_f64(skey[o.lo - off - 1]);
    const double xhi = key_f64(skey[o.hi - off - 1]);
    double qs = xlo;
    if (o.index > (double)o.lo && xhi != qs) {
        const double h = o.index - (double)o.lo;
        qs = q7_interp(h, qs, xhi);   // (no fma: ldw_dev.h)
    }
    const uint64_t kq = f64_key(qs);
    long long lo = 0, hi = m;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (skey[mid] < kq) lo = mid + 1; else hi = mid;
    }
    pick->disc_thresh = qs;
    pick->kstart = lo;
    pick->n_kept = m - lo;
}

// kept candidates keep their order key, the rest sink to the end of the second sort
__global__ void k_lr_mark(const uint64_t *__restrict__ skey, const uint64_t *__restrict__ sval,
                          const PickOut *__restrict__ pick, uint64_t *__restrict__ okey, uint64_t *__restrict__ oval,
                          long long m) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const bool keep = i >= pick->kstart;
    okey[i] = keep ? sval[i] : ~0ull;
    oval[i] = skey[i];
}

__global__ void k_lr_append(const uint64_t *__restrict__ okey, const uint64_t *__restrict__ oval,
                            const PickOut *__restrict__ pick, const int32_t *__restrict__ idx_f,
                            const int32_t *__restrict__ idx_t, int nf, const int64_t *__restrict__ lr_count,
                            int32_t *__restrict__ out_a, int32_t *__restrict__ out_b, double *__restrict__ out_mi) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= pick->n_kept) return;
    const uint64_t c = okey[i] & 0x3FFFFFFFFFFFFFFFull;
    const int a_loc = (int)(c % (uint64_t)nf), b_loc = (int)(c / (uint64_t)nf);
    const int64_t dst = *lr_count + i;
    out_a[dst] = idx_f[a_loc];
    out_b[dst] = idx_t[b_loc];
    out_mi[dst] = key_f64(oval[i]);
}

// ------------------------------------------------------------------------------------------------
// Selection without a sort (the common case: <= 32768 candidates above the guessed bucket), four small launches:
//   k_sel_thresh  ONE workgroup, every thread keeps 32 MI keys in registers.  The two order statistics of quantile type 7 by an
//                 11-bit-digit RADIX SELECT (6 passes, an LDS histogram each; a wave first folds the lanes that share its first
//                 lane's bin into one LDS atomic, because the top digits of the candidates nearly all coincide), the smallest key
//                 above for x[hi], the threshold with R's own rule.
//   k_sel_mark    one thread per candidate: every kept candidate sets its bit in a BITMAP over the block's row-order key space
//                 (bit = segment * nf * nt + a + b * nf) and counts itself in its 1024-bit chunk and its 65536-bit super-chunk.
//   k_sel_scatter one thread per candidate: rank = (super-chunks before, scanned per workgroup in LDS) + (chunks before, in the
//                 super-chunk) + (bits before, in the chunk) = the row's position in the reference's row order
//                 (R/computePairwiseMI.R:306-331: upper rows column-major, then lower) -> written straight to its final place.
//   k_sel_clear   zeroes the bitmap words and counters that were touched (all three arrays stay all-zero between blocks).
// Same results as the two radix sorts of the general path, which stays for blocks without a bucket guess, speculation misses,
// more candidates and key spaces beyond 2^32 bits.
// ------------------------------------------------------------------------------------------------
constexpr int SEL_CHUNK_BITS = 1024, SEL_CHUNK_WORDS = SEL_CHUNK_BITS / 32, SEL_SUPER = 64;   // super-chunk = 64 chunks
constexpr int SEL_DIGIT = 11, SEL_BINS = 1 << SEL_DIGIT;
constexpr int SEL_MAX = 1 << 20;          // candidates the sort-free path takes
constexpr int SEL_MAX_SUPER = 8192;       // super-chunks a scatter workgroup can scan in LDS: key spaces up to 2^29 bits

__device__ __forceinline__ uint64_t sel_order_key(uint64_t cv, uint64_t space) {   // cv = seg << 62 | (a + b * nf)
    return (cv >> 62) * space + (cv & 0x3FFFFFFFFFFFFFFFull);
}

constexpr int SEL_LIST = 6144;   // keys of the threshold's bucket a k_sel_thresh workgroup keeps in LDS (more: it re-reads the global list)

__device__ __forceinline__ void sel_thresh_body(const uint64_t *__restrict__ ckey, PickOut *__restrict__ pick) {
    __shared__ uint64_t list[SEL_LIST];
    __shared__ unsigned int hist[SEL_BINS];
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int s_n;
    __shared__ unsigned long long s_prefix, s_min, s_above;
    __shared__ long long s_k, s_less, s_eq;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const PickOut o = *pick;
    const long long m = (long long)o.n_cand;
    if (o.n <= 0 || m <= 0 || o.B_true >= NBINS) {
        if (t == 0) {
            pick->n_kept = 0;
            pick->kstart = -1;     // (as a key: above every candidate)
        }
        return;
    }
    // Phase A: the keys of the bucket that holds rank lo (the histogram that found it counts with the same mi_bucket) go to the
    // LDS list; the smallest key of the buckets above is x[hi] when rank hi leaves the bucket.  8 loads in flight per thread.
    if (t == 0) {
        s_n = 0u;
        s_above = ~0ull;
        s_prefix = 0ull;
        s_k = o.lo - o.n_below_true - 1;     // 0-based rank of x[lo] inside its bucket
        s_less = 0;
        s_eq = 0;
    }
    __syncthreads();
    unsigned long long above = ~0ull;
    for (long long base = 0; base < m; base += 8 * 1024) {
        uint64_t k8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long i = base + (long long)j * 1024 + t;
            k8[j] = i < m ? ckey[i] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long i = base + (long long)j * 1024 + t;
            const int bk = i < m ? mi_bucket(key_f64(k8[j])) : -1;
            if (bk > o.B_true && k8[j] < above) above = k8[j];
            const bool in = bk == o.B_true;
            const unsigned long long mk = __ballot(in);
            if (mk == 0ull) continue;
            unsigned int pos = 0;
            if (lane == __builtin_ctzll(mk)) pos = atomicAdd(&s_n, (unsigned int)__popcll(mk));
            pos = (unsigned int)__shfl((int)pos, __builtin_ctzll(mk)) + (unsigned int)__popcll(mk & ((1ull << lane) - 1ull));
            if (in && pos < SEL_LIST) list[pos] = k8[j];
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long y = __shfl_xor(above, d);
        above = y < above ? y : above;
    }
    if (lane == 0) atomicMin(&s_above, above);
    __syncthreads();
    const unsigned int n_in = s_n;
    const bool in_lds = n_in <= SEL_LIST;
    // visit every key of the bucket: the LDS list, or (huge tie groups) the global list once more
    auto n_items = [&]() -> long long { return in_lds ? (long long)n_in : m; };
    auto item = [&](long long i, uint64_t &k) -> bool {
        if (in_lds) {
            k = list[i];
            return true;
        }
        k = ckey[i];
        return mi_bucket(key_f64(k)) == o.B_true;
    };
    // Phase B: radix select of rank s_k among them, 11-bit digits from the top
    uint64_t mask = 0ull;
    const long long n_it = n_items();
    for (int pass = 0; pass < 6; ++pass) {
        const int shift = pass < 5 ? 64 - SEL_DIGIT * (pass + 1) : 0;     // 53, 42, 31, 20, 9, then the last 9 bits
        const unsigned int dmask = pass < 5 ? (unsigned int)SEL_BINS - 1u : 511u;
        for (int i = t; i < SEL_BINS; i += 1024) hist[i] = 0u;
        __syncthreads();
        const uint64_t prefix = s_prefix;
        for (long long i0 = 0; i0 < n_it; i0 += 1024) {
            const long long i = i0 + t;
            uint64_t k = 0ull;
            const bool in = i < n_it && item(i, k) && (k & mask) == prefix;
            const unsigned int bin = (unsigned int)(k >> shift) & dmask;
            const unsigned long long act = __ballot(in);
            if (act == 0ull) continue;
            const unsigned int b0 = (unsigned int)__shfl((int)bin, __builtin_ctzll(act));
            const unsigned long long same = __ballot(in && bin == b0);      // the top digits nearly all coincide: one atomic for them
            if (in && bin == b0) {
                if (lane == __builtin_ctzll(same)) atomicAdd(&hist[b0], (unsigned int)__popcll(same));
            } else if (in) {
                atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        // the bin that holds rank s_k: every thread owns two adjacent bins
        const unsigned int h0 = hist[2 * t], h1 = hist[2 * t + 1];
        unsigned int x = h0 + h1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        unsigned int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const long long incl = (long long)base + x, excl = incl - (h0 + h1), k = s_k;
        __syncthreads();
        if (k >= excl && k < incl) {        // exactly one thread
            const int b = (k < excl + h0) ? 0 : 1;
            const long long before = excl + (b ? h0 : 0);
            s_prefix = prefix | ((uint64_t)(2 * t + b) << shift);
            s_k = k - before;
            s_less += before;
            s_eq = b ? h1 : h0;
        }
        __syncthreads();
        mask |= (uint64_t)dmask << shift;
    }
    const uint64_t v_lo = s_prefix;                  // key of rank lo; s_less keys of the bucket are smaller, s_eq equal
    uint64_t v_hi = v_lo;
    const long long r_in = o.lo - o.n_below_true - 1;
    if (o.hi > o.lo && r_in + 1 >= s_less + s_eq) {  // x[hi] is the smallest key above v_lo: in the bucket, or the first one above it
        if (t == 0) s_min = s_above;
        __syncthreads();
        unsigned long long mn = ~0ull;
        for (long long i0 = 0; i0 < n_it; i0 += 1024) {
            const long long i = i0 + t;
            uint64_t k = 0ull;
            if (i < n_it && item(i, k) && k > v_lo && k < mn) mn = k;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned long long y = __shfl_xor(mn, d);
            mn = y < mn ? y : mn;
        }
        if (lane == 0) atomicMin(&s_min, mn);
        __syncthreads();
        v_hi = s_min;
    }
    if (t == 0) {
        const double xlo = key_f64(v_lo), xhi = key_f64(v_hi);
        double qs = xlo;
        if (o.index > (double)o.lo && xhi != qs) {   // stats::quantile type 7 (the interpolation is skipped on a tie)
            const double h = o.index - (double)o.lo;
            qs = q7_interp(h, qs, xhi);   // (no fma: ldw_dev.h)
        }
        pick->disc_thresh = qs;
        pick->n_kept = 0;                            // counted by k_sel_mark
        pick->kstart = (long long)f64_key(qs);       // (reused: the threshold key, read by the kernels below)
    }
}

__global__ __launch_bounds__(1024) void k_sel_thresh(const uint64_t *__restrict__ ckey, PickOut *__restrict__ pick) { sel_thresh_body(ckey, pick); }

// The selection of all reference blocks of a span in ONE launch per stage (blockIdx.y = segment): each segment has its own candidate
// list, pick record, bitmap and counters; the rows of segment k go behind those of segments 0 .. k-1 (their n_kept are final when the
// scatter starts), which is the order the reference appends in.
struct SelSpan {
    int n;
    const uint64_t *ck[LDW_SPAN_MAX], *cv[LDW_SPAN_MAX];
    PickOut *pick[LDW_SPAN_MAX];
    const int32_t *idx_t[LDW_SPAN_MAX];
    uint64_t space[LDW_SPAN_MAX];
    uint32_t *bitmap[LDW_SPAN_MAX], *chunks[LDW_SPAN_MAX], *supers[LDW_SPAN_MAX];
    int n_super[LDW_SPAN_MAX];
};
__global__ __launch_bounds__(1024) void k_sel_thresh_span(SelSpan S) { sel_thresh_body(S.ck[blockIdx.y], S.pick[blockIdx.y]); }

__device__ __forceinline__ void sel_mark_body(const uint64_t *__restrict__ ckey, const uint64_t *__restrict__ cval, PickOut *__restrict__ pick,
                                              uint64_t space, uint32_t *__restrict__ bitmap, uint32_t *__restrict__ chunk_cnt,
                                              uint32_t *__restrict__ super_cnt) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const bool keep = i < (long long)pick->n_cand && ckey[i] >= (uint64_t)pick->kstart;
    if (keep) {
        const uint64_t ok = sel_order_key(cval[i], space);
        atomicOr(&bitmap[ok >> 5], 1u << (ok & 31));
        atomicAdd(&chunk_cnt[ok / SEL_CHUNK_BITS], 1u);
        atomicAdd(&super_cnt[ok / (SEL_CHUNK_BITS * SEL_SUPER)], 1u);
    }
    const unsigned long long mk = __ballot(keep);
    if (mk != 0ull && (threadIdx.x & 63) == __builtin_ctzll(mk)) atomicAdd((unsigned long long *)&pick->n_kept, (unsigned long long)__popcll(mk));
}

__global__ __launch_bounds__(256) void k_sel_mark(const uint64_t *__restrict__ ckey, const uint64_t *__restrict__ cval, PickOut *__restrict__ pick,
                                                  uint64_t space, uint32_t *__restrict__ bitmap, uint32_t *__restrict__ chunk_cnt,
                                                  uint32_t *__restrict__ super_cnt) {
    sel_mark_body(ckey, cval, pick, space, bitmap, chunk_cnt, super_cnt);
}
__global__ __launch_bounds__(256) void k_sel_mark_span(SelSpan S) {
    const int y = blockIdx.y;
    sel_mark_body(S.ck[y], S.cv[y], S.pick[y], S.space[y], S.bitmap[y], S.chunks[y], S.supers[y]);
}

__device__ __forceinline__ void sel_scatter_body(const uint64_t *__restrict__ ckey, const uint64_t *__restrict__ cval, const PickOut *__restrict__ pick,
                                                 uint64_t space, const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ chunk_cnt,
                                                 const uint32_t *__restrict__ super_cnt, int n_super, const int32_t *__restrict__ idx_f,
                                                 const int32_t *__restrict__ idx_t, int nf, int64_t row_base,
                                                 int32_t *__restrict__ out_a, int32_t *__restrict__ out_b, double *__restrict__ out_mi) {
    __shared__ unsigned int spre[SEL_MAX_SUPER];
    __shared__ unsigned int wtot[4];
    // exclusive scan of the super-chunk counters, per workgroup: strips of consecutive counters per thread
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n_super + 255) / 256, s0 = t * per, s1 = s0 + per < n_super ? s0 + per : n_super;
    unsigned int sum = 0;
    for (int q = s0; q < s1; ++q) sum += super_cnt[q];
    unsigned int x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    unsigned int run = x - sum;
    for (int w = 0; w < wave; ++w) run += wtot[w];
    for (int q = s0; q < s1; ++q) {
        spre[q] = run;
        run += super_cnt[q];
    }
    __syncthreads();
    const long long i = blockIdx.x * (long long)blockDim.x + t;
    if (i >= (long long)pick->n_cand) return;
    const uint64_t k = ckey[i];
    if (k < (uint64_t)pick->kstart) return;
    const uint64_t cv = cval[i], ok = sel_order_key(cv, space);
    const uint64_t chunk = ok / SEL_CHUNK_BITS, sup = chunk / SEL_SUPER, w_end = ok >> 5;
    unsigned int rank = spre[sup];
    for (uint64_t q = sup * SEL_SUPER; q < chunk; ++q) rank += chunk_cnt[q];
    for (uint64_t w = chunk * SEL_CHUNK_WORDS; w < w_end; ++w) rank += __popc(bitmap[w]);
    rank += __popc(bitmap[w_end] & ((1u << (ok & 31)) - 1u));
    if ((long long)rank >= pick->n_kept) return;   // (cannot happen; never write past the rows reserved for this block)
    const uint64_t c = cv & 0x3FFFFFFFFFFFFFFFull;
    const int a_loc = (int)(c % (uint64_t)nf), b_loc = (int)(c / (uint64_t)nf);
    const int64_t dst = row_base + rank;
    out_a[dst] = idx_f[a_loc];
    out_b[dst] = idx_t[b_loc];
    out_mi[dst] = key_f64(k);
}
__global__ __launch_bounds__(256) void k_sel_scatter(const uint64_t *__restrict__ ckey, const uint64_t *__restrict__ cval, const PickOut *__restrict__ pick,
                                                     uint64_t space, const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ chunk_cnt,
                                                     const uint32_t *__restrict__ super_cnt, int n_super, const int32_t *__restrict__ idx_f,
                                                     const int32_t *__restrict__ idx_t, int nf, const int64_t *__restrict__ lr_count,
                                                     int32_t *__restrict__ out_a, int32_t *__restrict__ out_b, double *__restrict__ out_mi) {
    sel_scatter_body(ckey, cval, pick, space, bitmap, chunk_cnt, super_cnt, n_super, idx_f, idx_t, nf, *lr_count, out_a, out_b, out_mi);
}
__global__ __launch_bounds__(256) void k_sel_scatter_span(SelSpan S, const int32_t *__restrict__ idx_f, int nf, const int64_t *__restrict__ lr_count,
                                                          int32_t *__restrict__ out_a, int32_t *__restrict__ out_b, double *__restrict__ out_mi) {
    const int y = blockIdx.y;
    int64_t base = *lr_count;
    for (int j = 0; j < y; ++j) base += S.pick[j]->n_kept;   // (final: k_sel_mark_span of every segment has finished)
    sel_scatter_body(S.ck[y], S.cv[y], S.pick[y], S.space[y], S.bitmap[y], S.chunks[y], S.supers[y], S.n_super[y], idx_f, S.idx_t[y], nf, base, out_a, out_b,
                     out_mi);
}

__device__ __forceinline__ void sel_clear_body(const uint64_t *__restrict__ ckey, const uint64_t *__restrict__ cval, const PickOut *__restrict__ pick,
                                               uint64_t space, uint32_t *__restrict__ bitmap, uint32_t *__restrict__ chunk_cnt,
                                               uint32_t *__restrict__ super_cnt) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)pick->n_cand) return;
    if (ckey[i] < (uint64_t)pick->kstart) return;
    const uint64_t ok = sel_order_key(cval[i], space);
    bitmap[ok >> 5] = 0u;
    chunk_cnt[ok / SEL_CHUNK_BITS] = 0u;
    super_cnt[ok / (SEL_CHUNK_BITS * SEL_SUPER)] = 0u;
}

__global__ __launch_bounds__(256) void k_sel_clear(const uint64_t *__restrict__ ckey, const uint64_t *__restrict__ cval, const PickOut *__restrict__ pick,
                                                   uint64_t space, uint32_t *__restrict__ bitmap, uint32_t *__restrict__ chunk_cnt,
                                                   uint32_t *__restrict__ super_cnt) {
    sel_clear_body(ckey, cval, pick, space, bitmap, chunk_cnt, super_cnt);
}
__global__ __launch_bounds__(256) void k_sel_clear_span(SelSpan S) {
    const int y = blockIdx.y;
    sel_clear_body(S.ck[y], S.cv[y], S.pick[y], S.space[y], S.bitmap[y], S.chunks[y], S.supers[y]);
}
// running count and stats of every reference block of a span (one thread)
struct SpanDoneArgs {
    long long n_sr[LDW_SPAN_MAX];
};
__global__ void k_span_done(SelSpan S, SpanDoneArgs D, int64_t *lr_count, int64_t *stats_i, double *stats_d) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < S.n; ++k) {
        const PickOut *pk = S.pick[k];
        *lr_count += pk->n_kept;
        stats_i[k * 3 + 0] = pk->n;
        stats_i[k * 3 + 1] = pk->n_kept;
        stats_i[k * 3 + 2] = D.n_sr[k];
        stats_d[k] = pk->disc_thresh;
    }
}

// (a, b) of the short-range rows of ONE block of contiguous SNP ranges, from positions alone: the rows a pass emits there are a pure
// function of POS, g, sr_dist and the block geometry (R/computePairwiseMI.R:306-333) — what lets a multi-GPU gather send only the MI column of
// the short-range table (8 of 16 bytes per row) and rank 0 rebuild the index columns itself.  One wave per to-side SNP.
__global__ __launch_bounds__(256) void k_sr_fill(const ColInfo *__restrict__ cols, int nf, int nt, int fs0, int ts0, int lower_only, int64_t base,
                                                 int32_t *__restrict__ a_out, int32_t *__restrict__ b_out) {
    const int lane = threadIdx.x & 63, b_loc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b_loc >= nt) return;
    const ColInfo c = cols[b_loc];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int a_loc = c.s[k] + lane; a_loc < c.e[k]; a_loc += 64) {
            const int seg = pair_seg(a_loc, b_loc, lower_only);
            if (seg < 0) continue;
            const int64_t dst = base + (seg == 0 ? c.off_u + col_count(c, 0, a_loc) : c.off_l + col_count(c, b_loc + 1, a_loc));
            a_out[dst] = fs0 + a_loc;
            b_out[dst] = ts0 + b_loc;
        }
}

// running device-side counters and per-block stats
__global__ void k_block_done(PickOut *pick, int64_t *lr_count, int64_t n_sr_blk, int64_t *stats_i, double *stats_d) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    *lr_count += pick->n_kept;
    stats_i[0] = pick->n;
    stats_i[1] = pick->n_kept;
    stats_i[2] = n_sr_blk;
    stats_d[0] = pick->disc_thresh;
}

}  // namespace ldw

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

int upload_i32(ldw_ctx *c, ldw::DevBuf &buf, const std::vector<int32_t> &v) {
    if (int rc = buf.reserve(v.size() * 4 + 4)) return rc;
    if (!v.empty()) LDW_HIP(hipMemcpyAsync(buf.p, v.data(), v.size() * 4, hipMemcpyHostToDevice, c->stream));
    return LDW_OK;
}

// Row list of one side of a block.  SNPs are grouped by slot-count CLASS — 1, 2 or 4 indicator rows after padding
// (0 -> 1, 3 -> 4 with rows of zeros) — in their list order within a class, and every class starts on a 32-row
// boundary: a 32 x 32 MFMA tile then holds whole SNPs of one class on either side and no SNP straddles the 64 x 32
// sub-tile of a wave, which is what lets the fused kernel (ldw_fused.hip) run the epilogue on chip.  lrow[k] = row
// position of SNP k's first row; pos[p] = k at that position, -1 elsewhere; cls[g] = class of 32-row group g.
struct SideLists {
    std::vector<int32_t> rowlist, lrow, pos;
    std::vector<uint8_t> cls;
    int Rpad = 0;
};

// order1 (optional): the list positions of the SNPs of class 1 in the order their rows are to take (prep_block: by the weight of
// the minor state); the other classes, and class 1 without it, keep the list order.
int build_side(ldw_ctx *c, const int32_t *idx, int64_t n, SideLists &S, const std::vector<int32_t> *order1 = nullptr) {
    S.lrow.assign((size_t)n, 0);
    S.rowlist.clear();
    S.pos.clear();
    S.cls.clear();
    const int32_t zero_row = (int32_t)c->R;   // rows R .. R+TILE-1 of Mbits are zero
    for (int64_t k = 0; k < n; ++k)
        LDW_REQUIRE(idx[k] >= 0 && idx[k] < c->L, LDW_ERR_ARG, "SNP index %d out of range 0..%lld", idx[k], (long long)c->L - 1);
    for (int cl : {1, 2, 4}) {
        const bool ordered = cl == 1 && order1 != nullptr;
        const int64_t cnt = ordered ? (int64_t)order1->size() : n;
        for (int64_t kk = 0; kk < cnt; ++kk) {
            const int64_t k = ordered ? (int64_t)(*order1)[(size_t)kk] : kk;
            const int32_t a = idx[k];
            const int nr = c->h_row0[a + 1] - c->h_row0[a];
            const int mine = nr <= 1 ? 1 : (nr == 2 ? 2 : 4);
            if (mine != cl) continue;
            S.lrow[k] = (int32_t)S.rowlist.size();
            for (int q = 0; q < cl; ++q) {
                S.rowlist.push_back(q < nr ? c->h_row0[a] + q : zero_row);
                S.pos.push_back(q == 0 ? (int32_t)k : -1);
            }
        }
        while (S.rowlist.size() % 32) {
            S.rowlist.push_back(zero_row);
            S.pos.push_back(-1);
        }
        S.cls.resize(S.rowlist.size() / 32, (uint8_t)cl);
    }
    int64_t rp = ((int64_t)S.rowlist.size() + TILE - 1) / TILE * TILE;
    if (rp == 0) rp = TILE;
    LDW_REQUIRE(rp < 2000000000LL, LDW_ERR_ARG, "block too large");
    S.rowlist.resize((size_t)rp, zero_row);
    S.pos.resize((size_t)rp, -1);
    S.cls.resize((size_t)rp / 32, (uint8_t)1);
    S.Rpad = (int)rp;
    return LDW_OK;
}

// Short-range intervals of every to-side SNP (host, O(nt log nf)); returns the block's short-range row count.
// Requires the from-side list to be ascending in POS (true for contiguous blocks and for the order-preserving
// subsets of SR-only mode).  Every interval boundary is verified with the exact predicate of the reference
// (circ_len <= sr_dist); a column that fails the check is rebuilt by scanning.
int build_cols(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, bool diag, double sr_dist,
               std::vector<ColInfo> &cols, int64_t &n_sr_blk) {
    const double g = c->g;
    std::vector<double> pf((size_t)nf);
    for (int64_t a = 0; a < nf; ++a) {
        pf[a] = (double)c->h_POS[from_idx[a]];
        LDW_REQUIRE(a == 0 || pf[a] >= pf[a - 1], LDW_ERR_ARG, "from-side SNP list must be ascending in POS (position %lld)", (long long)a);
    }
    cols.resize((size_t)nt);
    {   // quick reject: when the two position ranges are farther apart than sr_dist both directly and around the
        // origin, no pair of the block is short-range (true for most off-diagonal blocks)
        double pt_min = (double)c->h_POS[to_idx[0]], pt_max = pt_min;
        for (int64_t b = 1; b < nt; ++b) {
            const double v = (double)c->h_POS[to_idx[b]];
            pt_min = v < pt_min ? v : pt_min;
            pt_max = v > pt_max ? v : pt_max;
        }
        const double pf_min = pf.front(), pf_max = pf.back();
        const bool apart = (pt_min - pf_max > sr_dist && pf_min + g - pt_max > sr_dist) ||
                           (pf_min - pt_max > sr_dist && pt_min + g - pf_max > sr_dist);
        if (apart && 2 * sr_dist < g) {
            ColInfo z;
            memset(&z, 0, sizeof(z));
            std::fill(cols.begin(), cols.end(), z);
            n_sr_blk = 0;
            return LDW_OK;
        }
    }
    int64_t total_u = 0, total_l = 0;
    std::vector<int32_t> cu((size_t)nt), cl((size_t)nt);
    for (int64_t b = 0; b < nt; ++b) {
        const double p1 = (double)c->h_POS[to_idx[b]];
        auto P = [&](int64_t a) { return circ_len(p1, pf[a], g) <= sr_dist; };
        ColInfo ci;
        memset(&ci, 0, sizeof(ci));
        // candidate intervals from the three position windows
        int64_t iv[3][2];
        iv[0][0] = 0;                                                                                // wrap-low: x <= p1 + sr - g
        iv[0][1] = std::upper_bound(pf.begin(), pf.end(), p1 + sr_dist - g) - pf.begin();
        iv[1][0] = std::lower_bound(pf.begin(), pf.end(), p1 - sr_dist) - pf.begin();               // centre
        iv[1][1] = std::upper_bound(pf.begin(), pf.end(), p1 + sr_dist) - pf.begin();
        iv[2][0] = std::lower_bound(pf.begin(), pf.end(), p1 - sr_dist + g) - pf.begin();           // wrap-high
        iv[2][1] = nf;
        // merge overlapping / touching intervals, keep ascending order
        int n = 0;
        int64_t m[3][2];
        for (int k = 0; k < 3; ++k) {
            if (iv[k][1] <= iv[k][0]) continue;
            if (n > 0 && iv[k][0] <= m[n - 1][1]) m[n - 1][1] = std::max(m[n - 1][1], iv[k][1]);
            else { m[n][0] = iv[k][0]; m[n][1] = iv[k][1]; ++n; }
        }
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) {
            ok = P(m[k][0]) && P(m[k][1] - 1);
            if (ok && m[k][0] > 0) ok = !P(m[k][0] - 1);
            if (ok && m[k][1] < nf) ok = !P(m[k][1]);
        }
        if (n == 0 && nf > 0) ok = !P(0) && !P(nf - 1);
        if (!ok) {  // rebuild from the exact predicate
            n = 0;
            int64_t a = 0;
            while (a < nf) {
                if (!P(a)) { ++a; continue; }
                int64_t e = a;
                while (e < nf && P(e)) ++e;
                LDW_REQUIRE(n < 3, LDW_ERR_ARG, "short-range partners of SNP %d form more than three index runs", to_idx[b]);
                m[n][0] = a; m[n][1] = e; ++n;
                a = e;
            }
        }
        for (int k = 0; k < 3; ++k) {
            ci.s[k] = k < n ? (int32_t)m[k][0] : 0;
            ci.e[k] = k < n ? (int32_t)m[k][1] : 0;
        }
        cu[b] = diag ? 0 : col_count(ci, 0, (int)std::min<int64_t>(b, nf));
        cl[b] = col_count(ci, (int)std::min<int64_t>(b + 1, nf), (int)nf);
        total_u += cu[b];
        total_l += cl[b];
        cols[b] = ci;
    }
    // row order of the block: all upper-segment columns, then all lower-segment columns (R/computePairwiseMI.R:306-310)
    int64_t ru = 0, rl = total_u;
    for (int64_t b = 0; b < nt; ++b) {
        cols[b].off_u = ru;
        cols[b].off_l = rl;
        ru += cu[b];
        rl += cl[b];
    }
    n_sr_blk = total_u + total_l;
    return LDW_OK;
}

bool same_list(const int32_t *a, int64_t na, const int32_t *b, int64_t nb) {
    if (na != nb) return false;
    for (int64_t i = 0; i < na; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

// device pointers of one block's index structures
struct DevPtrs {
    const int32_t *idx_f, *idx_t, *rl_f, *rl_t, *lrow_f, *lrow_t, *perm, *perm_t, *pos_f, *pos_t;
    const uint8_t *cls_f, *cls_t;
    const int32_t *cmax_f;      // mixed-precision path: per from-tile widest row-slot class, offsets of the low-limb blocks
    const int64_t *tile_base;
    const int32_t *tf_list;     // (tile, fs) pairs of the gathered GEMM's grid
    const uint8_t *band_mask;   // approximate path: [RTpad / 128][RFpad / 64] exact-GEMM tiles that hold a short-range pair
    int nf_tiles;               // tiles of 64 in the padded from-side order perm
    int gen_t0, gen_q0;         // first from-tile / column slot of the SNPs with >= 3 minor states or none (k_mi_screen_generic)
};

// host half of LoGeom
struct LoHost {
    int32_t n_lc[3] = {0, 0, 0}, uoff[3] = {0, 0, 0}, rowbase[3] = {0, 0, 0};
    int32_t RTlo = 0, ntiles = 0, n_tf = 0;
    int64_t glo_total = 0;
    int32_t n_tiles_cf[3] = {0, 0, 0};   // from-tiles whose widest row-slot class is 1, 2, 4
    int band_full = 0;                   // the exact GEMM has to cover every tile (a SNP with unflagged slots: its units are not screened)
    int apx = 0, slot = 0, diag = 0;     // approximate-GEMM path (ldw_apx.h) instead of the high-limb GEMM + gathered low limbs
    int ordered = 0;                     // rows of the one-row SNPs in order of the minor state's weight (prep_block: tile pruning)
    int fuse_ok = 0;                     // rows of one-row SNPs sit at their slot index in both row lists (no SNP without a row): the
                                         // GEMM's epilogue may apply the threshold table by row (ApxGemmArgs::fuse)
    int sr_sub = 0;                      // an SR sub-pass (short-range pairs of a span's corner segment): see HostBlock::sr_sub
    int sr_excl = 0;                     // the block's short-range pairs are evaluated by an SR sub-pass: keep them out of the candidates, list no unit for them
    int span = 0;                        // reference blocks on the to side (0: an ordinary block); sseg: their candidate lists / histograms
    const SpanSeg *sseg = nullptr;
    // r04: the threshold table this item's FIRST phase chose (null: none applies) — its second phase, queued after the first phases of later
    // items (which may rebuild the table for their level), takes exactly this one
    mutable const int2 *tab_p = nullptr;
    mutable bool tab_set = false;
};

void fill_epi_args(ldw_ctx *c, const DevPtrs &D, int64_t nf, int64_t nt, int RFpad, int quirk, const EmitArgs &E, const int64_t *G,
                   EpiArgs &A) {
    A.G = G;
    A.RFpad = RFpad;
    A.RTpad = 0;
    A.idx_f = D.idx_f;
    A.lrow_f = D.lrow_f;
    A.idx_t = D.idx_t;
    A.lrow_t = D.lrow_t;
    A.nf = (int)nf;
    A.nt = (int)nt;
    A.slot_meta = c->slot_meta.as<uint32_t>();
    A.slot_pfix = c->slot_pfix.as<int64_t>();
    A.r = c->r.as<double>();
    A.neff = c->neff;
    A.scale = std::ldexp(1.0, -c->frac_bits);
    A.quirk = quirk;
    A.gen_t0 = D.gen_t0;
    A.gen_q0 = D.gen_q0;
    A.colpack = A.colpack_hi = nullptr;
    A.rowpack = A.rowpack_hi = nullptr;
    A.rloc_f = A.rloc_t = nullptr;
    A.pl_pairs = nullptr;
    A.pl_n = nullptr;
    A.pl_cap = 0;
    A.row0 = c->row0.as<int32_t>();
    A.tab11 = nullptr;
    A.clean = nullptr;
    A.clean_stride = 0;
    A.sflag_f = A.sflag_t = nullptr;
    A.snp_sup = nullptr;
    A.tab_nb = 0;
    A.tab_c = 0;
    A.span = 0;
    A.sr_excl = 0;
    memset(A.sseg, 0, sizeof(A.sseg));
    A.E = E;
    memset(&A.lo, 0, sizeof(A.lo));
}

// GEMM + epilogue (or the histogram engine) of one block into ctx->MIblk; with E.cols set, the short-range
// scatter and the long-range histogram ride along.  Everything is asynchronous.
// stage events per block: [0] GEMM start, [1] GEMM end (GEMM stream); [4] epilogue start, [2] epilogue end, [3] selection
// end (main stream).  The GEMM of block b+1 runs beside the epilogue and selection of block b, so the stage times overlap.
constexpr int EVB = 6;
// which: 1 = GEMM only (on gstream, into Gbuf), 2 = epilogue only, 3 = both
constexpr int HI_LIMBS = 3, LO_LIMBS = 2;   // mixed-precision split of the 5 weight limbs
constexpr size_t PAIR_CAP = 1u << 18;       // most entries per pair list of the approximate path (PAIR_PATHS x PAIR_SHARDS lists)
// capacity of the pair lists of one block: an eighth of the block's pairs (a shard's fair share of ALL of them), between 2^12 and
// PAIR_CAP.  Small blocks (tests, several engines on one GPU) then take megabytes instead of the fixed 1.4 GB; a list that
// overflows makes the block fall back like a wrong guess (k_pick_bucket: spec_ok = 0), so results never depend on it.
// nseg > 4: the lists of a span of that many reference blocks (about 4e4 listed pairs per 10k x 10k block, most of them in ONE of the five
// paths: 3-state x 3-state SNP pairs) get twice the room.  g_pair_cap_override (ldw_set_pair_cap, tests only): a fixed small capacity, so
// that the overflow fallback can be exercised.
static std::atomic<uint32_t> g_pair_cap_override{0};
inline uint32_t pair_cap_for(int64_t nf, int64_t nt, int nseg = 0) {
    const uint32_t ovr = g_pair_cap_override.load();
    if (ovr) return ovr;
    const uint64_t top = nseg > 4 ? 2 * PAIR_CAP : PAIR_CAP;
    uint64_t want = (uint64_t)nf * (uint64_t)nt / PAIR_SHARDS + 1, cap = 1u << 12;
    while (cap < want && cap < top) cap <<= 1;
    return (uint32_t)cap;
}
// entries of the maybe list of an item (ApxGemmArgs::maybe).  r05: the WORST case — every region of 32 to-rows x 64 from-rows of the row
// rectangle hands over APX_MAYBE_MAX entries (0.56 B per entry of G', a seventh of G' itself) — so the list cannot overflow whatever the data
// look like.  r04 sized it for "one pair in a few thousand fails its table thresholds" (nf nt / 256 + 65 536), which holds on alignments full
// of rare states; on an alignment without them (MAF 0.2-0.5: the bench's adversarial leg) most regions hand over tens of entries, the list
// overflowed on nearly every block and each of them was redone on the plain path — 216 ms per pass against 155 ms for the plain path itself.
inline uint32_t maybe_cap_for(int64_t RTpad, int64_t RFpad) {
    if (const char *e = getenv("LDW_MAYBE_CAP")) {   // (tests: a list that overflows makes its block take the pair lists' overflow path)
        const long k = atol(e);
        if (k > 0) return (uint32_t)k;
    }
    return (uint32_t)std::min<int64_t>(((RTpad + 31) / 32) * ((RFpad + 63) / 64) * (int64_t)APX_MAYBE_MAX + 64, (int64_t)1 << 30);
}
// a block (or a span's segment) redone because a list overflowed (PickOut::over): counted, and a maybe list that overflowed — impossible at its
// worst-case capacity, so only under LDW_MAYBE_CAP — is switched off for the rest of the pass instead of overflowing item after item
inline void note_overflow(ldw_ctx *c, int over) {
    if (over & 1) ++c->pair_list_overflows;
    if (over & 2) {
        ++c->maybe_overflows;
        c->maybe_off = true;
    }
}
int launch_block_mi(ldw_ctx *c, const DevPtrs &D, int64_t nf, int64_t nt, int RFpad, int RTpad, int quirk, EmitArgs E,
                    hipEvent_t *ev, int which, ldw::DevBuf *Gb, hipStream_t gstream, unsigned long long *ghist,
                    const LoHost *mixed = nullptr) {
    ldw::DevBuf &Gbuf = Gb ? *Gb : c->G;
    if (!gstream) gstream = c->stream;
    if (which == 1) {   // the GEMM of a block, possibly on its own stream so that it overlaps the previous block's tail
        if (int rc = Gbuf.reserve((size_t)RFpad * RTpad * 8)) return rc;
        LDW_HIP(hipEventRecord(ev[0], gstream));
        // mixed-precision path: the block-wide GEMM carries the high limbs only
        if (int rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, D.rl_t, RTpad, D.rl_f, RFpad, Gbuf.as<int64_t>(),
                                      mixed ? HI_LIMBS : c->nlimbs, c->digits.as<int8_t>() + (mixed ? (int64_t)LO_LIMBS * c->KW * 64 : 0),
                                      E.lower_only, gstream))
            return rc;
        LDW_HIP(hipEventRecord(ev[1], gstream));
        LDW_HIP(hipEventRecord(ev[5], gstream));
        return LDW_OK;
    }
    const bool epilogue_only = which == 2;
    if (int rc = c->MIblk.reserve((size_t)nf * nt * 8)) return rc;
    E.MI = c->MIblk.as<double>();
    E.nf = (int)nf;
    dim3 egrid((unsigned)(D.nf_tiles > 0 ? D.nf_tiles : (nf + 63) / 64), (unsigned)((nt + EPI_COLS - 1) / EPI_COLS));
    LDW_REQUIRE(egrid.y <= 65535u, LDW_ERR_ARG, "nt too large for the epilogue grid");
    if (c->engine == LDW_ENGINE_HIST_STATES) {
        LDW_HIP(hipEventRecord(ev[0], c->stream));
        LDW_HIP(hipEventRecord(ev[1], c->stream));
#ifdef LDW_EXPERIMENTS
        if (int rc = launch_hist(c, D.idx_f, (int)nf, D.idx_t, (int)nt, c->pfix_state.as<int64_t>(), quirk, E.lower_only,
                                 c->MIblk.as<double>()))
            return rc;
#else
        LDW_REQUIRE(false, LDW_ERR_STATE, "LDW_ENGINE_HIST_STATES is not part of this build (LDW_EXPERIMENTS)");   // (unreachable: ldw_set_engine refuses)
#endif
        if (E.cols) {
            hipLaunchKernelGGL(k_post_mi, egrid, dim3(256), 0, c->stream, E, D.idx_f, D.idx_t, (int)nt, ghist);
            LDW_HIP(hipGetLastError());
        }
        LDW_HIP(hipEventRecord(ev[2], c->stream));
        return LDW_OK;
    }
    if (!epilogue_only) {
        if (int rc = Gbuf.reserve((size_t)RFpad * RTpad * 8)) return rc;
        LDW_HIP(hipEventRecord(ev[0], c->stream));
        if (c->engine == LDW_ENGINE_HIST) {   // the histogram formulation on bit planes: class-wise popcounts instead of the MFMA GEMM
            if (int rc = launch_cooc_popc(c, D.rl_t, RTpad, D.rl_f, RFpad, Gbuf.as<int64_t>(), E.lower_only, c->stream)) return rc;
        } else if (int rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, D.rl_t, RTpad, D.rl_f, RFpad, Gbuf.as<int64_t>(), c->nlimbs,
                                             c->digits.as<int8_t>(), E.lower_only, c->stream))
            return rc;
        LDW_HIP(hipEventRecord(ev[1], c->stream));
    }
    EpiArgs A;
    fill_epi_args(c, D, nf, nt, RFpad, quirk, E, Gbuf.as<int64_t>(), A);
    if (which == 2) LDW_HIP(hipEventRecord(ev[4], c->stream));
    if (A.E.scr_mode && A.E.cols && !A.E.write_dense) {
        // speculative mode: the lean fp32 screen lists the units that need the exact value, k_mi_units evaluates those
        const size_t n_units_max = (size_t)egrid.x * (size_t)nt;
        const size_t o_cnt = 64, o_flat = o_cnt + ((size_t)egrid.x * 12 + 63) / 64 * 64, o_tl = o_flat + 2 * n_units_max * 8;
        const int64_t list_stride = (int64_t)n_units_max;   // two flat lists: straight-line units, the others
        if (int rc = c->scr_units.reserve(o_tl + n_units_max * 4 + 64)) return rc;
        char *ub = c->scr_units.as<char>();
        unsigned int *n_units = reinterpret_cast<unsigned int *>(ub);
        uint64_t *units = reinterpret_cast<uint64_t *>(ub + o_flat);
        LDW_HIP(hipMemsetAsync(ub, 0, o_flat, c->stream));   // the unit counter and the per-(tile, class) counters
        if (mixed) {
            LDW_REQUIRE(mixed->ntiles == (int)egrid.x, LDW_ERR_STATE, "mixed-precision geometry does not match the epilogue grid");
            if (int rc = ensure_hi_marginals(c)) return rc;
            if (int rc = c->glo.reserve((size_t)mixed->glo_total * 4 + 64)) return rc;
            LoGeom &lo = A.lo;
            for (int k = 0; k < 3; ++k) {
                lo.n_lc[k] = mixed->n_lc[k];
                lo.uoff[k] = mixed->uoff[k];
                lo.rowbase[k] = mixed->rowbase[k];
            }
            lo.RTlo = mixed->RTlo;
            lo.ntiles = mixed->ntiles;
            lo.on = 1;
            lo.cmax_f = D.cmax_f;
            lo.tile_base = D.tile_base;
            lo.cnt = reinterpret_cast<unsigned int *>(ub + o_cnt);
            lo.tl = reinterpret_cast<uint32_t *>(ub + o_tl);
            lo.glo = c->glo.as<int32_t>();
            lo.slot_pfix_hi = c->slot_pfix_hi.as<int64_t>();
            lo.hi_shift = 8 * LO_LIMBS;
        }
        {   // per-block SNP constants in epilogue order (A.lo is final by now: the high-limb marginals are reachable)
            const int nf_slots = (int)egrid.x * 64;
            const size_t o_cph = ((size_t)nt * sizeof(ColMeta) + 255) / 256 * 256, o_rp = 2 * o_cph;
            const size_t o_rph = o_rp + ((size_t)nf_slots * sizeof(RowPack) + 255) / 256 * 256;
            const size_t o_rf = o_rph + ((size_t)nf_slots * sizeof(RowPack) + 255) / 256 * 256, o_rt = o_rf + ((size_t)nf * 4 + 255) / 256 * 256;
            if (int rc = c->packs.reserve(o_rt + (size_t)nt * 4 + 256)) return rc;
            char *pb = c->packs.as<char>();
            ColMeta *cp = reinterpret_cast<ColMeta *>(pb), *cph = reinterpret_cast<ColMeta *>(pb + o_cph);
            RowPack *rp = reinterpret_cast<RowPack *>(pb + o_rp), *rph = reinterpret_cast<RowPack *>(pb + o_rph);
            float *rlf = reinterpret_cast<float *>(pb + o_rf), *rlt = reinterpret_cast<float *>(pb + o_rt);
            const int nthr = std::max<int>((int)nt, nf_slots);
            hipLaunchKernelGGL(k_build_packs, dim3((unsigned)((nthr + 255) / 256), 2), dim3(256), 0, c->stream, A, D.perm, D.perm_t, nf_slots,
                               mixed ? 1 : 0, cp, cph, rp, rph, rlf, rlt);
            A.rloc_f = rlf;
            A.rloc_t = rlt;
            LDW_HIP(hipGetLastError());
            A.colpack = cp;
            A.colpack_hi = mixed ? cph : cp;
            A.rowpack = rp;
            A.rowpack_hi = mixed ? rph : rp;
        }
        const int rm = quirk == LDW_QUIRK_REFERENCE ? (nf == nt ? 1 : 2) : 0;
        if (rm == 0) hipLaunchKernelGGL((k_mi_screen<0, false>), egrid, dim3(256), 0, c->stream, A, D.perm, D.perm_t, units, n_units, list_stride);
        else if (rm == 1) hipLaunchKernelGGL((k_mi_screen<1, false>), egrid, dim3(256), 0, c->stream, A, D.perm, D.perm_t, units, n_units, list_stride);
        else hipLaunchKernelGGL((k_mi_screen<2, false>), egrid, dim3(256), 0, c->stream, A, D.perm, D.perm_t, units, n_units, list_stride);
        LDW_HIP(hipGetLastError());
        {   // the units outside k_mi_screen's domain: generic from-tiles x all columns, the other tiles x the generic columns
            const int gt0 = std::min<int>(A.gen_t0, (int)egrid.x);
            const int q0 = std::min<int>(A.gen_q0, (int)nt) / GEN_COLS * GEN_COLS;
            {
                GenRegions Rg;
                Rg.tile0_a = gt0;
                Rg.nt_a = (int)egrid.x - gt0;
                Rg.ncg_a = Rg.nt_a > 0 ? (int)((nt + GEN_COLS - 1) / GEN_COLS) : 0;
                Rg.nt_b = (gt0 > 0 && A.gen_q0 < (int)nt) ? gt0 : 0;
                Rg.q0_b = q0;
                Rg.ncg_b = Rg.nt_b > 0 ? (int)((nt - q0 + GEN_COLS - 1) / GEN_COLS) : 0;
                const long long nblk = (long long)Rg.nt_a * Rg.ncg_a + (long long)Rg.nt_b * Rg.ncg_b;
                if (nblk > 0)
                    hipLaunchKernelGGL(k_mi_screen_generic<false>, dim3((unsigned)nblk), dim3(256), 0, c->stream, A, D.perm, D.perm_t, units, n_units, list_stride, Rg);
            }
            LDW_HIP(hipGetLastError());
        }
        if (mixed) {   // low limbs of the listed units
            LoGemmArgs P;
            P.Mbits = c->Mbits.as<uint64_t>();
            P.KW = c->KW;
            P.Kpad = c->KW * 64;
            P.digits_lo = c->digits.as<int8_t>();
            P.perm_f = D.perm;
            P.idx_f = D.idx_f;
            P.perm_t = D.perm_t;
            P.idx_t = D.idx_t;
            P.row0 = c->row0.as<int32_t>();
            P.zero_row = (int32_t)c->R;
            P.nf = (int32_t)nf;
            P.nt = (int32_t)nt;
            P.tf_list = D.tf_list;
            P.lo = A.lo;
            if (int rc = launch_gemm_lo_units(c, P, mixed->n_tf, c->stream)) return rc;
        }
        UnitLists UL;
        memset(&UL, 0, sizeof(UL));
        UL.units[0] = units;
        UL.n[0] = n_units;
        hipLaunchKernelGGL(k_mi_units<true>, dim3(2048, 1), dim3(256), 0, c->stream, A, D.perm, D.perm_t, UL, ghist);
        UnitLists UG;
        memset(&UG, 0, sizeof(UG));
        UG.units[0] = units + list_stride;
        UG.n[0] = n_units + 1;
        hipLaunchKernelGGL(k_mi_units<false>, dim3(512, 1), dim3(256), 0, c->stream, A, D.perm, D.perm_t, UG, ghist);
        LDW_HIP(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_mi_epilogue, egrid, dim3(256), 0, c->stream, A, D.perm, D.perm_t, ghist);
        LDW_HIP(hipGetLastError());
    }
    LDW_HIP(hipEventRecord(ev[2], c->stream));
    return LDW_OK;
}

// ------------------------------------------------------------------------------------------------
// One block in the approximate-GEMM path (ldw_apx.h), in two phases so that the block-wide work of block b+1 runs beside
// the tail of block b:
//   phase 1 (stream gs): pack the bit panels of both row lists, the dual-digit int8 pass into the int32 block (not for an
//            SR-only pass: it screens without MI), per-block SNP constants, the approximate screens -> unit lists (units
//            that hold a short-range pair) and pair lists (the other long-range candidates), and the EXACT 5-limb GEMM of
//            the tiles that hold a short-range pair (band_mask: the short-range band is dense, a fifth of a diagonal block);
//   phase 2 (main stream): exact sums + fp64 MI + emission of the listed pairs (k_pair_sums, k_pair_mi) and the fp64
//            evaluation of the listed units from the exact tiles (k_mi_units).
// Everything phase 1 writes is per pipeline slot; both phases derive the same pointers from the slot's buffers.
// Events: ev[0] / ev[1] around the packing + approximate GEMM, ev[5] after the screens and the band GEMM (gs); ev[4] / ev[2]
// around phase 2.
// ------------------------------------------------------------------------------------------------
int launch_block_apx(ldw_ctx *c, const DevPtrs &D, int64_t nf, int64_t nt, int RFpad, int RTpad, int quirk, EmitArgs E, hipEvent_t *ev, int phase,
                     hipStream_t gs, unsigned long long *ghist, const LoHost *lo_h, void *zero_hist = nullptr, void *zero_pick = nullptr,
                     size_t zero_pick_bytes = 0, hipStream_t s2 = nullptr) {
    const int s = lo_h->slot;
    if (!s2) s2 = c->stream;   // stream of phase 2 (an SR sub-pass runs both of its phases on one stream)
    // an SR sub-pass has the slot's second set of list / constant buffers: it runs on the main stream while the item's long-range pass, whose
    // first phase filled the first set on the GEMM stream, still needs them for its second phase
    ldw::DevBuf &B_units = lo_h->sr_sub ? c->sub_units[s] : c->apx_units[s], &B_packs = lo_h->sr_sub ? c->sub_packs[s] : c->apx_packs[s],
                &B_bins = lo_h->sr_sub ? c->sub_bins[s] : c->apx_bins[s], &B_live = lo_h->sr_sub ? c->sub_live[s] : c->scr_live[s];
    E.nf = (int)nf;
    E.MI = nullptr;
    dim3 egrid((unsigned)(D.nf_tiles > 0 ? D.nf_tiles : (nf + 63) / 64), (unsigned)((nt + EPI_COLS - 1) / EPI_COLS));
    LDW_REQUIRE(egrid.y <= 65535u, LDW_ERR_ARG, "nt too large for the epilogue grid");
    LDW_REQUIRE(E.scr_mode && E.cols && !E.write_dense, LDW_ERR_STATE, "the approximate path needs the screen");
    const size_t n_units_max = (size_t)egrid.x * (size_t)nt;
    const size_t o_flat = 64;
    const int64_t list_stride = (int64_t)n_units_max;   // two flat lists: straight-line units, the others
    const int nf_slots = (int)egrid.x * 64;
    const size_t o_cph = ((size_t)nt * sizeof(ColMeta) + 255) / 256 * 256, o_rp = 2 * o_cph;
    const size_t o_rph = o_rp + ((size_t)nf_slots * sizeof(RowPack) + 255) / 256 * 256;
    const size_t o_rf = o_rph + ((size_t)nf_slots * sizeof(RowPack) + 255) / 256 * 256, o_rt = o_rf + ((size_t)nf * 4 + 255) / 256 * 256;
    const size_t o_pairs = 256;
    const bool use_pairs = E.scr_mode == 1 && E.do_lr;   // verify mode keeps whole units: it must see the dismissed ones
    // units are evaluated from EXACT sums: the 5-limb GEMM of the tiles they live in (all tiles when any unit can be listed)
    const bool need_exact = E.any_sr || !use_pairs || lo_h->band_full;
    // (an SR sub-pass only ever lists units that hold a short-range pair: the band's tiles are all the exact GEMM has to cover)
    const uint8_t *band = ((use_pairs || lo_h->sr_sub) && !lo_h->band_full) ? D.band_mask : nullptr;
    static const bool band_early_env = exp_env("LDW_BAND_LATE") == nullptr;
    // r04: the exact band GEMM in phase 1, on the GEMM stream (an SR sub-pass runs both phases on one stream anyway) — for alignments of at
    // least 4096 sequences, where the GEMMs are long: see screen_main below
    const int64_t swap_kw = [] { const char *e = getenv("LDW_QUEUE_SWAP_KW"); return e ? (int64_t)atol(e) : (int64_t)64; }();   // (per call: the tests lower it)
    const bool band_early = band_early_env && !lo_h->sr_sub && c->KW >= swap_kw;
    ldw::DevBuf &Gx = gx(c, s);
    if (phase == 1) {
        if (int rc = c->panel[s][0].reserve((size_t)RFpad * c->KW * 8)) return rc;
        if (!lo_h->diag)
            if (int rc = c->panel[s][1].reserve((size_t)RTpad * c->KW * 8)) return rc;
        if (E.do_lr)
            if (int rc = c->Gapx[s].reserve((size_t)RFpad * RTpad * 4)) return rc;
        if (int rc = B_units.reserve(o_flat + 2 * n_units_max * 8 + 64)) return rc;
        if (int rc = B_packs.reserve(o_rt + (size_t)nt * 4 + 256)) return rc;
        if (use_pairs)
            if (int rc = c->pairs[s].reserve(o_pairs + (size_t)PAIR_PATHS * PAIR_SHARDS * pair_cap_for(nf, nt, lo_h->span) * sizeof(PairEnt) +
                                             (size_t)maybe_cap_for(RTpad, RFpad) * sizeof(ApxMaybe) + 64))
                return rc;
        if (need_exact)
            if (int rc = Gx.reserve((size_t)RFpad * RTpad * 8)) return rc;
        if (int rc = B_bins.reserve(2 * ((size_t)RTpad + (size_t)RFpad) + (size_t)nt + (size_t)nf_slots + 256 + (size_t)(RTpad / 128) * (size_t)(RFpad / 64) * 4)) return rc;
        if (int rc = c->apx_clean[s].reserve((size_t)(RTpad / 32) * (size_t)(RFpad / 64) + 64)) return rc;
        if (!c->apx_skip.p) {
            if (int rc = c->apx_skip.reserve(64)) return rc;
            LDW_HIP(hipMemsetAsync(c->apx_skip.p, 0, 64, gs));
        }
    }
    if (phase == 1) {
        if (E.do_lr && c->tab11_on) {
            // threshold table of the biallelic pairs, one per block kind (diagonal blocks sit ~8 % below the others): valid for every
            // block whose level is at least the level it was built for — a higher level only widens the true interval, i.e. the
            // table gets looser, and how tight it is decides how many regions the GEMM's epilogue finds clean — so it is rebuilt as
            // soon as the block's level has left [tab_lo, 1.05 tab_lo] (one bucket = 0.5 %).  Phase 1 of every block runs in order
            // on ONE stream, so the rebuild cannot overtake a reader.
            const int kd = lo_h->diag ? 1 : 0;
            const double lo_blk = E.spec_lo - (double)E.scr_eps;
            if (!(c->tab11_lo[kd] > 0) || lo_blk < c->tab11_lo[kd] || lo_blk > 1.05 * c->tab11_lo[kd]) {
                constexpr int NBINS_T = 64;
                if (int rc = c->tab11[kd].reserve((size_t)4 * NBINS_T * NBINS_T * 8)) return rc;   // a ring of four tables
                c->tab11_cur[kd] = (c->tab11_cur[kd] + 1) & 3;   // (items in flight — at most the two before this one — keep the tables they were screened for)
                const double W = std::ldexp((double)c->total_fixed, -c->frac_bits);
                c->tab11_lo[kd] = 0.98 * lo_blk;   // buckets are 0.5 % wide: the next blocks of the kind may guess four buckets lower (six higher) without a rebuild (73 us)
                c->tab11_c = (float)(NBINS_T / std::sqrt(W + 1.0));
                c->tab11_nb = NBINS_T;
                const double sprime = std::ldexp(1.0, c->apx_e_last - c->frac_bits);
                hipLaunchKernelGGL(k_build_tab11, dim3(NBINS_T * NBINS_T * 32 / 256), dim3(256), 0, gs, W, c->tab11_lo[kd], c->apx_delta * 1.001,
                                   (c->apx_lost_units + 1.0) * sprime, sprime, NBINS_T, c->tab11_c, c->tab11[kd].as<int2>() + (size_t)c->tab11_cur[kd] * NBINS_T * NBINS_T);
                LDW_HIP(hipGetLastError());
                ++c->tab11_builds;
            }
        }
    }
    EpiArgs A;
    fill_epi_args(c, D, nf, nt, RFpad, quirk, E, reinterpret_cast<const int64_t *>(c->Gapx[s].p), A);
    A.lo.slot_pfix_hi = c->slot_papx.as<int64_t>();   // the screen derives its cells from the marginals of the approximate weights
    A.sr_excl = lo_h->sr_excl;
    if (lo_h->span) {
        LDW_REQUIRE(E.scr_mode == 1 && E.do_lr && !E.any_sr && !lo_h->band_full && lo_h->fuse_ok && lo_h->sseg, LDW_ERR_STATE, "a span needs long-range-only blocks and pair lists");
        A.span = lo_h->span;
        for (int k = 0; k < lo_h->span; ++k) A.sseg[k] = lo_h->sseg[k];
    }
    const int kd_tab = lo_h->diag ? 1 : 0;
    // (the table is built for RXY = 1, the floor of the reference's scrambled RXY = r r' / 4 as long as no SNP has r < 2)
    if (phase == 1 || !lo_h->tab_set) {
        lo_h->tab_p = nullptr;
        if (E.do_lr && c->tab11_on && c->tab11[kd_tab].p && c->tab11_lo[kd_tab] > 0 && E.spec_lo - (double)E.scr_eps >= c->tab11_lo[kd_tab] &&
            (quirk != LDW_QUIRK_REFERENCE || c->r_min >= 2.0))
            lo_h->tab_p = c->tab11[kd_tab].as<int2>() + (size_t)c->tab11_cur[kd_tab] * 64 * 64;
        lo_h->tab_set = true;
    }
    if (lo_h->tab_p) {   // (phase 2: the table of phase 1, whatever later items have built since)
        A.tab11 = lo_h->tab_p;
        A.tab_nb = c->tab11_nb;
        A.tab_c = c->tab11_c;
    }
    // long-range-only blocks: the GEMM applies the table itself and neither stores nor lets the screen read the regions that pass
    static const bool fuse_on = exp_env("LDW_NO_FUSE_TAB") == nullptr;
    const bool fuse = fuse_on && lo_h->fuse_ok && A.tab11 && A.tab_nb == 64 && (use_pairs || c->screen == 2) && E.do_lr && (!E.any_sr || (D.band_mask && !lo_h->band_full)) && RFpad % 64 == 0 &&   // (verify mode: the clean regions' units are listed as dismissed and checked in fp64)
                      2048 + (size_t)(c->KW / 2) * 256 + 64 * 64 * 8 + 1024 <= 65536;   // (the table shares the GEMM's LDS with the digit arrays)
    uint8_t *bin_t = B_bins.as<uint8_t>(), *bin_f = bin_t + RTpad;
    // pruning flags by row (zeroed per block: padding rows) and by epilogue slot
    // ... and, zeroed with the row flags, the count of the wave tiles the pruning leaves (k_apx_live_tiles), whose list ends the buffer
    uint8_t *rflag_t = bin_f + RFpad, *rflag_f = rflag_t + RTpad;
    unsigned int *n_live = reinterpret_cast<unsigned int *>(rflag_f + RFpad);
    uint8_t *sflag_t = rflag_f + RFpad + 16, *sflag_f = sflag_t + ((size_t)nt + 15) / 16 * 16;
    uint32_t *tile_list = reinterpret_cast<uint32_t *>(sflag_f + ((size_t)nf_slots + 15) / 16 * 16);
    const bool wide_prune = c->prune && c->snp_sup.p && E.do_lr && (use_pairs || c->screen == 2);
    if (wide_prune) {
        A.snp_sup = c->snp_sup.as<double>();
        A.sflag_f = sflag_f;
        A.sflag_t = sflag_t;
    }
    if (fuse) {
        A.clean = c->apx_clean[s].as<uint8_t>();
        A.clean_stride = RFpad / 64;
    }
    char *ub = B_units.as<char>();
    unsigned int *n_units = reinterpret_cast<unsigned int *>(ub);
    uint64_t *units = reinterpret_cast<uint64_t *>(ub + o_flat);
    char *pb = B_packs.as<char>();
    ColMeta *cp = reinterpret_cast<ColMeta *>(pb), *cph = reinterpret_cast<ColMeta *>(pb + o_cph);
    RowPack *rp = reinterpret_cast<RowPack *>(pb + o_rp), *rph = reinterpret_cast<RowPack *>(pb + o_rph);
    float *rlf = reinterpret_cast<float *>(pb + o_rf), *rlt = reinterpret_cast<float *>(pb + o_rt);
    if (use_pairs) {
        A.pl_n = c->pairs[s].as<unsigned int>();
        A.pl_pairs = reinterpret_cast<PairEnt *>(c->pairs[s].as<char>() + o_pairs);
        A.pl_cap = pair_cap_for(nf, nt, lo_h->span);
    }
    // r04: table-eligible regions with a few failing entries hand those entries over instead of being stored and screened (k_screen_maybe);
    // the counter lives in the zeroed header of the pair lists, the entries behind the lists
    const bool maybe_on = getenv("LDW_NO_MAYBE") == nullptr;   // (read per call: the tests switch it)
    // (only where the K loop is long enough to carry the epilogue's extra work — the mask of the failing entries is built for every eligible
    // region: at N = 616 the GEMM's launch went from 0.091 to 0.107 ms and the pass from 19.4 to 20.1 ms with it, at N = 5000 the launch does
    // not move and the pass gains 0.3-0.4 ms; building the mask only in failing regions was slower at both sizes)
    const bool use_maybe = maybe_on && !c->maybe_off && fuse && use_pairs && c->screen == 1 && !E.lower_only && c->KW >= 32;
    unsigned int *maybe_n = use_maybe ? c->pairs[s].as<unsigned int>() + 48 : nullptr;
    const unsigned int maybe_cap = maybe_cap_for(RTpad, RFpad);
    ApxMaybe *maybe_list = use_maybe ? reinterpret_cast<ApxMaybe *>(c->pairs[s].as<char>() + o_pairs + (size_t)PAIR_PATHS * PAIR_SHARDS * A.pl_cap * sizeof(PairEnt)) : nullptr;
    if (phase == 1) {
        if (E.do_lr) {
            if (int rc = launch_pack_panel(c, D.rl_f, RFpad, c->panel[s][0].as<uint64_t>(), gs, lo_h->diag ? nullptr : D.rl_t, RTpad,
                                           lo_h->diag ? nullptr : c->panel[s][1].as<uint64_t>()))
                return rc;
        }
        {   // the unit counters, the pair-list counters, this slot's histogram and pick record (submit_b skips its own memsets)
            ZeroArgs Z;
            memset(&Z, 0, sizeof(Z));
            Z.p[0] = reinterpret_cast<uint4 *>(ub);
            Z.n16[0] = (unsigned int)((o_flat + 15) / 16);
            if (use_pairs) {
                Z.p[1] = reinterpret_cast<uint4 *>(c->pairs[s].p);
                Z.n16[1] = (unsigned int)((o_pairs + 15) / 16);
            }
            if (fuse && E.lower_only) {   // diagonal block: the GEMM skips the tiles above the diagonal, whose regions must read "not clean"
                Z.p[4] = reinterpret_cast<uint4 *>(c->apx_clean[s].p);
                Z.n16[4] = (unsigned int)(((size_t)(RTpad / 32) * (size_t)(RFpad / 64) + 15) / 16);
            }
            Z.p[5] = reinterpret_cast<uint4 *>(rflag_t);
            Z.n16[5] = (unsigned int)(((size_t)RTpad + (size_t)RFpad) / 16 + 1);
            if (zero_hist) {
                Z.p[2] = reinterpret_cast<uint4 *>(zero_hist);
                Z.n16[2] = (unsigned int)((size_t)NBINS * 8 / 16 * (size_t)(lo_h->span ? lo_h->span : 1));   // (a span: one histogram per reference block)
                Z.p[3] = reinterpret_cast<uint4 *>(zero_pick);
                Z.n16[3] = (unsigned int)(zero_pick_bytes / 16);
            }
            hipLaunchKernelGGL(k_zero4, dim3(16), dim3(256), 0, gs, Z);
        }
        // the per-SNP constants in epilogue order — before the GEMM: its epilogue reads the table bins by row
        const int nthr = std::max<int>(std::max<int>((int)nt, nf_slots), std::max<int>(RTpad, RFpad));
        hipLaunchKernelGGL(k_build_packs, dim3((unsigned)((nthr + 255) / 256), 2), dim3(256), 0, gs, A, D.perm, D.perm_t, nf_slots, 1, cp, cph, rp, rph,
                           rlf, rlt, bin_t, bin_f, RTpad, RFpad, wide_prune ? sflag_f : nullptr, wide_prune ? sflag_t : nullptr, wide_prune ? rflag_f : nullptr,
                           wide_prune ? rflag_t : nullptr);
        LDW_HIP(hipGetLastError());
        ApxGemmArgs P;
        if (E.do_lr) {
            memset(&P, 0, sizeof(P));
            P.panel_f = c->panel[s][0].as<uint64_t>();
            P.panel_t = lo_h->diag ? P.panel_f : c->panel[s][1].as<uint64_t>();
            P.RTpad = RTpad;
            P.RFpad = RFpad;
            P.M2 = (int)(c->KW / 2);
            P.dig_a = c->dig_a.as<uint8_t>();
            P.dig_b = c->dig_b.as<uint8_t>();
            P.shift = c->apx_shift.as<int32_t>();
            P.fine = c->apx_fine ? 1 : 0;
            P.G = c->Gapx[s].as<int32_t>();
            P.lower_only = E.lower_only;
            if (fuse) {
                P.fuse = 1;
                static const bool std_kernel = getenv("LDW_APX_KERNEL") == nullptr && getenv("LDW_APX_TILE") == nullptr;   // (the experimental GEMM variants know no tile list)
                P.skip_ctr = (c->prune && lo_h->ordered && std_kernel) ? c->apx_skip.as<unsigned long long>() : nullptr;   // (list order: a tile spans every bin)
                if (P.skip_ctr) {
                    P.tile_list = tile_list;
                    P.n_live = n_live;
                }
                if (P.skip_ctr && wide_prune) {
                    P.rflag_t = rflag_t;
                    P.rflag_f = rflag_f;
                }
                P.bin_t = bin_t;
                P.bin_f = bin_f;
                P.tab = A.tab11;
                P.tab_nb = A.tab_nb;
                P.clean = c->apx_clean[s].as<uint8_t>();
                P.sr_mask = E.any_sr ? D.band_mask : nullptr;   // a block with a short-range corner: its band tiles stay with the screen
                if (use_maybe) {
                    P.maybe = maybe_list;
                    P.maybe_n = maybe_n;
                    P.maybe_cap = maybe_cap;
                }
            }
            if (P.skip_ctr)
                if (int rc = launch_apx_live_tiles(c, P, gs)) return rc;
        }
        LDW_HIP(hipEventRecord(ev[0], gs));   // ev[0] .. ev[1]: the approximate GEMM alone (its launch time is the roofline's)
        if (E.do_lr)
            if (int rc = launch_gemm_apx(c, P, gs)) return rc;
        LDW_HIP(hipEventRecord(ev[1], gs));
    }
    A.rloc_f = rlf;
    A.rloc_t = rlt;
    A.colpack = cp;
    A.colpack_hi = cph;
    A.rowpack = rp;
    A.rowpack_hi = rph;
    // LDW_SCREEN_MAIN=1 (experiment): the block's screen at the head of phase 2 on the main stream — beside the NEXT block's GEMM on the GEMM
    // stream — instead of behind its own GEMM
    // (r04, first attempt: a span's screen at the head of phase 2 went WRONG — phase 2 is queued after the first phase of LATER items, which may
    // rebuild the threshold table for their level; the screen then either found no table it might use and, with it, no clean flags — it read the
    // regions the GEMM's epilogue had not stored: 20-60 million pairs listed per span at C5, overflowing lists, 43 segments redone per pass — or
    // raced the rebuild.  Cured by LoHost::tab_p: every item keeps the table its first phase chose, out of a ring of four per kind.)
    // r04 (end of the round): with the table a per-item snapshot out of a ring (above) the screen may run at the head of phase 2 on the main
    // stream for every item, and the exact band GEMM moves the other way, into phase 1 on the GEMM stream: one queue holds the MFMA kernels, the
    // other the VALU- and latency-bound ones, which share the CUs better than two GEMMs do (C4, alternating runs: 33.6 / 33.2 against 34.2 / 34.2
    // ms; the band GEMM alone on the GEMM stream: 35.3 / 34.9; three more alternating pairs at the end: 33.6 / 33.2 / 33.4 against 34.0 / 33.8 /
    // 34.0; C5 819 against 824 ms, 3 misses per pass either way).  Only where the GEMMs are long (N >= 4096): at 85k x 616 the approximate GEMM
    // is 2.7 ms of a 19-ms pass, the GEMM queue would idle and the main queue carry everything — 20.9 / 21.1 against 19.3 / 19.1 ms — so short
    // alignments keep the old places.  LDW_SCREEN_GEMMQ=1 / LDW_BAND_LATE=1 restore them for any size.
    static const bool screen_main_env = exp_env("LDW_SCREEN_GEMMQ") == nullptr;
    const bool screen_main = screen_main_env && !lo_h->sr_sub && c->KW >= swap_kw;
    const int rm_s = quirk == LDW_QUIRK_REFERENCE ? (lo_h->span ? 3 : (nf == nt ? 1 : 2)) : 0;
#ifdef LDW_EXPERIMENTS
    // r04 experiment: list-driven screen (k_screen_tiles -> k_screen_live -> k_mi_screen_list) instead of one workgroup per (tile, column group)
    // (measured r04, C4, 10 cold steps per setting on one box: full grid 36.9 ms per pass, list-driven with 1536 / 4096 / 12288 / 32768 striding
    // workgroups 38.2 / 37.0 / 36.0 / 37.0: no gain — the dispatcher balances 86k short workgroups better than a strided list does, and the
    // two list kernels cost what the empty workgroups did; kept behind LDW_SCREEN_LIST=1 in the experiments build)
    static const bool screen_list = exp_env("LDW_SCREEN_LIST") != nullptr;
    const size_t o_ts = 64, o_live = o_ts + ((size_t)egrid.x * sizeof(TileState) + 63) / 64 * 64;
    if (screen_list)
        if (int rc = B_live.reserve(o_live + (size_t)egrid.x * egrid.y * 4 + 64)) return rc;
    unsigned int *n_live_scr = screen_list ? B_live.as<unsigned int>() : nullptr;
    TileState *ts_scr = screen_list ? reinterpret_cast<TileState *>(B_live.as<char>() + o_ts) : nullptr;
    uint32_t *live_scr = screen_list ? reinterpret_cast<uint32_t *>(B_live.as<char>() + o_live) : nullptr;
    static const size_t lgrid_max = [] { const char *e = exp_env("LDW_SCREEN_GRID"); return e ? (size_t)atol(e) : (size_t)1536; }();
    const unsigned lgrid = (unsigned)std::min<size_t>((size_t)egrid.x * egrid.y, lgrid_max);
#define LDW_SCREEN(RMv, ST)                                                                                                                       \
    do {                                                                                                                                          \
        if (screen_list) {                                                                                                                        \
            hipLaunchKernelGGL(k_screen_tiles, dim3(egrid.x), dim3(64), 0, ST, A, D.perm, (int)egrid.x, ts_scr, n_live_scr);                        \
            hipLaunchKernelGGL(k_screen_live<true>, dim3(egrid.y), dim3(256), 0, ST, A, D.perm_t, ts_scr, (int)egrid.x, live_scr, n_live_scr);    \
            hipLaunchKernelGGL((k_mi_screen_list<RMv, true>), dim3(lgrid), dim3(256), 0, ST, A, D.perm, D.perm_t, units, n_units, list_stride,    \
                               live_scr, n_live_scr);                                                                                             \
        } else {                                                                                                                                  \
            hipLaunchKernelGGL((k_mi_screen<RMv, true>), egrid, dim3(256), 0, ST, A, D.perm, D.perm_t, units, n_units, list_stride);                \
        }                                                                                                                                         \
    } while (0)
#else
    (void)B_live;
#define LDW_SCREEN(RMv, ST) hipLaunchKernelGGL((k_mi_screen<RMv, true>), egrid, dim3(256), 0, ST, A, D.perm, D.perm_t, units, n_units, list_stride)
#endif
    if (phase == 1) {
        if (!screen_main) {
            if (rm_s == 0) LDW_SCREEN(0, gs);
            else if (rm_s == 1) LDW_SCREEN(1, gs);
            else if (rm_s == 3) LDW_SCREEN(3, gs);
            else LDW_SCREEN(2, gs);
            LDW_HIP(hipGetLastError());
        }
        if (band_early && need_exact)
            if (int rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, D.rl_t, RTpad, D.rl_f, RFpad, Gx.as<int64_t>(), c->nlimbs, c->digits.as<int8_t>(),
                                          E.lower_only, gs, 0, -1, band))
                return rc;
        if (use_maybe && E.do_lr) {
            if (rm_s == 0) hipLaunchKernelGGL((k_screen_maybe<0>), dim3(512), dim3(256), 0, gs, A, maybe_list, maybe_n, maybe_cap);
            else if (rm_s == 1) hipLaunchKernelGGL((k_screen_maybe<1>), dim3(512), dim3(256), 0, gs, A, maybe_list, maybe_n, maybe_cap);
            else if (rm_s == 3) hipLaunchKernelGGL((k_screen_maybe<3>), dim3(512), dim3(256), 0, gs, A, maybe_list, maybe_n, maybe_cap);
            else hipLaunchKernelGGL((k_screen_maybe<2>), dim3(512), dim3(256), 0, gs, A, maybe_list, maybe_n, maybe_cap);
            LDW_HIP(hipGetLastError());
        }
        LDW_HIP(hipEventRecord(ev[5], gs));
        return LDW_OK;
    }
    // ---- phase 2 ----
    LDW_HIP(hipEventRecord(ev[4], s2));
    if (screen_main) {
        if (rm_s == 0) LDW_SCREEN(0, s2);
        else if (rm_s == 1) LDW_SCREEN(1, s2);
        else if (rm_s == 3) LDW_SCREEN(3, s2);
        else LDW_SCREEN(2, s2);
        LDW_HIP(hipGetLastError());
    }
#undef LDW_SCREEN
    // the units outside k_mi_screen's domain: generic from-tiles x all columns, the other tiles x the generic columns
    const int gt0 = std::min<int>(A.gen_t0, (int)egrid.x);
    const int q0 = std::min<int>(A.gen_q0, (int)nt) / GEN_COLS * GEN_COLS;
    {
        GenRegions Rg;
        Rg.tile0_a = gt0;
        Rg.nt_a = (int)egrid.x - gt0;
        Rg.ncg_a = Rg.nt_a > 0 ? (int)((nt + GEN_COLS - 1) / GEN_COLS) : 0;
        Rg.nt_b = (gt0 > 0 && A.gen_q0 < (int)nt) ? gt0 : 0;
        Rg.q0_b = q0;
        Rg.ncg_b = Rg.nt_b > 0 ? (int)((nt - q0 + GEN_COLS - 1) / GEN_COLS) : 0;
        const long long nblk = (long long)Rg.nt_a * Rg.ncg_a + (long long)Rg.nt_b * Rg.ncg_b;
        if (nblk > 0)
            hipLaunchKernelGGL(k_mi_screen_generic<true>, dim3((unsigned)nblk), dim3(256), 0, s2, A, D.perm, D.perm_t, units, n_units, list_stride, Rg);
    }
    LDW_HIP(hipGetLastError());
    if (need_exact && !band_early)
        if (int rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, D.rl_t, RTpad, D.rl_f, RFpad, Gx.as<int64_t>(), c->nlimbs, c->digits.as<int8_t>(),
                                      E.lower_only, s2, 0, -1, band))
            return rc;
    if (use_pairs) {
        if (int rc = c->pair_sums.reserve((size_t)PAIR_PATHS * PAIR_SHARDS * A.pl_cap * 16 * 8)) return rc;
        if (int rc = launch_pairs_exact(c, A, ghist, c->pair_sums.as<int64_t>(), s2)) return rc;
    }
    if (need_exact) {   // the listed units, from the exact tiles
        EpiArgs Ax = A;
        Ax.G = Gx.as<int64_t>();
        Ax.E.apx = 0;
        Ax.pl_pairs = nullptr;
        UnitLists UL;
        memset(&UL, 0, sizeof(UL));
        UL.units[0] = units;
        UL.n[0] = n_units;
        hipLaunchKernelGGL(k_mi_units<true>, dim3(2048, 1), dim3(256), 0, s2, Ax, D.perm, D.perm_t, UL, ghist);
        UnitLists UG;
        memset(&UG, 0, sizeof(UG));
        UG.units[0] = units + list_stride;
        UG.n[0] = n_units + 1;
        hipLaunchKernelGGL(k_mi_units<false>, dim3(512, 1), dim3(256), 0, s2, Ax, D.perm, D.perm_t, UG, ghist);
    }
    hipLaunchKernelGGL(k_apx_stats, dim3(1), dim3(64), 0, s2, n_units, A.pl_n, A.pl_cap, A.E.scr_viol + 1);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipEventRecord(ev[2], s2));
    return LDW_OK;
}

// lane order of the epilogue: from-side SNPs grouped by their number of indicator rows (1, 2, 3, 4, 0)
// The same for the from side of the epilogue kernels, which walk it in tiles of 64 (one wave): every class is padded
// with -1 to a multiple of 64, so no tile mixes slot counts — a mixed tile cannot take the straight-line code and
// would have all of its units listed for the fp64 kernel (and for the gathered low-limb GEMM).
void build_perm_tiles(ldw_ctx *c, const int32_t *from_idx, int64_t nf, std::vector<int32_t> &perm, const std::vector<int32_t> *order1 = nullptr) {
    perm.clear();
    for (int want : {1, 2, 3, 4, 0}) {
        if (want == 1 && order1) perm.insert(perm.end(), order1->begin(), order1->end());   // (every one-row SNP, in row order: build_side)
        else
            for (int64_t k = 0; k < nf; ++k)
                if (c->h_row0[from_idx[k] + 1] - c->h_row0[from_idx[k]] == want) perm.push_back((int32_t)k);
        if (want != 3)   // 3 and 4 rows share the generic code anyway
            while (perm.size() % 64) perm.push_back(-1);
    }
    if (perm.empty()) perm.assign(64, -1);
}

void build_perm(ldw_ctx *c, const int32_t *from_idx, int64_t nf, int32_t *perm, const std::vector<int32_t> *order1 = nullptr) {
    int64_t w = 0;
    for (int want : {1, 2, 3, 4, 0}) {
        if (want == 1 && order1) {
            for (int32_t k : *order1) perm[w++] = k;
            continue;
        }
        for (int64_t k = 0; k < nf; ++k)
            if (c->h_row0[from_idx[k] + 1] - c->h_row0[from_idx[k]] == want) perm[w++] = (int32_t)k;
    }
}

// The one-row SNPs of a list in ascending order of h_minor_w (ties in list order), cached per list: the all-pairs loop presents
// the same ten or fifty lists over and over, and a sort of 10^4 keys costs as much host time as the rest of prep_block.
// Entries are heap objects and are never removed while the row map lives (ensure_rows clears the cache): the pointers handed out stay
// valid whatever another thread adds; a full cache (256 lists) computes into the caller's own vector instead.
const std::vector<int32_t> *minor_weight_order(ldw_ctx *c, const int32_t *idx, int64_t n, std::vector<int32_t> &own) {
    {
        std::lock_guard<std::mutex> lk(c->order_mtx);
        for (auto &pe : c->order_cache)
            if ((int64_t)pe->idx.size() == n && pe->idx[0] == idx[0] && memcmp(pe->idx.data(), idx, (size_t)n * 4) == 0) return &pe->order;
    }
    std::vector<std::pair<int64_t, int32_t>> key;
    key.reserve((size_t)n);
    for (int64_t k = 0; k < n; ++k)
        if (c->h_row0[idx[k] + 1] - c->h_row0[idx[k]] == 1) key.emplace_back(c->h_minor_w[(size_t)idx[k]], (int32_t)k);
    std::sort(key.begin(), key.end());
    own.resize(key.size());
    for (size_t i = 0; i < key.size(); ++i) own[i] = key[i].second;
    std::lock_guard<std::mutex> lk(c->order_mtx);
    if (c->order_cache.size() >= 256) return &own;
    c->order_cache.emplace_back(new ldw_ctx::OrderCache());
    auto &e = *c->order_cache.back();
    e.idx.assign(idx, idx + n);
    e.order = own;
    return &e.order;
}

// dense MI of one block, synchronous staging through ctx-owned buffers (ldw_mi_block, ldw_joint_tables style)
int run_block_mi(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, int quirk,
                 EmitArgs E) {
    if (int rc = join_prepare(c)) return rc;
    if (int rc = ensure_rows(c)) return rc;
    LDW_REQUIRE(nf > 0 && nt > 0, LDW_ERR_ARG, "empty block (nf=%lld nt=%lld)", (long long)nf, (long long)nt);
    LDW_REQUIRE(nf <= 1000000 && nt <= 1000000, LDW_ERR_ARG, "block side too long");
    std::vector<int32_t> vf(from_idx, from_idx + nf), vt(to_idx, to_idx + nt), perm, perm_t((size_t)nt);
    SideLists SF, ST;
    if (int rc = build_side(c, from_idx, nf, SF)) return rc;
    if (int rc = build_side(c, to_idx, nt, ST)) return rc;
    build_perm_tiles(c, from_idx, nf, perm);
    build_perm(c, to_idx, nt, perm_t.data());
    if (int rc = upload_i32(c, c->perm_t, perm_t)) return rc;
    if (int rc = upload_i32(c, c->idx_f, vf)) return rc;
    if (int rc = upload_i32(c, c->idx_t, vt)) return rc;
    if (int rc = upload_i32(c, c->rowlist_f, SF.rowlist)) return rc;
    if (int rc = upload_i32(c, c->rowlist_t, ST.rowlist)) return rc;
    if (int rc = upload_i32(c, c->lrow_f, SF.lrow)) return rc;
    if (int rc = upload_i32(c, c->lrow_t, ST.lrow)) return rc;
    if (int rc = upload_i32(c, c->perm_f, perm)) return rc;
    if (int rc = c->hist[0].reserve((size_t)NBINS * 8)) return rc;
    LDW_HIP(hipStreamSynchronize(c->stream));  // pageable H2D copies are complete only after a sync
    DevPtrs D{c->idx_f.as<int32_t>(), c->idx_t.as<int32_t>(), c->rowlist_f.as<int32_t>(), c->rowlist_t.as<int32_t>(),
              c->lrow_f.as<int32_t>(), c->lrow_t.as<int32_t>(), c->perm_f.as<int32_t>(), c->perm_t.as<int32_t>(), nullptr, nullptr,
              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (int)(perm.size() / 64), 0x7FFFFFFF, 0x7FFFFFFF};
    E.write_dense = 1;
    E.spec_B = -1;
    // a symmetric block (same list on both sides) may be asked for in full: the GEMM then computes every tile
    return launch_block_mi(c, D, nf, nt, SF.Rpad, ST.Rpad, quirk, E, c->ev, 3, nullptr, nullptr, c->hist[0].as<unsigned long long>());
}

int ensure_links_capacity(ldw_ctx *c, int64_t sr_rows, int64_t lr_rows) {
    // the fused kernels of earlier blocks write the short-range table on the GEMM stream: a reallocation (rare: the
    // all-pairs driver sizes the table up front) waits for them and is complete before anything else is queued
    const bool grow_sr = (size_t)sr_rows * 8 > c->sr_mi.cap || (size_t)sr_rows * 4 > c->sr_a.cap || (size_t)sr_rows * 4 > c->sr_b.cap;
    if (grow_sr && c->gemm_stream) LDW_HIP(hipStreamSynchronize(c->gemm_stream));
    if (int rc = c->sr_a.reserve_keep((size_t)sr_rows * 4, (size_t)c->n_sr * 4, c->stream)) return rc;
    if (int rc = c->sr_b.reserve_keep((size_t)sr_rows * 4, (size_t)c->n_sr * 4, c->stream)) return rc;
    if (int rc = c->sr_mi.reserve_keep((size_t)sr_rows * 8, (size_t)c->n_sr * 8, c->stream)) return rc;
    if (grow_sr) LDW_HIP(hipStreamSynchronize(c->stream));
    if (c->lr_stream && ((size_t)lr_rows * 8 > c->lr_mi.cap || (size_t)lr_rows * 4 > c->lr_a.cap || (size_t)lr_rows * 4 > c->lr_b.cap))
        lr_stream_drain(c);   // (the writer thread reads these buffers on its own stream: everything handed over is on disk before they move)
    if (int rc = c->lr_a.reserve_keep((size_t)lr_rows * 4, (size_t)c->n_lr * 4, c->stream)) return rc;
    if (int rc = c->lr_b.reserve_keep((size_t)lr_rows * 4, (size_t)c->n_lr * 4, c->stream)) return rc;
    if (int rc = c->lr_mi.reserve_keep((size_t)lr_rows * 8, (size_t)c->n_lr * 8, c->stream)) return rc;
    return LDW_OK;
}

// layout of ctx->small during link selection
struct SmallLayout {
    int64_t *lr_count;   // running number of kept long-range rows (device side)
    ldw::PickOut *pick[LDW_NSLOT];   // one per pipeline slot
    int64_t *stats_i;    // [capacity][3]
    double *stats_d;     // [capacity]
};

constexpr size_t PICK_STRIDE = ((sizeof(ldw::PickOut) + 63) / 64) * 64;

void links_layout(ldw_ctx *c, SmallLayout &sl) {
    char *base = c->small.as<char>();
    sl.lr_count = reinterpret_cast<int64_t *>(base);
    // (every slot has room for the pick records of a span's segments: segment k of slot s at sl.pick[s] + k * PICK_STRIDE bytes)
    for (int k = 0; k < LDW_NSLOT; ++k) sl.pick[k] = reinterpret_cast<ldw::PickOut *>(base + 64 + (size_t)k * LDW_SPAN_MAX * PICK_STRIDE);
    sl.stats_i = reinterpret_cast<int64_t *>(base + 64 + (size_t)LDW_NSLOT * LDW_SPAN_MAX * PICK_STRIDE);
    sl.stats_d = reinterpret_cast<double *>(sl.stats_i + c->blk_capacity * 3);
}

// ---- one block of the link loop in phases so that the host work of block i+1 and (fused path) its whole kernel
// ---- overlap the selection of block i:
// ----   prep     pure host, into pinned memory
// ----   submit_a upload; unfused: the GEMM on the GEMM stream | fused: GEMM + epilogue + bucket pick + copy-back of the pick
// ----   submit_b unfused: epilogue + bucket pick + copy-back on the main stream | fused: nothing
// ----   finish   the one host round trip (candidate count), then sorts / threshold / append on the main stream
struct HostBlock {
    int64_t nf = 0, nt = 0, n_sr_blk = 0, n_lr_total = 0, blk_no = 0;
    int RFpad = 0, RTpad = 0, slot = 0;
    bool diag = false, fused = false, submitted = false;
    int nf_tiles = 0;          // tiles of 64 in the padded from-side order
    int gen_t0 = 0, gen_q0 = 0;
    bool mixed = false;        // high-limb GEMM + gathered low limbs (decided with the bucket guess at submit_a)
    bool apx = false;          // approximate GEMM + exact popcount sums of the listed units (ldw_apx.h); implies the lo geometry
    bool generic = false;      // a SNP list of the block is not ascending in POS: plain path + the k_gen_* pair list
    int guess = -1;            // bucket guess the block was submitted with
    LoHost lo;
    size_t o_cmax = 0, o_tbase = 0, o_tf = 0, o_band = 0;
    size_t o_idx_f = 0, o_idx_t = 0, o_rl_f = 0, o_rl_t = 0, o_lrow_f = 0, o_lrow_t = 0, o_perm = 0, o_perm_t = 0, o_cols = 0, o_pos_f = 0,
           o_pos_t = 0, o_cls_f = 0, o_cls_t = 0, total = 0;
    DevPtrs D{};
    EmitArgs E{};
    int spec_B = -1;
    // r04 span (ldw_epi.h): the to side is the concatenation of `span` reference blocks (0: an ordinary block)
    int span = 0;
    int32_t seg_start[LDW_SPAN_MAX] = {}, seg_nt[LDW_SPAN_MAX] = {};
    int64_t seg_lr_total[LDW_SPAN_MAX] = {};
    size_t cand_cap = 0;               // entries of every segment's candidate list
    SpanSeg sseg[LDW_SPAN_MAX] = {};
    int pin_slot = -1;                 // staging buffer the lists were built in (-1: the slot's own)
    bool force_plain = false;          // never speculate: a span's segment that is redone after a wrong guess
    bool span_alone = false;           // the span could not be submitted as one (no positive guess): its blocks run one by one in finish_span
    // r04b: short-range pairs of a span's CORNER segments (the neighbouring block pair of the row, the pair that closes the circle) are evaluated by an SR
    // sub-pass over that block alone, in list order (band GEMM + whole units: the r03 machinery in its SR-only form), queued in front of the span's
    // own kernels; the span treats the segment as long-range-only and keeps the short-range pairs out of its candidates (EpiArgs::sr_excl)
    bool sr_sub = false;               // this HostBlock IS such a sub-pass (list order, no ordering of its rows)
    bool lr_split = false;             // a single block (diagonal) whose short-range pairs go to an SR sub-pass: its own rows are ordered by weight like a
                                       // long-range-only block's, its screens keep the short-range pairs out (sr_excl), no band, no units
    size_t stage_base = 0;             // offset of its image in the item's staging buffer
    int64_t sr_base = 0;               // first row of the item's short-range rows (set when the item is submitted: submit order = block order)
    int64_t seg_n_sr[LDW_SPAN_MAX] = {};
    std::vector<HostBlock> subs;       // the SR sub-passes of a span (at most one per segment), subs_seg[i] = its segment
    std::vector<int> subs_seg;
    size_t stage_total = 0;            // bytes of the item's whole staging image (its own lists + those of its sub-passes); 0: total
    bool sr_base_fixed = false;        // sr_base was assigned by the caller (a span's segment that runs alone): do not touch the running row count
    std::vector<int32_t> span_from, span_to;   // the span's SNP lists (host): what a segment that runs on its own is prepared from
};

// the to side of a span: where each reference block starts in the concatenated list
struct SpanPlan {
    int nseg = 0;
    int32_t start[LDW_SPAN_MAX] = {}, nt[LDW_SPAN_MAX] = {};
};

// The speculative selection (bucket guess -> approximate GEMM / screen -> lists of the pairs that may pass) pays when the long-range filter
// keeps a small fraction of the pairs.  When lr_retain_links is a sizeable part of lr_links_approx — a small alignment with the default
// 1e6: 5000 SNPs keep 8 % — nearly every unit has to be listed and the lists cost more than evaluating everything: measured on 30k x 2k
// (tools/keep_frac_probe.py, warm passes, default against plain): 6.5 / 11.2 ms at 0.02 %, 21.2 / 21.6 at 0.5 %, 24.1 / 22.8 at 1 %, 84.7 / 24.9 at
// 5 %.  Above 0.7 % every block takes the plain path (5-limb GEMM, fp64 MI of every pair, full histogram), cold start included.
// Only the automatic choice is gated: ldw_set_fused(1) and ldw_set_path(1 | 2) are honoured as given.
static inline bool speculation_pays(const ldw_ctx *c, const ldw_mi_params *p) {
    if (c->fused || c->path_mode != 0) return true;
    return !(p->lr_links_approx > 0.0) || p->lr_retain_links < 0.007 * p->lr_links_approx;
}

int prep_block(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, const ldw_mi_params *p,
               int slot, int64_t blk_no, HostBlock &hb, const SpanPlan *sp = nullptr, int pin_slot = -1, bool sr_sub = false, size_t stage_base = 0,
               bool lr_split = false) {
    LDW_REQUIRE(nf > 0 && nt > 0, LDW_ERR_ARG, "empty block (nf=%lld nt=%lld)", (long long)nf, (long long)nt);
    LDW_REQUIRE(nf <= 1000000 && nt <= 1000000 && nf * nt < 2147483647LL, LDW_ERR_ARG, "block too large (%lld x %lld)",
                (long long)nf, (long long)nt);
    for (int64_t k = 0; k < nf; ++k)
        LDW_REQUIRE(from_idx[k] >= 0 && from_idx[k] < c->L, LDW_ERR_ARG, "SNP index %d out of range", from_idx[k]);
    for (int64_t k = 0; k < nt; ++k)
        LDW_REQUIRE(to_idx[k] >= 0 && to_idx[k] < c->L, LDW_ERR_ARG, "SNP index %d out of range", to_idx[k]);
    hb = HostBlock();
    hb.nf = nf;
    hb.nt = nt;
    hb.slot = slot;
    hb.blk_no = blk_no;
    hb.pin_slot = pin_slot;
    hb.sr_sub = sr_sub;
    hb.lr_split = lr_split;
    hb.stage_base = stage_base;
    hb.diag = same_list(from_idx, nf, to_idx, nt);
    SideLists SF, ST;
    std::vector<ColInfo> cols;
    auto ascending = [&](const int32_t *idx, int64_t n) {
        for (int64_t k = 1; k < n; ++k)
            if (c->h_POS[idx[k]] < c->h_POS[idx[k - 1]]) return false;
        return true;
    };
    hb.generic = !ascending(from_idx, nf) || !ascending(to_idx, nt);
    if (hb.generic) {   // no intervals: the pair list is made from the dense block with the predicate itself (submit_b)
        ColInfo z;
        memset(&z, 0, sizeof(z));
        cols.assign((size_t)nt, z);
        hb.n_sr_blk = 0;
    } else if (sp) {
        // a span: the intervals segment by segment (the long-range-only ones leave build_cols at its range test; one pass over the
        // concatenated list would take the slow path for all of them as soon as ONE segment is a corner block)
        cols.resize((size_t)nt);
        std::vector<ColInfo> ck;
        for (int k = 0; k < sp->nseg; ++k) {
            int64_t n_k = 0;
            if (int rc = build_cols(c, from_idx, nf, to_idx + sp->start[k], sp->nt[k], false, p->sr_dist, ck, n_k)) return rc;
            std::copy(ck.begin(), ck.end(), cols.begin() + sp->start[k]);
        }
        hb.n_sr_blk = 0;
    } else if (int rc = build_cols(c, from_idx, nf, to_idx, nt, hb.diag, p->sr_dist, cols, hb.n_sr_blk)) return rc;
    if (sp) {   // a span: every column learns its reference block and where that block starts in the concatenated to side
        LDW_REQUIRE(!hb.generic && !hb.diag && sp->nseg >= 1 && sp->nseg <= LDW_SPAN_MAX, LDW_ERR_STATE,
                    "span of %d blocks at block %lld cannot be formed", sp->nseg, (long long)blk_no);
        hb.n_sr_blk = 0;   // (the span itself emits no short-range row: its corner segments' pairs belong to their SR sub-passes; the intervals stay: sr_excl)
        hb.span = sp->nseg;
        for (int k = 0; k < sp->nseg; ++k) {
            hb.seg_start[k] = sp->start[k];
            hb.seg_nt[k] = sp->nt[k];
            hb.seg_lr_total[k] = nf * (int64_t)sp->nt[k] - std::min<int64_t>(nf, sp->nt[k]);   // (off-diagonal blocks drop their own diagonal: Q3)
            for (int32_t b = sp->start[k]; b < sp->start[k] + sp->nt[k]; ++b) {
                cols[(size_t)b].pad[0] = k;
                cols[(size_t)b].pad[1] = sp->start[k];
            }
        }
    }
    // Blocks without a short-range pair (most off-diagonal ones): nothing depends on the order of the rows within a class, so the
    // one-row SNPs are ordered by the weight of their minor state on both sides — rows and epilogue slots alike, which the table
    // test of the GEMM's epilogue requires anyway.  The wave tiles of the approximate GEMM then span few bins of the threshold
    // table and the tiles of the rare x rare corner are pruned whole (apx_tile_prunable).
    // An off-diagonal block WITH a short-range corner (neighbouring blocks, and the pair that closes the circle): the SNPs that have a
    // short-range partner in the block are a few hundred at the facing ends of the two lists.  They keep the list order — the band
    // of tiles the exact GEMM covers needs their partners contiguous — behind the ordered rest.  Diagonal blocks stay as they are:
    // every SNP has short-range partners there.
    const std::vector<int32_t> *ord_f = nullptr, *ord_t = nullptr;
    std::vector<int32_t> ord_f_own, ord_t_own, ord_f_full, ord_t_full;
    if (c->prune && !hb.generic && (!hb.diag || lr_split) && !sr_sub && c->engine == LDW_ENGINE_MFMA && c->apx_ok && !c->fused) {
        bool rowless = false;
        for (int64_t k = 0; k < nf && !rowless; ++k) rowless = c->h_row0[from_idx[k] + 1] == c->h_row0[from_idx[k]];
        for (int64_t k = 0; k < nt && !rowless; ++k) rowless = c->h_row0[to_idx[k] + 1] == c->h_row0[to_idx[k]];
        if (!rowless) {
            ord_f = minor_weight_order(c, from_idx, nf, ord_f_full);
            ord_t = minor_weight_order(c, to_idx, nt, ord_t_full);
            if (hb.n_sr_blk > 0 && !lr_split) {
                std::vector<int32_t> df((size_t)nf + 1, 0);
                std::vector<uint8_t> in_f((size_t)nf, 0), in_t((size_t)nt, 0);
                for (int64_t b = 0; b < nt; ++b)
                    for (int iv = 0; iv < 3; ++iv) {
                        const ColInfo &ci = cols[(size_t)b];
                        if (ci.e[iv] <= ci.s[iv]) continue;
                        in_t[(size_t)b] = 1;
                        ++df[(size_t)ci.s[iv]];
                        --df[(size_t)ci.e[iv]];
                    }
                int32_t run = 0;
                for (int64_t a = 0; a < nf; ++a) {
                    run += df[(size_t)a];
                    in_f[(size_t)a] = run > 0 ? 1 : 0;
                }
                auto rest_then_corner = [&](const std::vector<int32_t> &sorted, const std::vector<uint8_t> &corner, const int32_t *idx, int64_t n,
                                            std::vector<int32_t> &out) {
                    out.clear();
                    out.reserve(sorted.size());
                    for (int32_t k : sorted)
                        if (!corner[(size_t)k]) out.push_back(k);
                    for (int64_t k = 0; k < n; ++k)
                        if (corner[(size_t)k] && c->h_row0[idx[k] + 1] - c->h_row0[idx[k]] == 1) out.push_back((int32_t)k);
                };
                rest_then_corner(*ord_f, in_f, from_idx, nf, ord_f_own);
                rest_then_corner(*ord_t, in_t, to_idx, nt, ord_t_own);
                ord_f = &ord_f_own;
                ord_t = &ord_t_own;
            }
            c->sorted_blocks += sp ? sp->nseg : 1;
        }
    }
    LDW_REQUIRE(!sp || (ord_f && ord_t), LDW_ERR_STATE, "span at block %lld cannot be ordered", (long long)blk_no);
    if (int rc = build_side(c, from_idx, nf, SF, ord_f)) return rc;
    if (int rc = build_side(c, to_idx, nt, ST, ord_t)) return rc;
    hb.RFpad = SF.Rpad;
    hb.RTpad = ST.Rpad;
    hb.n_lr_total = (hb.diag ? nf * (nf - 1) / 2 : nf * nt - std::min(nf, nt)) - hb.n_sr_blk;
    auto al = [](size_t x) { return (x + 63) / 64 * 64; };
    size_t o = 0;
    hb.o_idx_f = o; o = al(o + (size_t)nf * 4);
    hb.o_idx_t = o; o = al(o + (size_t)nt * 4);
    hb.o_rl_f = o; o = al(o + SF.rowlist.size() * 4);
    hb.o_rl_t = o; o = al(o + ST.rowlist.size() * 4);
    hb.o_lrow_f = o; o = al(o + (size_t)nf * 4);
    hb.o_lrow_t = o; o = al(o + (size_t)nt * 4);
    std::vector<int32_t> pf;
    build_perm_tiles(c, from_idx, nf, pf, ord_f);
    hb.nf_tiles = (int)(pf.size() / 64);
    {   // where the SNPs with 1 or 2 indicator rows end in either order
        int64_t n12f = 0, n1 = 0, n2 = 0;
        for (int64_t k = 0; k < nf; ++k) {
            const int nr = c->h_row0[from_idx[k] + 1] - c->h_row0[from_idx[k]];
            n1 += nr == 1;
            n2 += nr == 2;
        }
        n12f = (n1 + 63) / 64 + (n2 + 63) / 64;   // build_perm_tiles pads each of the two classes to whole tiles
        hb.gen_t0 = (int)n12f;
        int64_t q12 = 0;
        for (int64_t k = 0; k < nt; ++k) {
            const int nr = c->h_row0[to_idx[k] + 1] - c->h_row0[to_idx[k]];
            q12 += nr == 1 || nr == 2;
        }
        hb.gen_q0 = (int)q12;
    }
    hb.o_perm = o; o = al(o + pf.size() * 4);
    hb.o_perm_t = o; o = al(o + (size_t)nt * 4);
    hb.o_cols = o; o = al(o + cols.size() * sizeof(ColInfo));
    hb.o_pos_f = o; o = al(o + SF.pos.size() * 4);
    hb.o_pos_t = o; o = al(o + ST.pos.size() * 4);
    hb.o_cls_f = o; o = al(o + SF.cls.size());
    hb.o_cls_t = o; o = al(o + ST.cls.size());
    // mixed-precision path: row-slot classes of the to side, widest class per from-tile, low-limb block offsets
    std::vector<int32_t> cmax((size_t)hb.nf_tiles, 1);
    std::vector<int64_t> tbase(cmax.size(), 0);
    {
        LoHost &lo = hb.lo;
        lo = LoHost();
        for (int64_t k = 0; k < nt; ++k) ++lo.n_lc[lo_class(c->h_row0[to_idx[k] + 1] - c->h_row0[to_idx[k]])];
        int32_t uo = 0, rb = 0;
        for (int lc = 0; lc < 3; ++lc) {
            lo.uoff[lc] = uo;
            lo.rowbase[lc] = rb;
            uo += lo.n_lc[lc];
            rb += (int32_t)((((int64_t)lo.n_lc[lc] << lc) + TILE - 1) / TILE * TILE);
        }
        lo.RTlo = rb;
        lo.ntiles = (int32_t)cmax.size();
        for (size_t t = 0; t < pf.size(); ++t) {
            if (pf[t] < 0) continue;
            const int32_t a = from_idx[pf[t]];
            const int cls = 1 << lo_class(c->h_row0[a + 1] - c->h_row0[a]);
            if (cls > cmax[t / 64]) cmax[t / 64] = cls;
        }
        int64_t tb = 0;
        for (size_t t = 0; t < cmax.size(); ++t) {
            tbase[t] = tb;
            tb += (int64_t)lo.RTlo * 64 * cmax[t];
            ++lo.n_tiles_cf[cmax[t] == 1 ? 0 : (cmax[t] == 2 ? 1 : 2)];
        }
        lo.slot = slot;
        lo.diag = hb.diag ? 1 : 0;
        lo.glo_total = tb;
    }
    std::vector<int32_t> tf;
    // last tiles first: the tile of the SNPs with >= 3 minor states (generic code, every unit listed, cmax 4) is the
    // critical path of the gathered GEMM and must not start last
    for (size_t t = cmax.size(); t-- > 0;)
        for (int fs = 0; fs < cmax[t]; ++fs) {
            tf.push_back((int32_t)t);
            tf.push_back(fs);
        }
    hb.lo.n_tf = (int32_t)(tf.size() / 2);
    // approximate path: tiles of the exact GEMM (128 to-side x 64 from-side rows) that hold a short-range pair.  POS ascends along
    // both lists and the row lists keep the list order within a slot-count class, so the partners of a to-side SNP are a
    // contiguous row range per class
    std::vector<uint8_t> band((size_t)(hb.RTpad / TILE) * (hb.RFpad / 64), 0);
    bool no_rowless = true;   // no SNP without an indicator row: a one-row SNP's row-list position is its slot (ApxGemmArgs::fuse)
    {
        const int ntx = hb.RFpad / 64;
        auto cls_of = [&](int32_t snp) { const int nr = c->h_row0[snp + 1] - c->h_row0[snp]; return nr <= 1 ? 0 : (nr == 2 ? 1 : 2); };
        std::vector<int32_t> pre[3];
        int32_t base[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) pre[k].assign((size_t)nf + 1, 0);
        for (int64_t a = 0; a < nf; ++a) {
            const int k = cls_of(from_idx[a]);
            for (int q = 0; q < 3; ++q) pre[q][(size_t)a + 1] = pre[q][(size_t)a] + (q == k ? 1 : 0);
            const uint32_t m = c->h_slot_meta[(size_t)from_idx[a]];
            const int n = (int)(m & 7);
            if ((n == 1 || n == 2) && (((m >> 3) & ((2u << n) - 1u)) != ((2u << n) - 1u))) hb.lo.band_full = 1;
            if (n == 0) hb.lo.band_full = 1;   // SNPs without a row sit among the one-row SNPs in the row list but last in the tiles
            if (n == 0) no_rowless = false;
        }
        for (int64_t b2 = 0; b2 < nt; ++b2) {
            const uint32_t m = c->h_slot_meta[(size_t)to_idx[b2]];
            const int n = (int)(m & 7);
            if (n == 0) no_rowless = false;
            if ((n == 1 || n == 2) && (((m >> 3) & ((2u << n) - 1u)) != ((2u << n) - 1u))) hb.lo.band_full = 1;
        }
        {   // first row of each class region of the from-side row list (build_side: classes 1, 2, 4 in this order, 32-row aligned)
            int64_t rows = 0;
            for (int k = 0; k < 3; ++k) {
                base[k] = (int32_t)rows;
                rows += (int64_t)pre[k][(size_t)nf] * (1 << k);
                rows = (rows + 31) / 32 * 32;
            }
        }
        auto mark = [&](int64_t t0, int64_t t1, int64_t f0, int64_t f1) {
            for (int64_t ty = t0 / TILE; ty <= (t1 - 1) / TILE; ++ty)
                for (int64_t tx = f0 / 64; tx <= (f1 - 1) / 64; ++tx) band[(size_t)ty * ntx + tx] = 1;
        };
        std::vector<int32_t> nxt1, prv1;   // first one-row SNP at or after / last one before a list position (ordered rows only)
        if (hb.n_sr_blk > 0 && !hb.lo.band_full && ord_f && !lr_split) {
            nxt1.assign((size_t)nf + 1, (int32_t)nf);
            prv1.assign((size_t)nf + 1, -1);
            for (int64_t a = nf - 1; a >= 0; --a) nxt1[(size_t)a] = cls_of(from_idx[a]) == 0 ? (int32_t)a : nxt1[(size_t)a + 1];
            for (int64_t a = 0; a < nf; ++a) prv1[(size_t)a + 1] = cls_of(from_idx[a]) == 0 ? (int32_t)a : prv1[(size_t)a];
        }
        if (hb.n_sr_blk > 0 && !hb.lo.band_full && !lr_split)
            for (int64_t b2 = 0; b2 < nt; ++b2) {
                const ColInfo &ci = cols[(size_t)b2];
                const int64_t rb0 = ST.lrow[(size_t)b2], rb1 = rb0 + (1 << cls_of(to_idx[b2]));
                for (int iv = 0; iv < 3; ++iv) {
                    if (ci.e[iv] <= ci.s[iv]) continue;
                    for (int k = 0; k < 3; ++k) {
                        const int32_t n0 = pre[k][(size_t)ci.s[iv]], n1 = pre[k][(size_t)ci.e[iv]];
                        if (n1 <= n0) continue;
                        // a listed unit is evaluated whole: all 64 from-side SNPs of its epilogue tile (build_perm_tiles: 64 SNPs of
                        // one class in list order = 64 << k consecutive rows), so the range grows to whole tiles; the tiles of the
                        // SNPs with 3 and 4 rows are ordered differently from their rows: their whole class region is kept
                        int64_t f0 = base[k] + (int64_t)(n0 / 64 * 64) * (1 << k), f1 = base[k] + (int64_t)((n1 + 63) / 64 * 64) * (1 << k);
                        if (k == 0 && ord_f) {
                            // ordered rows: the partners (corner SNPs, in list order among themselves) start at the row of the first
                            // one-row SNP of the interval and end at that of the last
                            const int64_t first = nxt1[(size_t)ci.s[iv]], last = prv1[(size_t)ci.e[iv]];
                            f0 = (int64_t)SF.lrow[(size_t)first] / 64 * 64;
                            f1 = ((int64_t)SF.lrow[(size_t)last] + 1 + 63) / 64 * 64;
                        }
                        if (k == 2) {
                            f0 = base[2];
                            f1 = base[2] + (int64_t)pre[2][(size_t)nf] * 4;
                        }
                        if (f1 > hb.RFpad) f1 = hb.RFpad;
                        mark(rb0, rb1, f0, f1);
                        if (hb.diag) mark(f0, f1, rb0, rb1);   // a pair may be read in mirrored roles (g_entry)
                    }
                }
            }
    }
    hb.lo.fuse_ok = no_rowless ? 1 : 0;
    hb.lo.ordered = ord_f != nullptr ? 1 : 0;
    hb.o_cmax = o; o = al(o + cmax.size() * 4);
    hb.o_tbase = o; o = al(o + tbase.size() * 8);
    hb.o_tf = o; o = al(o + tf.size() * 4);
    hb.o_band = o; o = al(o + band.size());
    hb.total = o;
    const int ps = pin_slot >= 0 ? pin_slot : slot;
    if (c->pin_cap[ps] < stage_base + o) {
        void *np = nullptr;
        LDW_HIP(hipHostMalloc(&np, (stage_base + o) * 2, hipHostMallocDefault));
        if (c->pin[ps] && stage_base) memcpy(np, c->pin[ps], stage_base);   // (the images of the item's earlier parts)
        if (c->pin[ps]) LDW_HIP(hipHostFree(c->pin[ps]));
        c->pin[ps] = np;
        c->pin_cap[ps] = (stage_base + o) * 2;
    }
    char *b = static_cast<char *>(c->pin[ps]) + stage_base;
    memcpy(b + hb.o_idx_f, from_idx, (size_t)nf * 4);
    memcpy(b + hb.o_idx_t, to_idx, (size_t)nt * 4);
    memcpy(b + hb.o_rl_f, SF.rowlist.data(), SF.rowlist.size() * 4);
    memcpy(b + hb.o_rl_t, ST.rowlist.data(), ST.rowlist.size() * 4);
    memcpy(b + hb.o_lrow_f, SF.lrow.data(), (size_t)nf * 4);
    memcpy(b + hb.o_lrow_t, ST.lrow.data(), (size_t)nt * 4);
    memcpy(b + hb.o_perm, pf.data(), pf.size() * 4);
    build_perm(c, to_idx, nt, reinterpret_cast<int32_t *>(b + hb.o_perm_t), ord_t);
    memcpy(b + hb.o_cols, cols.data(), cols.size() * sizeof(ColInfo));
    memcpy(b + hb.o_pos_f, SF.pos.data(), SF.pos.size() * 4);
    memcpy(b + hb.o_pos_t, ST.pos.data(), ST.pos.size() * 4);
    memcpy(b + hb.o_cls_f, SF.cls.data(), SF.cls.size());
    memcpy(b + hb.o_cls_t, ST.cls.data(), ST.cls.size());
    memcpy(b + hb.o_cmax, cmax.data(), cmax.size() * 4);
    memcpy(b + hb.o_tbase, tbase.data(), tbase.size() * 8);
    memcpy(b + hb.o_tf, tf.data(), tf.size() * 4);
    memcpy(b + hb.o_band, band.data(), band.size());
    return LDW_OK;
}

int launch_gather(ldw_ctx *c, const HostBlock &hb, const EmitArgs &E, const SmallLayout &sl) {
    GatherArgs S;
    S.MI = c->MIblk.as<double>();
    S.cols = E.cols;
    S.nf = (int)hb.nf;
    S.nt = (int)hb.nt;
    S.lower_only = E.lower_only;
    hipLaunchKernelGGL(k_lr_gather, dim3((unsigned)((hb.nt + 15) / 16)), dim3(256), 0, c->stream, S, sl.pick[hb.slot], E.ckey, E.cval);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

// Bound of |MI(exact sums) - MI(high-limb sums)| in nats.  Every cell of a joint table — indicator-row cell or derived by
// subtraction from the high-limb marginals — misses exactly the low limbs of ITS sequences, so sum_cells |dp| <= lo_abs_sum;
// dMI <= (1/den) sum_cells |dp| (|ln(pxy den / d)| + 1) with 0.5 <= pxy <= den, 0.25 <= d <= den^2, den >= neff.
double lo_bound(const ldw_ctx *c) {
    const double den = c->neff > 1.0 ? c->neff : 1.0;
    return c->lo_abs_sum * (2.0 * std::log(den + 12.5) + 3.0) / den;
}

// device image of a block's index structures (its part of the item's staging buffer)
static inline const char *stage_ptr(ldw_ctx *c, const HostBlock &hb) {
    return c->dstage[hb.pin_slot >= 0 ? hb.pin_slot : hb.slot].as<char>() + hb.stage_base;
}

// what the emission of a block's pairs needs (short-range table, candidate list of this slot)
int make_emit_args(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl, int spec_B) {
    const int s = hb.slot;
    const bool do_lr = !p->sr_only;
    const char *d = stage_ptr(c, hb);
    EmitArgs E;
    memset(&E, 0, sizeof(E));
    E.cols = reinterpret_cast<const ColInfo *>(d + hb.o_cols);
    E.nf = (int)hb.nf;
    E.lower_only = hb.diag ? 1 : 0;
    E.keep_sr = p->keep_sr ? 1 : 0;
    E.do_lr = do_lr ? 1 : 0;
    E.sr_base = (c->early_sr || hb.sr_base_fixed) ? hb.sr_base : c->n_sr;   // (early_sr: the rows were assigned when the item was submitted)
    E.sr_a = c->sr_a.as<int32_t>();
    E.sr_b = c->sr_b.as<int32_t>();
    E.sr_mi = c->sr_mi.as<double>();
    // Long-range candidates: with a bucket guess from an earlier block the epilogue appends them itself and the
    // dense MI block is neither written nor re-read; without one (first block, histogram engine) the dense block
    // is written and k_lr_gather collects them once the true bucket is known.
    hb.spec_B = spec_B;
    size_t cap = (size_t)hb.nf * hb.nt;  // worst case: every pair of the block
    if (hb.span) {
        // a span only ever runs speculatively: a segment's candidates all come from the span's pair lists (a segment whose guess was
        // wrong is redone on its own, with its own worst-case list)
        size_t seg_max = 0;
        for (int k = 0; k < hb.span; ++k) seg_max = std::max(seg_max, (size_t)hb.nf * (size_t)hb.seg_nt[k]);
        hb.cand_cap = std::min<size_t>(seg_max, (size_t)PAIR_PATHS * PAIR_SHARDS * pair_cap_for(hb.nf, hb.nt, hb.span));
        cap = hb.cand_cap * (size_t)hb.span;
    }
    if (do_lr && !hb.force_plain) {
        if (int rc = c->cand_key[s].reserve(cap * 8)) return rc;
        if (int rc = c->cand_val[s].reserve(cap * 8)) return rc;
    }
    if (hb.span) {
        if (int rc = c->hist[s].reserve((size_t)hb.span * NBINS * 8)) return rc;
        for (int k = 0; k < hb.span; ++k) {
            ldw::PickOut *pk = reinterpret_cast<ldw::PickOut *>(reinterpret_cast<char *>(sl.pick[s]) + (size_t)k * PICK_STRIDE);
            hb.sseg[k].n_cand = &pk->n_cand;
            hb.sseg[k].ckey = c->cand_key[s].as<uint64_t>() + (size_t)k * hb.cand_cap;
            hb.sseg[k].cval = c->cand_val[s].as<uint64_t>() + (size_t)k * hb.cand_cap;
            hb.sseg[k].ghist = c->hist[s].as<unsigned long long>() + (size_t)k * NBINS;
        }
    }
    E.write_dense = (do_lr && hb.spec_B < 0) ? 1 : 0;   // the dense block only feeds k_lr_gather; SR-only passes take the screen path
    E.spec_B = hb.spec_B;
    E.spec_lo = hb.spec_B > 0 ? bucket_lo(hb.spec_B) : -1e300;
    E.any_sr = (hb.n_sr_blk > 0 && !hb.span && !hb.lr_split) ? 1 : 0;   // (a span / a split block never emits a short-range row itself: sr_excl)
    E.n_cand = &sl.pick[s]->n_cand;
    E.ckey = c->cand_key[s].as<uint64_t>();
    E.cval = c->cand_val[s].as<uint64_t>();
    if (hb.force_plain && do_lr) {
        // a span's segment redone on its own: the slot's candidate buffers still hold the lists of the span's later segments
        if (int rc = c->miss_key.reserve((size_t)hb.nf * hb.nt * 8)) return rc;
        if (int rc = c->miss_val.reserve((size_t)hb.nf * hb.nt * 8)) return rc;
        E.ckey = c->miss_key.as<uint64_t>();
        E.cval = c->miss_val.as<uint64_t>();
    }
    // fp32 screen: only where a pair can be dismissed at all (speculative mode with a positive lower edge)
    E.scr_mode = (hb.spec_B > 0 || !do_lr) && !E.write_dense && (!do_lr || speculation_pays(c, p)) ? c->screen : 0;   // (no screen where nearly every unit would be listed)
    // the screen reads the top 31 bits of a joint sum; in the mixed-precision path the sums are those of the high-limb
    // weights (units of 2^(16 - F)) and the margin also covers what the low limbs can add (lo_bound)
    const int64_t tot = hb.mixed ? c->total_fixed_hi : c->total_fixed;
    int bits = 0;
    while (bits < 62 && (tot >> bits) != 0) ++bits;
    E.scr_shift = bits > 31 ? bits - 31 : 0;
    E.scr_scale = (float)std::ldexp(1.0, E.scr_shift - c->frac_bits + (hb.mixed ? 8 * LO_LIMBS : 0));
    E.scr_eps = SCREEN_EPS + (hb.mixed ? (float)lo_bound(c) : 0.0f);
    if (hb.apx) {
        // the screen reads int32 sums of the approximate weights in units of 2^e_last; its bound of the exact MI carries the
        // relative error delta of the weights and the units lost to truncation (full_cells_screen<.., APX>)
        const double den = c->neff > 1.0 ? c->neff : 1.0;
        E.apx = 1;
        E.apx_EG = (float)(c->apx_lost_units * 1.001);
        E.apx_dfac = (float)(1.01 * c->apx_delta / (1.0 - c->apx_delta));
        static const bool r02_bound = exp_env("LDW_SCREEN_R02_BOUND") != nullptr;   // A/B: the bound without the totals argument
        E.apx_unit = std::ldexp(1.0, c->apx_e_last - c->frac_bits);
        E.apx_s1 = (float)(E.apx_unit * (2.0 * std::log(den + 12.5) + (r02_bound ? 3.1 : 2.1)) / (1.0 - c->apx_delta) * 1.01);
        E.apx_c1 = r02_bound ? 1.02f : 0.02f;
        E.apx_W = r02_bound ? 0.0 : std::ldexp((double)c->total_fixed, -c->frac_bits);
        E.scr_shift = 0;
        E.scr_scale = (float)std::ldexp(1.0, c->apx_e_last - c->frac_bits);
        E.scr_eps = SCREEN_EPS;
    }
    E.scr_viol = reinterpret_cast<unsigned long long *>(sl.lr_count + 1);
    hb.E = E;
    return LDW_OK;
}

// bucket pick + copy-back of the pick of slot s on `st`
int launch_pick(ldw_ctx *c, const HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl, hipStream_t st) {
    const int s = hb.slot;
    if (hb.span) {
        PickSpanArgs S;
        memset(&S, 0, sizeof(S));
        for (int k = 0; k < hb.span; ++k) S.n_total[k] = (long long)hb.seg_lr_total[k];
        hipLaunchKernelGGL(k_pick_bucket_span, dim3((unsigned)hb.span), dim3(256), 0, st, c->hist[s].as<unsigned long long>(), p->lr_retain_links,
                           p->lr_links_approx, hb.spec_B, S, reinterpret_cast<char *>(sl.pick[s]), PICK_STRIDE,
                           reinterpret_cast<const unsigned int *>(c->pairs[s].p), pair_cap_for(hb.nf, hb.nt, hb.span));
        LDW_HIP(hipGetLastError());
        return LDW_OK;
    }
    if (!p->sr_only) {
        const bool pl = hb.apx && c->screen == 1;
        hipLaunchKernelGGL(k_pick_bucket, dim3(1), dim3(256), 0, st, c->hist[s].as<unsigned long long>(), p->lr_retain_links,
                           p->lr_links_approx, hb.spec_B, (long long)hb.n_lr_total, sl.pick[s],
                           pl ? reinterpret_cast<const unsigned int *>(c->pairs[s].p) : nullptr, pair_cap_for(hb.nf, hb.nt));
        LDW_HIP(hipGetLastError());
    }
    return LDW_OK;
}

// First phase: upload the block's index structures, then
//   unfused: the co-occurrence GEMM into this slot's G buffer on the GEMM stream.  Nothing there touches what the
//            previous block's epilogue / selection still uses, so it overlaps the tail of block b;
//   fused:   GEMM + epilogue in one kernel, the bucket pick and the copy-back of the pick, all on the GEMM stream:
//            the whole block overlaps the selection (sorts, host round trip) of the previous one.
static void fill_dev_ptrs(ldw_ctx *c, HostBlock &hb) {
    const char *d = stage_ptr(c, hb);
    auto I = [&](size_t off) { return reinterpret_cast<const int32_t *>(d + off); };
    auto B = [&](size_t off) { return reinterpret_cast<const uint8_t *>(d + off); };
    hb.D = DevPtrs{I(hb.o_idx_f), I(hb.o_idx_t), I(hb.o_rl_f), I(hb.o_rl_t), I(hb.o_lrow_f), I(hb.o_lrow_t), I(hb.o_perm), I(hb.o_perm_t),
                   I(hb.o_pos_f), I(hb.o_pos_t), B(hb.o_cls_f), B(hb.o_cls_t), I(hb.o_cmax),
                   reinterpret_cast<const int64_t *>(d + hb.o_tbase), I(hb.o_tf), B(hb.o_band), hb.nf_tiles, hb.gen_t0,
                   hb.gen_q0};
}

// The short-range pairs of one corner segment of a span: the block alone, in list order, through the SR-only form of the approximate path —
// per-SNP constants, k_mi_screen (lists the units that hold a short-range pair: no MI needed for that), the exact 5-limb GEMM of the band's
// tiles, k_mi_units (fp64 MI of the listed units; only the short-range pairs are emitted, to their final rows) — both phases on the GEMM
// given stream (the main one, in front of the item's second phase), with the slot's second set of list / constant buffers.
int launch_sr_sub(ldw_ctx *c, HostBlock &sub, const ldw_mi_params *p, const SmallLayout &sl, hipStream_t gs, int64_t sr_base) {
    fill_dev_ptrs(c, sub);
    ldw_mi_params q = *p;
    q.sr_only = 1;
    sub.sr_base = sr_base;
    sub.sr_base_fixed = true;
    sub.apx = true;
    sub.lo.apx = 1;
    sub.lo.sr_sub = 1;
    LDW_REQUIRE(!sub.lo.band_full && !sub.generic && sub.n_sr_blk > 0, LDW_ERR_STATE, "SR sub-pass of block %lld: unexpected block structure", (long long)sub.blk_no);
    if (int rc = make_emit_args(c, sub, &q, sl, -1)) return rc;
    hipEvent_t dummy[6] = {c->ev[3], c->ev[3], c->ev[3], c->ev[3], c->ev[3], c->ev[3]};
    if (int rc = launch_block_apx(c, sub.D, sub.nf, sub.nt, sub.RFpad, sub.RTpad, p->quirk_mode, sub.E, dummy, 1, gs, nullptr, &sub.lo)) return rc;
    if (int rc = launch_block_apx(c, sub.D, sub.nf, sub.nt, sub.RFpad, sub.RTpad, p->quirk_mode, sub.E, dummy, 2, gs, nullptr, &sub.lo, nullptr, nullptr, 0, gs)) return rc;
    ++c->span_sr_subs;
    return LDW_OK;
}

int submit_a(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl) {
    const int s = hb.slot;
    const int stg = hb.pin_slot >= 0 ? hb.pin_slot : s;   // (a span's segment that runs on its own is staged through the extra buffer)
    const size_t img = hb.stage_total ? hb.stage_total : hb.total;   // (a span's image includes the lists of its SR sub-passes)
    if (int rc = c->dstage[stg].reserve(img)) return rc;
    // the device image and the per-slot buffers (G, histogram, pick, candidates) were last used by the block two steps back
    if (c->done_recorded[s]) LDW_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_done[s], 0));
    LDW_HIP(hipMemcpyAsync(c->dstage[stg].p, c->pin[stg], img, hipMemcpyHostToDevice, c->copy_stream));
    LDW_HIP(hipEventRecord(c->ev_up[s], c->copy_stream));
    c->up_recorded[s] = true;
    fill_dev_ptrs(c, hb);
    hb.submitted = true;
    if (c->engine != LDW_ENGINE_MFMA || hb.generic) return LDW_OK;
    hipStream_t gs = c->overlap ? c->gemm_stream : c->stream;   // overlap off: the stages of all blocks run back to back
    LDW_HIP(hipStreamWaitEvent(gs, c->ev_up[s], 0));
    if (c->done_recorded[s]) LDW_HIP(hipStreamWaitEvent(gs, c->ev_done[s], 0));
    hipEvent_t *ev = &c->ev_pool[(size_t)hb.blk_no * EVB];
    const bool do_lr = !p->sr_only;
    int guess = do_lr ? ((speculation_pays(c, p) && !hb.force_plain) ? c->spec_B_next[hb.diag ? 1 : 0] : -1) : 0;
    if (hb.span > 1 && guess > 0) {
        // one guess serves every reference block of the span, and the next guess only arrives after all of them: LDW_SPAN_MARGIN=k lowers it by k
        // more buckets.  Not needed: at C5 (800 kept rows per block, the noisiest thresholds) 3 of 1275 blocks miss per pass with k = 0, as many
        // as block by block, and k = 4 lists 40 % more pairs (858 against 846 ms per pass)
        static const int extra_env = [] { const char *e = exp_env("LDW_SPAN_MARGIN"); return e ? atoi(e) : -1; }();
        const int extra = extra_env >= 0 ? extra_env : 0;
        guess = guess > extra ? guess - extra : 1;
    }
    if (hb.lr_split && !(guess > 0 && c->path_mode != 1 && c->apx_ok && c->screen == 1 && do_lr && !c->fused)) {
        // prepared for the split, but no positive guess for its kind (no probe for blocks this small, say): the plain path does the whole
        // block — short-range rows included — on the ordered rows (it reads every position through the row maps)
        hb.lr_split = false;
        hb.subs.clear();
        hb.subs_seg.clear();
        guess = do_lr ? -1 : 0;
    }
    if ((int64_t)c->ev_valid.size() < hb.blk_no + std::max(1, hb.span)) c->ev_valid.resize((size_t)(hb.blk_no + std::max(1, hb.span)), 1);
    c->ev_valid[(size_t)hb.blk_no] = 1;
    for (int k = 1; k < hb.span; ++k) c->ev_valid[(size_t)hb.blk_no + k] = 0;   // (the span's stage events are its first block's; a segment that runs alone records its own)
    if (c->early_sr && !hb.sr_base_fixed) {
        // this pass assigns short-range rows in SUBMIT order (= block order): the SR sub-passes of a span write theirs from the GEMM stream, ahead
        // of the second phase of the items before it
        int64_t sr_add = 0;
        if (p->keep_sr) {
            if (hb.span) for (int k = 0; k < hb.span; ++k) sr_add += hb.seg_n_sr[k];
            else sr_add = hb.n_sr_blk;
        }
        if (int rc = ensure_links_capacity(c, c->n_sr + sr_add, c->n_lr)) return rc;
        hb.sr_base = c->n_sr;
        c->n_sr += sr_add;
    }
    if (hb.span) {
        // a span runs the approximate path or not at all: should the state it was planned on have gone (no positive guess any more),
        // its reference blocks take the ordinary chain one after the other (finish_span)
        const bool can = c->path_mode != 1 && c->apx_ok && c->screen == 1 && do_lr && guess > 0 && !c->fused && c->engine == LDW_ENGINE_MFMA;
        if (!can) {
            hb.span_alone = true;
            return LDW_OK;
        }
    }
    hb.fused = c->fused && c->nlimbs <= 5 && (!do_lr || guess >= 0);
    if (hb.span) c->unfused_blocks += hb.span;
    else ++(hb.fused ? c->fused_blocks : c->unfused_blocks);
    hb.guess = guess;
    if (!hb.fused) {
        // mixed precision: with a bucket guess the block will run the screen, which only needs the high limbs; the low
        // limbs follow for the listed units only.  The guess is frozen here because the GEMM commits to it.
        hb.mixed = c->mixed && c->screen && c->nlimbs == HI_LIMBS + LO_LIMBS && do_lr && guess > 0 && lo_bound(c) < 1e-3 &&
                   c->N <= 60000;   // the low-limb sums are int32: |sum| <= N * 2^15
        // approximate GEMM + class-wise popcounts: any block that takes the screen path (a bucket guess exists, or the pass
        // is SR-only and needs no MI to screen at all)
        hb.apx = c->path_mode != 1 && c->apx_ok && c->screen && (do_lr ? guess > 0 : true) && hb.nt < (1 << 29);
        LDW_REQUIRE(c->path_mode != 2 || hb.apx || (do_lr && guess <= 0), LDW_ERR_STATE,
                    "ldw_set_path(2): the approximate path is not available (delta %.3g, %d weight classes, %lld sequences, screen %d)", c->apx_delta,
                    c->n_classes, (long long)c->N, c->screen);
        LDW_REQUIRE(!hb.lr_split || hb.apx, LDW_ERR_STATE, "block %lld was prepared for the split (SR sub-pass + ordered long-range pass) but cannot take the approximate path",
                    (long long)hb.blk_no);
        if (hb.apx) {
            hb.mixed = false;
            hb.lo.apx = 1;
            c->apx_blocks += hb.span ? hb.span : 1;
        }
        if (hb.span) {
            ++c->span_items;
            c->span_blocks += hb.span;
        }
        if (hb.mixed) ++c->mixed_blocks;
        if (hb.apx) {
            // phase 1 of the approximate path: panels, GEMM, SNP constants and the screens, all beside the previous block's tail.
            // The emission constants of phase 2 (table pointers, row base) are refreshed in submit_b.
            if (int rc = make_emit_args(c, hb, p, sl, do_lr ? guess : -1)) return rc;
            if (int rc = c->hist[s].reserve((size_t)NBINS * 8 * (size_t)(hb.span ? hb.span : 1))) return rc;
            hb.lo.span = hb.span;
            hb.lo.sseg = hb.sseg;
            hb.lo.sr_excl = ((hb.span && !hb.subs.empty()) || hb.lr_split) ? 1 : 0;
            if (int rc = launch_block_apx(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, hb.E, ev, 1, gs, nullptr, &hb.lo, c->hist[s].p, sl.pick[s],
                                          PICK_STRIDE * (size_t)(hb.span ? hb.span : 1)))
                return rc;
            LDW_HIP(hipEventRecord(c->ev_gemm[s], gs));
            return LDW_OK;
        }
        EmitArgs E;
        memset(&E, 0, sizeof(E));
        E.lower_only = hb.diag ? 1 : 0;
        E.do_lr = do_lr ? 1 : 0;
        if (int rc = launch_block_mi(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, E, ev, 1, &gx(c, s), gs, nullptr,
                                     hb.mixed ? &hb.lo : nullptr))
            return rc;
        LDW_HIP(hipEventRecord(c->ev_gemm[s], gs));
        return LDW_OK;
    }
    const int64_t sr_add = p->keep_sr ? hb.n_sr_blk : 0;
    if (int rc = ensure_links_capacity(c, c->n_sr + sr_add, c->n_lr)) return rc;
    if (int rc = c->hist[s].reserve((size_t)NBINS * 8)) return rc;
    if (int rc = make_emit_args(c, hb, p, sl, do_lr ? guess : -1)) return rc;
    c->n_sr += sr_add;
    LDW_HIP(hipMemsetAsync(c->hist[s].p, 0, (size_t)NBINS * 8, gs));
    LDW_HIP(hipMemsetAsync(sl.pick[s], 0, sizeof(ldw::PickOut), gs));
    FusedArgs F;
    F.Mbits = c->Mbits.as<uint64_t>();
    F.KW = c->KW;
    F.Kpad = c->KW * 64;
    F.rowlist_t = hb.D.rl_t;
    F.rowlist_f = hb.D.rl_f;
    F.digits = c->digits.as<int8_t>();
    F.pos_f = hb.D.pos_f;
    F.pos_t = hb.D.pos_t;
    F.cls_f = hb.D.cls_f;
    F.cls_t = hb.D.cls_t;
    F.ghist = c->hist[s].as<unsigned long long>();
    fill_epi_args(c, hb.D, hb.nf, hb.nt, hb.RFpad, p->quirk_mode, hb.E, nullptr, F.A);
    LDW_HIP(hipEventRecord(ev[0], gs));
#ifdef LDW_EXPERIMENTS
    if (int rc = launch_fused(c, F, hb.RFpad, hb.RTpad, c->nlimbs, gs)) return rc;
#else
    LDW_REQUIRE(false, LDW_ERR_STATE, "the fused kernel is not part of this build (LDW_EXPERIMENTS)");   // (unreachable: ldw_set_fused refuses)
#endif
    LDW_HIP(hipEventRecord(ev[1], gs));
    LDW_HIP(hipEventRecord(ev[4], gs));
    LDW_HIP(hipEventRecord(ev[2], gs));
    if (int rc = launch_pick(c, hb, p, sl, gs)) return rc;
    LDW_HIP(hipMemcpyAsync(c->pin_pick[s], sl.pick[s], sizeof(ldw::PickOut), hipMemcpyDeviceToHost, gs));
    LDW_HIP(hipEventRecord(c->ev_pick[s], gs));
    LDW_HIP(hipEventRecord(c->ev_gemm[s], gs));
    return LDW_OK;
}

// Second phase (unfused path only): epilogue, histogram pick and the copy-back of the pick on the main stream.
// A block in generic order (HostBlock::generic): dense MI of every pair by the plain path, then the reference's pair list from the
// dense block with the len predicate (k_gen_count -> host scan of the 2 nt column counts -> k_gen_emit_sr, k_pick_bucket,
// k_gen_gather).  Synchronous where it needs the counts; such blocks are the exception (the reference's own parser emits ascending
// positions), correctness is what matters here.
int submit_generic(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl) {
    const int s = hb.slot;
    const bool do_lr = !p->sr_only;
    LDW_HIP(hipStreamWaitEvent(c->stream, c->ev_up[s], 0));
    hipEvent_t *ev = &c->ev_pool[(size_t)hb.blk_no * EVB];
    EmitArgs E0;
    memset(&E0, 0, sizeof(E0));
    E0.write_dense = 1;
    E0.spec_B = -1;
    E0.lower_only = 0;   // every entry of the block (a diagonal block too: the pair list decides which ones are pairs)
    if (int rc = launch_block_mi(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, E0, ev, 3, &gx(c, s), nullptr, nullptr)) return rc;
    LDW_HIP(hipEventRecord(ev[5], c->stream));   // (ldw_links_end brackets ev[1] .. ev[5] and ev[4] .. ev[2]: every event of the block must be recorded)
    LDW_HIP(hipEventRecord(ev[4], c->stream));
    if (int rc = c->hist[s].reserve((size_t)NBINS * 8)) return rc;
    if (int rc = c->colcnt.reserve((size_t)hb.nt * 8 + (size_t)hb.nt * 16 + 64)) return rc;
    LDW_HIP(hipMemsetAsync(c->hist[s].p, 0, (size_t)NBINS * 8, c->stream));
    LDW_HIP(hipMemsetAsync(sl.pick[s], 0, sizeof(ldw::PickOut), c->stream));
    GenArgs S;
    S.MI = c->MIblk.as<double>();
    S.nf = (int)hb.nf;
    S.nt = (int)hb.nt;
    S.lower_only = hb.diag ? 1 : 0;
    S.keep_sr = p->keep_sr ? 1 : 0;
    S.do_lr = do_lr ? 1 : 0;
    S.idx_f = hb.D.idx_f;
    S.idx_t = hb.D.idx_t;
    S.POS = c->POS.as<int32_t>();
    S.g = c->g;
    S.sr_dist = p->sr_dist;
    int32_t *cnt_u = c->colcnt.as<int32_t>(), *cnt_l = cnt_u + hb.nt;
    int64_t *off_u = reinterpret_cast<int64_t *>(c->colcnt.as<char>() + (((size_t)hb.nt * 8 + 15) / 16 * 16)), *off_l = off_u + hb.nt;
    const unsigned gridc = (unsigned)((hb.nt + 3) / 4);
    hipLaunchKernelGGL(k_gen_count, dim3(gridc), dim3(256), 0, c->stream, S, cnt_u, cnt_l, c->hist[s].as<unsigned long long>());
    LDW_HIP(hipGetLastError());
    std::vector<int32_t> hc((size_t)hb.nt * 2);
    LDW_HIP(hipMemcpyAsync(hc.data(), cnt_u, (size_t)hb.nt * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    std::vector<int64_t> ho((size_t)hb.nt * 2);
    int64_t nu = 0, nl = 0;
    for (int64_t b = 0; b < hb.nt; ++b) nu += hc[(size_t)b];
    int64_t ru = 0, rl = nu;   // all upper rows first, then all lower rows
    for (int64_t b = 0; b < hb.nt; ++b) {
        ho[(size_t)b] = ru;
        ho[(size_t)hb.nt + b] = rl;
        ru += hc[(size_t)b];
        rl += hc[(size_t)hb.nt + b];
        nl += hc[(size_t)hb.nt + b];
    }
    hb.n_sr_blk = nu + nl;
    hb.n_lr_total = (hb.diag ? hb.nf * (hb.nf - 1) / 2 : hb.nf * hb.nt - std::min(hb.nf, hb.nt)) - hb.n_sr_blk;
    const int64_t sr_add = p->keep_sr ? hb.n_sr_blk : 0;
    if (int rc = ensure_links_capacity(c, c->n_sr + sr_add, c->n_lr)) return rc;
    if (sr_add > 0) {
        LDW_HIP(hipMemcpyAsync(off_u, ho.data(), (size_t)hb.nt * 16, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gen_emit_sr, dim3(gridc), dim3(256), 0, c->stream, S, off_u, off_l, (int64_t)c->n_sr, c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(),
                           c->sr_mi.as<double>());
        LDW_HIP(hipGetLastError());
        LDW_HIP(hipStreamSynchronize(c->stream));   // (ho is a host vector: the copy must be done before it goes away)
    }
    c->n_sr += sr_add;
    hb.spec_B = -1;
    if (do_lr) {
        const size_t cap = (size_t)hb.nf * hb.nt;
        if (int rc = c->cand_key[s].reserve(cap * 8)) return rc;
        if (int rc = c->cand_val[s].reserve(cap * 8)) return rc;
        if (int rc = launch_pick(c, hb, p, sl, c->stream)) return rc;
        hipLaunchKernelGGL(k_gen_gather, dim3(gridc), dim3(256), 0, c->stream, S, sl.pick[s], c->cand_key[s].as<uint64_t>(), c->cand_val[s].as<uint64_t>());
        LDW_HIP(hipGetLastError());
    }
    LDW_HIP(hipEventRecord(ev[2], c->stream));
    LDW_HIP(hipMemcpyAsync(c->pin_pick[s], sl.pick[s], sizeof(ldw::PickOut), hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipEventRecord(c->ev_pick[s], c->stream));
    ++c->generic_blocks;
    return LDW_OK;
}

int submit_b(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl) {
    if (hb.fused || hb.span_alone) return LDW_OK;
    if (hb.generic) return submit_generic(c, hb, p, sl);
    const int s = hb.slot;
    LDW_HIP(hipStreamWaitEvent(c->stream, c->ev_up[s], 0));
    if (c->engine == LDW_ENGINE_MFMA) LDW_HIP(hipStreamWaitEvent(c->stream, c->ev_gemm[s], 0));
    const bool do_lr = !p->sr_only;
    const int64_t sr_add = (p->keep_sr && !c->early_sr && !hb.sr_base_fixed) ? hb.n_sr_blk : 0;   // (early_sr: assigned in submit_a)
    if (int rc = ensure_links_capacity(c, c->n_sr + sr_add, c->n_lr)) return rc;
    if (int rc = c->hist[s].reserve((size_t)NBINS * 8)) return rc;
    if (!hb.apx) LDW_HIP(hipMemsetAsync(c->hist[s].p, 0, (size_t)NBINS * 8, c->stream));   // (the approximate path zeroed both in its first phase: k_zero4)
    if (int rc = make_emit_args(c, hb, p, sl, (hb.mixed || hb.apx) ? (do_lr ? hb.guess : -1) : ((do_lr && c->engine != LDW_ENGINE_HIST_STATES) ? c->spec_B_next[hb.diag ? 1 : 0] : -1)))   // (the bit-plane histogram engine shares the epilogue: it speculates like the MFMA engine)
        return rc;
    if (!hb.apx) LDW_HIP(hipMemsetAsync(sl.pick[s], 0, sizeof(ldw::PickOut), c->stream));
    hipEvent_t *ev = &c->ev_pool[(size_t)hb.blk_no * EVB];
    // the SR sub-passes of the item (a split diagonal block; the corner segments of a span) — on the MAIN stream, in front of the item's own second
    // phase: on the GEMM stream (r04b) they lengthened the longer of the two queues (39.8 against 36.0 ms per C4 pass with the corner blocks)
    if (hb.apx && !hb.subs.empty() && p->keep_sr) {
        if (!hb.span) {
            if (int rc = launch_sr_sub(c, hb.subs[0], p, sl, c->stream, hb.sr_base)) return rc;
        } else {
            int64_t base = hb.sr_base;
            size_t si = 0;
            for (int k = 0; k < hb.span; ++k) {
                if (si < hb.subs.size() && hb.subs_seg[si] == k) {
                    if (int rc = launch_sr_sub(c, hb.subs[si], p, sl, c->stream, base)) return rc;
                    ++si;
                }
                base += hb.seg_n_sr[k];
            }
        }
    }
    if (hb.apx) {
        hb.lo.span = hb.span;
        hb.lo.sseg = hb.sseg;
        hb.lo.sr_excl = ((hb.span && !hb.subs.empty()) || hb.lr_split) ? 1 : 0;
        if (int rc = launch_block_apx(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, hb.E, ev, 2, nullptr, c->hist[s].as<unsigned long long>(), &hb.lo))
            return rc;
    } else if (int rc = launch_block_mi(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, hb.E, ev, c->engine == LDW_ENGINE_MFMA ? 2 : 3,
                                        &gx(c, s), nullptr, c->hist[s].as<unsigned long long>(), hb.mixed ? &hb.lo : nullptr))
        return rc;
    c->n_sr += sr_add;
    if (int rc = launch_pick(c, hb, p, sl, c->stream)) return rc;
    if (do_lr && hb.spec_B < 0)
        if (int rc = launch_gather(c, hb, hb.E, sl)) return rc;
    LDW_HIP(hipMemcpyAsync(c->pin_pick[s], sl.pick[s], hb.span ? PICK_STRIDE * (size_t)hb.span : sizeof(ldw::PickOut), hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipEventRecord(c->ev_pick[s], c->stream));
    return LDW_OK;
}

// guesses for later blocks from the pick of a finished one: a little below this block's bucket
void update_guess(ldw_ctx *c, bool diag, const ldw::PickOut *hp, bool missed) {
    // buckets are 0.5 % wide: guess ~5 % below the threshold of the last block of the same kind.  Diagonal blocks
    // lose their closest pairs to the short-range table and sit ~8 % (16 buckets) lower than off-diagonal ones:
    // until a block of the other kind has been seen, its guess is derived from this one with a wider margin.
    const int kind = diag ? 1 : 0, other = kind ^ 1;
    // the margin follows what the thresholds of this kind have actually done: the spread of the last six of them plus two
    // buckets, between 4 and 10 (every bucket below the true one costs ~3000 more candidates on the C4 shape; a miss costs a
    // full non-speculative pass of the block, after which the history starts over)
    int &hn = c->spec_hist_n[kind];
    if (missed) hn = 0;
    c->spec_hist[kind][hn % 6] = hp->B_true;
    ++hn;
    int margin = 10;
    if (hn >= 3) {
        int lo = hp->B_true, hi = hp->B_true;
        for (int k = 0; k < (hn < 6 ? hn : 6); ++k) {
            lo = c->spec_hist[kind][k] < lo ? c->spec_hist[kind][k] : lo;
            hi = c->spec_hist[kind][k] > hi ? c->spec_hist[kind][k] : hi;
        }
        // few kept rows per block (many blocks: C5 keeps ~800 per block) make the threshold itself noisier
        const bool small = hp->n * (1.0 - hp->prob) < 5000.0;
        c->spec_small[kind] = small;
        margin = hi - lo + (small ? 4 : 2);
        const int mmin = small ? 6 : 4;
        margin = margin < mmin ? mmin : (margin > 10 ? 10 : margin);
    }
    c->spec_B_next[kind] = hp->B_true - margin > 0 ? hp->B_true - margin : 0;
    c->spec_seen[kind] = true;
    if (!c->spec_seen[other] && !c->spec_probed[other]) {   // (a probed guess of the other kind is better than one derived from this kind)
        const int g = diag ? hp->B_true - 10 : hp->B_true - 16 - 2 * 10;
        c->spec_B_next[other] = g > 0 ? g : 0;
    }
}

// The long-range selection of ONE reference block from its candidate list (m candidates; the pick record on the device knows the
// ranks): threshold, kept rows in the reference's row order appended at *lr_count, then k_block_done (running count, block stats).
struct SelIn {
    int64_t m, nf, nt, blk_no, n_sr_blk;
    uint64_t *ck, *cv;
    ldw::PickOut *pick;                 // device
    const int32_t *idx_f, *idx_t;       // device: the block's own index lists
};
int select_rows(ldw_ctx *c, const SelIn &S, bool do_lr, const SmallLayout &sl) {
    const int64_t m = do_lr ? S.m : 0;
    uint64_t *ck = S.ck, *cv = S.cv;
    const uint64_t sel_space = (uint64_t)S.nf * (uint64_t)S.nt;
    static const bool sel_fast_on = getenv("LDW_NO_FAST_SELECT") == nullptr;
    const long long n_words = (long long)((2 * sel_space + 31) / 32) + 1, n_chunks = (long long)((2 * sel_space + SEL_CHUNK_BITS - 1) / SEL_CHUNK_BITS) + 1;
    const long long n_super_ll = (n_chunks + SEL_SUPER - 1) / SEL_SUPER;
    if (do_lr && m > 0 && sel_fast_on && c->select_mode == 0 && m <= SEL_MAX && n_super_ll <= SEL_MAX_SUPER) {
        // the common case: radix select + bitmap ranks, four small launches, no sort (k_sel_thresh)
        if (int rc = ensure_links_capacity(c, c->n_sr, c->n_lr + m)) return rc;
        const int n_super = (int)n_super_ll;
        if ((size_t)n_words * 4 > c->sel_bitmap.cap || (size_t)n_chunks * 4 > c->sel_chunks.cap || (size_t)n_super * 4 > c->sel_prefix.cap) {
            // first use / a larger block: fresh zeroes
            if (int rc = c->sel_bitmap.reserve((size_t)n_words * 4)) return rc;
            if (int rc = c->sel_chunks.reserve((size_t)n_chunks * 4)) return rc;
            if (int rc = c->sel_prefix.reserve((size_t)SEL_MAX_SUPER * 4)) return rc;
            LDW_HIP(hipMemsetAsync(c->sel_bitmap.p, 0, c->sel_bitmap.cap, c->stream));
            LDW_HIP(hipMemsetAsync(c->sel_chunks.p, 0, c->sel_chunks.cap, c->stream));
            LDW_HIP(hipMemsetAsync(c->sel_prefix.p, 0, c->sel_prefix.cap, c->stream));
        }
        uint32_t *bm = c->sel_bitmap.as<uint32_t>(), *cc = c->sel_chunks.as<uint32_t>(), *sc = c->sel_prefix.as<uint32_t>();
        const unsigned gridm = (unsigned)((m + 255) / 256);
        hipLaunchKernelGGL(k_sel_thresh, dim3(1), dim3(1024), 0, c->stream, ck, S.pick);
        hipLaunchKernelGGL(k_sel_mark, dim3(gridm), dim3(256), 0, c->stream, ck, cv, S.pick, sel_space, bm, cc, sc);
        hipLaunchKernelGGL(k_sel_scatter, dim3(gridm), dim3(256), 0, c->stream, ck, cv, S.pick, sel_space, bm, cc, sc, n_super, S.idx_f, S.idx_t, (int)S.nf,
                           sl.lr_count, c->lr_a.as<int32_t>(), c->lr_b.as<int32_t>(), c->lr_mi.as<double>());
        hipLaunchKernelGGL(k_sel_clear, dim3(gridm), dim3(256), 0, c->stream, ck, cv, S.pick, sel_space, bm, cc, sc);
        LDW_HIP(hipGetLastError());
        c->n_lr += m;  // upper bound; the exact value is *lr_count
    } else if (do_lr && m > 0) {
        LDW_REQUIRE(m < 2147483647LL, LDW_ERR_SIZE, "too many quantile candidates (%lld)", (long long)m);
        if (int rc = ensure_links_capacity(c, c->n_sr, c->n_lr + m)) return rc;
        if (int rc = c->cand_key2.reserve((size_t)m * 8)) return rc;
        if (int rc = c->cand_val2.reserve((size_t)m * 8)) return rc;
        size_t tmp_bytes = 0;
        LDW_HIP(prim_sort_pairs(nullptr, tmp_bytes, ck, c->cand_key2.as<uint64_t>(), cv,
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        if (int rc = c->scratch.reserve(tmp_bytes)) return rc;
        LDW_HIP(prim_sort_pairs(c->scratch.p, tmp_bytes, ck, c->cand_key2.as<uint64_t>(), cv,
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        hipLaunchKernelGGL(k_lr_thresh, dim3(1), dim3(64), 0, c->stream, c->cand_key2.as<uint64_t>(), S.pick);
        LDW_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_lr_mark, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                           c->cand_key2.as<uint64_t>(), c->cand_val2.as<uint64_t>(), S.pick, ck, cv, (long long)m);
        LDW_HIP(hipGetLastError());
        LDW_HIP(prim_sort_pairs(c->scratch.p, tmp_bytes, ck, c->cand_key2.as<uint64_t>(), cv,
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        hipLaunchKernelGGL(k_lr_append, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                           c->cand_key2.as<uint64_t>(), c->cand_val2.as<uint64_t>(), S.pick, S.idx_f, S.idx_t, (int)S.nf,
                           sl.lr_count, c->lr_a.as<int32_t>(), c->lr_b.as<int32_t>(), c->lr_mi.as<double>());
        LDW_HIP(hipGetLastError());
        c->n_lr += m;  // upper bound; the exact value is *lr_count
    } else if (do_lr) {
        hipLaunchKernelGGL(k_lr_thresh, dim3(1), dim3(64), 0, c->stream, ck, S.pick);
        LDW_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_block_done, dim3(1), dim3(64), 0, c->stream, S.pick, sl.lr_count, S.n_sr_blk,
                       sl.stats_i + S.blk_no * 3, sl.stats_d + S.blk_no);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

int finish_span(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl);

int finish_block(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl) {
    if (hb.span) return finish_span(c, hb, p, sl);
    const bool do_lr = !p->sr_only;
    const int s = hb.slot;
    // ---- the one host round trip of the block: the candidate count sizes the sorts ----
    LDW_HIP(hipEventSynchronize(c->ev_pick[s]));
    if (hb.fused) LDW_HIP(hipStreamWaitEvent(c->stream, c->ev_gemm[s], 0));   // selection runs on the main stream
    ldw::PickOut *hp = static_cast<ldw::PickOut *>(c->pin_pick[s]);
    bool missed = false;
    if (do_lr && hb.spec_B >= 0 && hp->n > 0 && !hp->spec_ok) {
        // the bucket guess was above the true bucket: redo the epilogue non-speculatively (the short-range rows are
        // already final): full histogram, dense store, then pick and gather with the true bucket.  The plain two-kernel
        // path still has G; the fused and the mixed-precision ones have to run the (full) GEMM again.
        EmitArgs E = hb.E;
        E.write_dense = 1;
        E.spec_B = -1;
        E.keep_sr = 0;
        E.any_sr = hb.n_sr_blk > 0 ? 1 : 0;   // (a split block's speculative pass had it off — sr_excl —: the plain epilogue must keep its short-range pairs out of the histogram itself)
        E.scr_mode = 0;   // every pair goes into the histogram
        hb.spec_B = -1;
        LDW_HIP(hipMemsetAsync(c->hist[s].p, 0, (size_t)NBINS * 8, c->stream));
        LDW_HIP(hipMemsetAsync(sl.pick[s], 0, sizeof(ldw::PickOut), c->stream));
        hipEvent_t dummy[6] = {c->ev[3], c->ev[3], c->ev[4], c->ev[3], c->ev[5], c->ev[3]};   // keep the block's stage events as they are
        E.apx = 0;
        if (int rc = launch_block_mi(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, E, dummy, (hb.fused || hb.mixed || hb.apx) ? 3 : 2,
                                     &gx(c, s), nullptr, c->hist[s].as<unsigned long long>()))
            return rc;
        if (int rc = launch_pick(c, hb, p, sl, c->stream)) return rc;
        if (int rc = launch_gather(c, hb, E, sl)) return rc;
        LDW_HIP(hipMemcpyAsync(c->pin_pick[s], sl.pick[s], sizeof(ldw::PickOut), hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        ++c->spec_misses;
        note_overflow(c, hp->over);
        missed = true;
    }
    if (do_lr && hp->n > 0) update_guess(c, hb.diag, hp, missed || hb.force_plain);   // (force_plain: the redo of a span's segment whose guess was wrong)
    if ((int64_t)c->trace.size() <= hb.blk_no) c->trace.resize((size_t)hb.blk_no + 1);
    {
        ldw::BlockTrace &tr = c->trace[(size_t)hb.blk_no];
        tr.diag = hb.diag ? 1 : 0;
        tr.guess = hb.guess;
        tr.B_true = do_lr ? hp->B_true : -1;
        tr.path = hb.fused ? 3 : (hb.apx ? 2 : (hb.mixed ? 1 : 0));
        tr.missed = (missed || hb.force_plain) ? 1 : 0;
        tr.n_cand = do_lr ? (long long)hp->n_cand : 0;
    }
    if (c->lrc_recorded) {   // exact number of long-range rows kept by all EARLIER blocks
        LDW_HIP(hipEventSynchronize(c->ev_lrc));
        memcpy(&c->n_lr, c->pin_lrc, 8);
    }
    const char *d = stage_ptr(c, hb);
    SelIn S;
    S.m = do_lr ? (int64_t)hp->n_cand : 0;
    S.nf = hb.nf;
    S.nt = hb.nt;
    S.blk_no = hb.blk_no;
    S.n_sr_blk = hb.n_sr_blk;
    S.ck = hb.E.ckey ? hb.E.ckey : c->cand_key[s].as<uint64_t>();
    S.cv = hb.E.cval ? hb.E.cval : c->cand_val[s].as<uint64_t>();
    S.pick = sl.pick[s];
    S.idx_f = reinterpret_cast<const int32_t *>(d + hb.o_idx_f);
    S.idx_t = reinterpret_cast<const int32_t *>(d + hb.o_idx_t);
    if (int rc = select_rows(c, S, do_lr, sl)) return rc;
    LDW_HIP(hipMemcpyAsync(c->pin_lrc, sl.lr_count, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipEventRecord(c->ev_lrc, c->stream));
    c->lrc_recorded = true;
    LDW_HIP(hipEventRecord(c->ev_pool[(size_t)hb.blk_no * EVB + 3], c->stream));
    LDW_HIP(hipEventRecord(c->ev_done[s], c->stream));
    c->done_recorded[s] = true;
    return LDW_OK;
}

// One reference block of a span through the ordinary per-block chain (prep -> submit_a -> submit_b -> finish_block), synchronously, on the
// device buffers of the span's slot but staged through the extra staging buffer: the span's own lists are still read by the selection of
// its other segments.  force_plain: non-speculatively (the segment's guess was wrong).
int run_block_alone(ldw_ctx *c, const HostBlock &span, int k, const ldw_mi_params *p, const SmallLayout &sl, bool force_plain) {
    HostBlock hb;
    const int32_t *ti = span.span_to.data() + span.seg_start[k];
    LDW_HIP(hipStreamSynchronize(c->stream));                        // the extra staging buffer: free (an earlier segment may have used it)
    if (c->gemm_stream) LDW_HIP(hipStreamSynchronize(c->gemm_stream));
    if (int rc = prep_block(c, span.span_from.data(), span.nf, ti, span.seg_nt[k], p, span.slot, span.blk_no + k, hb, nullptr, LDW_NSLOT)) return rc;
    hb.force_plain = force_plain;
    // the segment's short-range rows: their place was assigned with the span; after a wrong guess (force_plain) the span's SR sub-pass has
    // already written them
    ldw_mi_params q = *p;
    if (force_plain) q.keep_sr = 0;
    hb.sr_base = span.sr_base;
    for (int j = 0; j < k; ++j) hb.sr_base += span.seg_n_sr[j];
    hb.sr_base_fixed = true;
    p = &q;
    if (int rc = submit_a(c, hb, p, sl)) return rc;
    if (int rc = submit_b(c, hb, p, sl)) return rc;
    if (int rc = finish_block(c, hb, p, sl)) return rc;
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

// The second half of a span: ONE host round trip for the pick records of all its reference blocks, then per block — in make_blocks
// order, which is the order the reference appends in (R/computePairwiseMI.R:103-116, :362) — the guess update, the selection of its
// candidates into the long-range table and its stats.  A block whose guess turned out too high (or all of them, when a pair list
// overflowed) is redone on its own, non-speculatively, in its place in that order.
int finish_span(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl) {
    const int s = hb.slot;
    if (hb.span_alone) {
        for (int k = 0; k < hb.span; ++k)
            if (int rc = run_block_alone(c, hb, k, p, sl, false)) return rc;
        LDW_HIP(hipEventRecord(c->ev_done[s], c->stream));
        c->done_recorded[s] = true;
        return LDW_OK;
    }
    LDW_HIP(hipEventSynchronize(c->ev_pick[s]));
    ldw::PickOut picks[LDW_SPAN_MAX];
    for (int k = 0; k < hb.span; ++k) memcpy(&picks[k], static_cast<const char *>(c->pin_pick[s]) + (size_t)k * PICK_STRIDE, sizeof(ldw::PickOut));
    if (c->lrc_recorded) {   // exact number of long-range rows kept by all EARLIER blocks (within the span: upper bounds add up)
        LDW_HIP(hipEventSynchronize(c->ev_lrc));
        memcpy(&c->n_lr, c->pin_lrc, 8);
    }
    const char *d = stage_ptr(c, hb);
    if ((int64_t)c->trace.size() < hb.blk_no + hb.span) c->trace.resize((size_t)(hb.blk_no + hb.span));
    {
        static const bool trace_on = getenv("LDW_BLOCK_TRACE") != nullptr;
        if (trace_on && picks[0].n > 0 && !picks[0].spec_ok) {   // a span that missed: its pair-list counters (an overflowing list fails every segment)
            unsigned int pl[PAIR_PATHS * PAIR_SHARDS];
            LDW_HIP(hipMemcpy(pl, c->pairs[s].p, sizeof(pl), hipMemcpyDeviceToHost));
            fprintf(stderr, "[ldw span at block %lld] guess %d, %d segments, pair-list capacity %u, counters by path:", (long long)hb.blk_no, hb.guess, hb.span, pair_cap_for(hb.nf, hb.nt, hb.span));
            for (int pth = 0; pth < PAIR_PATHS; ++pth) {
                unsigned long long tot = 0, mx = 0;
                for (int sh2 = 0; sh2 < PAIR_SHARDS; ++sh2) {
                    tot += pl[pth * PAIR_SHARDS + sh2];
                    mx = std::max<unsigned long long>(mx, pl[pth * PAIR_SHARDS + sh2]);
                }
                fprintf(stderr, "  [%d] total %llu max %llu", pth, tot, mx);
            }
            fprintf(stderr, "\n");
        }
    }
    // the common case — every guess held, every candidate set fits the sort-free selection — takes ONE launch per stage for all segments
    static const bool sel_fast_on = getenv("LDW_NO_FAST_SELECT") == nullptr && exp_env("LDW_NO_SPAN_SELECT") == nullptr;
    bool batched = sel_fast_on && c->select_mode == 0;
    long long m_max = 0, m_sum = 0;
    for (int k = 0; k < hb.span && batched; ++k) {
        const ldw::PickOut *hp = &picks[k];
        const uint64_t space = (uint64_t)hb.nf * (uint64_t)hb.seg_nt[k];
        const long long n_chunks = (long long)((2 * space + SEL_CHUNK_BITS - 1) / SEL_CHUNK_BITS) + 1, n_super = (n_chunks + SEL_SUPER - 1) / SEL_SUPER;
        batched = !(hp->n > 0 && !hp->spec_ok) && (long long)hp->n_cand <= SEL_MAX && n_super <= SEL_MAX_SUPER && (size_t)hp->n_cand <= hb.cand_cap;
        m_max = std::max<long long>(m_max, (long long)hp->n_cand);
        m_sum += (long long)hp->n_cand;
    }
    if (batched) {
        SelSpan S;
        memset(&S, 0, sizeof(S));
        SpanDoneArgs DA;
        memset(&DA, 0, sizeof(DA));
        S.n = hb.span;
        size_t w_off = 0, c_off = 0;
        size_t woff[LDW_SPAN_MAX], coff[LDW_SPAN_MAX];
        for (int k = 0; k < hb.span; ++k) {
            const uint64_t space = (uint64_t)hb.nf * (uint64_t)hb.seg_nt[k];
            const size_t n_words = (size_t)((2 * space + 31) / 32) + 1, n_chunks = (size_t)((2 * space + SEL_CHUNK_BITS - 1) / SEL_CHUNK_BITS) + 1;
            woff[k] = w_off;
            coff[k] = c_off;
            w_off += (n_words + 63) / 64 * 64;
            c_off += (n_chunks + 63) / 64 * 64;
            S.space[k] = space;
            S.n_super[k] = (int)((n_chunks + SEL_SUPER - 1) / SEL_SUPER);
        }
        if (w_off * 4 > c->sel_bitmap.cap || c_off * 4 > c->sel_chunks.cap || (size_t)LDW_SPAN_MAX * SEL_MAX_SUPER * 4 > c->sel_prefix.cap) {
            // first use / a larger span: fresh zeroes (all three arrays stay all-zero between uses: k_sel_clear)
            if (int rc = c->sel_bitmap.reserve(w_off * 4)) return rc;
            if (int rc = c->sel_chunks.reserve(c_off * 4)) return rc;
            if (int rc = c->sel_prefix.reserve((size_t)LDW_SPAN_MAX * SEL_MAX_SUPER * 4)) return rc;
            LDW_HIP(hipMemsetAsync(c->sel_bitmap.p, 0, c->sel_bitmap.cap, c->stream));
            LDW_HIP(hipMemsetAsync(c->sel_chunks.p, 0, c->sel_chunks.cap, c->stream));
            LDW_HIP(hipMemsetAsync(c->sel_prefix.p, 0, c->sel_prefix.cap, c->stream));
        }
        if (int rc = ensure_links_capacity(c, c->n_sr, c->n_lr + m_sum)) return rc;
        for (int k = 0; k < hb.span; ++k) {
            const ldw::PickOut *hp = &picks[k];
            if (hp->n > 0) update_guess(c, false, hp, false);
            ldw::BlockTrace &tr = c->trace[(size_t)hb.blk_no + k];
            tr = ldw::BlockTrace();
            tr.guess = hb.guess;
            tr.B_true = hp->B_true;
            tr.path = 4;   // span
            tr.n_cand = (long long)hp->n_cand;
            S.ck[k] = hb.sseg[k].ckey;
            S.cv[k] = hb.sseg[k].cval;
            S.pick[k] = reinterpret_cast<ldw::PickOut *>(reinterpret_cast<char *>(sl.pick[s]) + (size_t)k * PICK_STRIDE);
            S.idx_t[k] = reinterpret_cast<const int32_t *>(d + hb.o_idx_t) + hb.seg_start[k];
            S.bitmap[k] = c->sel_bitmap.as<uint32_t>() + woff[k];
            S.chunks[k] = c->sel_chunks.as<uint32_t>() + coff[k];
            S.supers[k] = c->sel_prefix.as<uint32_t>() + (size_t)k * SEL_MAX_SUPER;
            DA.n_sr[k] = hb.seg_n_sr[k];
        }
        const unsigned gridm = (unsigned)std::max<long long>(1, (m_max + 255) / 256);
        hipLaunchKernelGGL(k_sel_thresh_span, dim3(1, (unsigned)hb.span), dim3(1024), 0, c->stream, S);
        hipLaunchKernelGGL(k_sel_mark_span, dim3(gridm, (unsigned)hb.span), dim3(256), 0, c->stream, S);
        hipLaunchKernelGGL(k_sel_scatter_span, dim3(gridm, (unsigned)hb.span), dim3(256), 0, c->stream, S, reinterpret_cast<const int32_t *>(d + hb.o_idx_f), (int)hb.nf,
                           sl.lr_count, c->lr_a.as<int32_t>(), c->lr_b.as<int32_t>(), c->lr_mi.as<double>());
        hipLaunchKernelGGL(k_sel_clear_span, dim3(gridm, (unsigned)hb.span), dim3(256), 0, c->stream, S);
        hipLaunchKernelGGL(k_span_done, dim3(1), dim3(64), 0, c->stream, S, DA, sl.lr_count, sl.stats_i + hb.blk_no * 3, sl.stats_d + hb.blk_no);
        LDW_HIP(hipGetLastError());
        c->n_lr += m_sum;   // upper bound; the exact value is *lr_count
    }
    for (int k = 0; k < hb.span && !batched; ++k) {
        const ldw::PickOut *hp = &picks[k];
        const bool missed = hp->n > 0 && !hp->spec_ok;
        if (missed) {
            ++c->spec_misses;
            ++c->span_fallbacks;
            note_overflow(c, hp->over);
            // (the redo reads the exact row count of everything before it: the selections of the span's earlier segments are queued, not counted yet)
            LDW_HIP(hipMemcpyAsync(c->pin_lrc, sl.lr_count, 8, hipMemcpyDeviceToHost, c->stream));
            LDW_HIP(hipEventRecord(c->ev_lrc, c->stream));
            c->lrc_recorded = true;
            if (int rc = run_block_alone(c, hb, k, p, sl, true)) return rc;
            continue;
        }
        if (hp->n > 0) update_guess(c, false, hp, false);
        {
            ldw::BlockTrace &tr = c->trace[(size_t)hb.blk_no + k];
            tr = ldw::BlockTrace();
            tr.guess = hb.guess;
            tr.B_true = hp->B_true;
            tr.path = 4;   // span
            tr.n_cand = (long long)hp->n_cand;
        }
        LDW_REQUIRE((size_t)hp->n_cand <= hb.cand_cap, LDW_ERR_STATE, "span segment %d lists %llu candidates, capacity %zu", k, (unsigned long long)hp->n_cand, hb.cand_cap);
        SelIn S;
        S.m = (int64_t)hp->n_cand;
        S.nf = hb.nf;
        S.nt = hb.seg_nt[k];
        S.blk_no = hb.blk_no + k;
        S.n_sr_blk = hb.seg_n_sr[k];
        S.ck = hb.sseg[k].ckey;
        S.cv = hb.sseg[k].cval;
        S.pick = reinterpret_cast<ldw::PickOut *>(reinterpret_cast<char *>(sl.pick[s]) + (size_t)k * PICK_STRIDE);
        S.idx_f = reinterpret_cast<const int32_t *>(d + hb.o_idx_f);
        S.idx_t = reinterpret_cast<const int32_t *>(d + hb.o_idx_t) + hb.seg_start[k];
        if (int rc = select_rows(c, S, true, sl)) return rc;
    }
    LDW_HIP(hipMemcpyAsync(c->pin_lrc, sl.lr_count, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipEventRecord(c->ev_lrc, c->stream));
    c->lrc_recorded = true;
    LDW_HIP(hipEventRecord(c->ev_pool[(size_t)hb.blk_no * EVB + 3], c->stream));
    LDW_HIP(hipEventRecord(c->ev_done[s], c->stream));
    c->done_recorded[s] = true;
    return LDW_OK;
}

// Cold start: a bucket guess for a block kind from a LATTICE SAMPLE of one of its blocks.  Every PROBE_STRIDE-th SNP of both sides
// (about 2000 per side, spanning the whole block, so the sample has the block's own mix of distances) is run through the plain
// path — full-limb GEMM of the sample's rows, fp64 MI of every sampled pair, histogram of the long-range ones, bucket pick — and the
// bucket that holds the type-7 rank of the SAMPLE becomes the guess, minus a margin for the sampling noise (~400-800 pairs in the
// tail: +-0.3 bucket) and for the block-to-block drift.  ~0.2 ms, nothing is emitted, no table or counter of the pass is touched.
// Without it the first block of a pass (and of every rank of a multi-GPU pass) takes the non-speculative path — 5-limb GEMM of the
// whole block, fp64 MI of all 1e8 pairs, radix sorts: 3x the time of a speculative block — and the next one cannot overlap it.
// A guess that turns out too high costs what it always costs: the block is redone non-speculatively (spec_misses).
constexpr int64_t PROBE_SIDE = 2048;          // sampled SNPs per side
constexpr int64_t PROBE_MIN_PAIRS = 16000000; // smaller blocks are cheap enough without a guess
constexpr int PROBE_MARGIN = 6;

// r04: in two halves, so that the probes of both kinds are queued back to back and waited for ONCE (1.25 ms of a cold pass went into two
// prep / upload / run / wait round trips): probe `which` (0, 1) uses the device buffers of pipeline slot `which` and its own part of
// the last slot's pinned staging buffer.
struct Probe {
    HostBlock hb;
    int kind = 0, which = 0;
    bool queued = false;
};
int probe_enqueue(ldw_ctx *c, const int32_t *fi, int64_t nf, const int32_t *ti, int64_t nt, const ldw_mi_params *p, const SmallLayout &sl, int kind, int which,
                  size_t pin_base, Probe &P) {
    const int64_t stride = std::max<int64_t>(1, std::max(nf, nt) / PROBE_SIDE);
    std::vector<int32_t> sf, st;
    for (int64_t k = 0; k < nf; k += stride) sf.push_back(fi[k]);
    for (int64_t k = 0; k < nt; k += stride) st.push_back(ti[k]);
    ldw_mi_params q = *p;
    q.keep_sr = 0;
    HostBlock &hb = P.hb;
    P.kind = kind;
    P.which = which;
    P.queued = false;
    // (host staging of the LAST slot: the helper threads of ldw_mi_all_pairs are already building the first blocks' lists in the others)
    constexpr int PS = LDW_NSLOT - 1;
    if (int rc = prep_block(c, sf.data(), (int64_t)sf.size(), st.data(), (int64_t)st.size(), &q, PS, 0, hb, nullptr, -1, false, pin_base)) return rc;
    hb.slot = which;      // the DEVICE side of the probe is slot `which`'s (the blocks are submitted after the probes)
    hb.lo.slot = which;
    hb.stage_base = 0;    // (its device image starts its slot's staging buffer)
    if (hb.n_lr_total < 100000) return LDW_OK;   // too few long-range pairs in the sample to say anything
    if (int rc = c->dstage[which].reserve(hb.total)) return rc;
    LDW_HIP(hipMemcpyAsync(c->dstage[which].p, static_cast<const char *>(c->pin[PS]) + pin_base, hb.total, hipMemcpyHostToDevice, c->stream));
    fill_dev_ptrs(c, hb);
    if (int rc = c->hist[which].reserve((size_t)NBINS * 8)) return rc;
    if (int rc = make_emit_args(c, hb, &q, sl, -1)) return rc;
    hb.E.write_dense = 0;   // nothing reads the sample's MI values: only the histogram of the long-range ones
    LDW_HIP(hipMemsetAsync(c->hist[which].p, 0, (size_t)NBINS * 8, c->stream));
    LDW_HIP(hipMemsetAsync(sl.pick[which], 0, sizeof(ldw::PickOut), c->stream));
    // (in the CALLER's reading of RXY: under quirk Q1 the scrambled RXY — r of two other SNPs — lifts 3-state x 3-state pairs into the tail of an
    // off-diagonal block; a sample evaluated with the intended RXY sat 15 buckets = 7.5 % below the block's own threshold)
    if (int rc = launch_block_mi(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, hb.E, c->ev, 3, &gx(c, which), nullptr, c->hist[which].as<unsigned long long>()))
        return rc;
    hipLaunchKernelGGL(k_pick_bucket, dim3(1), dim3(256), 0, c->stream, c->hist[which].as<unsigned long long>(), p->lr_retain_links, p->lr_links_approx, -1,
                       (long long)hb.n_lr_total, sl.pick[which], (const unsigned int *)nullptr, 0u);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(c->pin_pick[which], sl.pick[which], sizeof(ldw::PickOut), hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemsetAsync(sl.pick[which], 0, sizeof(ldw::PickOut), c->stream));
    P.queued = true;
    return LDW_OK;
}
// after hipStreamSynchronize(c->stream)
void probe_collect(ldw_ctx *c, const Probe &P) {
    if (!P.queued) return;
    const ldw::PickOut *hp = static_cast<const ldw::PickOut *>(c->pin_pick[P.which]);
    if (hp->n > 0 && hp->B_true < NBINS) {
        const int g = hp->B_true - PROBE_MARGIN;
        c->spec_B_next[P.kind] = g > 0 ? g : 0;
        c->spec_hist_n[P.kind] = 0;
        c->spec_probed[P.kind] = true;
        ++c->probe_blocks;
        static const bool trace_on = getenv("LDW_BLOCK_TRACE") != nullptr;
        if (trace_on) fprintf(stderr, "[ldw probe] kind %d: sample %lld x %lld, %lld long-range pairs, bucket %d -> guess %d\n", P.kind, (long long)P.hb.nf, (long long)P.hb.nt,
                              (long long)P.hb.n_lr_total, hp->B_true, c->spec_B_next[P.kind]);
    }
}

// whether the next block can be submitted before the current one is finished: the fused path needs a bucket guess
bool can_submit_early(ldw_ctx *c, const HostBlock &hb, const ldw_mi_params *p) {
    if (!c->overlap) return false;
    if (!p->sr_only && !speculation_pays(c, p)) return true;   // (plain blocks need no guess)
    // no bucket guess for this kind of block yet (the first blocks of a cold pass): the block in flight is about to provide one —
    // submitted now, this block would take the non-speculative path (full 5-limb GEMM, fp64 for every pair: ~4 ms more)
    if (c->engine == LDW_ENGINE_MFMA && !p->sr_only && c->screen && c->spec_B_next[hb.diag ? 1 : 0] < 0) return false;
    if (c->engine != LDW_ENGINE_MFMA || !c->fused || c->nlimbs > 5 || p->sr_only) return true;
    return c->spec_B_next[hb.diag ? 1 : 0] >= 0;
}

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
        if (int rc = c->panel[s][0].reserve(RF * (size_t)KW * 8)) return rc;
        if (int rc = c->panel[s][1].reserve(RT * (size_t)KW * 8)) return rc;
        if (int rc = c->Gapx[s].reserve(RF * RT * 4)) return rc;
        if (int rc = c->apx_units[s].reserve(64 + 2 * n_units * 8 + 64)) return rc;
        if (int rc = c->apx_packs[s].reserve(2 * o_cph + 2 * o_rph + ((size_t)blk + (size_t)nt) * 4 + 1024)) return rc;
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
    int64_t h_viol = 0, h_apx[2] = {0, 0};
    LDW_HIP(hipMemcpyAsync(&h_lr, sl.lr_count, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(&h_viol, sl.lr_count + 1, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(h_apx, sl.lr_count + 2, 16, hipMemcpyDeviceToHost, c->stream));
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
        if (int rc = ensure_links_capacity(c, total_sr + 1024, 0)) return rc;
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
    std::vector<int32_t> fi, ti;
    std::vector<ColInfo> cols;
    int64_t base = 0;
    ldw::DevBuf dcols;
    struct Rel {
        ldw::DevBuf &b;
        ~Rel() { b.release(); }
    } rel{dcols};
    for (int64_t b = 0; b < nblocks; ++b) {
        const int32_t fs = blocks[b * 4 + 0], fe = blocks[b * 4 + 1], ts = blocks[b * 4 + 2], te = blocks[b * 4 + 3];
        LDW_REQUIRE(fs >= 1 && fe >= fs && fe <= c->L && ts >= 1 && te >= ts && te <= c->L, LDW_ERR_ARG, "block %lld = (%d,%d,%d,%d) outside 1..%lld", (long long)b, fs, fe,
                    ts, te, (long long)c->L);
        const int64_t nf = fe - fs + 1, nt = te - ts + 1;
        fi.resize((size_t)nf);
        ti.resize((size_t)nt);
        for (int64_t k = 0; k < nf; ++k) fi[(size_t)k] = fs - 1 + (int32_t)k;
        for (int64_t k = 0; k < nt; ++k) ti[(size_t)k] = ts - 1 + (int32_t)k;
        const bool diag = fs == ts && fe == te;
        int64_t n_blk = 0;
        if (int rc = build_cols(c, fi.data(), nf, ti.data(), nt, diag, sr_dist, cols, n_blk)) return rc;
        if (n_blk > 0 && a_out && b_out) {
            LDW_REQUIRE(base + n_blk <= capacity, LDW_ERR_SIZE, "ldw_sr_pairs_fill: capacity %lld < %lld rows", (long long)capacity, (long long)(base + n_blk));
            LDW_HIP(hipStreamSynchronize(c->stream));   // (cols of the block before is still being read)
            if (int rc = dcols.reserve(cols.size() * sizeof(ColInfo))) return rc;
            LDW_HIP(hipMemcpyAsync(dcols.p, cols.data(), cols.size() * sizeof(ColInfo), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_sr_fill, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, c->stream, dcols.as<ColInfo>(), (int)nf, (int)nt, fs - 1, ts - 1, diag ? 1 : 0, base,
                               a_out, b_out);
            LDW_HIP(hipGetLastError());
        }
        base += n_blk;
    }
    LDW_HIP(hipStreamSynchronize(c->stream));
    *n_out = base;
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
    out[3] = 0;
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
    c->n_red = c->n_pool = 0;   // whatever was derived from the old table is stale
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
