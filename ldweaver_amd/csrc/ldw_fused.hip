// Fused co-occurrence GEMM + MI epilogue: the fixed-point joint sums of a 128 x 64 row tile never leave the chip.
//
//   gemm_bits_kernel  (ldw_gemm_bits.hip)  writes G (8 B per row pair, ~1 GB per 10k x 10k block) to HBM and
//   k_mi_epilogue     (ldw_mi.hip)         reads it back; the first is bound by the matrix cores (+ its operand
//   expansion), the second by fp64 VALU issue, and because a GEMM wave owns half of a SIMD's register file the two
//   kernels cannot share a CU: run back to back they add up (C4: 1.70 + 0.92 ms per block).
//
// Here every wave finishes its 64 (to side) x 32 (from side) sub-tile, Horner-combines the limb accumulators, parks
// the int64 sums in a WAVE-PRIVATE LDS tile (aliased onto the operand buffers of the K loop, which are dead by then)
// and runs the epilogue of exactly those pairs: lane = to-side SNP, loop = from-side SNP (wave-uniform, its constants
// staged in LDS before the K loop).  Two workgroups are resident per CU, so the VALU-bound epilogue of one overlaps
// the MFMA loop of the other: that overlap is the point of the fusion, the saved G round trip comes on top.
//
// What makes this possible is the ORDER of the row lists (build_side in ldw_mi.hip): SNPs are grouped by slot-count
// class (1, 2 or 4 indicator rows after padding) and classes start on 32-row boundaries, so a 32 x 32 MFMA tile holds
// whole SNPs of ONE class on either side, no SNP straddles a wave's sub-tile, and nearly every wave runs one of the
// four straight-line variants (1|2 x 1|2 slots) without per-cell predication.
//
// Outputs are sparse, so only the speculative mode of the link selection is supported (EmitArgs::spec_B >= 0):
// short-range pairs go straight to their final rows of the sr table, long-range pairs at or above the guessed
// histogram bucket are appended to the candidate list and counted with global atomics (they are rare), everything
// below is accounted for analytically by k_pick_bucket.  Blocks without a guess and blocks whose guess turns out too
// high run the unfused pair of kernels.
#include "ldw_internal.h"
#include "ldw_dev.h"
#include "ldw_epi.h"
#include "ldw_gemm_tile.h"

namespace ldw {

constexpr int GS = 33;  // row stride (int64 words) of a wave's G tile in LDS: odd, so lanes on consecutive rows hit distinct bank pairs

// from-side SNP starting at a row position of this workgroup's tile (a_loc < 0: none starts here)
struct FromMeta {
    int32_t a_loc, sa;
    uint32_t ma;
    int32_t pad;
    double ra, rta;
    int64_t pa[5];
    double pXd[5];   // marginals as doubles / floats: converted once per workgroup, not once per wave and SNP
    float pXf[5];
    int32_t pad2;
};

template <int J>
union FusedSmem {
    GemmSmem<J> k;
    int64_t g[4][64 * GS];
};

__device__ __forceinline__ void load_from(const FromMeta &fmeta, double scale, RowSide &R, int &a_loc) {
    a_loc = __builtin_amdgcn_readfirstlane(fmeta.a_loc);
    R.sa = __builtin_amdgcn_readfirstlane(fmeta.sa);
    R.ma = (uint32_t)__builtin_amdgcn_readfirstlane((int)fmeta.ma);
    R.na = (int)(R.ma & 7);
    R.ra0 = 0;
    R.ra = fmeta.ra;
    R.rta = fmeta.rta;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        R.pa[k] = fmeta.pa[k];
        R.pXd[k] = fmeta.pXd[k];
        R.pXf[k] = fmeta.pXf[k];
    }
}

__device__ __forceinline__ bool slots_full(uint32_t m, int n) {
    return (int)(m & 7) == n && (((m >> 3) & ((2u << n) - 1u)) == ((2u << n) - 1u));
}

// U consecutive from-side SNPs of one class (NA slots) against this lane's to-side SNP (NB slots), every slot flagged,
// with the fp32 screen on.  A single wave per SIMD runs this code next to the K loop of the neighbouring workgroup, so
// nothing hides its latencies but its own instruction-level parallelism: the U joint tables are built and screened in
// ONE branch-free stretch (U independent chains of LDS reads, integer subtractions, v_log_f32), and only then — rarely —
// the fp64 value follows.  `rxy_q1` selects RXY = r[idx_f[b]] r[idx_t[a]] / 4 (quirk Q1 on a square block) over r_a r_b / 4;
// non-square blocks in reference mode take the generic path.  Whether a pair is short-range is read off the interval
// list of THIS lane's SNP even when a diagonal block meets the pair in mirrored roles: the relation is symmetric.
// Returns false if a SNP of the group does not have all of its NA slots flagged (the caller then takes them one by one).
template <int NA, int NB, int U>
__device__ __forceinline__ bool fused_group(const EpiArgs &A, const FromMeta *fg, const ColMeta &M, const int64_t *gl0, bool t_ok,
                                            int tpos, int fpos0, int b_loc, bool rxy_q1, unsigned long long *__restrict__ ghist) {
    const int lower_only = A.E.lower_only;
    RowSide R[U];
    int a_loc[U];
    bool ok = true;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        load_from(fg[u * NA], A.scale, R[u], a_loc[u]);
        ok = ok && (a_loc[u] < 0 || slots_full(R[u].ma, NA));
    }
    if (!ok) return false;
    FullCells<NA, NB> C[U];
    double rxy[U];
    bool act[U], need[U], is_sr[U];
    bool any = false;
    const bool test_sr = A.E.any_sr != 0, keep_sr = A.E.keep_sr != 0, do_lr = A.E.do_lr != 0;
    const float lo = (float)A.E.spec_lo - A.E.scr_eps;   // rounding of the difference is far inside the margin
#pragma unroll
    for (int u = 0; u < U; ++u) {
        act[u] = t_ok && a_loc[u] >= 0 && (lower_only ? fpos0 + u * NA > tpos : a_loc[u] != b_loc);
        full_cells<NA, NB>(R[u], M, gacc_plain(gl0 + u * NA, 1, GS), C[u]);
        rxy[u] = (rxy_q1 ? M.rq * R[u].rta : R[u].ra * M.rb) * 0.25;
        is_sr[u] = test_sr && col_is_sr(M.ci, a_loc[u]);
        const float ms = full_cells_screen<NA, NB>(A, R[u], M, rxy[u], C[u]);
        need[u] = act[u] && (is_sr[u] ? keep_sr : (do_lr && ms >= lo));
        any = any || need[u];
    }
    if (A.E.scr_mode == 1 && __ballot(any) == 0ull) return true;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (A.E.scr_mode == 1 && __ballot(need[u]) == 0ull) continue;
        const double mi = full_cells_mi<NA, NB>(A, R[u], M, rxy[u], C[u]);
        if (A.E.scr_mode == 2) {
            const bool w = is_sr[u] ? keep_sr : (do_lr && mi >= A.E.spec_lo);
            if (act[u] && !need[u] && w) atomicAdd(A.E.scr_viol, 1ull);
            need[u] = act[u];
        }
        if (need[u]) {
            if (lower_only && a_loc[u] < b_loc) {   // diagonal block, pair met in mirrored roles (row order is by class, not by index)
                const ColInfo ci = A.E.cols[a_loc[u]];
                emit_pair_spec(A.E, ci, b_loc, a_loc[u], M.sb, R[u].sa, mi, ghist);
            } else {
                emit_pair_spec(A.E, M.ci, a_loc[u], b_loc, R[u].sa, M.sb, mi, ghist);
            }
        }
    }
    return true;
}

template <int J>
__global__ __launch_bounds__(256, 2) void gemm_mi_fused_kernel(FusedArgs F) {
    const int bx = blockIdx.x, by = blockIdx.y;  // bx: from-side tile of 64 rows, by: to-side tile of 128 rows
    const int lower_only = F.A.E.lower_only;
    if (lower_only && bx * TILE_F4 + TILE_F4 - 1 < by * TILE) return;

    __shared__ FusedSmem<J> U;
    __shared__ FromMeta fm[TILE_F4];
    GemmSmem<J> &S = U.k;
    const bool square = F.A.nf == F.A.nt;

    // from-side constants of this tile: read after the K loop (dozens of barriers later)
    if (threadIdx.x < TILE_F4) {
        FromMeta m;
        m.a_loc = F.pos_f[bx * TILE_F4 + threadIdx.x];
        const int a = m.a_loc < 0 ? 0 : m.a_loc;
        m.sa = F.A.idx_f[a];
        m.ma = F.A.slot_meta[m.sa];
        m.pad = 0;
        m.ra = F.A.r[m.sa];
        m.rta = square ? F.A.r[F.A.idx_t[a]] : 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            m.pa[i] = F.A.slot_pfix[(int64_t)m.sa * 5 + i];
            m.pXd[i] = (double)m.pa[i] * F.A.scale;
            m.pXf[i] = (float)m.pXd[i];
        }
        m.pad2 = 0;
        fm[threadIdx.x] = m;
    }

    const uint64_t *__restrict__ Mbits = F.Mbits;
    const int64_t KW = F.KW, Kpad = F.Kpad;
    const int32_t *__restrict__ rowlist_t = F.rowlist_t;
    const int32_t *__restrict__ rowlist_f = F.rowlist_f;
    const int8_t *__restrict__ digits = F.digits;
#include "ldw_gemm_kloop.inc"

    // ---- limbs -> int64 fixed-point sums -> this wave's private LDS tile [64 to-rows][32 from-rows] ----
    int64_t *gw = U.g[wave];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int64_t gsum = (int64_t)acc[J - 1][m][e];
#pragma unroll
            for (int j = J - 2; j >= 0; --j) gsum = gsum * 256 + (int64_t)acc[j][m][e];
            gw[(m * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * GS + frow] = gsum;
        }
    }
    // the tile is written and read by this wave only: LDS operations of one wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- epilogue of the wave's sub-tile: lane = to-side SNP, loop = from-side SNP ----
    const EpiArgs &A = F.A;
    const int half = lane >> 5, l32 = lane & 31;
    const int ct = (int)F.cls_t[by * 4 + wm * 2 + half];          // slot-count class of this half-wave's 32-row group
    const int tin_raw = l32 * ct;
    const bool t_in = tin_raw < 32;
    const int tin = t_in ? tin_raw : 0;
    const int tpos = by * TILE + wm * 64 + half * 32 + tin;       // row position in the to-side list
    const int b_raw = t_in ? F.pos_t[tpos] : -1;
    const bool t_ok = b_raw >= 0;
    if (__ballot(t_ok) == 0ull) return;
    const int b_loc = t_ok ? b_raw : 0;
    ColMeta M;
    M.sb = A.idx_t[b_loc];
    M.mb = A.slot_meta[M.sb];
    M.rb0 = 0;
    M.bl = b_loc;
    M.rb = A.r[M.sb];
    M.rq = square ? A.r[A.idx_f[b_loc]] : 0.0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        M.pb[j] = A.slot_pfix[(int64_t)M.sb * 5 + j];
        M.pYd[j] = (double)M.pb[j] * A.scale;
        M.pYf[j] = (float)M.pYd[j];
    }
    M.pad2 = 0;
    M.ci = A.E.cols[b_loc];
    const int nb = (int)(M.mb & 7);
    const bool lane_full = nb == ct && (((M.mb >> 3) & ((2u << nb) - 1u)) == ((2u << nb) - 1u));
    const int ct0 = __builtin_amdgcn_readlane(ct, 0), ct1 = __builtin_amdgcn_readlane(ct, 32);
    const bool wave_fast = ct0 == ct1 && ct0 <= 2 && __ballot(t_ok && !lane_full) == 0ull;

    const int wv = __builtin_amdgcn_readfirstlane(wn);
    const int cf = (int)F.cls_f[bx * 2 + wv];
    const int nfrom = 32 / cf;
    const int64_t *gl = gw + (half * 32 + tin) * GS;
    // the grouped fast path needs the screen (speculative mode) and an RXY that is a plain product of two per-SNP values
    const bool rxy_q1 = A.quirk == LDW_QUIRK_REFERENCE;
    const bool group_ok = wave_fast && A.E.scr_mode != 0 && (square || !rxy_q1);
    constexpr int GU = 4;   // from-side SNPs per straight-line stretch (divides 32 / class)
    for (int i0 = 0; i0 < nfrom; i0 += GU) {
        const int fin0 = i0 * cf;
        const int fpos0 = bx * TILE_F4 + wv * 32 + fin0;
        if (group_ok) {
            const FromMeta *fg = &fm[wv * 32 + fin0];
            bool done;
            if (cf == 1) done = ct0 == 1 ? fused_group<1, 1, GU>(A, fg, M, gl + fin0, t_ok, tpos, fpos0, b_loc, rxy_q1, F.ghist)
                                         : fused_group<1, 2, GU>(A, fg, M, gl + fin0, t_ok, tpos, fpos0, b_loc, rxy_q1, F.ghist);
            else if (cf == 2) done = ct0 == 1 ? fused_group<2, 1, GU>(A, fg, M, gl + fin0, t_ok, tpos, fpos0, b_loc, rxy_q1, F.ghist)
                                              : fused_group<2, 2, GU>(A, fg, M, gl + fin0, t_ok, tpos, fpos0, b_loc, rxy_q1, F.ghist);
            else done = false;
            if (done) continue;
        }
        // generic path, one from-side SNP at a time: any slot counts, per-cell predication
        for (int i = i0; i < i0 + GU; ++i) {
            const int fin = i * cf;
            RowSide R;
            int a_loc;
            load_from(fm[wv * 32 + fin], A.scale, R, a_loc);
            if (a_loc < 0) continue;
            const int fpos = bx * TILE_F4 + wv * 32 + fin;
            const bool act = t_ok && (lower_only ? fpos > tpos : a_loc != b_loc);
            if (__ballot(act) == 0ull) continue;
            const double mi = pair_mi<4, 4>(A, R, M, a_loc, b_loc, square, gacc_plain(gl + fin, 1, GS));
            if (act) {
                if (lower_only && a_loc < b_loc) {   // diagonal block, pair met in mirrored roles (row order is by class, not by index)
                    const ColInfo ci = A.E.cols[a_loc];
                    emit_pair_spec(A.E, ci, b_loc, a_loc, M.sb, R.sa, mi, F.ghist);
                } else {
                    emit_pair_spec(A.E, M.ci, a_loc, b_loc, R.sa, M.sb, mi, F.ghist);
                }
            }
        }
    }
}

int launch_fused(ldw_ctx *ctx, const FusedArgs &F, int RFpad, int RTpad, int nlimbs, hipStream_t stream) {
    LDW_REQUIRE(RTpad % TILE == 0 && RFpad % TILE == 0 && F.KW > 0 && F.KW % 2 == 0, LDW_ERR_ARG,
                "launch_fused: padding violated (RT %d RF %d KW %lld)", RTpad, RFpad, (long long)F.KW);
    LDW_REQUIRE(nlimbs >= 1 && nlimbs <= 5, LDW_ERR_ARG, "launch_fused: nlimbs %d out of range", nlimbs);
    LDW_REQUIRE(F.A.E.cols && (!F.A.E.do_lr || F.A.E.spec_B >= 0), LDW_ERR_STATE, "launch_fused: needs the speculative selection mode");
    dim3 grid(RFpad / TILE_F4, RTpad / TILE), block(256);
    switch (nlimbs) {
    case 1: hipLaunchKernelGGL(gemm_mi_fused_kernel<1>, grid, block, 0, stream, F); break;
    case 2: hipLaunchKernelGGL(gemm_mi_fused_kernel<2>, grid, block, 0, stream, F); break;
    case 3: hipLaunchKernelGGL(gemm_mi_fused_kernel<3>, grid, block, 0, stream, F); break;
    case 4: hipLaunchKernelGGL(gemm_mi_fused_kernel<4>, grid, block, 0, stream, F); break;
    case 5: hipLaunchKernelGGL(gemm_mi_fused_kernel<5>, grid, block, 0, stream, F); break;
    }
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

}  // namespace ldw
