// Fixed-point weighted co-occurrence GEMM on the CDNA4 i8 matrix cores.
//
//   G[t][f] = sum_s V_s * [row t carries its state in sequence s] * [row f carries its state in s]
//
// with V_s the per-sequence weight quantised to J balanced base-256 digits d_j(s) in [-128,127]
// (V_s = sum_j d_j(s) 256^j).  One indicator matrix M (bytes 0x00 / 0xFF) feeds both operands: the
// N-dim operand is (M & d_j) = d_j where the indicator is set, the M-dim operand is M itself read as a
// signed byte (-1), so each limb accumulates  -sum_s d_j(s)[t][f]  EXACTLY in int32, and
//   G = -sum_j acc_j * 256^j      (Horner in int64).
// J limbs share one pass over K: the operand fragments are read from LDS once and masked J times on the
// VALU (4 v_and per limb per fragment), which multiplies the arithmetic per staged byte by J.
//
// This replaces the reference's 25 `tcrossprod(tX, CSR(tY))` products per block
// (R/computePairwiseMI.R:270-298, :391) — and, with J = 1 and unit digits on a sequence-major one-hot
// matrix, the five `crossprod`s of R/performPopulationStuctureCorrection.R:49-74.
//
// Tiling: 128 x 128 rows per workgroup (8 waves, 2 x 4), 64 x 32 per wave = 2 x 1 MFMA tiles of
// v_mfma_i32_32x32x32_i8, K staged through LDS in KSTEP-byte steps (register-staged double buffer).
// LDS rows are KSTEP bytes; the 16-B slot index is XOR-swizzled with the bank-row number so that each
// ds_read_b128 lane group touches 16 distinct slots of the 256-B bank row.
#include "ldw_internal.h"

namespace ldw {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int SPR = KSTEP / 16;          // 16-B slots per LDS row
constexpr int RPB = 256 / KSTEP;         // LDS rows per 256-B bank row
constexpr int NKK = KSTEP / 32;          // MFMA k-steps per stage
constexpr int PIECES = TILE * SPR / 512; // 16-B pieces per thread per operand tile
static_assert(KSTEP == 64 || KSTEP == 128, "KSTEP must be 64 or 128");

__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * KSTEP + ((slot ^ ((row / RPB) & (SPR - 1))) << 4);
}

template <int J>
__global__ __launch_bounds__(512, 2) void gemm_limb_kernel(const uint8_t *__restrict__ Mbase, int64_t Kpad,
                                                           const int32_t *__restrict__ rowlist_t,
                                                           const int32_t *__restrict__ rowlist_f,
                                                           const int8_t *__restrict__ digits,  // [J][Kpad]
                                                           int64_t *__restrict__ G, int RFpad, int lower_only,
                                                           int shift_bits, int accumulate) {
    const int bx = blockIdx.x, by = blockIdx.y;  // bx: from-side (lanes / N-dim), by: to-side (M-dim)
    if (lower_only && bx < by) return;

    __shared__ __attribute__((aligned(16))) uint8_t sT[2][TILE * KSTEP];
    __shared__ __attribute__((aligned(16))) uint8_t sF[2][TILE * KSTEP];
    __shared__ __attribute__((aligned(16))) int8_t sD[2][8 * KSTEP];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;  // 8 waves: 2 (to-side, M-dim) x 4 (from-side, N-dim)
    const int wm = wave >> 2, wn = wave & 3;

    // staging assignment: PIECES 16-B pieces per operand tile per thread
    const int srow = tid / SPR;  // row of the first piece; piece p adds p * (512 / SPR) rows
    const int sslot = tid % SPR;
    const uint8_t *gT[PIECES], *gF[PIECES];
    int wofs[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const int row = srow + p * (512 / SPR);
        gT[p] = Mbase + (int64_t)rowlist_t[by * TILE + row] * Kpad + sslot * 16;
        gF[p] = Mbase + (int64_t)rowlist_f[bx * TILE + row] * Kpad + sslot * 16;
        wofs[p] = lds_off(row, sslot);
    }
    // digits: J*KSTEP bytes per stage = J*SPR pieces of 16 B, loaded by the first J*SPR threads
    const bool dig_loader = tid < J * SPR;
    const int8_t *gD = digits + (int64_t)(tid / SPR) * Kpad + (tid % SPR) * 16;

    v16i acc[J][2];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][m][e] = 0;

    const int nk = (int)(Kpad / KSTEP);
    v4i rT[PIECES], rF[PIECES], rD;
    rD = v4i{0, 0, 0, 0};
    // prologue: stage 0
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        rT[p] = *reinterpret_cast<const v4i *>(gT[p]);
        rF[p] = *reinterpret_cast<const v4i *>(gF[p]);
    }
    if (dig_loader) rD = *reinterpret_cast<const v4i *>(gD);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        *reinterpret_cast<v4i *>(&sT[0][wofs[p]]) = rT[p];
        *reinterpret_cast<v4i *>(&sF[0][wofs[p]]) = rF[p];
    }
    if (dig_loader) *reinterpret_cast<v4i *>(&sD[0][tid * 16]) = rD;
    __syncthreads();

    const int frow = lane & 31;
    const int fh = lane >> 5;
    const int arow0 = wm * 64 + frow, arow1 = arow0 + 32, brow = wn * 32 + frow;

    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
#ifndef LDW_ABL_NOSTAGE
        if (ks + 1 < nk) {  // issue next stage's global loads early
            const int64_t ko = (int64_t)(ks + 1) * KSTEP;
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                rT[p] = *reinterpret_cast<const v4i *>(gT[p] + ko);
                rF[p] = *reinterpret_cast<const v4i *>(gF[p] + ko);
            }
            if (dig_loader) rD = *reinterpret_cast<const v4i *>(gD + ko);
        }
#endif
        // fragments of k-step kk+1 are read while the MFMAs of k-step kk run
        v4i a0[2], a1[2], b[2];
        a0[0] = *reinterpret_cast<const v4i *>(&sT[cur][lds_off(arow0, fh)]);
        a1[0] = *reinterpret_cast<const v4i *>(&sT[cur][lds_off(arow1, fh)]);
        b[0] = *reinterpret_cast<const v4i *>(&sF[cur][lds_off(brow, fh)]);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int slot = kk * 2 + fh;
            if (kk + 1 < NKK) {
                a0[(kk + 1) & 1] = *reinterpret_cast<const v4i *>(&sT[cur][lds_off(arow0, slot + 2)]);
                a1[(kk + 1) & 1] = *reinterpret_cast<const v4i *>(&sT[cur][lds_off(arow1, slot + 2)]);
                b[(kk + 1) & 1] = *reinterpret_cast<const v4i *>(&sF[cur][lds_off(brow, slot + 2)]);
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
#ifdef LDW_ABL_NODIG
                const v4i d = v4i{0x01010101 * (j + 1), 0x01010101, 0x01010101, 0x01010101};
#else
                const v4i d = *reinterpret_cast<const v4i *>(&sD[cur][j * KSTEP + slot * 16]);
#endif
                const v4i bm = b[kk & 1] & d;  // digit where the from-side indicator is set
                acc[j][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0[kk & 1], bm, acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1[kk & 1], bm, acc[j][1], 0, 0, 0);
            }
        }
#ifndef LDW_ABL_NOSTAGE
        if (ks + 1 < nk) {
            const int nxt = cur ^ 1;
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                *reinterpret_cast<v4i *>(&sT[nxt][wofs[p]]) = rT[p];
                *reinterpret_cast<v4i *>(&sF[nxt][wofs[p]]) = rF[p];
            }
            if (dig_loader) *reinterpret_cast<v4i *>(&sD[nxt][tid * 16]) = rD;
        }
#endif
#ifndef LDW_ABL_NOBARRIER
        __syncthreads();
#endif
    }

    // epilogue: Horner over limbs in int64, negate (the un-masked operand is -1), store.  C/D layout of
    // the 32x32 MFMA: col (N-dim, from-side row) = lane & 31, row (M-dim, to-side row) =
    // (e&3) + 8*(e>>2) + 4*(lane>>5).
    const int fcol = bx * TILE + wn * 32 + frow;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int trow = by * TILE + wm * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            int64_t gsum = (int64_t)acc[J - 1][m][e];
#pragma unroll
            for (int j = J - 2; j >= 0; --j) gsum = gsum * 256 + (int64_t)acc[j][m][e];
            int64_t *dst = &G[(int64_t)trow * RFpad + fcol];
            const int64_t val = -(gsum << shift_bits);
            *dst = accumulate ? (*dst + val) : val;
        }
    }
}

int launch_gemm(ldw_ctx *ctx, const int32_t *rowlist_t, int RTpad, const int32_t *rowlist_f, int RFpad,
                int64_t *G, int nlimbs, const int8_t *digits, const uint8_t *Mbase, int64_t Kpad,
                int lower_only, int accumulate) {
    LDW_REQUIRE(RTpad % TILE == 0 && RFpad % TILE == 0 && Kpad % KSTEP == 0 && Kpad > 0, LDW_ERR_ARG,
                "launch_gemm: tile padding violated (RT %d RF %d K %lld)", RTpad, RFpad, (long long)Kpad);
    LDW_REQUIRE(nlimbs >= 1 && nlimbs <= 6, LDW_ERR_ARG, "launch_gemm: nlimbs %d out of range", nlimbs);
    dim3 grid(RFpad / TILE, RTpad / TILE), block(512);
    // up to 5 limbs share one pass over K (160 accumulator registers per lane); 6 limbs run as 3 + 3
    int done = 0;
    while (done < nlimbs) {
        const int J = (nlimbs == 6) ? 3 : nlimbs;
        const int8_t *dg = digits + (int64_t)done * Kpad;
        const int shift = 8 * done, accum = (done > 0) || accumulate;
#define LDW_LAUNCH_J(JJ)                                                                              \
    case JJ:                                                                                          \
        hipLaunchKernelGGL(gemm_limb_kernel<JJ>, grid, block, 0, ctx->stream, Mbase, Kpad, rowlist_t, \
                           rowlist_f, dg, G, RFpad, lower_only, shift, accum);                        \
        break;
        switch (J) {
            LDW_LAUNCH_J(1)
            LDW_LAUNCH_J(2)
            LDW_LAUNCH_J(3)
            LDW_LAUNCH_J(4)
            LDW_LAUNCH_J(5)
        }
#undef LDW_LAUNCH_J
        LDW_HIP(hipGetLastError());
        done += J;
    }
    return LDW_OK;
}

}  // namespace ldw
