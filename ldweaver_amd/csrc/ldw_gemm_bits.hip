// Bit-packed fixed-point co-occurrence GEMM (MI path and Hamming weights).
//
//   G[t][f] = sum_s V_s [row t has bit s][row f has bit s],   V_s = sum_j d_j(s) 256^j   (J int8 digit limbs)
//
// The indicator matrix is stored as BITS (one uint64 per 64 sequences).  A first version kept it as bytes and was
// bound by what one CU can pull from L2 / Infinity Cache (16 KB per 64-sequence step and workgroup, ~33 GB/s per CU
// measured, i.e. ~2.1 ms per 11.6k x 11.6k x 5056 launch whatever the limb count), not by the matrix cores.  Bits
// cut the global and LDS traffic 8x: a whole 1024-sequence chunk of both operand tiles is 28 KB, loaded once per
// ~20k MFMA cycles, so the loop between two chunk barriers is pure ds_read_b64 -> expand -> mask -> MFMA with no
// global load and no barrier.
//
// Expansion, per 16 sequences (one MFMA operand fragment): two look-ups in a 256-entry LDS table (byte of bits ->
// 8 bytes of 0x01 for the M-dim operand, of 0xFF for the N-dim operand, built once per workgroup with
// (t * 0x204081) & 0x01010101 per nibble); the N-dim fragment is then ANDed with the J digit fragments.  Doing the
// expansion arithmetically (v_bfe, v_mul_u32_u24, v_and, ...: ~44 VALU per MFMA k-step and wave) made the loop
// VALU-issue bound (2.6 ms per C4 block); the look-ups bring it to 2.2 ms (63 % of the i8 peak).
//
// k-mapping: within a macro step of 128 sequences lane half h owns sequences [64h, 64h+64) — one 64-bit word —
// and MFMA k-step kk consumes its bits [16kk, 16kk+16); both operands and the digits use the same mapping, and
// a sum over sequences does not care about their order.
//
// Tiling: 128 (to side) x 64 (from side) rows per workgroup, 4 waves (2 x 2), 64 x 32 per wave = 2 MFMA tiles x J limbs.
#include "ldw_internal.h"
#include "ldw_gemm_tile.h"

namespace ldw {

// 256-thread workgroups (4 waves, 2 x 2), tile 128 (to side) x 64 (from side), 64 x 32 per wave.  Two workgroups are
// resident per CU (256 VGPRs per wave, 66.5 KB LDS each), so the prologue (LUT build, first chunk), the per-chunk
// barrier and the Horner/store tail of one overlap the MFMA loop of the other: 5 % faster than one 512-thread
// workgroup of 128 x 128 with the same wave tile.
template <int J>
__global__ __launch_bounds__(256, 2) void gemm_bits_kernel(const uint64_t *__restrict__ Mbits, int64_t KW,
                                                              const int32_t *__restrict__ rowlist_t,
                                                              const int32_t *__restrict__ rowlist_f,
                                                              const int8_t *__restrict__ digits, int64_t Kpad,
                                                              int64_t *__restrict__ G, int RFpad, int lower_only,
                                                              int shift_bits, int accumulate) {
    const int bx = blockIdx.x, by = blockIdx.y;  // bx: from-side tile of 64 rows, by: to-side tile of 128 rows
    if (lower_only && bx * TILE_F4 + TILE_F4 - 1 < by * TILE) return;

    __shared__ GemmSmem<J> S;
#include "ldw_gemm_kloop.inc"

    const int fcol = bx * TILE_F4 + wn * 32 + frow;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int trow = by * TILE + wm * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            int64_t gsum = (int64_t)acc[J - 1][m][e];
#pragma unroll
            for (int j = J - 2; j >= 0; --j) gsum = gsum * 256 + (int64_t)acc[j][m][e];
            int64_t *dst = &G[(int64_t)trow * RFpad + fcol];
            const int64_t val = gsum << shift_bits;
            *dst = accumulate ? (*dst + val) : val;
        }
    }
}

// bit rows: Mbits[row][w] bit i = (states[snp(row)][64 w + i] == state(row)); one thread per word
__global__ __launch_bounds__(256) void k_fill_rows_bits(const uint8_t *__restrict__ states, int64_t Npad,
                                                        const int32_t *__restrict__ rowinfo, int64_t KW,
                                                        uint64_t *__restrict__ Mbits) {
    const int64_t row = blockIdx.x;
    const int32_t info = rowinfo[row];
    const int64_t snp = info >> 3;
    const uint32_t st = (uint32_t)(info & 7);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(states + snp * Npad);
    for (int64_t w = threadIdx.x; w < KW; w += blockDim.x) {
        uint64_t bits = 0;
        if (w * 64 < Npad) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t x = src[w * 16 + q];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bits |= (uint64_t)(((x >> (8 * k)) & 0xFFu) == st) << (4 * q + k);
            }
        }
        Mbits[row * KW + w] = bits;
    }
}

int launch_gemm_bits(ldw_ctx *ctx, const uint64_t *Mbits, int64_t KW, const int32_t *rowlist_t, int RTpad, const int32_t *rowlist_f,
                     int RFpad, int64_t *G, int nlimbs, const int8_t *digits, int lower_only, hipStream_t stream) {
    if (!stream) stream = ctx->stream;
    const int64_t Kpad = KW * 64;
    LDW_REQUIRE(RTpad % TILE == 0 && RFpad % TILE == 0 && KW > 0 && KW % 2 == 0 && Kpad == KW * 64, LDW_ERR_ARG,
                "launch_gemm_bits: padding violated (RT %d RF %d KW %lld)", RTpad, RFpad, (long long)KW);
    LDW_REQUIRE(nlimbs >= 1 && nlimbs <= 6, LDW_ERR_ARG, "launch_gemm_bits: nlimbs %d out of range", nlimbs);
    dim3 grid(RFpad / TILE_F4, RTpad / TILE), block(256);
    int done = 0;
    while (done < nlimbs) {  // up to 5 limbs share one pass over K; 6 limbs run as 3 + 3
        const int J = (nlimbs == 6) ? 3 : nlimbs;
        const int8_t *dg = digits + (int64_t)done * Kpad;
        const int shift = 8 * done, accum = done > 0;
#define LDW_LAUNCH_B(JJ)                                                                                        \
    case JJ:                                                                                                    \
        hipLaunchKernelGGL(gemm_bits_kernel<JJ>, grid, block, 0, stream, Mbits, KW,                                 \
                           rowlist_t, rowlist_f, dg, Kpad, G, RFpad, lower_only, shift, accum);                 \
        break;
        switch (J) {
            LDW_LAUNCH_B(1)
            LDW_LAUNCH_B(2)
            LDW_LAUNCH_B(3)
            LDW_LAUNCH_B(4)
            LDW_LAUNCH_B(5)
        }
#undef LDW_LAUNCH_B
        LDW_HIP(hipGetLastError());
        done += J;
    }
    return LDW_OK;
}

int fill_rows_bits(ldw_ctx *c, const int32_t *d_rowinfo, int64_t R) {
    hipLaunchKernelGGL(k_fill_rows_bits, dim3((unsigned)R), dim3(256), 0, c->stream, c->states.as<uint8_t>(), c->Npad,
                       d_rowinfo, c->KW, c->Mbits.as<uint64_t>());
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

}  // namespace ldw
