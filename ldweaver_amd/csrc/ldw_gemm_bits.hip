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

namespace ldw {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BW_CHUNK = 16;        // 64-bit words of K per LDS chunk (1024 sequences)
constexpr int BROW = BW_CHUNK + 1;  // padded LDS row stride in words: odd, so 32 rows hit 32 distinct 8-B bank pairs

__device__ __forceinline__ v4i expand01(uint32_t b16) {
    v4i r;
    r[0] = (int)(__umul24(b16 & 0xFu, 0x204081u) & 0x01010101u);
    r[1] = (int)(__umul24((b16 >> 4) & 0xFu, 0x204081u) & 0x01010101u);
    r[2] = (int)(__umul24((b16 >> 8) & 0xFu, 0x204081u) & 0x01010101u);
    r[3] = (int)(__umul24((b16 >> 12) & 0xFu, 0x204081u) & 0x01010101u);
    return r;
}

// 256-thread workgroups (4 waves, 2 x 2), tile 128 (to side) x 64 (from side), 64 x 32 per wave.  Two workgroups are
// resident per CU (256 VGPRs per wave, 66.5 KB LDS each), so the prologue (LUT build, first chunk), the per-chunk
// barrier and the Horner/store tail of one overlap the MFMA loop of the other: 5 % faster than one 512-thread
// workgroup of 128 x 128 with the same wave tile.
constexpr int TILE_F4 = 64;
template <int J>
__global__ __launch_bounds__(256, 2) void gemm_bits_kernel(const uint64_t *__restrict__ Mbits, int64_t KW,
                                                              const int32_t *__restrict__ rowlist_t,
                                                              const int32_t *__restrict__ rowlist_f,
                                                              const int8_t *__restrict__ digits, int64_t Kpad,
                                                              int64_t *__restrict__ G, int RFpad, int lower_only,
                                                              int shift_bits, int accumulate) {
    const int bx = blockIdx.x, by = blockIdx.y;  // bx: from-side tile of 64 rows, by: to-side tile of 128 rows
    if (lower_only && bx * TILE_F4 + TILE_F4 - 1 < by * TILE) return;

    __shared__ __attribute__((aligned(16))) uint64_t sT[2][TILE * BROW];
    __shared__ __attribute__((aligned(16))) uint64_t sF[2][TILE_F4 * BROW];
    __shared__ __attribute__((aligned(16))) int8_t sD[2][J * BW_CHUNK * 64];
    __shared__ uint64_t lut01[256], lutFF[256];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // staging: to side 128 rows x 8 word pairs = 1024 pieces (4 per thread), from side 64 x 8 = 512 (2 per thread)
    const uint64_t *gT[4], *gF[2];
    int woT[4], woF[2], wpT[4], wpF[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = tid + 256 * p;
        const int row = idx >> 3;
        wpT[p] = idx & 7;
        gT[p] = Mbits + (int64_t)rowlist_t[by * TILE + row] * KW + 2 * wpT[p];
        woT[p] = row * BROW + 2 * wpT[p];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = tid + 256 * p;
        const int row = idx >> 3;
        wpF[p] = idx & 7;
        gF[p] = Mbits + (int64_t)rowlist_f[bx * TILE_F4 + row] * KW + 2 * wpF[p];
        woF[p] = row * BROW + 2 * wpF[p];
    }
    // digits of a chunk: J*64 pieces of 16 B over 256 threads
    constexpr int DP = (J * 64 + 255) / 256;
    const int8_t *gD[DP];
    int dOfs[DP], dq[DP];
    bool dOn[DP];
#pragma unroll
    for (int p = 0; p < DP; ++p) {
        const int idx = tid + 256 * p;
        dOn[p] = idx < J * 64;
        const int dj = idx >> 6;
        dq[p] = idx & 63;
        gD[p] = digits + (int64_t)(dOn[p] ? dj : 0) * Kpad + dq[p] * 16;
        dOfs[p] = dj * (BW_CHUNK * 64) + dq[p] * 16;
    }

    v16i acc[J][2];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][m][e] = 0;

    const int nchunk = (int)((KW + BW_CHUNK - 1) / BW_CHUNK);
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 rT[4], rF[2];
    v4i rD[DP];

    auto load_chunk = [&](int c) {
        const int64_t w0 = (int64_t)c * BW_CHUNK;
        const int cw = (int)((KW - w0) < BW_CHUNK ? (KW - w0) : BW_CHUNK);
#pragma unroll
        for (int p = 0; p < 4; ++p) rT[p] = (2 * wpT[p] < cw) ? *reinterpret_cast<const u64x2 *>(gT[p] + w0) : u64x2{0ull, 0ull};
#pragma unroll
        for (int p = 0; p < 2; ++p) rF[p] = (2 * wpF[p] < cw) ? *reinterpret_cast<const u64x2 *>(gF[p] + w0) : u64x2{0ull, 0ull};
#pragma unroll
        for (int p = 0; p < DP; ++p)
            rD[p] = (dOn[p] && dq[p] * 16 < cw * 64) ? *reinterpret_cast<const v4i *>(gD[p] + w0 * 64) : v4i{0, 0, 0, 0};
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            sT[buf][woT[p]] = rT[p][0];
            sT[buf][woT[p] + 1] = rT[p][1];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            sF[buf][woF[p]] = rF[p][0];
            sF[buf][woF[p] + 1] = rF[p][1];
        }
#pragma unroll
        for (int p = 0; p < DP; ++p)
            if (dOn[p]) *reinterpret_cast<v4i *>(&sD[buf][dOfs[p]]) = rD[p];
    };

    {
        const v4i lo = expand01((uint32_t)tid & 0xFFu);
        const uint64_t e01 = (uint64_t)(uint32_t)lo[0] | ((uint64_t)(uint32_t)lo[1] << 32);
        lut01[tid] = e01;
        lutFF[tid] = (e01 << 8) - e01;
    }
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fh = lane >> 5;
    const int a0o = (wm * 64 + frow) * BROW + fh, a1o = a0o + 32 * BROW, bo = (wn * 32 + frow) * BROW + fh;

    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunk) load_chunk(c + 1);
        const int64_t w0 = (int64_t)c * BW_CHUNK;
        const int nmac = (int)(((KW - w0) < BW_CHUNK ? (KW - w0) : BW_CHUNK) >> 1);
        for (int m = 0; m < nmac; ++m) {
            const uint64_t a0w = sT[cur][a0o + 2 * m], a1w = sT[cur][a1o + 2 * m], bw = sF[cur][bo + 2 * m];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
                const u64x2v a0q = {lut01[(a0w >> (16 * kk)) & 0xFFu], lut01[(a0w >> (16 * kk + 8)) & 0xFFu]};
                const u64x2v a1q = {lut01[(a1w >> (16 * kk)) & 0xFFu], lut01[(a1w >> (16 * kk + 8)) & 0xFFu]};
                const u64x2v bq = {lutFF[(bw >> (16 * kk)) & 0xFFu], lutFF[(bw >> (16 * kk + 8)) & 0xFFu]};
                const v4i a0 = __builtin_bit_cast(v4i, a0q), a1 = __builtin_bit_cast(v4i, a1q);
                const v4i bmask = __builtin_bit_cast(v4i, bq);
                const int doff = (2 * m + fh) * 64 + kk * 16;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const v4i d = *reinterpret_cast<const v4i *>(&sD[cur][j * (BW_CHUNK * 64) + doff]);
                    const v4i bm = bmask & d;
                    acc[j][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bm, acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bm, acc[j][1], 0, 0, 0);
                }
            }
        }
        if (c + 1 < nchunk) store_chunk(cur ^ 1);
        __syncthreads();
    }

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
