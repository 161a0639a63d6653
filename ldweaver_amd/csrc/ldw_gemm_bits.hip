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
#include "ldw_dev.h"
#include "ldw_epi.h"
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
                                                              int shift_bits, int accumulate, int by0,
                                                              const uint8_t *__restrict__ tile_mask, const uint32_t *__restrict__ tile_list) {
    int bx = blockIdx.x, by = blockIdx.y + by0;  // bx: from-side tile of 64 rows, by: to-side tile of 128 rows (grid.y may
                                                 // cover a strip of them starting at by0: sharded Hamming weights)
    if (tile_list) {
        // r06: the band launches of the approximate path run over the LIST of their live tiles (the host builds it with the mask: prep_block) — a 1-D grid of
        // exactly those.  With the mask alone the launch dispatched the whole 182 x 91 grid of a 10k x 10k block for the ~1 700 tiles of its short-range band,
        // and ~15 000 workgroups that leave at once still cost ~4 ns each: a third of the launch (profiles/r06_band_tile_list.txt).
        const uint32_t t = tile_list[blockIdx.x];
        bx = (int)(t & 0xFFFFu);
        by = (int)(t >> 16);
    } else {
        if (lower_only && bx * TILE_F4 + TILE_F4 - 1 < by * TILE) return;
        // tile_mask[by][bx] == 0: nobody reads this tile (approximate path: only the tiles that hold a short-range pair are needed)
        if (tile_mask && tile_mask[by * (RFpad / TILE_F4) + bx] == 0) return;
    }

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

// ------------------------------------------------------------------------------------------------
// Gathered low-limb GEMM of the mixed-precision path.  The block-wide GEMM runs with the 3 high limbs of the weights
// only (all the fp32 screen needs: C4 1.14 instead of 1.73 ms per block); the exact joint sums of the few units the
// screen lists (3-4 % of an off-diagonal block, the short-range band of a diagonal one) get their 2 low limbs here.
// Workgroup (chunk, tile, fs): 128 low-limb rows of one from-tile — the indicator rows of up to 128 / c listed to-side
// SNPs of one row-slot class c, read off the tile's unit list — against 64 of the tile's 64 * cmax from-side row slots.
// Same K loop, J = 2, row lists built in LDS.  Output: int32 (|sum| <= N * 2^15).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_lo_units_kernel(LoGemmArgs P) {
    // workgroup (x, y): y -> (tile, fs) through the host-built list of the 64-slot from-side sub-tiles that exist;
    // x strides over the chunks of 128 low-limb rows that the tile's unit lists actually fill, class by class
    const int tile = P.tf_list[2 * blockIdx.y], fs = P.tf_list[2 * blockIdx.y + 1];
    const int cmax = P.lo.cmax_f[tile];
    __shared__ GemmSmem<2> S;
    __shared__ int32_t lrows[TILE + TILE_F4];
    // chunk i of the tile (counted through the classes) goes to workgroup x = (i + y) mod gridDim.x: consecutive workgroup
    // ids land on different XCDs, and most tiles fill only a few chunks — without the rotation by y all of them would
    // sit on the first XCDs
    int seq = (int)(blockIdx.y % gridDim.x);
    for (int lc = 0; lc < 3; ++lc) {
        const int cnt = (int)P.lo.cnt[tile * 3 + lc];
        const int nch = (((cnt << lc) + TILE - 1) / TILE);
        for (int ch = 0; ch < nch; ++ch, ++seq) {
            if (seq % (int)gridDim.x != (int)blockIdx.x) continue;
            const int k0 = (ch * TILE) >> lc;            // first unit of this chunk
            const int r0 = P.lo.rowbase[lc] + ch * TILE; // its first low-limb row
            __syncthreads();                             // the previous chunk's row lists and operand tiles are done with
            {
                const int t0 = threadIdx.x;
                if (t0 < TILE) {
                    const int kk = k0 + (t0 >> lc), j = t0 & ((1 << lc) - 1);
                    int32_t row = P.zero_row;
                    if (kk < cnt) {
                        const int q = (int)(P.lo.tl[(int64_t)tile * P.nt + P.lo.uoff[lc] + kk] & 0x7FFFFFFFu);
                        const int snp = P.idx_t[P.perm_t[q]];
                        if (j < P.row0[snp + 1] - P.row0[snp]) row = P.row0[snp] + j;
                    }
                    lrows[t0] = row;
                } else if (t0 < TILE + TILE_F4) {
                    const int p = fs * TILE_F4 + (t0 - TILE);
                    const int ln = p / cmax, i = p - ln * cmax;
                    const int pf = P.perm_f[tile * 64 + ln];
                    int32_t row = P.zero_row;
                    if (pf >= 0) {
                        const int snp = P.idx_f[pf];
                        if (i < P.row0[snp + 1] - P.row0[snp]) row = P.row0[snp] + i;
                    }
                    lrows[t0] = row;
                }
            }
            __syncthreads();
            constexpr int J = 2;
            const uint64_t *__restrict__ Mbits = P.Mbits;
            const int64_t KW = P.KW, Kpad = P.Kpad;
            const int32_t *rowlist_t = lrows;
            const int32_t *rowlist_f = lrows + TILE;
            const int8_t *__restrict__ digits = P.digits_lo;
            const int bx = 0, by = 0;
#include "ldw_gemm_kloop.inc"

            const int64_t ld = (int64_t)TILE_F4 * cmax;
            int32_t *out = P.lo.glo + P.lo.tile_base[tile] + (int64_t)r0 * ld + fs * TILE_F4 + wn * 32 + frow;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int trow = wm * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    out[(int64_t)trow * ld] = acc[1][m][e] * 256 + acc[0][m][e];
                }
            }
        }
    }
}

int launch_gemm_lo_units(ldw_ctx *ctx, const LoGemmArgs &P, int n_tf, hipStream_t stream) {
    LDW_REQUIRE(n_tf > 0 && n_tf <= 65535 && P.KW > 0 && P.KW % 2 == 0 && P.tf_list, LDW_ERR_ARG, "launch_gemm_lo_units: bad geometry");
    // 32 workgroups per sub-tile stride over its chunks: the lists of an off-diagonal block fill ~5 chunks per tile, the
    // short-range band of a diagonal block a few dozen, and the one tile of SNPs with >= 3 minor states (generic code,
    // every unit listed) ~90 per sub-tile — that tile is the critical path
    hipLaunchKernelGGL(gemm_lo_units_kernel, dim3(32, (unsigned)n_tf), dim3(256), 0, stream, P);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

// bit rows: Mbits[row][w] bit i = (states[snp(row)][seq_perm[64 w + i]] == state(row)); one workgroup per row, the state
// row staged in LDS (the positions of a word are scattered over the row: weight order, ldw_set_weights)
__global__ __launch_bounds__(256) void k_fill_rows_bits(const uint8_t *__restrict__ states, int64_t Npad,
                                                        const int32_t *__restrict__ rowinfo, int64_t KW,
                                                        const int32_t *__restrict__ seq_perm, int use_lds,
                                                        uint64_t *__restrict__ Mbits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t srow[];
    const int64_t row = blockIdx.x;
    const int32_t info = rowinfo[row];
    const int64_t snp = info >> 3;
    const uint32_t st = (uint32_t)(info & 7);
    const uint8_t *src = states + snp * Npad;
    if (use_lds) {
        for (int64_t i = threadIdx.x; i < Npad / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(srow)[i] = reinterpret_cast<const uint4 *>(src)[i];
        __syncthreads();
    }
    for (int64_t w = threadIdx.x; w < KW; w += blockDim.x) {
        uint64_t bits = 0;
        const int32_t *sp = seq_perm + w * 64;
#pragma unroll 8
        for (int q = 0; q < 64; ++q) {
            const int32_t s = sp[q];
            const uint32_t x = s < 0 ? 255u : (use_lds ? (uint32_t)srow[s] : (uint32_t)src[s]);
            bits |= (uint64_t)(x == st) << q;
        }
        Mbits[row * KW + w] = bits;
    }
}

int launch_gemm_bits(ldw_ctx *ctx, const uint64_t *Mbits, int64_t KW, const int32_t *rowlist_t, int RTpad, const int32_t *rowlist_f,
                     int RFpad, int64_t *G, int nlimbs, const int8_t *digits, int lower_only, hipStream_t stream, int by0, int by1,
                     const uint8_t *tile_mask, const uint32_t *tile_list, int n_tile_list) {
    if (!stream) stream = ctx->stream;
    const int64_t Kpad = KW * 64;
    LDW_REQUIRE(RTpad % TILE == 0 && RFpad % TILE == 0 && KW > 0 && KW % 2 == 0 && Kpad == KW * 64, LDW_ERR_ARG,
                "launch_gemm_bits: padding violated (RT %d RF %d KW %lld)", RTpad, RFpad, (long long)KW);
    LDW_REQUIRE(nlimbs >= 1 && nlimbs <= 6, LDW_ERR_ARG, "launch_gemm_bits: nlimbs %d out of range", nlimbs);
    if (by1 < 0) by1 = RTpad / TILE;
    LDW_REQUIRE(by0 >= 0 && by0 < by1 && by1 <= RTpad / TILE, LDW_ERR_ARG, "launch_gemm_bits: bad tile-row range %d..%d", by0, by1);
    dim3 grid(RFpad / TILE_F4, by1 - by0), block(256);
    if (tile_list) {   // (the mask's tiles as a list: tile = by << 16 | bx, tiles above the diagonal of a lower_only launch already left out)
        LDW_REQUIRE(n_tile_list >= 0 && by0 == 0 && RFpad / TILE_F4 < 65536 && RTpad / TILE < 65536, LDW_ERR_ARG, "launch_gemm_bits: bad tile list");
        if (n_tile_list == 0) {
            ctx->gemm_stat[4] += 1;
            return LDW_OK;
        }
        grid = dim3((unsigned)n_tile_list, 1);
    }
    int done = 0;
    while (done < nlimbs) {  // up to 5 limbs share one pass over K; 6 limbs run as 3 + 3
        const int J = (nlimbs == 6) ? 3 : nlimbs;
        const int8_t *dg = digits + (int64_t)done * Kpad;
        const int shift = 8 * done, accum = done > 0;
#define LDW_LAUNCH_B(JJ)                                                                                        \
    case JJ:                                                                                                    \
        hipLaunchKernelGGL(gemm_bits_kernel<JJ>, grid, block, 0, stream, Mbits, KW,                                 \
                           rowlist_t, rowlist_f, dg, Kpad, G, RFpad, lower_only, shift, accum, by0, tile_mask, tile_list); \
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
    if (!tile_mask && !tile_list) {   // executed work (ldw_gemm_stats): workgroup tiles that do not leave at once, all limbs
        int64_t tiles = 0;
        const int nbx = RFpad / TILE_F4;
        for (int by = by0; by < by1; ++by) {
            if (!lower_only) tiles += nbx;
            else for (int bx = 0; bx < nbx; ++bx) tiles += (bx * TILE_F4 + TILE_F4 - 1 < by * TILE) ? 0 : 1;
        }
        ctx->gemm_stat[2] += 1;
        ctx->gemm_stat[3] += 2.0 * (double)tiles * TILE * TILE_F4 * (double)Kpad * nlimbs;
    } else {
        ctx->gemm_stat[4] += 1;
    }
    return LDW_OK;
}

int fill_rows_bits(ldw_ctx *c, const int32_t *d_rowinfo, int64_t R) {
    LDW_REQUIRE(c->seq_perm.p && (int64_t)c->h_seq_perm.size() == c->Npad, LDW_ERR_STATE, "fill_rows_bits: no sequence order (set the weights first)");
    const int use_lds = c->Npad <= 61440 ? 1 : 0;   // the default dynamic-LDS limit is 64 KB
    hipLaunchKernelGGL(k_fill_rows_bits, dim3((unsigned)R), dim3(256), use_lds ? (size_t)c->Npad : 0, c->stream, c->states.as<uint8_t>(), c->Npad,
                       d_rowinfo, c->KW, c->seq_perm.as<int32_t>(), use_lds, c->Mbits.as<uint64_t>());
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

}  // namespace ldw

namespace ldw {
void warm_gemm_bits() {   // ldw_ctx_reserve: load this translation unit's code object ahead of its first launch
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&gemm_bits_kernel<5>));
    (void)hipGetLastError();
}
}  // namespace ldw
