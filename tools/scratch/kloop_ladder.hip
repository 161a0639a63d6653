// gfx950: where does the K loop of gemm_apx_kernel lose the MFMA rate?  One k-step = 8 v_mfma_i32_32x32x32_i8 on a 4 x 2 tile of accumulators fed by 6 operand
// fragments, 2 waves per SIMD on every SIMD (the shipped kernel's shape); the fragments come from
//   level 0: loop-invariant registers                                   (MFMA alone)
//   level 1: + 24 VALU per k-step that do NOT feed the MFMAs            (issue competition only)
//   level 2: registers ANDed with a per-iteration mask                  (24 VALU feeding the MFMAs: VALU -> MFMA dependency)
//   level 3: a 256-entry LDS table read at indices taken from a rotating register word (12 index pairs = 24 VALU, 12 ds_read_b64), then the AND (24 VALU)
//   level 4: level 3 + the two digit vectors read from LDS per k-step (2 ds_read_b128)
//   level 41 / 42 / 43 (r06): level 4 with 1 / 2 / 3 of the six fragments expanded ARITHMETICALLY (per 4 positions: nibble, x 0x204081 & 0x01010101 -> a 0/1 byte per bit,
//            (t << 8) - t -> 0xFF bytes, AND: 6 VALU per dword instead of 2 + a table read per two dwords) — trading LDS reads for VALU
// Operands are sparse (about 15 % of the bytes non-zero) so that the clock stays at its nominal value (profiles/r05_mfma_rate_by_data.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
//   level 5: level 3 with the SAME indices in every lane (broadcast reads: no bank conflict at all)
//   REP = 16 / 32: the table replicated so that lane l reads slot l % REP of its entry (2-way conflicts by construction / none)
template <int LEVEL, int REP>
__global__ __launch_bounds__(256, 2) void k(const u64 *__restrict__ words, unsigned *out, int iters, const u64 *__restrict__ glut, const v4i *__restrict__ gdig) {
    __shared__ u64 lut[LEVEL == 9 ? 8192 : 256 * REP];   // level 9: 4096 entries of 16 B (12 bits -> 12 bytes + 4 unused)
    __shared__ __attribute__((aligned(16))) unsigned char dig[2][4096];
    {
        u64 e = 0;
        for (int b = 0; b < 8; ++b) e |= ((threadIdx.x >> b) & 1) ? (0xFFull << (8 * b)) : 0ull;
        if (LEVEL == 9) {
            for (int t = threadIdx.x; t < 4096; t += 256) {
                u64 lo = 0, hi = 0;
                for (int b = 0; b < 8; ++b) lo |= ((t >> b) & 1) ? (0xFFull << (8 * b)) : 0ull;
                for (int b = 0; b < 4; ++b) hi |= ((t >> (8 + b)) & 1) ? (0xFFull << (8 * b)) : 0ull;
                lut[2 * t] = lo;
                lut[2 * t + 1] = hi;
            }
        } else
            for (int q = 0; q < REP; ++q) lut[threadIdx.x * REP + q] = e;
        for (int i = threadIdx.x; i < 4096; i += 256) { dig[0][i] = (unsigned char)(1 + (i * 37) % 100); dig[1][i] = (unsigned char)(1 + (i * 53) % 100); }
    }
    __syncthreads();
    v16i acc[4][2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    u64 w[6];
    for (int r = 0; r < 6; ++r) w[r] = words[(blockIdx.x * 6 + r) * 256 + (LEVEL == 5 ? 0 : threadIdx.x)];
    const u64 *lt = lut + (threadIdx.x % REP);
    v4i base[6];
    for (int r = 0; r < 6; ++r) {
        const u64 a = lt[(w[r] & 0xFF) * REP], b = lt[((w[r] >> 8) & 0xFF) * REP];
        base[r] = v4i{(int)a, (int)(a >> 32), (int)b, (int)(b >> 32)} & v4i{0x11223344, 0x0A0B0C0D, 0x21314151, 0x07060504};
    }
    unsigned side[8];
    for (int i = 0; i < 8; ++i) side[i] = threadIdx.x * 2654435761u + i;
    const int fh = (threadIdx.x >> 5) & 1;
    for (int it = 0; it < iters; ++it) {
        v4i f[6];
        if (LEVEL <= 1) {
#pragma unroll
            for (int r = 0; r < 6; ++r) f[r] = base[r];
            if (LEVEL == 1) {
#pragma unroll
                for (int q = 0; q < 24; ++q) side[q & 7] = (side[q & 7] & 0x7F7F7F7Fu) + 0x01010101u;
            }
        } else if (LEVEL == 2) {
            const int m = 0x7F7F7F7F ^ (it & 0x0F0F0F0F);
            const v4i mk = {m, m ^ 0x01010101, m ^ 0x02020202, m ^ 0x03030303};
#pragma unroll
            for (int r = 0; r < 6; ++r) f[r] = base[r] & mk;
        } else {
            v4i da = {0x11223344, 0x0A0B0C0D, 0x21314151, 0x07060504}, db = da;
            if (LEVEL == 4 || LEVEL >= 40) {
                da = *reinterpret_cast<const v4i *>(&dig[0][((it & 63) * 64 + fh * 16) & 4080]);
                db = *reinterpret_cast<const v4i *>(&dig[1][((it & 63) * 64 + fh * 16) & 4080]);
            }
            if (LEVEL == 8) {   // the digit vectors of both lane halves through the SCALAR cache (uniform addresses), selected per lane half
                const int o = (it & 63) * 4;
                const v4i a0 = gdig[o], a1 = gdig[o + 1], b0 = gdig[o + 2], b1 = gdig[o + 3];
                da = fh ? a1 : a0;
                db = fh ? b1 : b0;
            }
            const int sh = (it & 3) * 16;
            if (LEVEL == 9) {
                // 8 reads of 12 bits each = 96 positions = the 6 fragments of this k-step; 3 dwords of a read are data: fragments are made of them by renaming
                v4i q[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) q[r] = *reinterpret_cast<const v4i *>(&lut[2 * ((w[r % 6] >> ((sh + 12 * (r / 6)) & 31)) & 0xFFF)]);
                f[0] = v4i{q[0][0], q[0][1], q[0][2], q[1][0]} & da;
                f[1] = v4i{q[1][1], q[1][2], q[2][0], q[2][1]} & da;
                f[2] = v4i{q[2][2], q[3][0], q[3][1], q[3][2]} & da;
                f[3] = v4i{q[4][0], q[4][1], q[4][2], q[5][0]} & da;
                f[4] = v4i{q[5][1], q[5][2], q[6][0], q[6][1]} & db;
                f[5] = v4i{q[6][2], q[7][0], q[7][1], q[7][2]} & db;
            } else
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                if (LEVEL >= 40 && r >= 6 - (LEVEL - 40)) {   // arithmetic expansion of the 16 positions at bit sh of w[r]
                    const v4i dg = r < 4 ? da : db;
                    v4i o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned nib = (unsigned)(w[r] >> (sh + 4 * q)) & 0xFu;
                        const unsigned t = __umul24(nib, 0x204081u) & 0x01010101u;
                        o[q] = (int)((t << 8) - t) & dg[q];
                    }
                    f[r] = o;
                    continue;
                }
                // level 6: the second read of every fragment, level 7: both reads of fragments 4 and 5 (a third of the reads) go through the vector L1
                const bool ga = LEVEL == 7 && r >= 4, gb = LEVEL == 6 || (LEVEL == 7 && r >= 4);
                const u64 a = ga ? glut[(w[r] >> sh) & 0xFF] : lt[((w[r] >> sh) & 0xFF) * REP];
                const u64 b = gb ? glut[(w[r] >> (sh + 8)) & 0xFF] : lt[((w[r] >> (sh + 8)) & 0xFF) * REP];
                f[r] = v4i{(int)a, (int)(a >> 32), (int)b, (int)(b >> 32)} & (r < 4 ? da : db);
            }
            if ((it & 3) == 3) {
#pragma unroll
                for (int r = 0; r < 6; ++r) w[r] = (w[r] << 7) | (w[r] >> 57);   // (new indices for the next four k-steps)
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[i], f[4 + j], acc[i][j], 0, 0, 0);
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= side[i];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) s ^= (unsigned)acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same rungs on a 3 x 3 tile of accumulators: 6 fragments for 9 MFMAs per k-step
template <int LEVEL, int REP>
__global__ __launch_bounds__(256, 2) void k33(const u64 *__restrict__ words, unsigned *out, int iters, const u64 *__restrict__ glut, const v4i *__restrict__ gdig) {
    __shared__ u64 lut[LEVEL == 9 ? 8192 : 256 * REP];   // level 9: 4096 entries of 16 B (12 bits -> 12 bytes + 4 unused)
    __shared__ __attribute__((aligned(16))) unsigned char dig[2][4096];
    {
        u64 e = 0;
        for (int b = 0; b < 8; ++b) e |= ((threadIdx.x >> b) & 1) ? (0xFFull << (8 * b)) : 0ull;
        if (LEVEL == 9) {
            for (int t = threadIdx.x; t < 4096; t += 256) {
                u64 lo = 0, hi = 0;
                for (int b = 0; b < 8; ++b) lo |= ((t >> b) & 1) ? (0xFFull << (8 * b)) : 0ull;
                for (int b = 0; b < 4; ++b) hi |= ((t >> (8 + b)) & 1) ? (0xFFull << (8 * b)) : 0ull;
                lut[2 * t] = lo;
                lut[2 * t + 1] = hi;
            }
        } else
            for (int q = 0; q < REP; ++q) lut[threadIdx.x * REP + q] = e;
        for (int i = threadIdx.x; i < 4096; i += 256) { dig[0][i] = (unsigned char)(1 + (i * 37) % 100); dig[1][i] = (unsigned char)(1 + (i * 53) % 100); }
    }
    __syncthreads();
    v16i acc[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    u64 w[6];
    for (int r = 0; r < 6; ++r) w[r] = words[(blockIdx.x * 6 + r) * 256 + (LEVEL == 5 ? 0 : threadIdx.x)];
    const u64 *lt = lut + (threadIdx.x % REP);
    v4i base[6];
    for (int r = 0; r < 6; ++r) {
        const u64 a = lt[(w[r] & 0xFF) * REP], b = lt[((w[r] >> 8) & 0xFF) * REP];
        base[r] = v4i{(int)a, (int)(a >> 32), (int)b, (int)(b >> 32)} & v4i{0x11223344, 0x0A0B0C0D, 0x21314151, 0x07060504};
    }
    unsigned side[8];
    for (int i = 0; i < 8; ++i) side[i] = threadIdx.x * 2654435761u + i;
    const int fh = (threadIdx.x >> 5) & 1;
    for (int it = 0; it < iters; ++it) {
        v4i f[6];
        if (LEVEL <= 1) {
#pragma unroll
            for (int r = 0; r < 6; ++r) f[r] = base[r];
            if (LEVEL == 1) {
#pragma unroll
                for (int q = 0; q < 24; ++q) side[q & 7] = (side[q & 7] & 0x7F7F7F7Fu) + 0x01010101u;
            }
        } else if (LEVEL == 2) {
            const int m = 0x7F7F7F7F ^ (it & 0x0F0F0F0F);
            const v4i mk = {m, m ^ 0x01010101, m ^ 0x02020202, m ^ 0x03030303};
#pragma unroll
            for (int r = 0; r < 6; ++r) f[r] = base[r] & mk;
        } else {
            v4i da = {0x11223344, 0x0A0B0C0D, 0x21314151, 0x07060504}, db = da;
            if (LEVEL == 4) {
                da = *reinterpret_cast<const v4i *>(&dig[0][((it & 63) * 64 + fh * 16) & 4080]);
                db = *reinterpret_cast<const v4i *>(&dig[1][((it & 63) * 64 + fh * 16) & 4080]);
            }
            if (LEVEL == 8) {   // the digit vectors of both lane halves through the SCALAR cache (uniform addresses), selected per lane half
                const int o = (it & 63) * 4;
                const v4i a0 = gdig[o], a1 = gdig[o + 1], b0 = gdig[o + 2], b1 = gdig[o + 3];
                da = fh ? a1 : a0;
                db = fh ? b1 : b0;
            }
            const int sh = (it & 3) * 16;
            if (LEVEL == 9) {
                // 8 reads of 12 bits each = 96 positions = the 6 fragments of this k-step; 3 dwords of a read are data: fragments are made of them by renaming
                v4i q[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) q[r] = *reinterpret_cast<const v4i *>(&lut[2 * ((w[r % 6] >> ((sh + 12 * (r / 6)) & 31)) & 0xFFF)]);
                f[0] = v4i{q[0][0], q[0][1], q[0][2], q[1][0]} & da;
                f[1] = v4i{q[1][1], q[1][2], q[2][0], q[2][1]} & da;
                f[2] = v4i{q[2][2], q[3][0], q[3][1], q[3][2]} & da;
                f[3] = v4i{q[4][0], q[4][1], q[4][2], q[5][0]} & da;
                f[4] = v4i{q[5][1], q[5][2], q[6][0], q[6][1]} & db;
                f[5] = v4i{q[6][2], q[7][0], q[7][1], q[7][2]} & db;
            } else
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                // level 6: the second read of every fragment, level 7: both reads of fragments 4 and 5 (a third of the reads) go through the vector L1
                const bool ga = LEVEL == 7 && r >= 4, gb = LEVEL == 6 || (LEVEL == 7 && r >= 4);
                const u64 a = ga ? glut[(w[r] >> sh) & 0xFF] : lt[((w[r] >> sh) & 0xFF) * REP];
                const u64 b = gb ? glut[(w[r] >> (sh + 8)) & 0xFF] : lt[((w[r] >> (sh + 8)) & 0xFF) * REP];
                f[r] = v4i{(int)a, (int)(a >> 32), (int)b, (int)(b >> 32)} & (r < 3 ? da : db);
            }
            if ((it & 3) == 3) {
#pragma unroll
                for (int r = 0; r < 6; ++r) w[r] = (w[r] << 7) | (w[r] >> 57);   // (new indices for the next four k-steps)
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[i], f[3 + j], acc[i][j], 0, 0, 0);
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= side[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int e = 0; e < 16; ++e) s ^= (unsigned)acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same rungs on a 5 x 2 tile of accumulators: 7 fragments for 10 MFMAs per k-step
template <int LEVEL, int REP>
__global__ __launch_bounds__(256, 2) void k52(const u64 *__restrict__ words, unsigned *out, int iters, const u64 *__restrict__ glut, const v4i *__restrict__ gdig) {
    __shared__ u64 lut[LEVEL == 9 ? 8192 : 256 * REP];   // level 9: 4096 entries of 16 B (12 bits -> 12 bytes + 4 unused)
    __shared__ __attribute__((aligned(16))) unsigned char dig[2][4096];
    {
        u64 e = 0;
        for (int b = 0; b < 8; ++b) e |= ((threadIdx.x >> b) & 1) ? (0xFFull << (8 * b)) : 0ull;
        if (LEVEL == 9) {
            for (int t = threadIdx.x; t < 4096; t += 256) {
                u64 lo = 0, hi = 0;
                for (int b = 0; b < 8; ++b) lo |= ((t >> b) & 1) ? (0xFFull << (8 * b)) : 0ull;
                for (int b = 0; b < 4; ++b) hi |= ((t >> (8 + b)) & 1) ? (0xFFull << (8 * b)) : 0ull;
                lut[2 * t] = lo;
                lut[2 * t + 1] = hi;
            }
        } else
            for (int q = 0; q < REP; ++q) lut[threadIdx.x * REP + q] = e;
        for (int i = threadIdx.x; i < 4096; i += 256) { dig[0][i] = (unsigned char)(1 + (i * 37) % 100); dig[1][i] = (unsigned char)(1 + (i * 53) % 100); }
    }
    __syncthreads();
    v16i acc[5][2];
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    u64 w[7];
    for (int r = 0; r < 7; ++r) w[r] = words[(blockIdx.x * 6 + (r % 6)) * 256 + (LEVEL == 5 ? 0 : threadIdx.x)];
    const u64 *lt = lut + (threadIdx.x % REP);
    v4i base[7];
    for (int r = 0; r < 7; ++r) {
        const u64 a = lt[(w[r] & 0xFF) * REP], b = lt[((w[r] >> 8) & 0xFF) * REP];
        base[r] = v4i{(int)a, (int)(a >> 32), (int)b, (int)(b >> 32)} & v4i{0x11223344, 0x0A0B0C0D, 0x21314151, 0x07060504};
    }
    unsigned side[8];
    for (int i = 0; i < 8; ++i) side[i] = threadIdx.x * 2654435761u + i;
    const int fh = (threadIdx.x >> 5) & 1;
    for (int it = 0; it < iters; ++it) {
        v4i f[7];
        if (LEVEL <= 1) {
#pragma unroll
            for (int r = 0; r < 7; ++r) f[r] = base[r];
            if (LEVEL == 1) {
#pragma unroll
                for (int q = 0; q < 24; ++q) side[q & 7] = (side[q & 7] & 0x7F7F7F7Fu) + 0x01010101u;
            }
        } else if (LEVEL == 2) {
            const int m = 0x7F7F7F7F ^ (it & 0x0F0F0F0F);
            const v4i mk = {m, m ^ 0x01010101, m ^ 0x02020202, m ^ 0x03030303};
#pragma unroll
            for (int r = 0; r < 7; ++r) f[r] = base[r] & mk;
        } else {
            v4i da = {0x11223344, 0x0A0B0C0D, 0x21314151, 0x07060504}, db = da;
            if (LEVEL == 4) {
                da = *reinterpret_cast<const v4i *>(&dig[0][((it & 63) * 64 + fh * 16) & 4080]);
                db = *reinterpret_cast<const v4i *>(&dig[1][((it & 63) * 64 + fh * 16) & 4080]);
            }
            if (LEVEL == 8) {   // the digit vectors of both lane halves through the SCALAR cache (uniform addresses), selected per lane half
                const int o = (it & 63) * 4;
                const v4i a0 = gdig[o], a1 = gdig[o + 1], b0 = gdig[o + 2], b1 = gdig[o + 3];
                da = fh ? a1 : a0;
                db = fh ? b1 : b0;
            }
            const int sh = (it & 3) * 16;
            if (LEVEL == 9) {
                // 8 reads of 12 bits each = 96 positions = the 6 fragments of this k-step; 3 dwords of a read are data: fragments are made of them by renaming
                v4i q[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) q[r] = *reinterpret_cast<const v4i *>(&lut[2 * ((w[r % 6] >> ((sh + 12 * (r / 6)) & 31)) & 0xFFF)]);
                f[0] = v4i{q[0][0], q[0][1], q[0][2], q[1][0]} & da;
                f[1] = v4i{q[1][1], q[1][2], q[2][0], q[2][1]} & da;
                f[2] = v4i{q[2][2], q[3][0], q[3][1], q[3][2]} & da;
                f[3] = v4i{q[4][0], q[4][1], q[4][2], q[5][0]} & da;
                f[4] = v4i{q[5][1], q[5][2], q[6][0], q[6][1]} & db;
                f[5] = v4i{q[6][2], q[7][0], q[7][1], q[7][2]} & db;
            } else
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                // level 6: the second read of every fragment, level 7: both reads of fragments 4 and 5 (a third of the reads) go through the vector L1
                const bool ga = LEVEL == 7 && r >= 4, gb = LEVEL == 6 || (LEVEL == 7 && r >= 4);
                const u64 a = ga ? glut[(w[r] >> sh) & 0xFF] : lt[((w[r] >> sh) & 0xFF) * REP];
                const u64 b = gb ? glut[(w[r] >> (sh + 8)) & 0xFF] : lt[((w[r] >> (sh + 8)) & 0xFF) * REP];
                f[r] = v4i{(int)a, (int)(a >> 32), (int)b, (int)(b >> 32)} & (r < 5 ? da : db);
            }
            if ((it & 3) == 3) {
#pragma unroll
                for (int r = 0; r < 7; ++r) w[r] = (w[r] << 7) | (w[r] >> 57);   // (new indices for the next four k-steps)
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[i], f[5 + j], acc[i][j], 0, 0, 0);
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= side[i];
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) s ^= (unsigned)acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int LEVEL, int REP = 1>
static void run(const u64 *d, unsigned *o, int g, int iters, const char *what, const u64 *glut = nullptr, const v4i *gdig = nullptr) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<LEVEL, REP>), dim3(g), dim3(256), 0, 0, d, o, iters, glut, gdig);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double per = (double)ms * 1e6 / ((double)g * 4 / 1024.0 * iters * 8);
    printf("level %d  %-78s %7.2f ms  %.2f ns per MFMA per SIMD = %.2f of the nominal rate\n", LEVEL, what, ms, per, 13.333 / per);
}
template <int LEVEL>
static void run52(const u64 *d, unsigned *o, int g, int iters, const char *what) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k52<LEVEL, 1>), dim3(g), dim3(256), 0, 0, d, o, iters, (const u64 *)nullptr, (const v4i *)nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double per = (double)ms * 1e6 / ((double)g * 4 / 1024.0 * iters * 10);
    printf("5 x 2 tile, level %d  %-66s %7.2f ms  %.2f ns per MFMA per SIMD = %.2f of the nominal rate\n", LEVEL, what, ms, per, 13.333 / per);
}
template <int LEVEL>
static void run33(const u64 *d, unsigned *o, int g, int iters, const char *what) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k33<LEVEL, 1>), dim3(g), dim3(256), 0, 0, d, o, iters, (const u64 *)nullptr, (const v4i *)nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double per = (double)ms * 1e6 / ((double)g * 4 / 1024.0 * iters * 9);
    printf("3 x 3 tile, level %d  %-66s %7.2f ms  %.2f ns per MFMA per SIMD = %.2f of the nominal rate\n", LEVEL, what, ms, per, 13.333 / per);
}
int main() {
    const int g = 512 * 4, iters = 4000;
    const size_t n = (size_t)g * 6 * 256;
    u64 *h = (u64 *)malloc(n * 8), *d;
    unsigned *o;
    srand(11);
    for (size_t i = 0; i < n; ++i) {
        u64 v = 0;
        for (int b = 0; b < 64; ++b) v |= (u64)(rand() % 100 < 15) << b;
        h[i] = v;
    }
    if (hipMalloc(&d, n * 8) != hipSuccess || hipMalloc(&o, (size_t)g * 256 * 4) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
    u64 hl[256], *gl;
    for (int t = 0; t < 256; ++t) {
        u64 e = 0;
        for (int b = 0; b < 8; ++b) e |= ((t >> b) & 1) ? (0xFFull << (8 * b)) : 0ull;
        hl[t] = e;
    }
    if (hipMalloc(&gl, 2048) != hipSuccess) return 1;
    (void)hipMemcpy(gl, hl, 2048, hipMemcpyHostToDevice);
    run<0>(d, o, g, iters, "fragments loop-invariant (MFMA alone)");
    run<1>(d, o, g, iters, "+ 24 independent VALU per k-step");
    run<2>(d, o, g, iters, "24 VALU (AND with a changing mask) FEEDING the MFMAs");
    run<3>(d, o, g, iters, "12 table reads (ds_read_b64) + 24 index VALU + 24 AND feeding the MFMAs");
    run<4>(d, o, g, iters, "level 3 + the two digit vectors from LDS (2 ds_read_b128 per k-step)");
    run<41>(d, o, g, iters, "level 4 with ONE of the six fragments expanded arithmetically (10 table reads)");
    run<42>(d, o, g, iters, "level 4 with TWO fragments expanded arithmetically (8 table reads)");
    run<43>(d, o, g, iters, "level 4 with THREE fragments expanded arithmetically (6 table reads)");
    run<46>(d, o, g, iters, "level 4 with ALL SIX fragments expanded arithmetically (no table read)");
    run<5>(d, o, g, iters, "level 3 with the same indices in every lane (broadcast: no bank conflicts)");
    run<3, 16>(d, o, g, iters, "level 3, table replicated 16 x (2-way conflicts by construction)");
    run<3, 32>(d, o, g, iters, "level 3, table replicated 32 x (conflict-free, 64 KB)");
    run<4, 32>(d, o, g, iters, "level 4, table replicated 32 x");
    v4i hd[256], *gd;
    for (int t = 0; t < 256; ++t) hd[t] = v4i{0x11223344 + t, 0x0A0B0C0D, 0x21314151, 0x07060504 + t};
    if (hipMalloc(&gd, sizeof(hd)) != hipSuccess) return 1;
    (void)hipMemcpy(gd, hd, sizeof(hd), hipMemcpyHostToDevice);
    run33<0>(d, o, g, iters, "fragments loop-invariant");
    run33<3>(d, o, g, iters, "12 table reads + 24 index VALU + 24 AND per 9 MFMAs");
    run33<4>(d, o, g, iters, "+ the two digit vectors from LDS");
    run52<0>(d, o, g, iters, "fragments loop-invariant");
    run52<3>(d, o, g, iters, "14 table reads + 28 index VALU + 28 AND per 10 MFMAs");
    run52<4>(d, o, g, iters, "+ the two digit vectors from LDS");
    run<9>(d, o, g, iters, "8 ds_read_b128 from a 4096-entry table (12 bits -> 12 bytes, 64 KB) instead of 12 ds_read_b64");
    run<8>(d, o, g, iters, "level 3 + the digit vectors through the scalar cache (s_load) and a select per lane half", gl, gd);
    run<6>(d, o, g, iters, "level 3 with HALF of the table reads through the vector L1 (global_load_dwordx2)", gl);
    run<7>(d, o, g, iters, "level 3 with a THIRD of the table reads through the vector L1", gl);
    return 0;
}
