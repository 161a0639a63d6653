// K loop of the bit-packed fixed-point co-occurrence GEMM for one 128 (to side) x 64 (from side) workgroup tile,
// shared by gemm_bits_kernel (ldw_gemm_bits.hip: G goes to HBM) and the fused GEMM + MI epilogue kernel
// (ldw_fused.hip: G stays on chip).  See ldw_gemm_bits.hip for the formulation.
#pragma once
#include "ldw_internal.h"

namespace ldw {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BW_CHUNK = 16;        // 64-bit words of K per LDS chunk (1024 sequences)
constexpr int BROW = BW_CHUNK + 1;  // padded LDS row stride in words: odd, so 32 rows hit 32 distinct 8-B bank pairs
constexpr int TILE_F4 = 64;         // from-side rows per workgroup tile (to side: TILE = 128)

__device__ __forceinline__ v4i expand01(uint32_t b16) {
    v4i r;
    r[0] = (int)(__umul24(b16 & 0xFu, 0x204081u) & 0x01010101u);
    r[1] = (int)(__umul24((b16 >> 4) & 0xFu, 0x204081u) & 0x01010101u);
    r[2] = (int)(__umul24((b16 >> 8) & 0xFu, 0x204081u) & 0x01010101u);
    r[3] = (int)(__umul24((b16 >> 12) & 0xFu, 0x204081u) & 0x01010101u);
    return r;
}

template <int J>
struct GemmSmem {
    __attribute__((aligned(16))) uint64_t sT[2][TILE * BROW];
    __attribute__((aligned(16))) uint64_t sF[2][TILE_F4 * BROW];
    __attribute__((aligned(16))) int8_t sD[2][J * BW_CHUNK * 64];
    uint64_t lut01[256], lutFF[256];
};

}  // namespace ldw
