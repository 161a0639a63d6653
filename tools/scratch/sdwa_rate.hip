// gfx950: (1) issue rate of v_lshlrev_b32_sdwa (byte select) against v_bfe_u32 (+ v_lshlrev_b32) alone; (2) the same instructions BESIDE MFMAs
// (8 v_mfma_i32_32x32x32_i8 + NV of them per iteration, 2 workgroups of 4 waves per CU = 2 waves per SIMD) — does an SDWA instruction overlap
// with a running MFMA the way a plain VALU instruction does?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ unsigned op(unsigned a) {
    unsigned t;
    if (MODE == 0) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(t) : "s"(3u), "v"(a));
    else if (MODE == 1) asm volatile("v_bfe_u32 %0, %1, 8, 8\n\tv_lshlrev_b32 %0, 3, %0" : "=v"(t) : "v"(a));
    else asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(t) : "v"(a));
    return t;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_alone(unsigned *out, int iters) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = op<MODE>(a[i]) + 0x01010101u * (unsigned)(i + 1);
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// NV operations of kind MODE (MODE 3: none) between 8 MFMAs on 8 independent accumulators
template <int MODE, int NV>
__global__ __launch_bounds__(256, 2) void k_mix(unsigned *out, int iters) {
    v16i acc[8];
    for (int j = 0; j < 8; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u;
    v4i fa = {(int)threadIdx.x, 1, 2, 3}, fb = {4, 5, 6, (int)threadIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc[j], 0, 0, 0);
            if (MODE < 3) {
#pragma unroll
                for (int v = 0; v < NV / 8; ++v) a[(j + v) & 7] = op<MODE>(a[(j + v) & 7]) + 0x01010101u;
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    for (int j = 0; j < 8; ++j)
        for (int e = 0; e < 16; ++e) s ^= (unsigned)acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    unsigned *d;
    CK(hipMalloc(&d, 4096 * 256 * 4));
    const int iters = 2000;
    printf("alone (64 x (op + add) per iteration, %d iterations, 4096 workgroups):\n", iters);
    printf("  sdwa        %.3f ms\n", time_ms([&] { hipLaunchKernelGGL(k_alone<0>, dim3(4096), dim3(256), 0, 0, d, iters); }));
    printf("  bfe + lshl  %.3f ms\n", time_ms([&] { hipLaunchKernelGGL(k_alone<1>, dim3(4096), dim3(256), 0, 0, d, iters); }));
    printf("  bfe         %.3f ms\n", time_ms([&] { hipLaunchKernelGGL(k_alone<2>, dim3(4096), dim3(256), 0, 0, d, iters); }));
    const int g = 512 * 8;   // 8 rounds of 2 workgroups per CU
    printf("beside MFMAs (8 MFMA 32x32x32 i8 + NV x (op + add) per iteration, %d iterations, %d workgroups, 2 per CU):\n", iters, g);
    printf("  MFMA only          %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((k_mix<3, 0>), dim3(g), dim3(256), 0, 0, d, iters); }));
    printf("  + 16 sdwa          %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((k_mix<0, 16>), dim3(g), dim3(256), 0, 0, d, iters); }));
    printf("  + 16 bfe           %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((k_mix<2, 16>), dim3(g), dim3(256), 0, 0, d, iters); }));
    printf("  + 16 (bfe + lshl)  %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((k_mix<1, 16>), dim3(g), dim3(256), 0, 0, d, iters); }));
    printf("  + 32 sdwa          %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((k_mix<0, 32>), dim3(g), dim3(256), 0, 0, d, iters); }));
    printf("  + 32 bfe           %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((k_mix<2, 32>), dim3(g), dim3(256), 0, 0, d, iters); }));
    printf("  + 32 (bfe + lshl)  %.3f ms\n", time_ms([&] { hipLaunchKernelGGL((k_mix<1, 32>), dim3(g), dim3(256), 0, 0, d, iters); }));
    return 0;
}
