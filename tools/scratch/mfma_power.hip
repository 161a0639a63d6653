// gfx950: does the int8 MFMA rate depend on the DATA?  8 independent v_mfma_i32_32x32x32_i8 per iteration, 2 workgroups of 4 waves per CU (2 waves per SIMD),
// operands (a) all zero, (b) small constants, (c) random bytes, (d) random bytes re-drawn from a 4-entry ring each iteration; ~40 ms per launch so that the
// power management has time to react.  Prints the time per MFMA per SIMD in ns and the implied fraction of 32 cycles at 2.4 GHz (13.33 ns).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void k(const v4i *__restrict__ src, unsigned *out, int iters, int ring) {
    v16i acc[8];
    for (int j = 0; j < 8; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    v4i fa[4], fb[4];
    for (int r = 0; r < 4; ++r) {
        fa[r] = src[(r * 2 + 0) * 256 + threadIdx.x];
        fb[r] = src[(r * 2 + 1) * 256 + threadIdx.x];
    }
    for (int it = 0; it < iters; ++it) {
        const int r = ring ? (it & 3) : 0;
        const v4i a = r == 0 ? fa[0] : (r == 1 ? fa[1] : (r == 2 ? fa[2] : fa[3]));
        const v4i b = r == 0 ? fb[0] : (r == 1 ? fb[1] : (r == 2 ? fb[2] : fb[3]));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
    }
    unsigned s = 0;
    for (int j = 0; j < 8; ++j)
        for (int e = 0; e < 16; ++e) s ^= (unsigned)acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int n = 8 * 256;
    v4i *h = (v4i *)malloc(n * sizeof(v4i)), *d;
    unsigned *o;
    if (hipMalloc(&d, n * sizeof(v4i)) != hipSuccess || hipMalloc(&o, 4096 * 256 * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, g = 512 * 4;
    const char *names[8] = {"zeros", "small constants", "random bytes", "random bytes, 4-entry ring", "bytes 0..127, 10 % non-zero", "bytes 0..127, 30 % non-zero",
                            "bytes 0..127, 50 % non-zero", "bytes 0..127, all non-zero"};
    for (int mode = 0; mode < 8; ++mode) {
        srand(7);
        const int dens = mode == 4 ? 10 : (mode == 5 ? 30 : (mode == 6 ? 50 : 100));
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < 4; ++q) {
                if (mode < 4) h[i][q] = mode == 0 ? 0 : (mode == 1 ? 0x01020304 : (int)((unsigned)rand() * 2654435761u + (unsigned)rand()));
                else {
                    unsigned w = 0;
                    for (int b = 0; b < 4; ++b) w |= (unsigned)((rand() % 100 < dens) ? 1 + rand() % 127 : 0) << (8 * b);
                    h[i][q] = (int)w;
                }
            }
        (void)hipMemcpy(d, h, n * sizeof(v4i), hipMemcpyHostToDevice);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d, o, iters, mode == 3 ? 1 : 0);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        // per SIMD: g workgroups * 4 waves / 1024 SIMDs, each iters * 8 MFMAs
        const double per = (double)ms * 1e6 / ((double)g * 4 / 1024.0 * iters * 8);
        printf("%-28s %.2f ms per launch, %.2f ns per MFMA per SIMD = %.2f of the 32-cycle rate at 2.4 GHz (%.0f TOP/s chip-wide)\n", names[mode], ms, per, 13.333 / per,
               (double)g * 4 * iters * 8 * 65536.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
