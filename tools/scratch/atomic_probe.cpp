// throughput of scattered global atomicAdd (u32) into tables of different sizes: the histogram passes of a sort-free quantile selection
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *tab, uint64_t mask, int64_t n, int per) {
    int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * per;
    uint64_t x = (uint64_t)i0 * 0x9E3779B97F4A7C15ull + 12345;
    for (int k = 0; k < per && i0 + k < n; ++k) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        atomicAdd(&tab[x & mask], 1u);
    }
}
int main() {
    const int64_t n = 1ll << 30;
    for (int bits : {16, 22, 26}) {   // 256 KB, 16 MB, 256 MB tables
        uint32_t *tab; size_t sz = (size_t)4 << bits;
        hipMalloc(&tab, sz); hipMemset(tab, 0, sz);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const int per = 16; const int64_t threads = n / per;
        k<<<dim3((unsigned)(threads / 256)), dim3(256)>>>(tab, ((uint64_t)1 << bits) - 1, n, per);
        hipEventRecord(a);
        k<<<dim3((unsigned)(threads / 256)), dim3(256)>>>(tab, ((uint64_t)1 << bits) - 1, n, per);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("table %4zu MB: %.1f ms for 2^30 atomics = %.1f G atomics/s\n", sz >> 20, ms, n / (ms * 1e-3) / 1e9);
        hipFree(tab);
    }
    return 0;
}
