#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template <class K, class V> int run(int64_t n, int b0, int b1, hipStream_t s, const char *what) {
    std::vector<K> hk(n), ok(n); std::vector<V> hv(n), ov(n);
    std::mt19937_64 rng(1);
    for (int64_t i = 0; i < n; ++i) { hk[i] = (K)(((rng() % 20000) << 16) | (rng() & 0xFFFF)); hv[i] = (V)i; }
    K *dk, *dk2; V *dv, *dv2; void *tmp; size_t tb = 0;
    CK(hipMalloc(&dk, n*sizeof(K))); CK(hipMalloc(&dk2, n*sizeof(K))); CK(hipMalloc(&dv, n*sizeof(V))); CK(hipMalloc(&dv2, n*sizeof(V)));
    CK(hipMemcpy(dk, hk.data(), n*sizeof(K), hipMemcpyHostToDevice)); CK(hipMemcpy(dv, hv.data(), n*sizeof(V), hipMemcpyHostToDevice));
    CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dk2, dv, dv2, n, b0, b1, s));
    CK(hipMalloc(&tmp, tb));
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, dk, dk2, dv, dv2, n, b0, b1, s));
    CK(hipMemcpyAsync(ok.data(), dk2, n*sizeof(K), hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(ov.data(), dv2, n*sizeof(V), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    const uint64_t mask = (b1 - b0 >= 64) ? ~0ull : (((1ull << (b1 - b0)) - 1) << b0);
    int64_t bad = 0, unstable = 0, mism = 0;
    for (int64_t i = 1; i < n; ++i) {
        uint64_t a = (uint64_t)ok[i-1] & mask, b = (uint64_t)ok[i] & mask;
        if (b < a) ++bad; else if (a == b && ov[i] < ov[i-1]) ++unstable;
    }
    for (int64_t i = 0; i < n; ++i) if (hk[(size_t)ov[i]] != ok[i]) ++mism;
    printf("%-28s n=%lld bits[%d,%d) tmp=%zu bad=%lld unstable=%lld mismatch=%lld\n", what, (long long)n, b0, b1, tb, (long long)bad, (long long)unstable, (long long)mism);
    hipFree(dk); hipFree(dk2); hipFree(dv); hipFree(dv2); hipFree(tmp);
    return 0;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int64_t n : {1000ll, 695825ll, 5000000ll}) {
        run<uint32_t, uint64_t>(n, 16, 32, s, "u32 key, u64 val");
        run<uint32_t, uint64_t>(n, 0, 32, s, "u32 key, u64 val");
        run<uint32_t, uint32_t>(n, 16, 32, s, "u32 key, u32 val");
        run<uint64_t, uint32_t>(n, 16, 32, s, "u64 key, u32 val");
        run<uint64_t, uint32_t>(n, 0, 64, s, "u64 key, u32 val");
        run<uint64_t, uint64_t>(n, 32, 48, s, "u64 key, u64 val");
    }
    return 0;
}
