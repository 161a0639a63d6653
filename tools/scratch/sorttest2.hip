#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
struct Pay { uint64_t k; uint32_t tag; uint32_t idx; };
int run(int64_t n, hipStream_t s) {
    std::vector<uint16_t> hk(n), ok(n); std::vector<Pay> hv(n), ov(n);
    std::mt19937_64 rng(1);
    for (int64_t i = 0; i < n; ++i) { hk[i] = (uint16_t)(rng() % 20000); hv[i] = Pay{rng(), (uint32_t)rng(), (uint32_t)i}; }
    uint16_t *dk, *dk2; Pay *dv, *dv2; void *tmp; size_t tb = 0;
    CK(hipMalloc(&dk, n*2)); CK(hipMalloc(&dk2, n*2)); CK(hipMalloc(&dv, n*sizeof(Pay))); CK(hipMalloc(&dv2, n*sizeof(Pay)));
    CK(hipMemcpy(dk, hk.data(), n*2, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, hv.data(), n*sizeof(Pay), hipMemcpyHostToDevice));
    CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dk2, dv, dv2, n, 0, 16, s));
    CK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, dk, dk2, dv, dv2, n, 0, 16, s));
    hipEventRecord(e1, s);
    CK(hipMemcpyAsync(ok.data(), dk2, n*2, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(ov.data(), dv2, n*sizeof(Pay), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int64_t bad = 0, unstable = 0, mism = 0;
    for (int64_t i = 1; i < n; ++i) { if (ok[i] < ok[i-1]) ++bad; else if (ok[i] == ok[i-1] && ov[i].idx < ov[i-1].idx) ++unstable; }
    for (int64_t i = 0; i < n; ++i) { const Pay &p = hv[ov[i].idx]; if (hk[ov[i].idx] != ok[i] || p.k != ov[i].k || p.tag != ov[i].tag) ++mism; }
    printf("u16 key + 16B payload n=%lld tmp=%zu bad=%lld unstable=%lld mismatch=%lld  %.3f ms (%.1f GB/s of 2x18B/row)\n", (long long)n, tb, (long long)bad, (long long)unstable, (long long)mism, ms, n*36.0/ms*1e-6);
    hipFree(dk); hipFree(dk2); hipFree(dv); hipFree(dv2); hipFree(tmp);
    return 0;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int64_t n : {1ll, 1000ll, 50000ll, 99999ll, 100001ll, 695825ll, 1000001ll, 5000000ll, 100000000ll}) run(n, s);
    return 0;
}
