// does device memory that has just been freed come back slowly?  hipMalloc + first touch of N GB, fresh against right after a hipFree of the same amount; the same for pinned host memory
// (r05: an engine created right after another one was destroyed waits ~1 s in its first calls: tools/hamm_probe.py)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256 * 4096) p[i * 4096 < n ? i * 4096 : 0] = 1;
}
static void round_dev(const char *tag, int nbuf, size_t gb_each) {
    std::vector<void *> ps((size_t)nbuf, nullptr);
    double t0 = now();
    for (auto &p : ps) hipMalloc(&p, gb_each << 30);
    double t1 = now();
    for (auto &p : ps) hipMemsetAsync(p, 0, gb_each << 30, 0);
    hipDeviceSynchronize();
    double t2 = now();
    for (auto &p : ps) hipFree(p);
    double t3 = now();
    printf("%-28s %d x %zu GB: hipMalloc %.1f ms, memset of all of it %.1f ms, hipFree %.1f ms\n", tag, nbuf, gb_each, t1 - t0, t2 - t1, t3 - t2);
}
int main() {
    hipFree(0);
    round_dev("device, fresh", 4, 4);
    round_dev("device, right after the free", 4, 4);
    round_dev("device, again", 4, 4);
    round_dev("device, fresh, larger", 4, 12);
    round_dev("device, right after, larger", 4, 12);
    for (int rep = 0; rep < 3; ++rep) {
        void *p = nullptr;
        double t0 = now();
        hipHostMalloc(&p, (size_t)512 << 20, hipHostMallocDefault);
        double t1 = now();
        hipHostFree(p);
        printf("hipHostMalloc 512 MB (rep %d): %.1f ms, free %.1f ms\n", rep, t1 - t0, now() - t1);
    }
    return 0;
}
