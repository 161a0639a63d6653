// how long do hipMalloc / hipHostMalloc / hipFree take as a function of size (the first-pass allocation cost of the engine: DESIGN.md 8)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(0);
    void *w; hipMalloc(&w, 1 << 20); hipFree(w);
    for (size_t mb : {1, 16, 256, 1024, 4096, 8192}) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, mb << 20);
        double t1 = now();
        hipMemset(p, 0, 64); hipDeviceSynchronize();
        double t2 = now();
        hipFree(p);
        double t3 = now();
        printf("hipMalloc %5zu MB: %.3f ms (rc %d)  first touch %.3f ms  hipFree %.3f ms\n", mb, t1 - t0, (int)e, t2 - t1, t3 - t2);
    }
    for (size_t mb : {1, 8, 32}) {
        void *p = nullptr;
        double t0 = now();
        hipHostMalloc(&p, mb << 20, hipHostMallocDefault);
        double t1 = now();
        hipHostFree(p);
        printf("hipHostMalloc %3zu MB: %.3f ms  free %.3f ms\n", mb, t1 - t0, now() - t1);
    }
    double t0 = now();
    std::vector<hipEvent_t> ev(400);
    for (auto &e : ev) hipEventCreate(&e);
    printf("400 x hipEventCreate: %.3f ms\n", now() - t0);
    t0 = now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("hipStreamCreate: %.3f ms\n", now() - t0);
    return 0;
}
