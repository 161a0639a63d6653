// Small host/device helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace ldw {

// len = 0.5*g - abs((pos1 - pos2) %% g - 0.5*g)   (R/computePairwiseMI.R:330; R's floored %%)
__host__ __device__ __forceinline__ double circ_len(double pos1, double pos2, double g) {
    const double x = pos1 - pos2;
    double d = x - floor(x / g) * g;
    if (d < 0) d += g;
    if (d >= g) d -= g;
    return 0.5 * g - fabs(d - 0.5 * g);
}

// R evaluates these expressions one IEEE operation at a time (its C code is not fma-contracted); hipcc contracts a * b + c by
// default — even through __dmul_rn / __dadd_rn, which are plain operators in the HIP headers — so the three expressions whose
// last bit reaches an output (a block's threshold) are compiled with contraction off.
//   stats::quantile type 7:  index = 1 + (n - 1) p ;  qs = (1 - h) x[lo] + h x[hi]          (R/computePairwiseMI.R:354, :422)
//   prob = 1 - ((lr_retain_links * (n_lr_links / lr_links_approx)) / n_lr_links)               (R/computePairwiseMI.R:352)
__host__ __device__ __forceinline__ double q7_index(double n_minus_1, double p) {
#pragma clang fp contract(off)
    const double t = n_minus_1 * p;
    return 1.0 + t;
}
__host__ __device__ __forceinline__ double q7_interp(double h, double xlo, double xhi) {
#pragma clang fp contract(off)
    const double w = 1.0 - h;
    const double a = w * xlo;
    const double b = h * xhi;
    return a + b;
}
__host__ __device__ __forceinline__ double lr_prob(double lr_retain, double n, double lr_approx) {
#pragma clang fp contract(off)
    const double f = n / lr_approx;
    const double k = lr_retain * f;
    const double q = k / n;
    return 1.0 - q;
}

// order-preserving map double -> uint64 (ascending)
__host__ __device__ __forceinline__ uint64_t f64_key(double v) {
    uint64_t u;
    memcpy(&u, &v, 8);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ __forceinline__ double key_f64(uint64_t k) {
    uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double v;
    memcpy(&v, &u, 8);
    return v;
}

}  // namespace ldw
