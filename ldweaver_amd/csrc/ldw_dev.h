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
