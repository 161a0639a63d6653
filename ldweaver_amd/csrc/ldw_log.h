// The fp64 logarithm of the MI evaluation as ONE source for the device (ldw_epi.h) and for a host program (tests/host/log_check.cpp, built and run by
// tests/test_log_host.py with g++): the fold, the polynomial and the reciprocal's Newton step are these lines on both sides; only the reciprocal
// estimate differs — v_rcp_f64 on the device (good to 4.5e-8, measured on gfx950), 1 / x with an injected relative error of that size on the host.
#pragma once
#include <cmath>
#include <cstdint>
#if defined(__HIPCC__)
#define LDW_LOG_FN __host__ __device__ __forceinline__
#else
#define LDW_LOG_FN inline
#endif

namespace ldw {

LDW_LOG_FN int dbl_hi(double x) { return (int)(__builtin_bit_cast(uint64_t, x) >> 32); }
LDW_LOG_FN int dbl_lo(double x) { return (int)(uint32_t)__builtin_bit_cast(uint64_t, x); }
LDW_LOG_FN double dbl_make(int hi, int lo) { return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); }

// one Newton step on a reciprocal estimate r0 of x: relative error e -> e^2
LDW_LOG_FN double rcp_newton(double x, double r0) { return std::fma(std::fma(-x, r0, 1.0), r0, r0); }

// HALF of log(N / D) for positive, finite, normal doubles with ONE reciprocal and no floating-point compare (r06; before: D rescaled by the
// exponent difference, then two fp64 multiplications and compares against sqrt 2 to fold N / D' into [1/sqrt2, sqrt2], 8 Taylor terms).
// The high words of N and D read as integers are the piecewise-linear logarithm L(x) = e + m (exponent + mantissa fraction), within
// [-0.0861, 0] of log2 x; k = round(L(N) - L(D)) therefore leaves log2(N / (D 2^k)) in [-0.5862, 0.5862] (the low words' 2^-20 included),
// N / D' in [0.666, 1.502], and with s = (N - D') / (N + D'), |s| <= 0.2006:  log(N / D') = 2 atanh(s) = 2 s (1 + z q(z)), z = s^2 <= 0.0403.
// q is a degree-6 polynomial interpolated at the Chebyshev nodes of [0, 0.2006^2] (tools/scratch/log_poly_fit.py, 60-digit arithmetic:
// the logarithm's truncation error is < 3e-17 absolute; a degree-5 one would give 2.6e-15).  The factor 2 is left to the caller, which
// divides the sum over the cells by den / 2 instead of den.  RCP: x -> an estimate of 1 / x (refined here by one Newton step).
// s_out (optional): the folded argument, for the host check of |s| <= 0.2006.
template <class RCP>
LDW_LOG_FN double half_log_ratio_core(double N, double D, RCP rcp_estimate, double *s_out = nullptr) {
    const int hn = dbl_hi(N), hd = dbl_hi(D);
    const int k20 = (hn - hd + 0x80000) & (int)0xFFF00000;     // k 2^20
    const double Dp = dbl_make(hd + k20, dbl_lo(D));           // D 2^k
    const double sum = N + Dp;
    const double s = (N - Dp) * rcp_newton(sum, rcp_estimate(sum));
    if (s_out) *s_out = s;
    const double z = s * s;
    double p = 0x1.35c3cc8164535p-4;
    p = std::fma(p, z, 0x1.38feb8144a860p-4);
    p = std::fma(p, z, 0x1.746be3c11806ap-4);
    p = std::fma(p, z, 0x1.c71c3c1108301p-4);
    p = std::fma(p, z, 0x1.24924952daa42p-3);
    p = std::fma(p, z, 0x1.999999997bbebp-3);
    p = std::fma(p, z, 0x1.555555555556ep-2);
    p = p * z;  // atanh(s)/s - 1
    return std::fma((double)k20, 0x1p-20 * (0.5 * 0.693147180559945309417), std::fma(s, p, s));   // (k 2^20 converts exactly; the constant is ln2 / 2 scaled by a power of two)
}

}  // namespace ldw
