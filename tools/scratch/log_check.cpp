// Host check of ldw_epi.h's fast_half_log_ratio (same operations; v_rcp_f64 modelled as 1/x with a 4.5e-8 relative error): the largest |s| the
// integer fold leaves and the largest error against the long-double logarithm over random and adversarial operand pairs.
//   g++ -O2 -mfma -o /tmp/log_check tools/scratch/log_check.cpp && /tmp/log_check
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
static int hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (int)(u >> 32); }
static int lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (int)(uint32_t)u; }
static double mk(int h, int l) { uint64_t u = ((uint64_t)(uint32_t)h << 32) | (uint32_t)l; double x; memcpy(&x, &u, 8); return x; }
static double rcp_model(double x, double rel) { double r = (1.0 / x) * (1.0 + rel); return std::fma(std::fma(-x, r, 1.0), r, r); }
static double smax = 0;
static double half_log_ratio(double N, double D, double rel) {
    const int hn = hi(N), hd = hi(D);
    const int k = (hn - hd + 0x80000) >> 20;
    const double Dp = mk(hd + (k << 20), lo(D));
    const double s = (N - Dp) * rcp_model(N + Dp, rel);
    if (std::fabs(s) > smax) smax = std::fabs(s);
    const double z = s * s;
    double p = 0x1.35c3cc8164535p-4;
    p = std::fma(p, z, 0x1.38feb8144a860p-4);
    p = std::fma(p, z, 0x1.746be3c11806ap-4);
    p = std::fma(p, z, 0x1.c71c3c1108301p-4);
    p = std::fma(p, z, 0x1.24924952daa42p-3);
    p = std::fma(p, z, 0x1.999999997bbebp-3);
    p = std::fma(p, z, 0x1.555555555556ep-2);
    p = p * z;
    return std::fma((double)k, 0.5 * 0.693147180559945309417, std::fma(s, p, s));
}
int main() {
    std::mt19937_64 g(1988);
    std::uniform_real_distribution<double> ue(-40.0, 40.0), um(1.0, 2.0), ur(-4.5e-8, 4.5e-8);
    double worst_abs = 0, worst_rel = 0;
    for (long it = 0; it < 40000000; ++it) {
        double N, D;
        if (it & 1) {   // ratio near the fold points and near 1
            D = std::ldexp(um(g), (int)ue(g));
            const double t = (it & 2) ? 1.0 : ((it & 4) ? 1.5 : 0.6667);
            N = D * t * (1.0 + 1e-3 * (um(g) - 1.5)) * std::ldexp(1.0, (int)(ue(g) / 4));
        } else {
            N = std::ldexp(um(g), (int)ue(g));
            D = std::ldexp(um(g), (int)ue(g));
        }
        const double v = 2.0 * half_log_ratio(N, D, ur(g));
        const long double ref = logl((long double)N / (long double)D);   // (the quotient's rounding: 5e-20 relative)
        const double e = (double)fabsl((long double)v - ref);
        if (e > worst_abs) worst_abs = e;
        const double r = (double)(e / fmaxl(fabsl(ref), 1e-300L));
        if (fabsl(ref) > 1e-3 && r > worst_rel) worst_rel = r;
    }
    printf("largest |s| %.6f   largest |error| %.3e   largest relative error (|log| > 1e-3) %.3e\n", smax, worst_abs, worst_rel);
    return smax < 0.2006 ? 0 : 1;
}
