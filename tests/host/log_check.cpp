// Host check of the fp64 logarithm of the MI evaluation — the SAME source the device compiles (ldweaver_amd/csrc/ldw_log.h: fold, polynomial, Newton step);
// only the reciprocal estimate is modelled: 1 / x with a relative error of up to 4.5e-8 (what v_rcp_f64 delivers on gfx950).  Prints the largest |s| the
// integer fold leaves and the largest errors against the long-double logarithm over random and adversarial operand pairs; exit code 0 iff the bounds the
// header states hold.   usage: log_check [pairs = 4e7]        (tests/test_log_host.py builds it with g++ -O2 -mfma -std=c++17 and runs 4e6 pairs)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../ldweaver_amd/csrc/ldw_log.h"

struct HostRcpEstimate {
    double rel;
    double operator()(double x) const { return (1.0 / x) * (1.0 + rel); }
};

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 40000000L;
    std::mt19937_64 g(1988);
    std::uniform_real_distribution<double> ue(-40.0, 40.0), um(1.0, 2.0), ur(-4.5e-8, 4.5e-8);
    double worst_abs = 0, worst_rel = 0, smax = 0;
    for (long it = 0; it < n; ++it) {
        double N, D;
        if (it & 1) {   // ratios near the fold points (2/3, 3/2) and near 1
            D = std::ldexp(um(g), (int)ue(g));
            const double t = (it & 2) ? 1.0 : ((it & 4) ? 1.5 : 0.6667);
            N = D * t * (1.0 + 1e-3 * (um(g) - 1.5)) * std::ldexp(1.0, (int)(ue(g) / 4));
        } else {
            N = std::ldexp(um(g), (int)ue(g));
            D = std::ldexp(um(g), (int)ue(g));
        }
        double s = 0;
        const double v = 2.0 * ldw::half_log_ratio_core(N, D, HostRcpEstimate{ur(g)}, &s);
        if (std::fabs(s) > smax) smax = std::fabs(s);
        const long double ref = logl((long double)N / (long double)D);   // (the quotient's rounding: 5e-20 relative)
        const double e = (double)fabsl((long double)v - ref);
        if (e > worst_abs) worst_abs = e;
        const double r = (double)(e / fmaxl(fabsl(ref), 1e-300L));
        if (fabsl(ref) > 1e-3 && r > worst_rel) worst_rel = r;
    }
    printf("pairs %ld  largest |s| %.6f  largest |error| %.3e  largest relative error (|log| > 1e-3) %.3e\n", n, smax, worst_abs, worst_rel);
    return (smax <= 0.2006 && worst_rel < 5e-15 && worst_abs < 1e-13) ? 0 : 1;
}
