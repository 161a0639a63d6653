// MI epilogue, link selection and the block drivers (twins of perform_MI_computation_ACGTN,
// R/computePairwiseMI.R:167-386, and of the block loop of perform_MI_computation, :103-116).
//
// Per block:  GEMM (ldw_gemm_bits.hip) -> k_mi_epilogue: one thread per SNP pair turns the fixed-point joint
// sums into MI (src/computeMI.cpp:19), writes the dense MI block, scatters the short-range links straight
// to their final rows and histograms the long-range MI values in LDS -> k_pick_bucket: ranks of the two
// order statistics of quantile type 7 and the histogram bucket holding them -> k_lr_gather: every
// long-range pair at or above that bucket -> two radix sorts (by MI, then by reference row order) with
// k_lr_thresh in between -> k_lr_append.  The short-range test needs no arithmetic per pair: POS is
// ascending, so the partners of a to-side SNP within sr_dist (circularly) are at most three index
// intervals of the from-side list, found on the host by binary search (ColInfo).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <hipcub/hipcub.hpp>

#include "ldw_internal.h"
#include "ldw_dev.h"

using namespace ldw;

namespace ldw {

// ------------------------------------------------------------------------------------------------
// helpers shared by host and device
// ------------------------------------------------------------------------------------------------
// Histogram bucket of an MI value: monotone non-decreasing in mi, 128 buckets per octave (0.5 % wide) from 2^-20
// up, taken straight from the IEEE-754 bits (exponent + 7 mantissa bits) — no floating-point arithmetic.  Values
// below 2^-20 (and negatives, which quirk Q1 can produce) fall in bucket 0.
constexpr int BUCKET_OFF = (1023 - 20) << 7;
__host__ __device__ __forceinline__ int mi_bucket(double mi) {
    long long u;
    memcpy(&u, &mi, 8);
    if (u <= 0) return 0;  // -x, -0, +0
    const int b = (int)(u >> 45) - BUCKET_OFF;
    return b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
}
// lower edge of bucket B (B >= 1)
__host__ __device__ __forceinline__ double bucket_lo(int B) {
    const long long u = (long long)(B + BUCKET_OFF) << 45;
    double v;
    memcpy(&v, &u, 8);
    return v;
}

// Short-range partners of one to-side SNP: up to three disjoint, ascending index intervals [s,e) of the
// from-side list, plus the first row of its upper (a_loc < b_loc) and lower (a_loc > b_loc) segment in
// the short-range table (relative to the block's base row).
struct ColInfo {
    int32_t s[3], e[3];
    int32_t pad[2];
    int64_t off_u, off_l;
};

__host__ __device__ __forceinline__ bool col_is_sr(const ColInfo &c, int a) {
    return (a >= c.s[0] && a < c.e[0]) || (a >= c.s[1] && a < c.e[1]) || (a >= c.s[2] && a < c.e[2]);
}
// number of short-range partners in [lo, hi)
__host__ __device__ __forceinline__ int col_count(const ColInfo &c, int lo, int hi) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = c.s[k] > lo ? c.s[k] : lo, b = c.e[k] < hi ? c.e[k] : hi;
        n += b > a ? b - a : 0;
    }
    return n;
}

// which segment a pair belongs to: 0 = upper (a<b, off-diagonal blocks only), 1 = lower (a>b), -1 = not a pair
__host__ __device__ __forceinline__ int pair_seg(int a_loc, int b_loc, int lower_only) {
    if (a_loc == b_loc) return -1;
    if (a_loc > b_loc) return 1;
    return lower_only ? -1 : 0;
}

// ------------------------------------------------------------------------------------------------
// fp64 helpers of the epilogue: no IEEE division, no libm call.  Accuracy ~2e-15; the
// epilogue is not bit-matched to the reference (tolerance 1e-6 on MI), see DESIGN.md.
// ------------------------------------------------------------------------------------------------
// v_rcp_f64 is good to 4.5e-8 (measured on gfx950); one Newton step brings it to 2e-15
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// exact conversion of an integer 0 <= n < 2^52 to double (two integer ops and one add instead of the
// multi-instruction int64 -> f64 sequence)
__device__ __forceinline__ double u52_to_double(int64_t n) {
    return __longlong_as_double(n | 0x4330000000000000LL) - 4503599627370496.0;
}

// log(N / D) for positive, finite, normal doubles with ONE reciprocal: D is rescaled by a power of two so that
// N / D' lies in [1/sqrt2, sqrt2]; then log(N/D') = 2 atanh(s), s = (N - D')/(N + D'), |s| <= 0.1716, odd series
// to s^17 (absolute error < 1e-15), and log(N/D) = k ln2 + log(N/D').
__device__ __forceinline__ double fast_log_ratio(double N, double D) {
    const int hn = __double2hiint(N), hd = __double2hiint(D);
    int k = ((hn >> 20) & 0x7FF) - ((hd >> 20) & 0x7FF);
    double Dp = __hiloint2double(hd + (k << 20), __double2loint(D));   // D * 2^k: same exponent as N
    // N / Dp is in (1/2, 2): fold it into [1/sqrt2, sqrt2]
    const bool big = N > Dp * 1.4142135623730951, small = N * 1.4142135623730951 < Dp;
    const int adj = big ? 1 : (small ? -1 : 0);
    Dp = __hiloint2double(__double2hiint(Dp) + (adj << 20), __double2loint(Dp));
    k += adj;
    const double s = (N - Dp) * fast_rcp(N + Dp);
    const double z = s * s;
    double p = 1.0 / 17.0;           // truncation z^9/19 <= 9e-16 relative to 2s
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = p * z;  // atanh(s)/s - 1
    const double lm = fma(s + s, p, s + s);
    return fma((double)k, 0.693147180559945309417, lm);
}

// ------------------------------------------------------------------------------------------------
// what happens to one finished pair: dense store, short-range scatter, long-range histogram
// ------------------------------------------------------------------------------------------------
struct EmitArgs {
    double *MI;            // dense block, column-major nf x nt
    const ColInfo *cols;   // null: dense store only (ldw_mi_block)
    int nf, lower_only, keep_sr, do_lr;
    int write_dense;       // store the dense MI block (needed by k_lr_gather; off when the gather is speculative)
    int spec_B;            // >= 0: append long-range pairs with bucket >= spec_B to the candidate list right here,
                           //       and histogram ONLY those (the pairs below are counted analytically)
    int any_sr;            // 0: no pair of this block is short-range (skip the interval tests)
    double spec_lo;        // lower edge of bucket spec_B minus a guard: cheap reject before the bucket arithmetic
    int64_t sr_base;
    int32_t *sr_a, *sr_b;
    double *sr_mi;
    unsigned long long *n_cand;
    uint64_t *ckey, *cval;
};

__device__ __forceinline__ void emit_pair(const EmitArgs &E, const ColInfo &c, int a_loc, int b_loc, int sa, int sb,
                                          double mi, unsigned int *sh_hist) {
    if (E.write_dense) E.MI[(int64_t)a_loc + (int64_t)b_loc * E.nf] = mi;
    if (!E.cols) return;
    const int seg = pair_seg(a_loc, b_loc, E.lower_only);
    if (seg < 0) return;
    if (E.any_sr && col_is_sr(c, a_loc)) {
        if (E.keep_sr) {
            const int64_t dst = E.sr_base + (seg == 0 ? c.off_u + col_count(c, 0, a_loc) : c.off_l + col_count(c, b_loc + 1, a_loc));
            E.sr_a[dst] = sa;
            E.sr_b[dst] = sb;
            E.sr_mi[dst] = mi;
        }
    } else if (E.do_lr) {
        if (E.spec_B >= 0) {
            // speculative mode: only the (rare) pairs at or above the guessed bucket are histogrammed and appended
            if (mi >= E.spec_lo) {
                const int bk = mi_bucket(mi);
                if (bk >= E.spec_B) {
                    atomicAdd(&sh_hist[bk], 1u);
                    const unsigned long long p = atomicAdd(E.n_cand, 1ull);
                    E.ckey[p] = f64_key(mi);
                    E.cval[p] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)E.nf);
                }
            }
        } else {
            atomicAdd(&sh_hist[mi_bucket(mi)], 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MI epilogue: one thread per SNP pair; a wave = 64 consecutive from-side SNPs at one to-side SNP and walks
// EPI_COLS/4 consecutive to-side SNPs, so everything indexed by the to-side SNP is wave-uniform.
// ------------------------------------------------------------------------------------------------
constexpr int EPI_COLS = 128;  // to-side SNPs per workgroup (4 waves x 32)

struct EpiArgs {
    const int64_t *G;
    int RFpad;
    const int32_t *idx_f, *lrow_f, *idx_t, *lrow_t;
    int nf, nt;
    const uint32_t *slot_meta;
    const int64_t *slot_pfix;
    const double *r;
    double neff, scale;
    int quirk;
    EmitArgs E;
};

// everything the epilogue needs about one to-side SNP, staged in LDS once per workgroup so that the
// per-pair loop has no dependent global loads except its G entries
struct ColMeta {
    int32_t sb;
    uint32_t mb;
    int32_t rb0, pad;
    double rb;      // r of the to-side SNP
    double rq;      // Q1 on square blocks: r[idx_f[b_loc]]
    double pYd[5];
    int64_t pb[5];
    ColInfo ci;
};

// per-lane constants of the from-side SNP
struct RowSide {
    int sa, na;
    uint32_t ma;
    int64_t ra0;
    double ra, rta;  // rta: Q1 on square blocks, r[idx_t[a_loc]]
    int64_t pa[5];
    double pXd[5];
};

// MI of one pair.  NAM / NB bound the unrolled slot loops (na <= NAM for every lane of the wave, nb <= NB);
// the run-time slot counts still mask the individual cells.
template <int NAM, int NB>
__device__ __forceinline__ double pair_mi(const EpiArgs &A, const RowSide &R, const ColMeta &M, int a_loc, int b_loc,
                                          bool square) {
    const int na = R.na, nb = M.mb & 7;
    const uint32_t ma = R.ma, mb = M.mb;
    // joint sums of the row slots (from G), their row / column sums
    int64_t g[NAM][NB], rs[NAM], cs[NB];
#pragma unroll
    for (int i = 0; i < NAM; ++i) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) cs[j] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NAM; ++i) {
            int64_t v = 0;
            if (i < na && j < nb) v = A.G[((int64_t)M.rb0 + j) * A.RFpad + R.ra0 + i];
            g[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int64_t pa_drop = 0;
#pragma unroll
    for (int i = 0; i <= NAM; ++i)
        if (i == na) pa_drop = R.pa[i];
    int64_t dd = pa_drop;
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (j < nb) dd -= M.pb[j] - cs[j];

    const double ra = R.ra, rb = M.rb;
    const double den = A.neff + (ra * rb) * 0.5;  // R/computePairwiseMI.R:260
    double RXY;
    if (A.quirk == LDW_QUIRK_REFERENCE) {
        // rft is nt x nf but read by the linear index c = a + b*nf of the nf x nt matrix (Q1):
        // 0.25 * rf[c / nt] * rt[c % nt]; on square blocks c / nt = b_loc and c % nt = a_loc
        if (square) {
            RXY = (M.rq * R.rta) * 0.25;
        } else {
            const uint32_t c = (uint32_t)a_loc + (uint32_t)b_loc * (uint32_t)A.nf;
            const uint32_t q = c / (uint32_t)A.nt;
            RXY = (A.r[A.idx_f[q]] * A.r[A.idx_t[c - q * (uint32_t)A.nt]]) * 0.25;
        }
    } else {
        RXY = (ra * rb) * 0.25;
    }
    const double rX = 0.5 * ra, rY = 0.5 * rb;

    // sum over cells of pxy * log(pxy / (pX pY + RXY + pX rX + pY rY) * den), divided by den at the end
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i <= NAM; ++i) {
        if (i <= na && ((ma >> (3 + i)) & 1)) {
            const double pX = R.pXd[i];
            const double pXr = fma(pX, rX, RXY);
#pragma unroll
            for (int j = 0; j <= NB; ++j) {
                if (j <= nb && ((mb >> (3 + j)) & 1)) {
                    int64_t nfix;
                    if (i < NAM && j < NB && i < na && j < nb) nfix = g[i < NAM ? i : 0][j < NB ? j : 0];
                    else if (i < NAM && i < na) nfix = R.pa[i] - rs[i < NAM ? i : 0];   // j == nb
                    else if (j < NB && j < nb) nfix = M.pb[j] - cs[j < NB ? j : 0];     // i == na
                    else nfix = dd;
                    const double pY = M.pYd[j];
                    const double pxy = fma(u52_to_double(nfix), A.scale, 0.5);
                    const double d = fma(pY, rY, fma(pX, pY, pXr));
                    acc = fma(pxy, fast_log_ratio(pxy * den, d), acc);
                }
            }
        }
    }
    return acc * fast_rcp(den);
}

// Straight-line variant for the common case: every lane has exactly NA row slots, the column has exactly NB,
// and every slot of both SNPs is flagged in uqe — no per-cell predication, every index static.
template <int NA, int NB>
__device__ __forceinline__ double pair_mi_full(const EpiArgs &A, const RowSide &R, const ColMeta &M, int a_loc, int b_loc,
                                               bool square) {
    int64_t g[NA > 0 ? NA : 1][NB > 0 ? NB : 1], rs[NA > 0 ? NA : 1], cs[NB > 0 ? NB : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) cs[j] = 0;
    const int64_t base = (int64_t)M.rb0 * A.RFpad + R.ra0;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int64_t v = A.G[base + (int64_t)j * A.RFpad + i];
            g[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int64_t dd = R.pa[NA];
#pragma unroll
    for (int j = 0; j < NB; ++j) dd -= M.pb[j] - cs[j];

    const double ra = R.ra, rb = M.rb;
    const double den = A.neff + (ra * rb) * 0.5;
    double RXY;
    if (A.quirk == LDW_QUIRK_REFERENCE) {
        if (square) {
            RXY = (M.rq * R.rta) * 0.25;
        } else {
            const uint32_t c = (uint32_t)a_loc + (uint32_t)b_loc * (uint32_t)A.nf;
            const uint32_t q = c / (uint32_t)A.nt;
            RXY = (A.r[A.idx_f[q]] * A.r[A.idx_t[c - q * (uint32_t)A.nt]]) * 0.25;
        }
    } else {
        RXY = (ra * rb) * 0.25;
    }
    const double rX = 0.5 * ra, rY = 0.5 * rb;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i <= NA; ++i) {
        const double pX = R.pXd[i];
        const double pXr = fma(pX, rX, RXY);
#pragma unroll
        for (int j = 0; j <= NB; ++j) {
            int64_t nfix;
            if (i < NA && j < NB) nfix = g[i < NA ? i : 0][j < NB ? j : 0];
            else if (i < NA) nfix = R.pa[i] - rs[i < NA ? i : 0];
            else if (j < NB) nfix = M.pb[j] - cs[j < NB ? j : 0];
            else nfix = dd;
            const double pY = M.pYd[j];
            const double pxy = fma(u52_to_double(nfix), A.scale, 0.5);
            const double d = fma(pY, rY, fma(pX, pY, pXr));
            acc = fma(pxy, fast_log_ratio(pxy * den, d), acc);
        }
    }
    return acc * fast_rcp(den);
}

__global__ __launch_bounds__(256) void k_mi_epilogue(EpiArgs A, const int32_t *__restrict__ perm_f,
                                                     unsigned long long *__restrict__ ghist) {
    __shared__ unsigned int sh_hist[NBINS];
    __shared__ ColMeta cm[EPI_COLS];
    const bool use_hist = A.E.cols && A.E.do_lr;
    if (use_hist)
        for (int i = threadIdx.x; i < NBINS; i += 256) sh_hist[i] = 0;
    const bool square = A.nf == A.nt;
    if (threadIdx.x < EPI_COLS) {
        const int b_loc = blockIdx.y * EPI_COLS + threadIdx.x;
        if (b_loc < A.nt) {
            ColMeta m;
            m.sb = A.idx_t[b_loc];
            m.mb = A.slot_meta[m.sb];
            m.rb0 = A.lrow_t[b_loc];
            m.pad = 0;
            m.rb = A.r[m.sb];
            m.rq = square ? A.r[A.idx_f[b_loc]] : 0.0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                m.pb[j] = A.slot_pfix[(int64_t)m.sb * 5 + j];
                m.pYd[j] = (double)m.pb[j] * A.scale;
            }
            if (A.E.cols) m.ci = A.E.cols[b_loc];
            cm[threadIdx.x] = m;
        }
    }
    __syncthreads();

    // lanes walk the from-side SNPs in an order that groups equal slot counts, so that a wave runs the same cells
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = threadIdx.x >> 6;
    const bool a_ok = t < A.nf;
    const int a_loc = perm_f[a_ok ? t : A.nf - 1];
    RowSide R;
    R.sa = A.idx_f[a_loc];
    R.ma = A.slot_meta[R.sa];
    R.na = a_ok ? (int)(R.ma & 7) : 0;
    R.ra0 = A.lrow_f[a_loc];
    R.ra = A.r[R.sa];
    R.rta = (square && a_ok) ? A.r[A.idx_t[a_loc]] : 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        R.pa[i] = A.slot_pfix[(int64_t)R.sa * 5 + i];
        R.pXd[i] = (double)R.pa[i] * A.scale;
    }
    const int na_max = (__ballot(R.na > 2) != 0ull) ? 4 : ((__ballot(R.na > 1) != 0ull) ? 2 : 1);
    // wave-uniform: every active lane has the same slot count (1 or 2) and all its slots flagged in uqe
    const int na0 = __builtin_amdgcn_readfirstlane(R.na);
    const bool a_full = a_ok && R.na == na0 && (((R.ma >> 3) & ((2u << na0) - 1u)) == ((2u << na0) - 1u));
    const bool wave_full = (na0 == 1 || na0 == 2) && __ballot(!a_full) == 0ull;

    const int c_first = wave * (EPI_COLS / 4);
    const int b_base = blockIdx.y * EPI_COLS;
    int n_it = A.nt - (b_base + c_first);
    n_it = n_it > EPI_COLS / 4 ? EPI_COLS / 4 : n_it;
    for (int it = 0; it < n_it; ++it) {
        const int cl = c_first + it;
        const int b_loc = b_base + cl;
        if (!a_ok) continue;
        if (A.E.lower_only && a_loc <= b_loc) continue;
        const ColMeta &M = cm[cl];
        const int nb = __builtin_amdgcn_readfirstlane((int)(M.mb & 7));
        double mi;
        const bool b_full = ((M.mb >> 3) & ((2u << nb) - 1u)) == ((2u << nb) - 1u);
        if (wave_full && b_full && (nb == 1 || nb == 2)) {
            if (na0 == 1) mi = nb == 1 ? pair_mi_full<1, 1>(A, R, M, a_loc, b_loc, square) : pair_mi_full<1, 2>(A, R, M, a_loc, b_loc, square);
            else mi = nb == 1 ? pair_mi_full<2, 1>(A, R, M, a_loc, b_loc, square) : pair_mi_full<2, 2>(A, R, M, a_loc, b_loc, square);
        } else if (na_max == 1) {
            if (nb <= 1) mi = pair_mi<1, 1>(A, R, M, a_loc, b_loc, square);
            else if (nb == 2) mi = pair_mi<1, 2>(A, R, M, a_loc, b_loc, square);
            else mi = pair_mi<1, 4>(A, R, M, a_loc, b_loc, square);
        } else if (na_max == 2) {
            if (nb <= 1) mi = pair_mi<2, 1>(A, R, M, a_loc, b_loc, square);
            else if (nb == 2) mi = pair_mi<2, 2>(A, R, M, a_loc, b_loc, square);
            else mi = pair_mi<2, 4>(A, R, M, a_loc, b_loc, square);
        } else {
            if (nb <= 1) mi = pair_mi<4, 1>(A, R, M, a_loc, b_loc, square);
            else if (nb == 2) mi = pair_mi<4, 2>(A, R, M, a_loc, b_loc, square);
            else mi = pair_mi<4, 4>(A, R, M, a_loc, b_loc, square);
        }
        emit_pair(A.E, M.ci, a_loc, b_loc, R.sa, M.sb, mi, sh_hist);
    }
    if (use_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < NBINS; i += 256)
            if (sh_hist[i]) atomicAdd(&ghist[i], (unsigned long long)sh_hist[i]);
    }
}

// the same emission for an MI block produced elsewhere (LDW_ENGINE_HIST)
__global__ __launch_bounds__(256) void k_post_mi(EmitArgs E, const int32_t *__restrict__ idx_f,
                                                 const int32_t *__restrict__ idx_t, int nt,
                                                 unsigned long long *__restrict__ ghist) {
    __shared__ unsigned int sh_hist[NBINS];
    for (int i = threadIdx.x; i < NBINS; i += 256) sh_hist[i] = 0;
    __syncthreads();
    const int a_loc = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = threadIdx.x >> 6;
    const int b_first = blockIdx.y * EPI_COLS + wave * (EPI_COLS / 4);
    if (a_loc < E.nf) {
        const int sa = idx_f[a_loc];
        for (int it = 0; it < EPI_COLS / 4; ++it) {
            const int b_loc = b_first + it;
            if (b_loc >= nt) break;
            if (E.lower_only && a_loc <= b_loc) continue;
            const ColInfo c = E.cols[b_loc];
            emit_pair(E, c, a_loc, b_loc, sa, idx_t[b_loc], E.MI[(int64_t)a_loc + (int64_t)b_loc * E.nf], sh_hist);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += 256)
        if (sh_hist[i]) atomicAdd(&ghist[i], (unsigned long long)sh_hist[i]);
}

// ------------------------------------------------------------------------------------------------
// joint tables for explicit pairs (test / inspection API): table[p][X][Y] from G entry (p, p)
// ------------------------------------------------------------------------------------------------
__global__ void k_tables(const int64_t *__restrict__ G, int RFpad, const int32_t *idx_f, const int32_t *lrow_f,
                         const int32_t *idx_t, const int32_t *lrow_t, int np, const uint32_t *slot_meta,
                         const int64_t *marg /* [L][5] by slot */, int64_t *out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    const int sa = idx_f[p], sb = idx_t[p];
    const uint32_t ma = slot_meta[sa], mb = slot_meta[sb];
    const int na = ma & 7, nb = mb & 7;
    int64_t cell[5][5];
    for (int i = 0; i <= 4; ++i)
        for (int j = 0; j <= 4; ++j) cell[i][j] = 0;
    int64_t rs[5] = {0, 0, 0, 0, 0}, cs[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            const int64_t v = G[((int64_t)lrow_t[p] + j) * RFpad + lrow_f[p] + i];
            cell[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    int64_t dd = marg[(int64_t)sa * 5 + na];
    for (int i = 0; i < na; ++i) cell[i][nb] = marg[(int64_t)sa * 5 + i] - rs[i];
    for (int j = 0; j < nb; ++j) {
        cell[na][j] = marg[(int64_t)sb * 5 + j] - cs[j];
        dd -= cell[na][j];
    }
    cell[na][nb] = dd;
    int64_t *o = out + (int64_t)p * 25;
    for (int k = 0; k < 25; ++k) o[k] = 0;
    for (int i = 0; i <= na; ++i)
        for (int j = 0; j <= nb; ++j) {
            const int X = (ma >> (8 + 3 * i)) & 7, Y = (mb >> (8 + 3 * j)) & 7;
            o[X * 5 + Y] = cell[i][j];
        }
}

// unit-weight marginals by slot = state counts reordered
__global__ void k_slot_counts(const int32_t *counts, const uint32_t *slot_meta, int64_t L, int64_t *out) {
    const int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (a >= L) return;
    const uint32_t m = slot_meta[a];
    const int n = m & 7;
    for (int i = 0; i < 5; ++i) out[a * 5 + i] = (i <= n) ? counts[a * 5 + ((m >> (8 + 3 * i)) & 7)] : 0;
}

// ------------------------------------------------------------------------------------------------
// long-range selection
// ------------------------------------------------------------------------------------------------
struct PickOut {
    long long n;        // number of long-range pairs in the block
    long long lo, hi;   // 1-based ranks of the two order statistics of quantile type 7
    long long n_below;  // pairs in buckets below B
    double index, prob;
    int B;              // first bucket gathered
    int spec_ok;        // the speculative candidate list of the epilogue covers bucket B
    int B_true, pad2;   // bucket that holds rank lo (before the speculative override)
    unsigned long long n_cand;  // filled by k_lr_gather
    long long n_kept;           // filled by k_lr_thresh
    double disc_thresh;
    long long kstart;
};

// prob and quantile ranks of R/computePairwiseMI.R:352-354 (stats::quantile type 7), then the bucket
// holding rank lo.  One workgroup: chunked prefix sum over the NBINS counters.
// In speculative mode (spec_B >= 0) the histogram only holds buckets >= spec_B; n_total is the block's long-range
// pair count known to the host, and everything below spec_B is lumped into one virtual bucket.
__global__ __launch_bounds__(256) void k_pick_bucket(const unsigned long long *__restrict__ hist, double lr_retain,
                                                     double lr_approx, int spec_B, long long n_total,
                                                     PickOut *__restrict__ out) {
    __shared__ long long part[256];
    __shared__ long long s_lo;
    constexpr int PER = NBINS / 256;
    const int t = threadIdx.x;
    long long loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        loc[k] = (long long)hist[t * PER + k];
        sum += loc[k];
    }
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        long long above = 0;
        for (int i = 0; i < 256; ++i) above += part[i];
        // speculative mode: pairs below spec_B were not histogrammed
        long long run = spec_B >= 0 ? n_total - above : 0;
        for (int i = 0; i < 256; ++i) {
            const long long v = part[i];
            part[i] = run;
            run += v;
        }
        PickOut o;
        memset(&o, 0, sizeof(o));
        o.n_cand = out->n_cand;  // speculative candidates appended by the epilogue
        o.n = run;
        o.B = NBINS;
        o.B_true = NBINS;
        o.disc_thresh = nan("");
        if (run > 0) {
            const double dn = (double)run;
            double prob = 1.0 - ((lr_retain * (dn / lr_approx)) / dn);
            if (!(prob > 0.0)) prob = 0.0;
            o.prob = prob;
            o.index = 1.0 + (dn - 1.0) * prob;
            o.lo = (long long)floor(o.index);
            o.hi = (long long)ceil(o.index);
        }
        *out = o;
        s_lo = o.lo;
    }
    __syncthreads();
    const long long lo = s_lo;
    if (lo <= 0) return;
    __shared__ int s_B;
    long long cum = part[t];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (cum < lo && cum + loc[k] >= lo) {  // exactly one (thread, k) satisfies this
            out->B = t * PER + k;
            out->B_true = t * PER + k;
            out->n_below = cum;
            s_B = t * PER + k;
        }
        cum += loc[k];
    }
    __syncthreads();
    if (spec_B < 0) return;
    // speculative list = every long-range pair with bucket >= spec_B: usable iff rank lo lies at or above spec_B,
    // i.e. iff some bucket >= spec_B was found to hold it (the unhistogrammed mass below spec_B precedes part[0])
    if (lo > part[0]) {
        cum = part[t];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (t * PER + k == spec_B) {
                out->B = spec_B;
                out->n_below = cum;
                out->spec_ok = 1;
            }
            cum += loc[k];
        }
    } else if (t == 0) {
        out->n_cand = 0;  // guess too high: the fallback gather starts from an empty list
    }
}

struct GatherArgs {
    const double *MI;
    const ColInfo *cols;
    int nf, nt, lower_only;
};

// gather every long-range pair whose bucket is >= B: (MI key, order key)
__global__ __launch_bounds__(256) void k_lr_gather(GatherArgs S, PickOut *__restrict__ pick, uint64_t *__restrict__ ckey,
                                                   uint64_t *__restrict__ cval) {
    const int B = pick->B;
    if (B >= NBINS) return;
    const double lo_val = B > 0 ? bucket_lo(B) : 0.0;
    const int b0 = blockIdx.x * 16;
    for (int bb = 0; bb < 16; ++bb) {
        const int b_loc = b0 + bb;
        if (b_loc >= S.nt) break;
        const ColInfo c = S.cols[b_loc];
        const double *col = S.MI + (int64_t)b_loc * S.nf;
        for (int a_loc = threadIdx.x; a_loc < S.nf; a_loc += 256) {
            const int seg = pair_seg(a_loc, b_loc, S.lower_only);
            if (seg < 0) continue;  // also skips the never-written part of a diagonal block
            const double mi = col[a_loc];
            if (B > 0 && mi < lo_val) continue;  // cheap reject; the bucket test below decides
            if (col_is_sr(c, a_loc)) continue;
            if (mi_bucket(mi) < B) continue;
            const unsigned long long p = atomicAdd(&pick->n_cand, 1ull);
            ckey[p] = f64_key(mi);
            cval[p] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)S.nf);
        }
    }
}

// candidates sorted ascending by MI: quantile type 7, then first kept index
__global__ void k_lr_thresh(const uint64_t *__restrict__ skey, PickOut *__restrict__ pick) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PickOut o = *pick;
    const long long m = (long long)o.n_cand;
    if (o.n <= 0 || m <= 0) {
        pick->n_kept = 0;
        pick->kstart = 0;
        return;
    }
    const long long off = o.n - m;  // ranks below the candidate set
    const double xlo = key_f64(skey[o.lo - off - 1]);
    const double xhi = key_f64(skey[o.hi - off - 1]);
    double qs = xlo;
    if (o.index > (double)o.lo && xhi != qs) {
        const double h = o.index - (double)o.lo;
        qs = (1.0 - h) * qs + h * xhi;
    }
    const uint64_t kq = f64_key(qs);
    long long lo = 0, hi = m;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (skey[mid] < kq) lo = mid + 1; else hi = mid;
    }
    pick->disc_thresh = qs;
    pick->kstart = lo;
    pick->n_kept = m - lo;
}

// kept candidates keep their order key, the rest sink to the end of the second sort
__global__ void k_lr_mark(const uint64_t *__restrict__ skey, const uint64_t *__restrict__ sval,
                          const PickOut *__restrict__ pick, uint64_t *__restrict__ okey, uint64_t *__restrict__ oval,
                          long long m) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const bool keep = i >= pick->kstart;
    okey[i] = keep ? sval[i] : ~0ull;
    oval[i] = skey[i];
}

__global__ void k_lr_append(const uint64_t *__restrict__ okey, const uint64_t *__restrict__ oval,
                            const PickOut *__restrict__ pick, const int32_t *__restrict__ idx_f,
                            const int32_t *__restrict__ idx_t, int nf, const int64_t *__restrict__ lr_count,
                            int32_t *__restrict__ out_a, int32_t *__restrict__ out_b, double *__restrict__ out_mi) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= pick->n_kept) return;
    const uint64_t c = okey[i] & 0x3FFFFFFFFFFFFFFFull;
    const int a_loc = (int)(c % (uint64_t)nf), b_loc = (int)(c / (uint64_t)nf);
    const int64_t dst = *lr_count + i;
    out_a[dst] = idx_f[a_loc];
    out_b[dst] = idx_t[b_loc];
    out_mi[dst] = key_f64(oval[i]);
}

// running device-side counters and per-block stats
__global__ void k_block_done(PickOut *pick, int64_t *lr_count, int64_t n_sr_blk, int64_t *stats_i, double *stats_d) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    *lr_count += pick->n_kept;
    stats_i[0] = pick->n;
    stats_i[1] = pick->n_kept;
    stats_i[2] = n_sr_blk;
    stats_d[0] = pick->disc_thresh;
}

}  // namespace ldw

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

int upload_i32(ldw_ctx *c, ldw::DevBuf &buf, const std::vector<int32_t> &v) {
    if (int rc = buf.reserve(v.size() * 4 + 4)) return rc;
    if (!v.empty()) LDW_HIP(hipMemcpyAsync(buf.p, v.data(), v.size() * 4, hipMemcpyHostToDevice, c->stream));
    return LDW_OK;
}

// row lists / local row offsets of one side
int build_side(ldw_ctx *c, const int32_t *idx, int64_t n, std::vector<int32_t> &rowlist, std::vector<int32_t> &lrow,
               int &Rpad) {
    lrow.resize((size_t)n);
    rowlist.clear();
    for (int64_t k = 0; k < n; ++k) {
        const int32_t a = idx[k];
        LDW_REQUIRE(a >= 0 && a < c->L, LDW_ERR_ARG, "SNP index %d out of range 0..%lld", a, (long long)c->L - 1);
        lrow[k] = (int32_t)rowlist.size();
        for (int32_t rr = c->h_row0[a]; rr < c->h_row0[a + 1]; ++rr) rowlist.push_back(rr);
    }
    int64_t rp = ((int64_t)rowlist.size() + TILE - 1) / TILE * TILE;
    if (rp == 0) rp = TILE;
    LDW_REQUIRE(rp < 2000000000LL, LDW_ERR_ARG, "block too large");
    rowlist.resize((size_t)rp, (int32_t)c->R);  // padding rows point at the zero rows behind M
    Rpad = (int)rp;
    return LDW_OK;
}

// Short-range intervals of every to-side SNP (host, O(nt log nf)); returns the block's short-range row count.
// Requires the from-side list to be ascending in POS (true for contiguous blocks and for the order-preserving
// subsets of SR-only mode).  Every interval boundary is verified with the exact predicate of the reference
// (circ_len <= sr_dist); a column that fails the check is rebuilt by scanning.
int build_cols(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, bool diag, double sr_dist,
               std::vector<ColInfo> &cols, int64_t &n_sr_blk) {
    const double g = c->g;
    std::vector<double> pf((size_t)nf);
    for (int64_t a = 0; a < nf; ++a) {
        pf[a] = (double)c->h_POS[from_idx[a]];
        LDW_REQUIRE(a == 0 || pf[a] >= pf[a - 1], LDW_ERR_ARG, "from-side SNP list must be ascending in POS (position %lld)", (long long)a);
    }
    cols.resize((size_t)nt);
    {   // quick reject: when the two position ranges are farther apart than sr_dist both directly and around the
        // origin, no pair of the block is short-range (true for most off-diagonal blocks)
        double pt_min = (double)c->h_POS[to_idx[0]], pt_max = pt_min;
        for (int64_t b = 1; b < nt; ++b) {
            const double v = (double)c->h_POS[to_idx[b]];
            pt_min = v < pt_min ? v : pt_min;
            pt_max = v > pt_max ? v : pt_max;
        }
        const double pf_min = pf.front(), pf_max = pf.back();
        const bool apart = (pt_min - pf_max > sr_dist && pf_min + g - pt_max > sr_dist) ||
                           (pf_min - pt_max > sr_dist && pt_min + g - pf_max > sr_dist);
        if (apart && 2 * sr_dist < g) {
            ColInfo z;
            memset(&z, 0, sizeof(z));
            std::fill(cols.begin(), cols.end(), z);
            n_sr_blk = 0;
            return LDW_OK;
        }
    }
    int64_t total_u = 0, total_l = 0;
    std::vector<int32_t> cu((size_t)nt), cl((size_t)nt);
    for (int64_t b = 0; b < nt; ++b) {
        const double p1 = (double)c->h_POS[to_idx[b]];
        auto P = [&](int64_t a) { return circ_len(p1, pf[a], g) <= sr_dist; };
        ColInfo ci;
        memset(&ci, 0, sizeof(ci));
        // candidate intervals from the three position windows
        int64_t iv[3][2];
        iv[0][0] = 0;                                                                                // wrap-low: x <= p1 + sr - g
        iv[0][1] = std::upper_bound(pf.begin(), pf.end(), p1 + sr_dist - g) - pf.begin();
        iv[1][0] = std::lower_bound(pf.begin(), pf.end(), p1 - sr_dist) - pf.begin();               // centre
        iv[1][1] = std::upper_bound(pf.begin(), pf.end(), p1 + sr_dist) - pf.begin();
        iv[2][0] = std::lower_bound(pf.begin(), pf.end(), p1 - sr_dist + g) - pf.begin();           // wrap-high
        iv[2][1] = nf;
        // merge overlapping / touching intervals, keep ascending order
        int n = 0;
        int64_t m[3][2];
        for (int k = 0; k < 3; ++k) {
            if (iv[k][1] <= iv[k][0]) continue;
            if (n > 0 && iv[k][0] <= m[n - 1][1]) m[n - 1][1] = std::max(m[n - 1][1], iv[k][1]);
            else { m[n][0] = iv[k][0]; m[n][1] = iv[k][1]; ++n; }
        }
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) {
            ok = P(m[k][0]) && P(m[k][1] - 1);
            if (ok && m[k][0] > 0) ok = !P(m[k][0] - 1);
            if (ok && m[k][1] < nf) ok = !P(m[k][1]);
        }
        if (n == 0 && nf > 0) ok = !P(0) && !P(nf - 1);
        if (!ok) {  // rebuild from the exact predicate
            n = 0;
            int64_t a = 0;
            while (a < nf) {
                if (!P(a)) { ++a; continue; }
                int64_t e = a;
                while (e < nf && P(e)) ++e;
                LDW_REQUIRE(n < 3, LDW_ERR_ARG, "short-range partners of SNP %d form more than three index runs", to_idx[b]);
                m[n][0] = a; m[n][1] = e; ++n;
                a = e;
            }
        }
        for (int k = 0; k < 3; ++k) {
            ci.s[k] = k < n ? (int32_t)m[k][0] : 0;
            ci.e[k] = k < n ? (int32_t)m[k][1] : 0;
        }
        cu[b] = diag ? 0 : col_count(ci, 0, (int)std::min<int64_t>(b, nf));
        cl[b] = col_count(ci, (int)std::min<int64_t>(b + 1, nf), (int)nf);
        total_u += cu[b];
        total_l += cl[b];
        cols[b] = ci;
    }
    // row order of the block: all upper-segment columns, then all lower-segment columns (R/computePairwiseMI.R:306-310)
    int64_t ru = 0, rl = total_u;
    for (int64_t b = 0; b < nt; ++b) {
        cols[b].off_u = ru;
        cols[b].off_l = rl;
        ru += cu[b];
        rl += cl[b];
    }
    n_sr_blk = total_u + total_l;
    return LDW_OK;
}

bool same_list(const int32_t *a, int64_t na, const int32_t *b, int64_t nb) {
    if (na != nb) return false;
    for (int64_t i = 0; i < na; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

// device pointers of one block's index structures
struct DevPtrs {
    const int32_t *idx_f, *idx_t, *rl_f, *rl_t, *lrow_f, *lrow_t, *perm;
};

// GEMM + epilogue (or the histogram engine) of one block into ctx->MIblk; with E.cols set, the short-range
// scatter and the long-range histogram ride along.  ev[0..2] are recorded before the GEMM, between the two
// kernels and after the epilogue.  Everything is asynchronous on ctx->stream.
// stage events per block: [0] GEMM start, [1] GEMM end (GEMM stream); [4] epilogue start, [2] epilogue end, [3] selection
// end (main stream).  The GEMM of block b+1 runs beside the epilogue and selection of block b, so the stage times overlap.
constexpr int EVB = 5;
// which: 1 = GEMM only (on gstream, into Gbuf), 2 = epilogue only, 3 = both
int launch_block_mi(ldw_ctx *c, const DevPtrs &D, int64_t nf, int64_t nt, int RFpad, int RTpad, int quirk, EmitArgs E,
                    hipEvent_t *ev, int which = 3, ldw::DevBuf *Gb = nullptr, hipStream_t gstream = nullptr) {
    ldw::DevBuf &Gbuf = Gb ? *Gb : c->G;
    if (!gstream) gstream = c->stream;
    if (which == 1) {   // the GEMM of a block, possibly on its own stream so that it overlaps the previous block's tail
        if (int rc = Gbuf.reserve((size_t)RFpad * RTpad * 8)) return rc;
        LDW_HIP(hipEventRecord(ev[0], gstream));
        if (int rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, D.rl_t, RTpad, D.rl_f, RFpad, Gbuf.as<int64_t>(), c->nlimbs,
                                      c->digits.as<int8_t>(), E.lower_only, gstream))
            return rc;
        LDW_HIP(hipEventRecord(ev[1], gstream));
        return LDW_OK;
    }
    const bool epilogue_only = which == 2;
    if (int rc = c->MIblk.reserve((size_t)nf * nt * 8)) return rc;
    if (int rc = c->hist.reserve((size_t)NBINS * 8)) return rc;
    E.MI = c->MIblk.as<double>();
    E.nf = (int)nf;
    dim3 egrid((unsigned)((nf + 63) / 64), (unsigned)((nt + EPI_COLS - 1) / EPI_COLS));
    LDW_REQUIRE(egrid.y <= 65535u, LDW_ERR_ARG, "nt too large for the epilogue grid");
    unsigned long long *ghist = c->hist.as<unsigned long long>();
    if (c->engine == LDW_ENGINE_HIST) {
        LDW_HIP(hipEventRecord(ev[0], c->stream));
        LDW_HIP(hipEventRecord(ev[1], c->stream));
        if (int rc = launch_hist(c, D.idx_f, (int)nf, D.idx_t, (int)nt, c->pfix_state.as<int64_t>(), quirk, E.lower_only,
                                 c->MIblk.as<double>()))
            return rc;
        if (E.cols) {
            hipLaunchKernelGGL(k_post_mi, egrid, dim3(256), 0, c->stream, E, D.idx_f, D.idx_t, (int)nt, ghist);
            LDW_HIP(hipGetLastError());
        }
        LDW_HIP(hipEventRecord(ev[2], c->stream));
        return LDW_OK;
    }
    if (!epilogue_only) {
        if (int rc = Gbuf.reserve((size_t)RFpad * RTpad * 8)) return rc;
        LDW_HIP(hipEventRecord(ev[0], c->stream));
        if (int rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, D.rl_t, RTpad, D.rl_f, RFpad, Gbuf.as<int64_t>(), c->nlimbs,
                                      c->digits.as<int8_t>(), E.lower_only, c->stream))
            return rc;
        LDW_HIP(hipEventRecord(ev[1], c->stream));
    }
    EpiArgs A;
    A.G = Gbuf.as<int64_t>();
    A.RFpad = RFpad;
    A.idx_f = D.idx_f;
    A.lrow_f = D.lrow_f;
    A.idx_t = D.idx_t;
    A.lrow_t = D.lrow_t;
    A.nf = (int)nf;
    A.nt = (int)nt;
    A.slot_meta = c->slot_meta.as<uint32_t>();
    A.slot_pfix = c->slot_pfix.as<int64_t>();
    A.r = c->r.as<double>();
    A.neff = c->neff;
    A.scale = std::ldexp(1.0, -c->frac_bits);
    A.quirk = quirk;
    A.E = E;
    if (which == 2) LDW_HIP(hipEventRecord(ev[4], c->stream));
    hipLaunchKernelGGL(k_mi_epilogue, egrid, dim3(256), 0, c->stream, A, D.perm, ghist);
    LDW_HIP(hipGetLastError());
    if (which != 2 || c->engine != LDW_ENGINE_HIST) LDW_HIP(hipEventRecord(ev[2], c->stream));
    return LDW_OK;
}

// lane order of the epilogue: from-side SNPs grouped by their number of indicator rows (1, 2, 3, 4, 0)
void build_perm(ldw_ctx *c, const int32_t *from_idx, int64_t nf, int32_t *perm) {
    int64_t w = 0;
    for (int want : {1, 2, 3, 4, 0})
        for (int64_t k = 0; k < nf; ++k)
            if (c->h_row0[from_idx[k] + 1] - c->h_row0[from_idx[k]] == want) perm[w++] = (int32_t)k;
}

// dense MI of one block, synchronous staging through ctx-owned buffers (ldw_mi_block, ldw_joint_tables style)
int run_block_mi(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, int quirk,
                 EmitArgs E) {
    if (int rc = ensure_rows(c)) return rc;
    LDW_REQUIRE(nf > 0 && nt > 0, LDW_ERR_ARG, "empty block (nf=%lld nt=%lld)", (long long)nf, (long long)nt);
    LDW_REQUIRE(nf <= 1000000 && nt <= 1000000, LDW_ERR_ARG, "block side too long");
    std::vector<int32_t> vf(from_idx, from_idx + nf), vt(to_idx, to_idx + nt), rl_f, rl_t, lr_f, lr_t, perm((size_t)nf);
    int RFpad = 0, RTpad = 0;
    if (int rc = build_side(c, from_idx, nf, rl_f, lr_f, RFpad)) return rc;
    if (int rc = build_side(c, to_idx, nt, rl_t, lr_t, RTpad)) return rc;
    build_perm(c, from_idx, nf, perm.data());
    if (int rc = upload_i32(c, c->idx_f, vf)) return rc;
    if (int rc = upload_i32(c, c->idx_t, vt)) return rc;
    if (int rc = upload_i32(c, c->rowlist_f, rl_f)) return rc;
    if (int rc = upload_i32(c, c->rowlist_t, rl_t)) return rc;
    if (int rc = upload_i32(c, c->lrow_f, lr_f)) return rc;
    if (int rc = upload_i32(c, c->lrow_t, lr_t)) return rc;
    if (int rc = upload_i32(c, c->perm_f, perm)) return rc;
    LDW_HIP(hipStreamSynchronize(c->stream));  // pageable H2D copies are complete only after a sync
    DevPtrs D{c->idx_f.as<int32_t>(), c->idx_t.as<int32_t>(), c->rowlist_f.as<int32_t>(), c->rowlist_t.as<int32_t>(),
              c->lrow_f.as<int32_t>(), c->lrow_t.as<int32_t>(), c->perm_f.as<int32_t>()};
    E.write_dense = 1;
    E.spec_B = -1;
    return launch_block_mi(c, D, nf, nt, RFpad, RTpad, quirk, E, c->ev);
}

int ensure_links_capacity(ldw_ctx *c, int64_t sr_rows, int64_t lr_rows) {
    if (int rc = c->sr_a.reserve_keep((size_t)sr_rows * 4, (size_t)c->n_sr * 4, c->stream)) return rc;
    if (int rc = c->sr_b.reserve_keep((size_t)sr_rows * 4, (size_t)c->n_sr * 4, c->stream)) return rc;
    if (int rc = c->sr_mi.reserve_keep((size_t)sr_rows * 8, (size_t)c->n_sr * 8, c->stream)) return rc;
    if (int rc = c->lr_a.reserve_keep((size_t)lr_rows * 4, (size_t)c->n_lr * 4, c->stream)) return rc;
    if (int rc = c->lr_b.reserve_keep((size_t)lr_rows * 4, (size_t)c->n_lr * 4, c->stream)) return rc;
    if (int rc = c->lr_mi.reserve_keep((size_t)lr_rows * 8, (size_t)c->n_lr * 8, c->stream)) return rc;
    return LDW_OK;
}

// layout of ctx->small during link selection
struct SmallLayout {
    int64_t *lr_count;   // running number of kept long-range rows (device side)
    ldw::PickOut *pick;
    int64_t *stats_i;    // [capacity][3]
    double *stats_d;     // [capacity]
};

void links_layout(ldw_ctx *c, SmallLayout &sl) {
    char *base = c->small.as<char>();
    sl.lr_count = reinterpret_cast<int64_t *>(base);
    sl.pick = reinterpret_cast<ldw::PickOut *>(base + 64);
    sl.stats_i = reinterpret_cast<int64_t *>(base + 64 + ((sizeof(ldw::PickOut) + 63) / 64) * 64);
    sl.stats_d = reinterpret_cast<double *>(sl.stats_i + c->blk_capacity * 3);
}

// ---- one block of the link loop in three phases so that the host work of block i+1 overlaps the GPU work of
// ---- block i:  prep (pure host, into pinned memory)  ->  submit (upload + kernels up to the candidate gather)
// ---- ->  finish (the one host round trip: candidate count, then sorts / threshold / append)
struct HostBlock {
    int64_t nf = 0, nt = 0, n_sr_blk = 0, n_lr_total = 0, blk_no = 0;
    int RFpad = 0, RTpad = 0, slot = 0;
    bool diag = false;
    size_t o_idx_f = 0, o_idx_t = 0, o_rl_f = 0, o_rl_t = 0, o_lrow_f = 0, o_lrow_t = 0, o_perm = 0, o_cols = 0, total = 0;
    // set by submit_block, used by the fallback of finish_block
    DevPtrs D{};
    EmitArgs E{};
    int spec_B = -1;
};

int prep_block(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, const ldw_mi_params *p,
               int slot, int64_t blk_no, HostBlock &hb) {
    LDW_REQUIRE(nf > 0 && nt > 0, LDW_ERR_ARG, "empty block (nf=%lld nt=%lld)", (long long)nf, (long long)nt);
    LDW_REQUIRE(nf <= 1000000 && nt <= 1000000 && nf * nt < 2147483647LL, LDW_ERR_ARG, "block too large (%lld x %lld)",
                (long long)nf, (long long)nt);
    for (int64_t k = 0; k < nf; ++k)
        LDW_REQUIRE(from_idx[k] >= 0 && from_idx[k] < c->L, LDW_ERR_ARG, "SNP index %d out of range", from_idx[k]);
    for (int64_t k = 0; k < nt; ++k)
        LDW_REQUIRE(to_idx[k] >= 0 && to_idx[k] < c->L, LDW_ERR_ARG, "SNP index %d out of range", to_idx[k]);
    hb.nf = nf;
    hb.nt = nt;
    hb.slot = slot;
    hb.blk_no = blk_no;
    hb.diag = same_list(from_idx, nf, to_idx, nt);
    std::vector<int32_t> rl_f, rl_t, lr_f, lr_t;
    if (int rc = build_side(c, from_idx, nf, rl_f, lr_f, hb.RFpad)) return rc;
    if (int rc = build_side(c, to_idx, nt, rl_t, lr_t, hb.RTpad)) return rc;
    std::vector<ColInfo> cols;
    if (int rc = build_cols(c, from_idx, nf, to_idx, nt, hb.diag, p->sr_dist, cols, hb.n_sr_blk)) return rc;
    hb.n_lr_total = (hb.diag ? nf * (nf - 1) / 2 : nf * nt - std::min(nf, nt)) - hb.n_sr_blk;
    auto al = [](size_t x) { return (x + 63) / 64 * 64; };
    size_t o = 0;
    hb.o_idx_f = o; o = al(o + (size_t)nf * 4);
    hb.o_idx_t = o; o = al(o + (size_t)nt * 4);
    hb.o_rl_f = o; o = al(o + rl_f.size() * 4);
    hb.o_rl_t = o; o = al(o + rl_t.size() * 4);
    hb.o_lrow_f = o; o = al(o + (size_t)nf * 4);
    hb.o_lrow_t = o; o = al(o + (size_t)nt * 4);
    hb.o_perm = o; o = al(o + (size_t)nf * 4);
    hb.o_cols = o; o = al(o + cols.size() * sizeof(ColInfo));
    hb.total = o;
    if (c->pin_cap[slot] < o) {
        if (c->pin[slot]) LDW_HIP(hipHostFree(c->pin[slot]));
        c->pin[slot] = nullptr;
        c->pin_cap[slot] = 0;
        LDW_HIP(hipHostMalloc(&c->pin[slot], o * 2, hipHostMallocDefault));
        c->pin_cap[slot] = o * 2;
    }
    char *b = static_cast<char *>(c->pin[slot]);
    memcpy(b + hb.o_idx_f, from_idx, (size_t)nf * 4);
    memcpy(b + hb.o_idx_t, to_idx, (size_t)nt * 4);
    memcpy(b + hb.o_rl_f, rl_f.data(), rl_f.size() * 4);
    memcpy(b + hb.o_rl_t, rl_t.data(), rl_t.size() * 4);
    memcpy(b + hb.o_lrow_f, lr_f.data(), (size_t)nf * 4);
    memcpy(b + hb.o_lrow_t, lr_t.data(), (size_t)nt * 4);
    build_perm(c, from_idx, nf, reinterpret_cast<int32_t *>(b + hb.o_perm));
    memcpy(b + hb.o_cols, cols.data(), cols.size() * sizeof(ColInfo));
    return LDW_OK;
}

int launch_gather(ldw_ctx *c, const HostBlock &hb, const EmitArgs &E, const SmallLayout &sl) {
    GatherArgs S;
    S.MI = c->MIblk.as<double>();
    S.cols = E.cols;
    S.nf = (int)hb.nf;
    S.nt = (int)hb.nt;
    S.lower_only = E.lower_only;
    hipLaunchKernelGGL(k_lr_gather, dim3((unsigned)((hb.nt + 15) / 16)), dim3(256), 0, c->stream, S, sl.pick,
                       c->cand_key.as<uint64_t>(), c->cand_val.as<uint64_t>());
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

// First half of a block: upload its index structures and run the co-occurrence GEMM into this slot's G buffer on
// the GEMM stream.  Nothing here touches what the previous block's epilogue / selection still uses, so the GEMM of
// block b+1 overlaps the tail of block b (epilogue wind-down, the launch-bound selection kernels, the host round trip).
int submit_gemm(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p) {
    const int s = hb.slot;
    if (int rc = c->dstage[s].reserve(hb.total)) return rc;
    // the device image and the G buffer of this slot were last read by the block two steps back
    if (c->done_recorded[s]) LDW_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_done[s], 0));
    LDW_HIP(hipMemcpyAsync(c->dstage[s].p, c->pin[s], hb.total, hipMemcpyHostToDevice, c->copy_stream));
    LDW_HIP(hipEventRecord(c->ev_up[s], c->copy_stream));
    const char *d = c->dstage[s].as<char>();
    auto I = [&](size_t off) { return reinterpret_cast<const int32_t *>(d + off); };
    hb.D = DevPtrs{I(hb.o_idx_f), I(hb.o_idx_t), I(hb.o_rl_f), I(hb.o_rl_t), I(hb.o_lrow_f), I(hb.o_lrow_t), I(hb.o_perm)};
    if (c->engine != LDW_ENGINE_MFMA) return LDW_OK;
    hipStream_t gs = c->overlap ? c->gemm_stream : c->stream;   // overlap off: the stages of all blocks run back to back
    LDW_HIP(hipStreamWaitEvent(gs, c->ev_up[s], 0));
    if (c->done_recorded[s]) LDW_HIP(hipStreamWaitEvent(gs, c->ev_done[s], 0));
    EmitArgs E;
    memset(&E, 0, sizeof(E));
    E.lower_only = hb.diag ? 1 : 0;
    if (int rc = launch_block_mi(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, E, &c->ev_pool[(size_t)hb.blk_no * EVB], 1,
                                 s ? &c->G2 : &c->G, gs))
        return rc;
    LDW_HIP(hipEventRecord(c->ev_gemm[s], gs));
    return LDW_OK;
}

// Second half: epilogue, histogram pick and the copy-back of the pick on the main stream.
int submit_block(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl) {
    const int s = hb.slot;
    LDW_HIP(hipStreamWaitEvent(c->stream, c->ev_up[s], 0));
    if (c->engine == LDW_ENGINE_MFMA) LDW_HIP(hipStreamWaitEvent(c->stream, c->ev_gemm[s], 0));
    const char *d = c->dstage[s].as<char>();
    const DevPtrs D = hb.D;
    const bool do_lr = !p->sr_only;
    const int64_t sr_add = p->keep_sr ? hb.n_sr_blk : 0;
    if (int rc = ensure_links_capacity(c, c->n_sr + sr_add, c->n_lr)) return rc;
    if (int rc = c->hist.reserve((size_t)NBINS * 8)) return rc;
    LDW_HIP(hipMemsetAsync(c->hist.p, 0, (size_t)NBINS * 8, c->stream));
    EmitArgs E;
    memset(&E, 0, sizeof(E));
    E.cols = reinterpret_cast<const ColInfo *>(d + hb.o_cols);
    E.lower_only = hb.diag ? 1 : 0;
    E.keep_sr = p->keep_sr ? 1 : 0;
    E.do_lr = do_lr ? 1 : 0;
    E.sr_base = c->n_sr;
    E.sr_a = c->sr_a.as<int32_t>();
    E.sr_b = c->sr_b.as<int32_t>();
    E.sr_mi = c->sr_mi.as<double>();
    // Long-range candidates: with a bucket guess from the previous block the epilogue appends them itself and the
    // dense MI block is neither written nor re-read; without one (first block, histogram engine) the dense block
    // is written and k_lr_gather collects them once the true bucket is known.
    hb.spec_B = (do_lr && c->engine == LDW_ENGINE_MFMA) ? c->spec_B_next[hb.diag ? 1 : 0] : -1;
    const size_t cap = (size_t)hb.nf * hb.nt;  // worst case: every pair of the block
    if (do_lr) {
        if (int rc = c->cand_key.reserve(cap * 8)) return rc;
        if (int rc = c->cand_val.reserve(cap * 8)) return rc;
    }
    E.write_dense = hb.spec_B < 0 ? 1 : 0;
    E.spec_B = hb.spec_B;
    E.spec_lo = hb.spec_B > 0 ? bucket_lo(hb.spec_B) : -1e300;
    E.any_sr = hb.n_sr_blk > 0 ? 1 : 0;
    E.n_cand = &sl.pick->n_cand;
    E.ckey = c->cand_key.as<uint64_t>();
    E.cval = c->cand_val.as<uint64_t>();
    hb.E = E;
    LDW_HIP(hipMemsetAsync(sl.pick, 0, sizeof(ldw::PickOut), c->stream));
    hipEvent_t *ev = &c->ev_pool[(size_t)hb.blk_no * EVB];
    if (int rc = launch_block_mi(c, D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, E, ev, c->engine == LDW_ENGINE_MFMA ? 2 : 3,
                                 s ? &c->G2 : &c->G))
        return rc;
    c->n_sr += sr_add;
    if (do_lr) {
        hipLaunchKernelGGL(k_pick_bucket, dim3(1), dim3(256), 0, c->stream, c->hist.as<unsigned long long>(),
                           p->lr_retain_links, p->lr_links_approx, hb.spec_B, (long long)hb.n_lr_total, sl.pick);
        LDW_HIP(hipGetLastError());
        if (hb.spec_B < 0)
            if (int rc = launch_gather(c, hb, E, sl)) return rc;
    }
    LDW_HIP(hipMemcpyAsync(c->pin_pick, sl.pick, sizeof(ldw::PickOut), hipMemcpyDeviceToHost, c->stream));
    // exact number of long-range rows kept by all EARLIER blocks (this block's append has not run yet)
    LDW_HIP(hipMemcpyAsync(static_cast<char *>(c->pin_pick) + sizeof(ldw::PickOut), sl.lr_count, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipEventRecord(c->ev_pick, c->stream));
    return LDW_OK;
}

int finish_block(ldw_ctx *c, HostBlock &hb, const ldw_mi_params *p, const SmallLayout &sl) {
    const bool do_lr = !p->sr_only;
    // ---- the one host round trip of the block: the candidate count sizes the sorts ----
    LDW_HIP(hipEventSynchronize(c->ev_pick));
    ldw::PickOut *hp = static_cast<ldw::PickOut *>(c->pin_pick);
    if (do_lr && hb.spec_B >= 0 && hp->n > 0 && !hp->spec_ok) {
        // the bucket guess was above the true bucket: redo the epilogue non-speculatively (G is still intact, the
        // short-range rows are already final): full histogram, dense store, then pick and gather with the true bucket
        EmitArgs E = hb.E;
        E.write_dense = 1;
        E.spec_B = -1;
        E.keep_sr = 0;
        LDW_HIP(hipMemsetAsync(c->hist.p, 0, (size_t)NBINS * 8, c->stream));
        LDW_HIP(hipMemsetAsync(sl.pick, 0, sizeof(ldw::PickOut), c->stream));
        hipEvent_t dummy[5] = {c->ev[3], c->ev[3], c->ev[4], c->ev[3], c->ev[5]};   // keep the block's stage events as they are
        if (int rc = launch_block_mi(c, hb.D, hb.nf, hb.nt, hb.RFpad, hb.RTpad, p->quirk_mode, E, dummy, 2, hb.slot ? &c->G2 : &c->G))
            return rc;
        hipLaunchKernelGGL(k_pick_bucket, dim3(1), dim3(256), 0, c->stream, c->hist.as<unsigned long long>(),
                           p->lr_retain_links, p->lr_links_approx, -1, 0LL, sl.pick);
        LDW_HIP(hipGetLastError());
        if (int rc = launch_gather(c, hb, E, sl)) return rc;
        LDW_HIP(hipMemcpyAsync(c->pin_pick, sl.pick, sizeof(ldw::PickOut), hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        ++c->spec_misses;
    }
    if (do_lr && hp->n > 0) {  // next block's guess: a little below this block's bucket
        // buckets are 0.5 % wide: guess ~5 % below the threshold of the last block of the same kind (diagonal blocks
        // lose their closest pairs to the short-range table and sit ~8 % lower than off-diagonal ones)
        const int margin = 10;
        c->spec_B_next[hb.diag ? 1 : 0] = hp->B_true - margin > 0 ? hp->B_true - margin : 0;
    }
    const int64_t m = do_lr ? (int64_t)hp->n_cand : 0;
    {   // tighten the upper bound on the long-range row count: exact up to the previous block + this block's candidates
        int64_t exact_before = 0;
        memcpy(&exact_before, static_cast<char *>(c->pin_pick) + sizeof(ldw::PickOut), 8);
        c->n_lr = exact_before;
    }
    const int64_t nf = hb.nf;
    const char *d = c->dstage[hb.slot].as<char>();
    const int32_t *idx_f = reinterpret_cast<const int32_t *>(d + hb.o_idx_f), *idx_t = reinterpret_cast<const int32_t *>(d + hb.o_idx_t);
    if (do_lr && m > 0) {
        LDW_REQUIRE(m < 2147483647LL, LDW_ERR_SIZE, "too many quantile candidates (%lld)", (long long)m);
        if (int rc = ensure_links_capacity(c, c->n_sr, c->n_lr + m)) return rc;
        if (int rc = c->cand_key2.reserve((size_t)m * 8)) return rc;
        if (int rc = c->cand_val2.reserve((size_t)m * 8)) return rc;
        size_t tmp_bytes = 0;
        LDW_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, c->cand_key.as<uint64_t>(),
                                                   c->cand_key2.as<uint64_t>(), c->cand_val.as<uint64_t>(),
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        if (int rc = c->scratch.reserve(tmp_bytes)) return rc;
        LDW_HIP(hipcub::DeviceRadixSort::SortPairs(c->scratch.p, tmp_bytes, c->cand_key.as<uint64_t>(),
                                                   c->cand_key2.as<uint64_t>(), c->cand_val.as<uint64_t>(),
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        hipLaunchKernelGGL(k_lr_thresh, dim3(1), dim3(64), 0, c->stream, c->cand_key2.as<uint64_t>(), sl.pick);
        LDW_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_lr_mark, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                           c->cand_key2.as<uint64_t>(), c->cand_val2.as<uint64_t>(), sl.pick,
                           c->cand_key.as<uint64_t>(), c->cand_val.as<uint64_t>(), (long long)m);
        LDW_HIP(hipGetLastError());
        LDW_HIP(hipcub::DeviceRadixSort::SortPairs(c->scratch.p, tmp_bytes, c->cand_key.as<uint64_t>(),
                                                   c->cand_key2.as<uint64_t>(), c->cand_val.as<uint64_t>(),
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        hipLaunchKernelGGL(k_lr_append, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                           c->cand_key2.as<uint64_t>(), c->cand_val2.as<uint64_t>(), sl.pick, idx_f, idx_t, (int)nf,
                           sl.lr_count, c->lr_a.as<int32_t>(), c->lr_b.as<int32_t>(), c->lr_mi.as<double>());
        LDW_HIP(hipGetLastError());
        c->n_lr += m;  // upper bound; the exact value is *lr_count
    } else if (do_lr) {
        hipLaunchKernelGGL(k_lr_thresh, dim3(1), dim3(64), 0, c->stream, c->cand_key.as<uint64_t>(), sl.pick);
        LDW_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_block_done, dim3(1), dim3(64), 0, c->stream, sl.pick, sl.lr_count, hb.n_sr_blk,
                       sl.stats_i + hb.blk_no * 3, sl.stats_d + hb.blk_no);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipEventRecord(c->ev_pool[(size_t)hb.blk_no * EVB + 3], c->stream));
    LDW_HIP(hipEventRecord(c->ev_done[hb.slot], c->stream));
    c->done_recorded[hb.slot] = true;
    return LDW_OK;
}

}  // namespace

extern "C" {

int ldw_mi_block(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, int quirk_mode,
                 double *MI_out, int on_device) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(from_idx && to_idx && MI_out, LDW_ERR_ARG, "ldw_mi_block: null argument");
    LDW_REQUIRE(quirk_mode == LDW_QUIRK_REFERENCE || quirk_mode == LDW_QUIRK_INTENDED, LDW_ERR_ARG, "bad quirk mode");
    EmitArgs E;
    memset(&E, 0, sizeof(E));
    if (int rc = run_block_mi(c, from_idx, nf, to_idx, nt, quirk_mode, E)) return rc;
    LDW_HIP(hipMemcpyAsync(MI_out, c->MIblk.p, (size_t)nf * nt * 8, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    float t01 = 0, t12 = 0;
    LDW_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
    LDW_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
    c->last_ms[0] = t01;
    c->last_ms[1] = t12;
    c->last_ms[2] = 0;
    c->last_ms[3] = t01 + t12;
    return LDW_OK;
}

int ldw_joint_tables(ldw_ctx *c, const int32_t *pair_a, const int32_t *pair_b, int64_t np, int64_t *counts_out,
                     int64_t *fixed_out, int *frac_bits_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(pair_a && pair_b && np > 0, LDW_ERR_ARG, "ldw_joint_tables: bad argument");
    if (int rc = ensure_rows(c)) return rc;
    if (frac_bits_out) *frac_bits_out = c->frac_bits;
    const int64_t CH = 1024;
    std::vector<int8_t> ones((size_t)c->Npad, 1);
    ldw::DevBuf d_ones, d_out, d_cmarg;
    int rc = LDW_OK;
    auto cleanup = [&]() { d_ones.release(); d_out.release(); d_cmarg.release(); };
    if ((rc = d_ones.reserve((size_t)c->Npad)) || (rc = d_out.reserve((size_t)CH * 25 * 8)) ||
        (rc = d_cmarg.reserve((size_t)c->L * 40))) {
        cleanup();
        return rc;
    }
    hipError_t he = hipMemcpyAsync(d_ones.p, ones.data(), ones.size(), hipMemcpyHostToDevice, c->stream);
    if (he != hipSuccess) { cleanup(); return ldw::hip_fail(he, "memcpy ones", __FILE__, __LINE__); }
    hipLaunchKernelGGL(k_slot_counts, dim3((unsigned)((c->L + 255) / 256)), dim3(256), 0, c->stream,
                       c->counts.as<int32_t>(), c->slot_meta.as<uint32_t>(), c->L, d_cmarg.as<int64_t>());
    for (int64_t p0 = 0; p0 < np && rc == LDW_OK; p0 += CH) {
        const int64_t n = std::min(CH, np - p0);
        std::vector<int32_t> rl_f, rl_t, lr_f, lr_t;
        int RFpad = 0, RTpad = 0;
        if ((rc = build_side(c, pair_a + p0, n, rl_f, lr_f, RFpad))) break;
        if ((rc = build_side(c, pair_b + p0, n, rl_t, lr_t, RTpad))) break;
        if ((rc = upload_i32(c, c->rowlist_f, rl_f)) || (rc = upload_i32(c, c->rowlist_t, rl_t)) ||
            (rc = upload_i32(c, c->lrow_f, lr_f)) || (rc = upload_i32(c, c->lrow_t, lr_t)))
            break;
        std::vector<int32_t> vf(pair_a + p0, pair_a + p0 + n), vt(pair_b + p0, pair_b + p0 + n);
        if ((rc = upload_i32(c, c->idx_f, vf)) || (rc = upload_i32(c, c->idx_t, vt))) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = LDW_ERR_HIP; break; }
        if ((rc = c->G.reserve((size_t)RFpad * RTpad * 8))) break;
        for (int pass = 0; pass < 2 && rc == LDW_OK; ++pass) {
            int64_t *host_out = pass == 0 ? counts_out : fixed_out;
            if (!host_out) continue;
            rc = launch_gemm_bits(c, c->Mbits.as<uint64_t>(), c->KW, c->rowlist_t.as<int32_t>(), RTpad, c->rowlist_f.as<int32_t>(), RFpad, c->G.as<int64_t>(),
                                  pass == 0 ? 1 : c->nlimbs, pass == 0 ? d_ones.as<int8_t>() : c->digits.as<int8_t>(), 0);
            if (rc) break;
            hipLaunchKernelGGL(k_tables, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, c->G.as<int64_t>(), RFpad,
                               c->idx_f.as<int32_t>(), c->lrow_f.as<int32_t>(), c->idx_t.as<int32_t>(),
                               c->lrow_t.as<int32_t>(), (int)n, c->slot_meta.as<uint32_t>(),
                               pass == 0 ? d_cmarg.as<int64_t>() : c->slot_pfix.as<int64_t>(), d_out.as<int64_t>());
            he = hipMemcpyAsync(host_out + p0 * 25, d_out.p, (size_t)n * 25 * 8, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) rc = ldw::hip_fail(he, "joint tables copy", __FILE__, __LINE__);
        }
    }
    cleanup();
    return rc;
}

int ldw_links_begin(ldw_ctx *c, int64_t nblocks_capacity) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(nblocks_capacity > 0, LDW_ERR_ARG, "ldw_links_begin: capacity must be positive");
    if (int rc = ensure_rows(c)) return rc;  // uses ctx->small for staging; link bookkeeping takes it over below
    const size_t need = 64 + ((sizeof(ldw::PickOut) + 63) / 64) * 64 + (size_t)nblocks_capacity * 32 + 64;
    if (int rc = c->small.reserve(need)) return rc;
    LDW_HIP(hipMemsetAsync(c->small.p, 0, need, c->stream));
    if (!c->copy_stream) {
        LDW_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            LDW_HIP(hipEventCreateWithFlags(&c->ev_up[k], hipEventDisableTiming));
            LDW_HIP(hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
        }
        LDW_HIP(hipEventCreateWithFlags(&c->ev_pick, hipEventDisableTiming));
        LDW_HIP(hipHostMalloc(&c->pin_pick, sizeof(ldw::PickOut) + 64, hipHostMallocDefault));
        LDW_HIP(hipStreamCreateWithFlags(&c->gemm_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) LDW_HIP(hipEventCreateWithFlags(&c->ev_gemm[k], hipEventDisableTiming));
    }
    // everything queued on the main stream so far (row map, weights) must be visible to the GEMM stream
    LDW_HIP(hipEventRecord(c->ev_up[0], c->stream));
    LDW_HIP(hipStreamWaitEvent(c->gemm_stream, c->ev_up[0], 0));
    while ((int64_t)c->ev_pool.size() < nblocks_capacity * EVB) {
        hipEvent_t e;
        LDW_HIP(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    c->done_recorded[0] = c->done_recorded[1] = false;
    c->n_sr = 0;
    c->n_lr = 0;
    c->stats.clear();
    c->blk_capacity = nblocks_capacity;
    c->blk_cursor = 0;
    for (int i = 0; i < 4; ++i) c->last_ms[i] = 0;
    return LDW_OK;
}

static int links_check(ldw_ctx *c, const ldw_mi_params *p) {
    LDW_REQUIRE(c->blk_capacity > 0 && c->blk_cursor < c->blk_capacity, LDW_ERR_STATE,
                "call ldw_links_begin with enough capacity first");
    LDW_REQUIRE(p->sr_only || p->lr_links_approx > 0, LDW_ERR_ARG, "lr_links_approx must be positive");
    LDW_REQUIRE(p->quirk_mode == LDW_QUIRK_REFERENCE || p->quirk_mode == LDW_QUIRK_INTENDED, LDW_ERR_ARG, "bad quirk mode");
    LDW_REQUIRE(c->have_meta, LDW_ERR_STATE, "SNP meta data (POS, g) not set");
    return LDW_OK;
}

int ldw_mi_block_links(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt,
                       const ldw_mi_params *p) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(from_idx && to_idx && p, LDW_ERR_ARG, "ldw_mi_block_links: null argument");
    if (int rc = links_check(c, p)) return rc;
    SmallLayout sl;
    links_layout(c, sl);
    HostBlock hb;
    if (int rc = prep_block(c, from_idx, nf, to_idx, nt, p, (int)(c->blk_cursor & 1), c->blk_cursor, hb)) return rc;
    if (int rc = submit_gemm(c, hb, p)) return rc;
    if (int rc = submit_block(c, hb, p, sl)) return rc;
    if (int rc = finish_block(c, hb, p, sl)) return rc;
    ++c->blk_cursor;
    return LDW_OK;
}

int ldw_links_end(ldw_ctx *c) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(c->blk_capacity > 0, LDW_ERR_STATE, "ldw_links_end without ldw_links_begin");
    SmallLayout sl;
    links_layout(c, sl);
    const int64_t nb = c->blk_cursor;
    int64_t h_lr = 0;
    std::vector<int64_t> si((size_t)nb * 3 + 1);
    std::vector<double> sd((size_t)nb + 1);
    LDW_HIP(hipMemcpyAsync(&h_lr, sl.lr_count, 8, hipMemcpyDeviceToHost, c->stream));
    if (nb > 0) {
        LDW_HIP(hipMemcpyAsync(si.data(), sl.stats_i, (size_t)nb * 24, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipMemcpyAsync(sd.data(), sl.stats_d, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
    }
    LDW_HIP(hipStreamSynchronize(c->stream));
    c->n_lr = h_lr;
    c->stats.resize((size_t)nb);
    for (int64_t b = 0; b < nb; ++b) {
        c->stats[b].n_lr_total = si[b * 3 + 0];
        c->stats[b].n_lr_kept = si[b * 3 + 1];
        c->stats[b].n_sr = si[b * 3 + 2];
        c->stats[b].disc_thresh = sd[b];
        float t01 = 0, t12 = 0, t23 = 0;
        hipEvent_t *ev = &c->ev_pool[(size_t)b * EVB];
        LDW_HIP(hipEventElapsedTime(&t01, ev[0], ev[1]));
        LDW_HIP(hipEventElapsedTime(&t12, ev[c->engine == LDW_ENGINE_MFMA ? 4 : 1], ev[2]));
        LDW_HIP(hipEventElapsedTime(&t23, ev[2], ev[3]));
        c->last_ms[0] += t01;
        c->last_ms[1] += t12;
        c->last_ms[2] += t23;
        c->last_ms[3] += t01 + t12 + t23;
    }
    c->blk_capacity = 0;
    return LDW_OK;
}

int ldw_mi_all_pairs(ldw_ctx *c, const int32_t *blocks, int64_t nblocks, const ldw_mi_params *p, int reset) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(blocks && p && nblocks > 0, LDW_ERR_ARG, "ldw_mi_all_pairs: bad argument");
    LDW_REQUIRE(reset, LDW_ERR_ARG, "ldw_mi_all_pairs: appending to earlier calls is not supported (reset must be 1)");
    if (int rc = ldw_links_begin(c, nblocks)) return rc;
    if (int rc = links_check(c, p)) return rc;
    SmallLayout sl;
    links_layout(c, sl);
    std::vector<int32_t> fi, ti;
    auto fill = [&](int64_t b) -> int {
        const int32_t fs = blocks[b * 4 + 0], fe = blocks[b * 4 + 1], ts = blocks[b * 4 + 2], te = blocks[b * 4 + 3];
        LDW_REQUIRE(fs >= 1 && fe >= fs && fe <= c->L && ts >= 1 && te >= ts && te <= c->L, LDW_ERR_ARG,
                    "block %lld = (%d,%d,%d,%d) outside 1..%lld", (long long)b, fs, fe, ts, te, (long long)c->L);
        fi.resize((size_t)(fe - fs + 1));
        ti.resize((size_t)(te - ts + 1));
        for (int32_t k = fs; k <= fe; ++k) fi[k - fs] = k - 1;
        for (int32_t k = ts; k <= te; ++k) ti[k - ts] = k - 1;
        return LDW_OK;
    };
    if (p->keep_sr && nblocks > 200) {  // big runs: size the short-range table once (its row count follows from POS alone)
                                        // instead of growing it geometrically, which would double-buffer tens of GB
        int64_t total_sr = 0;
        std::vector<ColInfo> cols;
        for (int64_t b = 0; b < nblocks; ++b) {
            if (int rc = fill(b)) return rc;
            int64_t n_sr_blk = 0;
            const bool diag = same_list(fi.data(), (int64_t)fi.size(), ti.data(), (int64_t)ti.size());
            if (int rc = build_cols(c, fi.data(), (int64_t)fi.size(), ti.data(), (int64_t)ti.size(), diag, p->sr_dist, cols, n_sr_blk))
                return rc;
            total_sr += n_sr_blk;
        }
        if (int rc = ensure_links_capacity(c, total_sr, 0)) return rc;
    }
    // software pipeline: while the GPU works on block b, the host prepares block b+1
    HostBlock hb[2];
    if (int rc = fill(0)) return rc;
    if (int rc = prep_block(c, fi.data(), (int64_t)fi.size(), ti.data(), (int64_t)ti.size(), p, 0, 0, hb[0])) return rc;
    if (int rc = submit_gemm(c, hb[0], p)) return rc;
    for (int64_t b = 0; b < nblocks; ++b) {
        const int s = (int)(b & 1);
        if (int rc = submit_block(c, hb[s], p, sl)) return rc;          // epilogue + pick of block b (main stream)
        if (b + 1 < nblocks) {
            if (int rc = fill(b + 1)) return rc;
            if (int rc = prep_block(c, fi.data(), (int64_t)fi.size(), ti.data(), (int64_t)ti.size(), p, s ^ 1, b + 1, hb[s ^ 1]))
                return rc;
            if (c->overlap)
                if (int rc = submit_gemm(c, hb[s ^ 1], p)) return rc;   // GEMM of block b+1 (GEMM stream) runs beside them
        }
        if (int rc = finish_block(c, hb[s], p, sl)) return rc;          // round trip + selection of block b
        if (!c->overlap && b + 1 < nblocks)
            if (int rc = submit_gemm(c, hb[s ^ 1], p)) return rc;       // overlap off: strictly one block after the other
        ++c->blk_cursor;
    }
    return ldw_links_end(c);
}

int ldw_set_overlap(ldw_ctx *c, int on) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    c->overlap = on != 0;
    return LDW_OK;
}

int ldw_links_count(ldw_ctx *c, int which, int64_t *n_out) {
    LDW_REQUIRE(c && n_out && (which == 0 || which == 1), LDW_ERR_ARG, "ldw_links_count: bad argument");
    *n_out = which == 0 ? c->n_sr : c->n_lr;
    return LDW_OK;
}

int ldw_links_fetch(ldw_ctx *c, int which, int32_t *a_out, int32_t *b_out, double *MI_out, int64_t capacity,
                    int on_device) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(which == 0 || which == 1, LDW_ERR_ARG, "ldw_links_fetch: which must be 0 (sr) or 1 (lr)");
    const int64_t n = which == 0 ? c->n_sr : c->n_lr;
    LDW_REQUIRE(capacity >= n, LDW_ERR_SIZE, "ldw_links_fetch: capacity %lld < %lld rows", (long long)capacity, (long long)n);
    if (n == 0) return LDW_OK;
    LDW_REQUIRE(a_out && b_out && MI_out, LDW_ERR_ARG, "ldw_links_fetch: null output");
    const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    ldw::DevBuf &A = which == 0 ? c->sr_a : c->lr_a, &B = which == 0 ? c->sr_b : c->lr_b, &M = which == 0 ? c->sr_mi : c->lr_mi;
    LDW_HIP(hipMemcpyAsync(a_out, A.p, (size_t)n * 4, k, c->stream));
    LDW_HIP(hipMemcpyAsync(b_out, B.p, (size_t)n * 4, k, c->stream));
    LDW_HIP(hipMemcpyAsync(MI_out, M.p, (size_t)n * 8, k, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_block_stats(ldw_ctx *c, int64_t nblocks, int64_t *n_lr_total, int64_t *n_lr_kept, int64_t *n_sr,
                    double *disc_thresh) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    LDW_REQUIRE((int64_t)c->stats.size() == nblocks, LDW_ERR_ARG, "ldw_block_stats: last call processed %lld blocks, not %lld",
                (long long)c->stats.size(), (long long)nblocks);
    for (int64_t b = 0; b < nblocks; ++b) {
        if (n_lr_total) n_lr_total[b] = c->stats[b].n_lr_total;
        if (n_lr_kept) n_lr_kept[b] = c->stats[b].n_lr_kept;
        if (n_sr) n_sr[b] = c->stats[b].n_sr;
        if (disc_thresh) disc_thresh[b] = c->stats[b].disc_thresh;
    }
    return LDW_OK;
}

}  // extern "C"
