// Short-range p-value model and ARACNE on the device-resident short-range link table: the O(#links) parts of
// mergeNsort_sr_links (R/computePairwiseMI.R:400-495) and runARACNE (R/io_functions.R:101-164).
//
// The table stays in HBM (C5: 2.25e9 rows, 36 GB); what crosses to the host are the per-(cluster, len) order
// statistics (<= nclust x sr_dist doubles), five sums per cluster, and the reduced link set.  The two tiny
// numerical steps in between — the log-log least-squares fit (:428) and the two-parameter beta MLE (:452) — stay
// with the caller (they are O(sr_dist) and O(1)); the library offers the data reductions either side of them:
//
//   ldw_sr_len_quantiles : rows -> (len, cluster, cluster) tags; two stable radix sorts (MI, then len); one
//                          workgroup per len counts the members of every cluster and picks the two order
//                          statistics quantile type 7 interpolates between           (:417-424)
//   ldw_sr_excess_stats  : diff = MI - mean_dist[len] (positional index, quirk Q5); n, sum x, sum x^2,
//                          sum log x, sum log(1-x) of the positive excesses per cluster — the sufficient
//                          statistics of the beta likelihood, reduced in a fixed order      (:444-452)
//   ldw_sr_pvalues       : srp = -log P_beta(X > diff) per row and cluster (continued fraction, log-space tail),
//                          max over the clusters a link belongs to (:475-486), cut at srp_cutoff, ARACNE pool
//                          MI >= min(MI kept) (:489-490)
//   ldw_aracne_device    : CSR adjacency of the pool by one radix sort; one wave per link to check intersects the
//                          two sorted neighbour lists
//
// HBM-bound integer/byte work apart from the log/continued-fraction arithmetic of the p-values.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include "ldw_prim.h"
#include <thread>
#include <vector>

#include "ldw_internal.h"
#include "ldw_dev.h"

using namespace ldw;

namespace ldw {

constexpr int SRM_MAXCL = 255;   // cluster ids 1..255 fit the 8-bit tags
constexpr int SRM_GRID = 2048;   // fixed grid of the row-streaming kernels (deterministic reduction order)

struct RowTag {
    int len;      // integer len, 0 = not in (0, sr_dist)
    int c1, c2;   // clust1 (to side, pos1), clust2 (from side, pos2)
};

__device__ __forceinline__ RowTag row_tag(int32_t a, int32_t b, const int32_t *__restrict__ POS,
                                          const int32_t *__restrict__ paint, double g, double sr_dist) {
    RowTag t;
    const int64_t gi = (int64_t)g;
    if ((double)gi == g && gi > 0) {
        // integral genome length (the rule: ldw_sr_len_quantiles requires it): POS are integers, so circ_len's 0.5 g - |d - 0.5 g| is min(d, g - d)
        // exactly — no fp64 division per row (r04: k_sr_stats / k_sr_pval stream 2.25e9 rows at C5)
        int64_t d = ((int64_t)POS[b] - (int64_t)POS[a]) % gi;
        if (d < 0) d += gi;
        const int64_t len = d < gi - d ? d : gi - d;
        t.len = (len > 0 && (double)len < sr_dist) ? (int)len : 0;
    } else {
        const double len = circ_len((double)POS[b], (double)POS[a], g);
        t.len = (len > 0.0 && len < sr_dist) ? (int)len : 0;
    }
    t.c1 = paint[b];
    t.c2 = paint[a];
    return t;
}

// ------------------------------------------------------------------------------------------------
// quantiles per (cluster, len)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sr_tag(const int32_t *__restrict__ sa, const int32_t *__restrict__ sb,
                                                const double *__restrict__ smi, int64_t n, const int32_t *__restrict__ POS,
                                                const int32_t *__restrict__ paint, double g, double sr_dist,
                                                uint32_t *__restrict__ pack, uint64_t *__restrict__ key) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const RowTag t = row_tag(sa[i], sb[i], POS, paint, g, sr_dist);
        pack[i] = ((uint32_t)t.len << 16) | ((uint32_t)t.c1 << 8) | (uint32_t)t.c2;
        key[i] = f64_key(smi[i]);
    }
}

// payload of the second (by len) sort: the MI key and the tag, 12 bytes
struct SrPay {
    uint32_t klo, khi, tag;
};

__global__ __launch_bounds__(256) void k_sr_split(const uint32_t *__restrict__ pack, const uint64_t *__restrict__ key, int64_t n,
                                                  uint16_t *__restrict__ len16, SrPay *__restrict__ pay) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t p = pack[i];
        const uint64_t k = key[i];
        len16[i] = (uint16_t)(p >> 16);
        pay[i] = SrPay{(uint32_t)k, (uint32_t)(k >> 32), p};
    }
}

// off[l] = first sorted row with len >= l, l = 0..S+1
__global__ void k_sr_seg_offsets(const uint16_t *__restrict__ pack, int64_t n, int S, int64_t *__restrict__ off) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > S + 1) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int)pack[mid] < l) lo = mid + 1;
        else hi = mid;
    }
    off[l] = lo;
}

// One workgroup per len.  q[(c*S + l-1)*2 + {0,1}] = the order statistics of rank floor(h), ceil(h), h = (n-1)*prob,
// among the MI values of the links of that len that touch cluster c+1; cnt[c*S + l-1] = n.
// r05: `gtot` (may be null) — the rows in `pay` are only the TOP of every group (the candidates rank 0 received from all ranks: ldw_sr_quantiles_merge);
// gtot[c*S + l-1] = the group's size over ALL ranks, the rows that are not here all lie below every row that is, so a wanted rank moves down by
// (gtot - members here); a rank that falls outside what is here (the senders' bound was wrong) is counted in *viol and leaves NaN.
__global__ __launch_bounds__(256) void k_sr_quant(const SrPay *__restrict__ pay, const int64_t *__restrict__ off, int S, int nclust, double prob,
                                                  double *__restrict__ q, int64_t *__restrict__ cnt, const int64_t *__restrict__ gtot,
                                                  unsigned int *__restrict__ viol) {
    __shared__ unsigned long long tot[SRM_MAXCL + 1], run[SRM_MAXCL + 1], tlo[SRM_MAXCL + 1], thi[SRM_MAXCL + 1];
    __shared__ unsigned int ccnt[SRM_MAXCL + 1];
    __shared__ unsigned int wcnt[4];
    const int l = blockIdx.x + 1;
    const int64_t beg = off[l], end = off[l + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int c = tid; c <= SRM_MAXCL; c += 256) {
        tot[c] = 0;
        run[c] = 0;
        ccnt[c] = 0;
    }
    __syncthreads();
    for (int64_t i = beg + tid; i < end; i += 256) {
        const uint32_t p = pay[i].tag;
        const int c1 = (p >> 8) & 0xFF, c2 = p & 0xFF;
        atomicAdd(&tot[c1], 1ull);
        if (c2 != c1) atomicAdd(&tot[c2], 1ull);
    }
    __syncthreads();
    for (int c = tid + 1; c <= nclust; c += 256) {
        const unsigned long long here = tot[c];
        const unsigned long long n = gtot ? (unsigned long long)gtot[(int64_t)(c - 1) * S + (l - 1)] : here;
        cnt[(int64_t)(c - 1) * S + (l - 1)] = (int64_t)n;
        if (n) {
            const double index = q7_index((double)(n - 1), prob);   // R's 1-based index, rounded as R rounds it (no fma: ldw_dev.h)
            const unsigned long long below = n - (here < n ? here : n), lo = (unsigned long long)floor(index) - 1ull, hi = (unsigned long long)ceil(index) - 1ull;
            if (here > n || lo < below) {   // (only with gtot: more members here than the group has, or the rank lies among the rows that stayed at home)
                tlo[c] = thi[c] = ~0ull;
                if (viol) atomicAdd(viol, 1u);
            } else {
                tlo[c] = lo - below;
                thi[c] = hi - below;
            }
        } else {
            tlo[c] = thi[c] = ~0ull;
            if (here && viol) atomicAdd(viol, 1u);
        }
    }
    __syncthreads();
    for (int64_t base = beg; base < end; base += 256) {
        const int64_t i = base + tid;
        int c1 = 0, c2 = 0;
        uint64_t k = 0;
        if (i < end) {
            const SrPay py = pay[i];
            c1 = (py.tag >> 8) & 0xFF;
            c2 = py.tag & 0xFF;
            k = ((uint64_t)py.khi << 32) | py.klo;
            atomicAdd(&ccnt[c1], 1u);
            if (c2 != c1) atomicAdd(&ccnt[c2], 1u);
        }
        __syncthreads();
        for (int c = 1; c <= nclust; ++c) {   // uniform: everything tested here lives in LDS
            const unsigned long long r0 = run[c], cc = ccnt[c];
            const bool hit_lo = tlo[c] >= r0 && tlo[c] < r0 + cc, hit_hi = thi[c] >= r0 && thi[c] < r0 + cc;
            if (!(hit_lo || hit_hi)) continue;
            const bool m = i < end && (c1 == c || c2 == c);
            const unsigned long long bal = __ballot(m);
            if (lane == 0) wcnt[wv] = (unsigned int)__popcll(bal);
            __syncthreads();
            unsigned long long rank = r0 + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv; ++w) rank += wcnt[w];
            if (m) {
                const int64_t o = ((int64_t)(c - 1) * S + (l - 1)) * 2;
                if (rank == tlo[c]) q[o] = key_f64(k);
                if (rank == thi[c]) q[o + 1] = key_f64(k);
            }
            __syncthreads();
        }
        for (int c = tid + 1; c <= nclust; c += 256) {
            run[c] += ccnt[c];
            ccnt[c] = 0;
        }
        if (tid == 0) ccnt[0] = 0;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// r04: the same order statistics WITHOUT sorting the table by MI.  The quantiles are 2 order statistics per (cluster, len) — 3 x 19 999
// pairs of numbers out of up to 2.25e9 rows — and the 64-bit radix sort by MI alone moved 192 B per row (8 passes x 24 B: 140 of the 227 ms
// of a C5 job's quantile step).  What is left: ONE sort, by len (16-bit keys, 2 radix passes), reading its keys and its 12-byte payload
// {MI key, cluster tags} straight from the table through transform iterators (no tagging pass), and a radix SELECT per len:
//   sweep A   members and the AND / OR of the members' keys per cluster (wave-aggregated: no atomics in the loop) -> ranks wanted, common prefix
//   sweep B.. histogram of the next 11 key bits below the common prefix, per cluster, in LDS -> the bucket holding the rank; repeated on
//             the bucket until it holds at most SEL_CAP keys (typically once: the prefix skips the bits all MI values share)
//   sweep C   the bucket's keys into LDS + the smallest key ABOVE the bucket (the second order statistic when the first is the bucket's last)
//   then the rank within the bucket by counting (O(m^2 / 64) LDS reads per cluster, m <= 512).
// Values are == the sorted path's (they are order statistics; equal keys are equal values).  Used for nclust <= SEL_MAXCL; more clusters
// (or >= 2^32 rows) take the two-sort path above.
// ------------------------------------------------------------------------------------------------
constexpr int SEL_MAXCL = 4, SEL_BITS = 11, SEL_BINS = 1 << SEL_BITS, SEL_CAP = 512;

struct SrRowsView {
    const int32_t *sa, *sb;
    const double *smi;
    const int32_t *POS, *paint;
    double g, sr_dist;
};
// (the sort evaluates both functors once per row and pass: the key needs the two positions only — in integers: POS and g are integers, so
// circ_len's 0.5 g - |d - 0.5 g| is min(d, g - d) exactly — the payload the two cluster ids only)
struct SrLenOf {
    SrRowsView v;
    __host__ __device__ uint16_t operator()(int64_t i) const {
        const int64_t gi = (int64_t)v.g;
        int64_t d = ((int64_t)v.POS[v.sb[i]] - (int64_t)v.POS[v.sa[i]]) % gi;
        if (d < 0) d += gi;
        const int64_t len = d < gi - d ? d : gi - d;
        return (len > 0 && (double)len < v.sr_dist) ? (uint16_t)len : (uint16_t)0;
    }
};
struct SrPayOf {
    SrRowsView v;
    __host__ __device__ SrPay operator()(int64_t i) const {
        const uint64_t k = f64_key(v.smi[i]);
        return SrPay{(uint32_t)k, (uint32_t)(k >> 32), ((uint32_t)v.paint[v.sb[i]] << 8) | (uint32_t)v.paint[v.sa[i]]};
    }
};

constexpr int SEL_NT = 512;   // threads per workgroup: 3 workgroups per CU (LDS) = 6 waves per SIMD streaming the segment
__global__ __launch_bounds__(SEL_NT) void k_sr_select(const SrPay *__restrict__ pay, const int64_t *__restrict__ off, int S, int nclust, double prob,
                                                   double *__restrict__ q, int64_t *__restrict__ cnt) {
    __shared__ unsigned int hist[SEL_MAXCL][SEL_BINS];
    __shared__ unsigned long long cand[SEL_MAXCL][SEL_CAP];
    __shared__ unsigned long long tot[SEL_MAXCL], kand[SEL_MAXCL], kor[SEL_MAXCL], above[SEL_MAXCL], pfx[SEL_MAXCL], msk[SEL_MAXCL], rnk[SEL_MAXCL];
    __shared__ unsigned int ncand[SEL_MAXCL], cbk[SEL_MAXCL];
    __shared__ int shf[SEL_MAXCL], state[SEL_MAXCL], two[SEL_MAXCL];   // state 0: refining, 1: bucket fits (gather), 2: bucket of equal keys, 3: done / empty
    __shared__ int n_refining;
    const int l = blockIdx.x + 1;
    const int64_t beg = off[l], end = off[l + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < SEL_MAXCL) {
        tot[tid] = 0;
        kand[tid] = ~0ull;
        kor[tid] = 0;
        above[tid] = ~0ull;
        ncand[tid] = 0;
    }
    __syncthreads();
    // ---- sweep A
    {
        unsigned long long n_c[SEL_MAXCL], a_c[SEL_MAXCL], o_c[SEL_MAXCL];
#pragma unroll
        for (int c = 0; c < SEL_MAXCL; ++c) {
            n_c[c] = 0;
            a_c[c] = ~0ull;
            o_c[c] = 0;
        }
        for (int64_t base = beg; base < end; base += 2 * SEL_NT) {   // two rows per thread in flight
            const int64_t i0 = base + tid, i1 = i0 + SEL_NT;
            const SrPay pa = i0 < end ? pay[i0] : SrPay{0u, 0u, 0u}, pb = i1 < end ? pay[i1] : SrPay{0u, 0u, 0u};   // (tag 0: no cluster)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const SrPay py = h ? pb : pa;
                const int c1 = (py.tag >> 8) & 0xFF, c2 = py.tag & 0xFF;
                const unsigned long long k = ((unsigned long long)py.khi << 32) | py.klo;
#pragma unroll
                for (int c = 0; c < SEL_MAXCL; ++c) {
                    const bool m = c1 == c + 1 || c2 == c + 1;
                    n_c[c] += (unsigned long long)__popcll(__ballot(m));   // (wave-uniform count)
                    if (m) {
                        a_c[c] &= k;
                        o_c[c] |= k;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < SEL_MAXCL; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a_c[c] &= __shfl_xor(a_c[c], o);
                o_c[c] |= __shfl_xor(o_c[c], o);
            }
            if (lane == 0 && c < nclust) {
                atomicAdd(&tot[c], n_c[c]);
                atomicAnd(&kand[c], a_c[c]);
                atomicOr(&kor[c], o_c[c]);
            }
        }
    }
    __syncthreads();
    if (tid < nclust) {
        const int c = tid;
        const unsigned long long n = tot[c];
        cnt[(int64_t)c * S + (l - 1)] = (int64_t)n;
        if (n == 0) {
            state[c] = 3;   // (the host pre-filled q with NaN)
        } else {
            const double index = q7_index((double)(n - 1), prob);   // R's 1-based index, rounded as R rounds it (no fma: ldw_dev.h)
            const unsigned long long tlo = (unsigned long long)floor(index) - 1ull, thi = (unsigned long long)ceil(index) - 1ull;
            rnk[c] = tlo;
            two[c] = thi != tlo;
            const unsigned long long diff = kand[c] ^ kor[c];
            if (diff == 0) {   // every member has the same MI
                const int64_t o = ((int64_t)c * S + (l - 1)) * 2;
                q[o] = q[o + 1] = key_f64(kand[c]);
                state[c] = 3;
            } else {
                const int top = 63 - __clzll((long long)diff);          // highest bit in which two members differ
                const int sh = top + 1 > SEL_BITS ? top + 1 - SEL_BITS : 0;
                shf[c] = sh;
                msk[c] = sh + SEL_BITS >= 64 ? 0ull : ~((1ull << (sh + SEL_BITS)) - 1ull);
                pfx[c] = kand[c] & msk[c];
                state[c] = 0;
            }
        }
    } else if (tid < SEL_MAXCL) {
        state[tid] = 3;
    }
    __syncthreads();
    // ---- histogram sweeps until every cluster's bucket fits
    for (;;) {
        if (tid == 0) {
            int r = 0;
            for (int c = 0; c < SEL_MAXCL; ++c) r += state[c] == 0;
            n_refining = r;
        }
        for (int b = tid; b < SEL_MAXCL * SEL_BINS; b += SEL_NT) (&hist[0][0])[b] = 0;
        __syncthreads();
        if (n_refining == 0) break;
        for (int64_t i0 = beg + tid; i0 < end; i0 += 2 * SEL_NT) {   // two rows per thread in flight
            const int64_t i1 = i0 + SEL_NT;
            const SrPay pa = pay[i0], pb = i1 < end ? pay[i1] : SrPay{0u, 0u, 0u};   // (tag 0: no cluster)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const SrPay py = h ? pb : pa;
                const int c1 = (py.tag >> 8) & 0xFF, c2 = py.tag & 0xFF;
                const unsigned long long k = ((unsigned long long)py.khi << 32) | py.klo;
#pragma unroll
                for (int c = 0; c < SEL_MAXCL; ++c)
                    if ((c1 == c + 1 || c2 == c + 1) && state[c] == 0 && (k & msk[c]) == pfx[c]) atomicAdd(&hist[c][(k >> shf[c]) & (SEL_BINS - 1)], 1u);
            }
        }
        __syncthreads();
        // wave c: the bucket of cluster c that holds the rank (32 bins per lane, wave scan)
        if (wv < SEL_MAXCL && state[wv] == 0) {
            const int c = wv;
            unsigned int mine = 0;
            for (int b = 0; b < SEL_BINS / 64; ++b) mine += hist[c][lane * (SEL_BINS / 64) + b];
            unsigned int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const unsigned long long r = rnk[c];
            const bool here = (unsigned long long)(incl - mine) <= r && r < (unsigned long long)incl;
            if (here) {   // exactly one lane
                unsigned int before = incl - mine;
                int b = lane * (SEL_BINS / 64);
                while ((unsigned long long)before + hist[c][b] <= r) before += hist[c][b++];
                const unsigned int cb = hist[c][b];
                const int sh = shf[c];
                rnk[c] = r - before;
                pfx[c] |= (unsigned long long)b << sh;
                msk[c] |= (unsigned long long)(SEL_BINS - 1) << sh;
                cbk[c] = cb;
                if (cb <= (unsigned int)SEL_CAP) state[c] = 1;
                else if (sh == 0) state[c] = 2;   // all 64 bits fixed: the bucket's keys are equal
                else {
                    const int ns = sh > SEL_BITS ? sh - SEL_BITS : 0;   // (the next digit may overlap bits already fixed: harmless, they match)
                    shf[c] = ns;
                }
            }
        }
        __syncthreads();
    }
    // ---- sweep C: the buckets' keys and the smallest key above each bucket
    {
        bool any = false;
        for (int c = 0; c < SEL_MAXCL; ++c) any = any || state[c] == 1 || state[c] == 2;
        if (any) {
            for (int64_t i0 = beg + tid; i0 < end; i0 += 2 * SEL_NT) {
                const int64_t i1 = i0 + SEL_NT;
                const SrPay pa = pay[i0], pb = i1 < end ? pay[i1] : SrPay{0u, 0u, 0u};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const SrPay py = h ? pb : pa;
                    const int c1 = (py.tag >> 8) & 0xFF, c2 = py.tag & 0xFF;
                    const unsigned long long k = ((unsigned long long)py.khi << 32) | py.klo;
#pragma unroll
                    for (int c = 0; c < SEL_MAXCL; ++c) {
                        if (!(c1 == c + 1 || c2 == c + 1) || state[c] == 3) continue;
                        if ((k & msk[c]) == pfx[c]) {
                            if (state[c] == 1) cand[c][atomicAdd(&ncand[c], 1u)] = k;
                        } else if (k > pfx[c] && k < above[c]) {   // (k > pfx and outside the bucket: above it; the read of `above` is only a filter)
                            atomicMin(&above[c], k);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- the rank within the bucket
    if (wv < SEL_MAXCL && (state[wv] == 1 || state[wv] == 2)) {
        const int c = wv;
        const int64_t o = ((int64_t)c * S + (l - 1)) * 2;
        const unsigned long long r = rnk[c];
        if (state[c] == 2) {
            if (lane == 0) {
                q[o] = key_f64(pfx[c]);
                q[o + 1] = key_f64((!two[c] || r + 1 < (unsigned long long)cbk[c]) ? pfx[c] : above[c]);
            }
        } else {
            const unsigned int m = ncand[c];
            if (lane == 0 && two[c] && r + 1 >= (unsigned long long)m) q[o + 1] = key_f64(above[c]);
            for (unsigned int i = lane; i < m; i += 64) {
                const unsigned long long k = cand[c][i];
                unsigned int less = 0, eq = 0;
                for (unsigned int j = 0; j < m; ++j) {
                    const unsigned long long kj = cand[c][j];
                    less += kj < k;
                    eq += kj == k;
                }
                if ((unsigned long long)less <= r && r < (unsigned long long)less + eq) q[o] = key_f64(k);
                if (!two[c]) {
                    if ((unsigned long long)less <= r && r < (unsigned long long)less + eq) q[o + 1] = key_f64(k);
                } else if ((unsigned long long)less <= r + 1 && r + 1 < (unsigned long long)less + eq) {
                    q[o + 1] = key_f64(k);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// positive excesses over the fitted decay: sufficient statistics of the beta likelihood
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// excess of a link over cluster c's fitted decay; NaN when the link has no valid len or the table has no entry
__device__ __forceinline__ double excess(double mi, int len, int c, const double *__restrict__ md, int S) {
    if (len <= 0 || c < 1) return __builtin_nan("");
    return mi - md[(int64_t)(c - 1) * S + (len - 1)];
}

// The rows of one workgroup.  seg == null: the table cut into gridDim.x contiguous strips.  r05, seg != null: segment blockIdx.x / strips of the table
// (seg[2 k], seg[2 k + 1]) = [first row, end) — the rows of ONE reference block — cut into `strips` strips: a segment's partial sums are then a function
// of that block's rows alone, whichever rank holds them and whatever else its table holds (ldw_sr_excess_stats_blocks).
__device__ __forceinline__ void strip_of(int64_t n, const int64_t *__restrict__ seg, int strips, int64_t &beg, int64_t &end) {
    if (seg) {
        const int k = blockIdx.x / strips, j = blockIdx.x % strips;
        const int64_t b0 = seg[2 * k], b1 = seg[2 * k + 1];
        const int64_t per = (b1 - b0 + strips - 1) / strips;
        beg = b0 + (int64_t)j * per < b1 ? b0 + (int64_t)j * per : b1;
        end = beg + per < b1 ? beg + per : b1;
    } else {
        const int64_t per = (n + gridDim.x - 1) / gridDim.x;   // contiguous strip per workgroup
        beg = (int64_t)blockIdx.x * per;
        end = beg + per < n ? beg + per : n;
    }
}

// r04: the same statistics for nclust <= 4 with the sums kept PER LANE over the workgroup's whole strip (5 running sums per cluster in
// registers, predicated adds) and reduced once at the end — lanes in index order, then the four waves, as before the workgroups on the host:
// still a fixed order, bit-identical from run to run.  k_sr_stats peels the clusters of every wave of 64 rows and reduces five doubles across
// the wave each time (and evaluates both logarithms for all 64 lanes): 33 ms for the 2.25e9 rows of C5.
template <int NC>
__global__ __launch_bounds__(256) void k_sr_stats_small(const int32_t *__restrict__ sa, const int32_t *__restrict__ sb,
                                                        const double *__restrict__ smi, int64_t n, const int32_t *__restrict__ POS,
                                                        const int32_t *__restrict__ paint, double g, double sr_dist,
                                                        const double *__restrict__ md, int S, int nclust, double *__restrict__ part,
                                                        const int64_t *__restrict__ seg, int strips) {
    __shared__ double red[4][NC][5];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int64_t beg, end;
    strip_of(n, seg, strips, beg, end);
    double a[NC][5];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) a[c][k] = 0.0;
    for (int64_t i = beg + tid; i < end; i += 256) {
        const RowTag t = row_tag(sa[i], sb[i], POS, paint, g, sr_dist);
        const double mi = smi[i];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = s == 0 ? t.c1 : (t.c2 != t.c1 ? t.c2 : 0);
            const double x = c ? excess(mi, t.len, c, md, S) : 0.0;
            if (c && x > 0) {   // (few rows lie above the fitted decay: the logarithms run for those lanes only)
                const double lx = log(x), l1x = log1p(-x);
#pragma unroll
                for (int cc = 0; cc < NC; ++cc)
                    if (c == cc + 1) {
                        a[cc][0] += 1.0;
                        a[cc][1] += x;
                        a[cc][2] += x * x;
                        a[cc][3] += lx;
                        a[cc][4] += l1x;
                    }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const double v = wave_sum(a[c][k]);   // (xor butterfly: the same tree on every run)
            if (lane == 0) red[wv][c][k] = v;
        }
    __syncthreads();
    for (int k = tid; k < nclust * 5; k += 256) {
        const int c = k / 5, j = k % 5;
        part[(int64_t)blockIdx.x * nclust * 5 + k] = ((red[0][c][j] + red[1][c][j]) + red[2][c][j]) + red[3][c][j];
    }
}

// part[(blockIdx.x * nclust + c) * 5 + k]
__global__ __launch_bounds__(256) void k_sr_stats(const int32_t *__restrict__ sa, const int32_t *__restrict__ sb,
                                                  const double *__restrict__ smi, int64_t n, const int32_t *__restrict__ POS,
                                                  const int32_t *__restrict__ paint, double g, double sr_dist,
                                                  const double *__restrict__ md, int S, int nclust,
                                                  double *__restrict__ part, const int64_t *__restrict__ seg, int strips) {
    extern __shared__ double acc[];   // [4 waves][nclust][5]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int k = tid; k < 4 * nclust * 5; k += 256) acc[k] = 0.0;
    __syncthreads();
    double *mine = acc + (int64_t)wv * nclust * 5;
    int64_t beg, end;
    strip_of(n, seg, strips, beg, end);
    for (int64_t base = beg; base < end; base += 256) {
        const int64_t i = base + tid;
        int cs[2] = {0, 0};
        double ds[2] = {0, 0};
        if (i < end) {
            const RowTag t = row_tag(sa[i], sb[i], POS, paint, g, sr_dist);
            const double mi = smi[i];
            const double d1 = excess(mi, t.len, t.c1, md, S);
            if (d1 > 0) {
                cs[0] = t.c1;
                ds[0] = d1;
            }
            if (t.c2 != t.c1) {
                const double d2 = excess(mi, t.len, t.c2, md, S);
                if (d2 > 0) {
                    cs[1] = t.c2;
                    ds[1] = d2;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int c = cs[s];
            const double x = ds[s];
            const double lx = c ? log(x) : 0.0, l1x = c ? log1p(-x) : 0.0;
            unsigned long long todo = __ballot(c != 0);
            while (todo) {   // peel the distinct clusters of this wave, lowest lane first: fixed order
                const int src = __ffsll((long long)todo) - 1;
                const int c0 = __shfl(c, src);
                const bool in = c == c0;
                const double s0 = wave_sum(in ? 1.0 : 0.0), s1 = wave_sum(in ? x : 0.0), s2 = wave_sum(in ? x * x : 0.0),
                             s3 = wave_sum(in ? lx : 0.0), s4 = wave_sum(in ? l1x : 0.0);
                if (lane == 0) {
                    double *a = mine + (int64_t)(c0 - 1) * 5;
                    a[0] += s0;
                    a[1] += s1;
                    a[2] += s2;
                    a[3] += s3;
                    a[4] += s4;
                }
                todo &= ~__ballot(in);
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < nclust * 5; k += 256)
        part[(int64_t)blockIdx.x * nclust * 5 + k] = ((acc[k] + acc[nclust * 5 + k]) + acc[2 * nclust * 5 + k]) + acc[3 * nclust * 5 + k];
}

// ------------------------------------------------------------------------------------------------
// -log P_beta(X > x) for shape (a, b): continued fraction of the regularised incomplete beta function
// (modified Lentz), evaluated on the side where it converges fast, tail kept in log space
// ------------------------------------------------------------------------------------------------
__device__ double beta_cf(double a, double b, double x) {
    const double tiny = 1e-300;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 1000; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 4e-16) break;
    }
    return h;
}

__device__ double neg_log_beta_sf(double x, double a, double b, double lbeta) {
    if (!(x > 0.0)) return 0.0;
    if (x >= 1.0) return __builtin_inf();
    const double lfront = a * log(x) + b * log1p(-x) - lbeta;
    if (x < (a + 1.0) / (a + b + 2.0)) {
        const double I = exp(lfront) * beta_cf(a, b, x) / a;   // lower tail, well below 1 here
        return -log1p(-I);
    }
    return -(lfront + log(beta_cf(b, a, 1.0 - x) / b));
}

struct SrCounters {
    unsigned long long n_red, n_pool, min_key;
};

// r04: per cluster the excess below which -log P(X > x) cannot exceed the cut-off.  The tail is increasing in x, so a bisection of the very
// function the rows are judged with finds the crossing; k_sr_pval evaluates the continued fraction only for rows at or above 0.999999 of it
// (the function is good to ~1e-10 relative: the margin is four orders wider) and judges THOSE exactly as before — 0.07 % of the rows of a C5
// table instead of every row with a positive excess.  dstar = 0: every positive excess is evaluated (cut-off below the tail's start).
__global__ void k_sr_dstar(const double *__restrict__ shape, int nclust, double cutoff, double *__restrict__ dstar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nclust) return;
    const double a = shape[c * 3], b = shape[c * 3 + 1], lb = shape[c * 3 + 2];
    double lo = 0.0, hi = 1.0;
    if (!(cutoff > 0.0) || !(neg_log_beta_sf(1e-300, a, b, lb) <= cutoff)) {
        dstar[c] = 0.0;
        return;
    }
    for (int it = 0; it < 200 && hi - lo > 1e-15 * hi; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (neg_log_beta_sf(mid, a, b, lb) > cutoff) hi = mid;
        else lo = mid;
    }
    dstar[c] = lo * 0.999999;   // (lo: the tail there is still <= the cut-off)
}

// MODE 0: count reduced rows and min MI key; MODE 1: also write them.  meta = clust_c | first_cluster << 8 | dup << 16
template <int MODE>
__global__ __launch_bounds__(256) void k_sr_pval(const int32_t *__restrict__ sa, const int32_t *__restrict__ sb,
                                                 const double *__restrict__ smi, int64_t n, const int32_t *__restrict__ POS,
                                                 const int32_t *__restrict__ paint, double g, double sr_dist,
                                                 const double *__restrict__ md, int S, const double *__restrict__ shape,
                                                 double cutoff, SrCounters *__restrict__ ctr, int64_t *__restrict__ red_row,
                                                 uint32_t *__restrict__ red_meta, double *__restrict__ red_srp, int64_t cap,
                                                 const double *__restrict__ dstar) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const RowTag t = row_tag(sa[i], sb[i], POS, paint, g, sr_dist);
        const double mi = smi[i];
        const double d1 = excess(mi, t.len, t.c1, md, S);
        const bool v1 = d1 > 0;
        if (dstar) {   // (a row is kept only if the tail of one of its clusters exceeds the cut-off: neither excess reaches its cluster's crossing -> dropped)
            bool may = v1 && d1 >= dstar[t.c1 - 1];
            if (!may && t.c2 != t.c1) {
                const double d2 = excess(mi, t.len, t.c2, md, S);
                may = d2 > 0 && d2 >= dstar[t.c2 - 1];
            }
            if (!may) continue;
        }
        double s = 0.0;
        int cc = 0, first = 0;
        if (v1) {
            const double *sh = shape + (int64_t)(t.c1 - 1) * 3;
            s = neg_log_beta_sf(d1, sh[0], sh[1], sh[2]);
            cc = first = t.c1;
        }
        const bool dup = t.c2 != t.c1;
        if (dup) {
            const double d2 = excess(mi, t.len, t.c2, md, S);
            if (d2 > 0) {
                const double *sh = shape + (int64_t)(t.c2 - 1) * 3;
                const double s2 = neg_log_beta_sf(d2, sh[0], sh[1], sh[2]);
                // which.max over the rows of the group in cluster order: ties go to the smaller cluster id
                if (!v1 || s2 > s || (s2 == s && t.c2 < cc)) {
                    s = s2;
                    cc = t.c2;
                }
                first = (!v1 || t.c2 < first) ? t.c2 : first;
            }
        }
        if (cc && s > cutoff) {
            const unsigned long long slot = atomicAdd(&ctr->n_red, 1ull);
            atomicMin(&ctr->min_key, (unsigned long long)f64_key(mi));
            if (MODE == 1 && (int64_t)slot < cap) {
                red_row[slot] = i;
                red_meta[slot] = (uint32_t)cc | ((uint32_t)first << 8) | ((uint32_t)dup << 16);
                red_srp[slot] = s;
            }
        }
    }
}

// ARACNE pool: rows of the merged table (positive excess in some cluster) with MI >= min(MI kept)
template <int MODE>
__global__ __launch_bounds__(256) void k_sr_pool(const int32_t *__restrict__ sa, const int32_t *__restrict__ sb,
                                                 const double *__restrict__ smi, int64_t n, const int32_t *__restrict__ POS,
                                                 const int32_t *__restrict__ paint, double g, double sr_dist,
                                                 const double *__restrict__ md, int S, SrCounters *__restrict__ ctr,
                                                 int32_t *__restrict__ pa, int32_t *__restrict__ pb, double *__restrict__ pmi,
                                                 int64_t cap) {
    const uint64_t kmin = ctr->min_key;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double mi = smi[i];
        if (f64_key(mi) < kmin) continue;
        const int32_t a = sa[i], b = sb[i];
        const RowTag t = row_tag(a, b, POS, paint, g, sr_dist);
        bool v = excess(mi, t.len, t.c1, md, S) > 0;
        if (!v && t.c2 != t.c1) v = excess(mi, t.len, t.c2, md, S) > 0;
        if (!v) continue;
        const unsigned long long slot = atomicAdd(&ctr->n_pool, 1ull);
        if (MODE == 1 && (int64_t)slot < cap) {
            pa[slot] = a;
            pb[slot] = b;
            pmi[slot] = mi;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ARACNE
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ar_edges(const int32_t *__restrict__ pa, const int32_t *__restrict__ pb,
                                                  const double *__restrict__ pmi, int64_t n, uint64_t *__restrict__ key,
                                                  double *__restrict__ val) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t a = (uint32_t)pa[i], b = (uint32_t)pb[i];
        const double m = pmi[i];
        key[2 * i] = (a << 32) | b;
        val[2 * i] = m;
        key[2 * i + 1] = (b << 32) | a;
        val[2 * i + 1] = m;
    }
}

// off[v] = first directed edge whose node is >= v, v = 0..L
__global__ void k_ar_offsets(const uint64_t *__restrict__ key, int64_t n2, int64_t L, int64_t *__restrict__ off) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > L) return;
    int64_t lo = 0, hi = n2;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(key[mid] >> 32) < v) lo = mid + 1;
        else hi = mid;
    }
    off[v] = lo;
}

// One wave per link (X,Z) = (to side, from side): lanes stride over the shorter neighbour list and binary-search the
// longer one.  flag = 0 iff some common neighbour Y has MI(X,Z) < MI(X,Y) and MI(X,Z) < MI(Z,Y)  (src/computeMI.cpp:63-77).
__global__ __launch_bounds__(256) void k_ar_check(const int64_t *__restrict__ red_row, int64_t n_red,
                                                  const int32_t *__restrict__ sa, const int32_t *__restrict__ sb,
                                                  const double *__restrict__ smi, const uint64_t *__restrict__ key,
                                                  const double *__restrict__ val, const int64_t *__restrict__ off,
                                                  uint8_t *__restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_red) return;
    const int64_t row = red_row[w];
    const int32_t X = sb[row], Z = sa[row];
    const double mi0 = smi[row];
    int64_t xs = off[X], xe = off[X + 1], zs = off[Z], ze = off[Z + 1];
    if (xe - xs > ze - zs) {   // iterate over the shorter list
        int64_t t = xs; xs = zs; zs = t;
        t = xe; xe = ze; ze = t;
    }
    bool indirect = false;
    for (int64_t i = xs + lane; i < xe && !indirect; i += 64) {
        const uint32_t y = (uint32_t)key[i];
        if (!(mi0 < val[i])) continue;
        int64_t lo = zs, hi = ze;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((uint32_t)key[mid] < y) lo = mid + 1;
            else hi = mid;
        }
        if (lo < ze && (uint32_t)key[lo] == y && mi0 < val[lo]) indirect = true;
    }
    const bool any = __ballot(indirect) != 0ull;
    if (lane == 0) flags[w] = any ? 0 : 1;
}

__global__ void k_red_gather(const int64_t *__restrict__ row, int64_t n, const int32_t *__restrict__ sa, const int32_t *__restrict__ sb,
                             const double *__restrict__ smi, int32_t *__restrict__ a, int32_t *__restrict__ b, double *__restrict__ mi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = row[i];
    a[i] = sa[r];
    b[i] = sb[r];
    mi[i] = smi[r];
}

// Buffers of the select path.  (r04, tried and dropped: reserving them on a side thread while the MI pass runs, with and without touching the
// new memory there.  The first quantile step of a context costs ~20 ms more than a later one (2.4 GB of fresh device memory at C4), but the
// wait only moved — from hipMalloc to the first sort — and the memset took 5 ms from the MI pass: tools/quant_cold_probe.py, tools/job_profile.py --cold.)
static size_t srm_sort_temp_bytes(int64_t n) {
    const SrRowsView V{nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, 1.0};
    auto kin = rocprim::make_transform_iterator(rocprim::counting_iterator<int64_t>(0), SrLenOf{V});
    auto vin = rocprim::make_transform_iterator(rocprim::counting_iterator<int64_t>(0), SrPayOf{V});
    size_t tb = 0;
    if (rocprim::radix_sort_pairs(nullptr, tb, kin, (uint16_t *)nullptr, vin, (SrPay *)nullptr, (size_t)n, 0u, 16u, (hipStream_t)0) != hipSuccess) return 0;
    return tb;
}
static int srm_reserve_select(ldw_ctx *c, int64_t n) {
    if (int rc = c->srm_pack2.reserve((size_t)n * 2 + 64)) return rc;
    if (int rc = c->srm_pay.reserve((size_t)n * sizeof(SrPay))) return rc;
    return c->srm_tmp.reserve(srm_sort_temp_bytes(n) + 256);
}
// ------------------------------------------------------------------------------------------------
// r05: the short-range model with the rows LEFT on the rank that computed them (docs/HISTORY.md 7b).  What travels instead of the table:
// per (cluster, len) group the rows at or above a bound that is known to lie below the group's order statistics (k_sr_tail: ~7 % of the MI
// column), five sums per reference block and cluster, the kept links and the ARACNE pool.
// ------------------------------------------------------------------------------------------------
// Rows at or above their group's bound, group id = (len - 1) * nclust + (cluster - 1) (len-major: what the merging side sorts by).  A row
// whose two SNPs lie in different clusters is a member of both groups (R/computePairwiseMI.R:411-414) and is listed in each.
// MODE 0: cur[g] += members passing; MODE 1: out[cur[g]++] = MI (cur preset to the groups' first output positions).  NaN bound: nothing passes.
template <int MODE>
__global__ __launch_bounds__(256) void k_sr_tail(const int32_t *__restrict__ sa, const int32_t *__restrict__ sb, const double *__restrict__ smi,
                                                 int64_t n, const int32_t *__restrict__ POS, const int32_t *__restrict__ paint, double g,
                                                 double sr_dist, const double *__restrict__ lower, int S, int nclust,
                                                 unsigned long long *__restrict__ cur, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const RowTag t = row_tag(sa[i], sb[i], POS, paint, g, sr_dist);
        if (t.len <= 0 || t.len > S) continue;
        const double mi = smi[i];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = s == 0 ? t.c1 : (t.c2 != t.c1 ? t.c2 : 0);
            if (c < 1 || c > nclust) continue;
            if (!(mi >= lower[(int64_t)(c - 1) * S + (t.len - 1)])) continue;
            const unsigned long long pos = atomicAdd(&cur[(int64_t)(t.len - 1) * nclust + (c - 1)], 1ull);
            if (MODE == 1) out[pos] = mi;
        }
    }
}

// One source's candidates (values grouped len-major, goff = exclusive scan of its group counts, G + 1 entries) as tagged rows of the
// two-sort path: pack = len << 16 | c << 8 | c (a member of ONE cluster: a row of two clusters was listed once per group), key = key of MI.
__global__ __launch_bounds__(256) void k_tail_unpack(const double *__restrict__ mi, int64_t m, const int64_t *__restrict__ goff, int G, int nclust,
                                                     int64_t base, uint32_t *__restrict__ pack, uint64_t *__restrict__ key) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = G;   // the group whose range [goff[g], goff[g + 1]) holds i: last g with goff[g] <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (goff[mid] <= i) lo = mid;
            else hi = mid;
        }
        const uint32_t l = (uint32_t)(lo / nclust) + 1u, c = (uint32_t)(lo % nclust) + 1u;
        pack[base + i] = (l << 16) | (c << 8) | c;
        key[base + i] = f64_key(mi[i]);
    }
}

__global__ void k_iota64(int64_t *__restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// The two-sort path from tagged rows on: c->srm_pack[i] = len << 16 | clust1 << 8 | clust2, c->srm_key[i] = key of MI, n rows (k_sr_tag, or the
// candidates of all ranks: k_tail_unpack).  Two stable LSD sorts, both over the FULL width of their key type: by MI (u64 keys, tags as values), then
// by len (u16 keys, {MI key, tag} as values) -> ordered by (len, MI).  rocPRIM 7.2's merge-sort path mis-sorts u32 keys on a partial bit range at
// mid sizes (reproduced standalone through the hipCUB interface), so no begin_bit/end_bit tricks here.  q: host, 2 * nclust * S (NaN-filled by
// the caller: uploaded first); gtot / viol_out: see k_sr_quant.
static int quant_reserve_sorts(ldw_ctx *c, int64_t n, int32_t S, size_t cells) {
    if (int rc = c->srm_pack.reserve((size_t)n * 4)) return rc;
    if (int rc = c->srm_pack2.reserve((size_t)n * 4)) return rc;
    if (int rc = c->srm_key.reserve((size_t)n * 8)) return rc;
    if (int rc = c->srm_key2.reserve((size_t)n * 8)) return rc;
    if (int rc = c->srm_off.reserve((size_t)(S + 2) * 8)) return rc;
    if (int rc = c->srm_q.reserve(cells * 16)) return rc;
    if (int rc = c->srm_n.reserve(cells * 8 + 64)) return rc;
    if (int rc = c->srm_pay.reserve((size_t)n * sizeof(SrPay))) return rc;
    return c->srm_pay2.reserve((size_t)n * sizeof(SrPay));
}
static int quant_two_sorts(ldw_ctx *c, int64_t n, int32_t S, int nclust, double prob, const int64_t *d_gtot, std::vector<double> &q, int64_t *n_out,
                           unsigned int *viol_out) {
    const size_t cells = (size_t)nclust * S;
    uint32_t *pack = c->srm_pack.as<uint32_t>(), *pack2 = c->srm_pack2.as<uint32_t>();
    uint64_t *key = c->srm_key.as<uint64_t>(), *key2 = c->srm_key2.as<uint64_t>();
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 65536);
    uint16_t *len16 = reinterpret_cast<uint16_t *>(pack), *len16b = len16 + n;   // pack is free after the first sort
    SrPay *pay = c->srm_pay.as<SrPay>(), *pay2 = c->srm_pay2.as<SrPay>();
    size_t t1 = 0, t2 = 0;
    LDW_HIP(prim_sort_pairs(nullptr, t1, key, key2, pack, pack2, n, 0, 64, c->stream));
    LDW_HIP(prim_sort_pairs(nullptr, t2, len16, len16b, pay, pay2, n, 0, 16, c->stream));
    if (int rc = c->scratch.reserve(std::max(t1, t2))) return rc;
    size_t tb = c->scratch.cap;
    LDW_HIP(prim_sort_pairs(c->scratch.p, tb, key, key2, pack, pack2, n, 0, 64, c->stream));
    hipLaunchKernelGGL(k_sr_split, dim3(grid), dim3(256), 0, c->stream, pack2, key2, n, len16, pay);
    tb = c->scratch.cap;
    LDW_HIP(prim_sort_pairs(c->scratch.p, tb, len16, len16b, pay, pay2, n, 0, 16, c->stream));
    hipLaunchKernelGGL(k_sr_seg_offsets, dim3((S + 2 + 255) / 256), dim3(256), 0, c->stream, len16b, n, S, c->srm_off.as<int64_t>());
    LDW_HIP(hipMemcpyAsync(c->srm_q.p, q.data(), cells * 16, hipMemcpyHostToDevice, c->stream));
    unsigned int *d_viol = reinterpret_cast<unsigned int *>(c->srm_n.as<char>() + cells * 8);   // (the 64 spare bytes behind the counts)
    if (viol_out) LDW_HIP(hipMemsetAsync(d_viol, 0, 4, c->stream));
    hipLaunchKernelGGL(k_sr_quant, dim3(S), dim3(256), 0, c->stream, pay2, c->srm_off.as<int64_t>(), S, nclust, prob,
                       c->srm_q.as<double>(), c->srm_n.as<int64_t>(), d_gtot, viol_out ? d_viol : (unsigned int *)nullptr);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(q.data(), c->srm_q.p, cells * 16, hipMemcpyDeviceToHost, c->stream));
    if (n_out) LDW_HIP(hipMemcpyAsync(n_out, c->srm_n.p, cells * 8, hipMemcpyDeviceToHost, c->stream));
    if (viol_out) LDW_HIP(hipMemcpyAsync(viol_out, d_viol, 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}
static int sr_ready(ldw_ctx *c, const char *who) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(c->have_meta, LDW_ERR_STATE, "%s: ldw_set_snp_meta has not been called", who);
    return LDW_OK;
}

}  // namespace ldw

extern "C" {

int ldw_sr_len_quantiles(ldw_ctx *c, int nclust, double sr_dist, double prob, int32_t S, double *q_lo_out, double *q_hi_out,
                         int64_t *n_out) {
    if (int rc = sr_ready(c, "ldw_sr_len_quantiles")) return rc;
    if (int rc = ldw::join_prepare(c)) return rc;   // (reads and may borrow Gapx[k], which ldw_ctx_reserve's side thread sizes)
    LDW_REQUIRE(nclust >= 1 && nclust <= SRM_MAXCL, LDW_ERR_ARG, "ldw_sr_len_quantiles: nclust must be in 1..%d", SRM_MAXCL);
    LDW_REQUIRE(sr_dist > 1 && sr_dist <= 65535, LDW_ERR_ARG, "ldw_sr_len_quantiles: sr_dist must be in (1, 65535]");
    LDW_REQUIRE(prob >= 0 && prob <= 1, LDW_ERR_ARG, "ldw_sr_len_quantiles: prob outside [0,1]");
    LDW_REQUIRE(S == (int32_t)std::ceil(sr_dist) - 1, LDW_ERR_ARG, "ldw_sr_len_quantiles: S must be ceil(sr_dist)-1 = %d",
                (int)std::ceil(sr_dist) - 1);
    LDW_REQUIRE(c->g == std::floor(c->g), LDW_ERR_ARG, "ldw_sr_len_quantiles: the genome length must be an integer");
    LDW_REQUIRE(q_lo_out && q_hi_out && n_out, LDW_ERR_ARG, "ldw_sr_len_quantiles: null output");
    LDW_REQUIRE(c->paint_min >= 1 && c->paint_max <= nclust, LDW_ERR_ARG,
                "ldw_sr_len_quantiles: cds_var$paint must lie in 1..nclust (found %d..%d, nclust %d)", (int)c->paint_min,
                (int)c->paint_max, nclust);
    // (POS may be in any order and may repeat: every row's len is computed from its two positions, rows with len outside
    // (0, sr_dist) are left out exactly as R/computePairwiseMI.R:414 does)
    const int64_t n = c->n_sr;
    c->srm_S = S;
    c->srm_nclust = nclust;
    c->srm_sr_dist = sr_dist;
    const size_t cells = (size_t)nclust * S;
    std::vector<double> q(cells * 2, std::nan(""));
    if (n == 0) {
        for (size_t k = 0; k < cells; ++k) {
            q_lo_out[k] = q_hi_out[k] = std::nan("");
            n_out[k] = 0;
        }
        return LDW_OK;
    }
    static const bool host_timing = getenv("LDW_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_0 = now();
    const bool force_sort = getenv("LDW_SR_QUANT_SORT") != nullptr;   // (A/B and tests: the two-sort path for any nclust; read per call)
    if (nclust <= SEL_MAXCL && n < (int64_t)0xFFFFFFFFll && !force_sort) {
        // r04: one sort (by len) fed from the table itself, then a radix select per len (k_sr_select)
        // Working memory: the three G' blocks of the MI pass's pipeline slots (4.3 GB each for 10k-SNP blocks) lie idle once the pass is over,
        // and the quantile step of a job follows that pass: it borrows them when its arrays fit (C4: 0.2 + 1.1 + 1.3 GB) instead of allocating
        // 2.4 GB of fresh device memory, whose first use cost the first job of a context ~20 ms (job leg: 24-28 ms against 4-5 ms for the
        // same call on memory that had been used before).  Stream order makes it safe: ldw_mi_all_pairs returns with its streams drained.
        const size_t need[3] = {(size_t)n * 2 + 64, (size_t)n * sizeof(SrPay), srm_sort_temp_bytes(n) + 256};
        void *mem[3];
        size_t tmp_cap;
        const bool borrow = LDW_NSLOT >= 3 && c->blk_capacity == 0 /* (no link pass open: its slots are really idle) */ && c->Gapx[0].cap >= need[0] && c->Gapx[1].cap >= need[1] && c->Gapx[2].cap >= need[2] && ldw::exp_env("LDW_SR_QUANT_OWN") == nullptr;
        const auto t_a = now();
        if (borrow) {
            if (c->gemm_stream) LDW_HIP(hipStreamSynchronize(c->gemm_stream));   // (idle already; the G' blocks belong to that stream's kernels)
            for (int k = 0; k < 3; ++k) mem[k] = c->Gapx[k].p;
            tmp_cap = c->Gapx[2].cap;
        } else {
            if (int rc = srm_reserve_select(c, n)) return rc;
            mem[0] = c->srm_pack2.p;
            mem[1] = c->srm_pay.p;
            mem[2] = c->srm_tmp.p;
            tmp_cap = c->srm_tmp.cap;
        }
        const auto t_b = now();
        if (int rc = c->srm_off.reserve((size_t)(S + 2) * 8)) return rc;
        if (int rc = c->srm_q.reserve(cells * 16)) return rc;
        if (int rc = c->srm_n.reserve(cells * 8)) return rc;
        if (host_timing) fprintf(stderr, "[ldw] sr quantiles: temp query %.2f ms, gemm stream sync / reserve %.2f, small reserves %.2f\n", ms(t_0, t_a), ms(t_a, t_b), ms(t_b, now()));
        const SrRowsView V{c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(), c->sr_mi.as<double>(), c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, sr_dist};
        auto kin = rocprim::make_transform_iterator(rocprim::counting_iterator<int64_t>(0), SrLenOf{V});
        auto vin = rocprim::make_transform_iterator(rocprim::counting_iterator<int64_t>(0), SrPayOf{V});
        uint16_t *len_sorted = static_cast<uint16_t *>(mem[0]);
        SrPay *pay_sorted = static_cast<SrPay *>(mem[1]);
        size_t tb = tmp_cap;
        const auto t_1 = now();
        if (host_timing) LDW_HIP(hipStreamSynchronize(c->stream));
        const auto t_1b = now();
        LDW_HIP(rocprim::radix_sort_pairs(mem[2], tb, kin, len_sorted, vin, pay_sorted, (size_t)n, 0u, 16u, c->stream));
        if (host_timing) {
            const auto t_2 = now();
            LDW_HIP(hipStreamSynchronize(c->stream));
            fprintf(stderr, "[ldw] sr quantiles%s: reserve %.2f ms (scratch %.1f MB), stream drain %.2f, sort enqueue %.2f + wait %.2f\n", borrow ? " (memory borrowed from the pass)" : "", ms(t_0, t_1), (double)tb / 1e6,
                    ms(t_1, t_1b), ms(t_1b, t_2), ms(t_2, now()));
        }
        hipLaunchKernelGGL(k_sr_seg_offsets, dim3((S + 2 + 255) / 256), dim3(256), 0, c->stream, len_sorted, n, S, c->srm_off.as<int64_t>());
        LDW_HIP(hipMemcpyAsync(c->srm_q.p, q.data(), cells * 16, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_sr_select, dim3(S), dim3(SEL_NT), 0, c->stream, pay_sorted, c->srm_off.as<int64_t>(), S, nclust, prob, c->srm_q.as<double>(),
                           c->srm_n.as<int64_t>());
        LDW_HIP(hipGetLastError());
        LDW_HIP(hipMemcpyAsync(q.data(), c->srm_q.p, cells * 16, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipMemcpyAsync(n_out, c->srm_n.p, cells * 8, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < cells; ++k) {
            q_lo_out[k] = q[2 * k];
            q_hi_out[k] = q[2 * k + 1];
        }
        return LDW_OK;
    }
    if (int rc = quant_reserve_sorts(c, n, S, cells)) return rc;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(k_sr_tag, dim3(grid), dim3(256), 0, c->stream, c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(),
                       c->sr_mi.as<double>(), n, c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, sr_dist, c->srm_pack.as<uint32_t>(), c->srm_key.as<uint64_t>());
    if (int rc = quant_two_sorts(c, n, S, nclust, prob, nullptr, q, n_out, nullptr)) return rc;
    for (size_t k = 0; k < cells; ++k) {
        q_lo_out[k] = q[2 * k];
        q_hi_out[k] = q[2 * k + 1];
    }
    return LDW_OK;
}

// ARACNE pool of the context's short-range table: rows with a positive excess in some cluster and MI key >= min_key (R/computePairwiseMI.R:489-490);
// needs the fitted decay on the device (upload_md) and the geometry of the last ldw_sr_len_quantiles call.  Sets c->n_pool.
static int build_pool(ldw_ctx *c, unsigned long long min_key) {
    const int64_t n = c->n_sr;
    c->n_pool = 0;
    if (n == 0) return LDW_OK;
    if (int rc = c->srm_cnt.reserve(sizeof(SrCounters))) return rc;
    SrCounters *d = c->srm_cnt.as<SrCounters>();
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 16384);
    {
        const size_t floor_rows = (size_t)std::min<int64_t>(n, (int64_t)1 << 22);
        if (int rc = c->pool_a.reserve(floor_rows * 4)) return rc;
        if (int rc = c->pool_b.reserve(floor_rows * 4)) return rc;
        if (int rc = c->pool_mi.reserve(floor_rows * 8)) return rc;
    }
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t cap = std::min<int64_t>((int64_t)(c->pool_a.cap / 4), std::min<int64_t>((int64_t)(c->pool_b.cap / 4), (int64_t)(c->pool_mi.cap / 8)));
        SrCounters h2 = {0, 0, min_key};
        LDW_HIP(hipMemcpyAsync(d, &h2, sizeof(h2), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_sr_pool<1>, dim3(grid), dim3(256), 0, c->stream, c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(), c->sr_mi.as<double>(), n,
                           c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, c->srm_sr_dist, c->srm_md.as<double>(), c->srm_S, d, c->pool_a.as<int32_t>(),
                           c->pool_b.as<int32_t>(), c->pool_mi.as<double>(), cap);
        LDW_HIP(hipGetLastError());
        SrCounters got;
        LDW_HIP(hipMemcpyAsync(&got, d, sizeof(got), hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        c->n_pool = (int64_t)got.n_pool;
        if (c->n_pool <= cap) break;
        LDW_REQUIRE(pass == 0, LDW_ERR_STATE, "short-range pool changed size between passes");
        if (int rc = c->pool_a.reserve((size_t)c->n_pool * 4)) return rc;
        if (int rc = c->pool_b.reserve((size_t)c->n_pool * 4)) return rc;
        if (int rc = c->pool_mi.reserve((size_t)c->n_pool * 8)) return rc;
    }
    return LDW_OK;
}

static int upload_md(ldw_ctx *c, int nclust, int32_t S, const double *mean_dist, const char *who) {
    LDW_REQUIRE(nclust == c->srm_nclust && S == c->srm_S, LDW_ERR_STATE, "%s: nclust/S differ from the last ldw_sr_len_quantiles call", who);
    LDW_REQUIRE(mean_dist, LDW_ERR_ARG, "%s: null mean_dist", who);
    if (int rc = c->srm_md.reserve((size_t)nclust * S * 8)) return rc;
    LDW_HIP(hipMemcpyAsync(c->srm_md.p, mean_dist, (size_t)nclust * S * 8, hipMemcpyHostToDevice, c->stream));
    return LDW_OK;
}

int ldw_sr_excess_stats(ldw_ctx *c, int nclust, int32_t S, const double *mean_dist, double *stats_out) {
    if (int rc = sr_ready(c, "ldw_sr_excess_stats")) return rc;
    if (int rc = upload_md(c, nclust, S, mean_dist, "ldw_sr_excess_stats")) return rc;
    LDW_REQUIRE(stats_out, LDW_ERR_ARG, "ldw_sr_excess_stats: null output");
    const int64_t n = c->n_sr;
    for (int k = 0; k < nclust * 5; ++k) stats_out[k] = 0.0;
    if (n == 0) return LDW_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, SRM_GRID);
    const size_t pbytes = (size_t)grid * nclust * 5 * 8;
    if (int rc = c->srm_part.reserve(pbytes)) return rc;
    if (nclust <= 4 && getenv("LDW_SR_STATS_PEEL") == nullptr)
        hipLaunchKernelGGL(k_sr_stats_small<4>, dim3(grid), dim3(256), 0, c->stream, c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(), c->sr_mi.as<double>(), n,
                           c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, c->srm_sr_dist, c->srm_md.as<double>(), S, nclust, c->srm_part.as<double>(),
                           (const int64_t *)nullptr, 0);
    else
    hipLaunchKernelGGL(k_sr_stats, dim3(grid), dim3(256), (size_t)4 * nclust * 5 * 8, c->stream, c->sr_a.as<int32_t>(),
                       c->sr_b.as<int32_t>(), c->sr_mi.as<double>(), n, c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g,
                       c->srm_sr_dist, c->srm_md.as<double>(), S, nclust, c->srm_part.as<double>(), (const int64_t *)nullptr, 0);
    LDW_HIP(hipGetLastError());
    std::vector<double> part((size_t)grid * nclust * 5);
    LDW_HIP(hipMemcpyAsync(part.data(), c->srm_part.p, pbytes, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    for (int b = 0; b < grid; ++b)   // fixed order: the result does not depend on scheduling
        for (int k = 0; k < nclust * 5; ++k) stats_out[k] += part[(size_t)b * nclust * 5 + k];
    return LDW_OK;
}

int ldw_sr_pvalues(ldw_ctx *c, int nclust, int32_t S, const double *mean_dist, const double *shape, double srp_cutoff,
                   int64_t *n_red_out, int64_t *n_pool_out, double *min_mi_out) {
    if (int rc = sr_ready(c, "ldw_sr_pvalues")) return rc;
    if (int rc = upload_md(c, nclust, S, mean_dist, "ldw_sr_pvalues")) return rc;
    LDW_REQUIRE(shape && n_red_out, LDW_ERR_ARG, "ldw_sr_pvalues: null argument");   // (n_pool_out null: no pool — the caller builds it from a minimum over all ranks, ldw_sr_pool_build)
    for (int k = 0; k < nclust; ++k)
        LDW_REQUIRE(shape[3 * k] > 0 && shape[3 * k + 1] > 0 && std::isfinite(shape[3 * k + 2]), LDW_ERR_ARG,
                    "ldw_sr_pvalues: cluster %d has an invalid beta shape", k + 1);
    const int64_t n = c->n_sr;
    c->n_red = c->n_pool = 0;
    c->red_from_lr = false;
    *n_red_out = 0;
    if (n_pool_out) *n_pool_out = 0;
    if (min_mi_out) *min_mi_out = std::nan("");
    if (n == 0) return LDW_OK;
    if (int rc = c->srm_shape.reserve((size_t)nclust * 32)) return rc;
    if (int rc = c->srm_cnt.reserve(sizeof(SrCounters))) return rc;
    LDW_HIP(hipMemcpyAsync(c->srm_shape.p, shape, (size_t)nclust * 24, hipMemcpyHostToDevice, c->stream));
    SrCounters h = {0, 0, ~0ull};
    SrCounters *d = c->srm_cnt.as<SrCounters>();
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 16384);
    const int32_t *sa = c->sr_a.as<int32_t>(), *sb = c->sr_b.as<int32_t>();
    const double *smi = c->sr_mi.as<double>();
    const int32_t *POS = c->POS.as<int32_t>(), *paint = c->paint.as<int32_t>();
    // pass 1 writes into whatever capacity is there; a second pass runs only if it was too small.  r04: a floor of 4 M rows (144 MB in all)
    // under both outputs, so that the FIRST job of a context does not pay the two kernels twice (C5: 1.5 M rows kept of 2.25e9, 50 ms per pass)
    {
        const size_t floor_rows = (size_t)std::min<int64_t>(n, (int64_t)1 << 22);
        if (int rc = c->red_row.reserve(floor_rows * 8)) return rc;
        if (int rc = c->red_meta.reserve(floor_rows * 4)) return rc;
        if (int rc = c->red_srp.reserve(floor_rows * 8)) return rc;
        if (int rc = c->pool_a.reserve(floor_rows * 4)) return rc;
        if (int rc = c->pool_b.reserve(floor_rows * 4)) return rc;
        if (int rc = c->pool_mi.reserve(floor_rows * 8)) return rc;
    }
    // (srm_shape: 3 doubles per cluster, the crossings behind them)
    double *d_dstar = nullptr;
    if (getenv("LDW_SR_PVAL_ALL") == nullptr) {
        d_dstar = c->srm_shape.as<double>() + (size_t)nclust * 3;
        hipLaunchKernelGGL(k_sr_dstar, dim3((nclust + 63) / 64), dim3(64), 0, c->stream, c->srm_shape.as<double>(), nclust, srp_cutoff, d_dstar);
        LDW_HIP(hipGetLastError());
    }
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t cap = (int64_t)(c->red_row.cap / 8);
        LDW_HIP(hipMemcpyAsync(d, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_sr_pval<1>, dim3(grid), dim3(256), 0, c->stream, sa, sb, smi, n, POS, paint, c->g, c->srm_sr_dist,
                           c->srm_md.as<double>(), S, c->srm_shape.as<double>(), srp_cutoff, d, c->red_row.as<int64_t>(),
                           c->red_meta.as<uint32_t>(), c->red_srp.as<double>(),
                           std::min<int64_t>(cap, std::min<int64_t>((int64_t)(c->red_meta.cap / 4), (int64_t)(c->red_srp.cap / 8))), d_dstar);
        LDW_HIP(hipGetLastError());
        SrCounters got;
        LDW_HIP(hipMemcpyAsync(&got, d, sizeof(got), hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        c->n_red = (int64_t)got.n_red;
        h.min_key = got.min_key;
        const int64_t have = std::min<int64_t>(cap, std::min<int64_t>((int64_t)(c->red_meta.cap / 4), (int64_t)(c->red_srp.cap / 8)));
        if (c->n_red <= have) break;
        LDW_REQUIRE(pass == 0, LDW_ERR_STATE, "ldw_sr_pvalues: reduced set changed size between passes");
        if (int rc = c->red_row.reserve((size_t)c->n_red * 8)) return rc;
        if (int rc = c->red_meta.reserve((size_t)c->n_red * 4)) return rc;
        if (int rc = c->red_srp.reserve((size_t)c->n_red * 8)) return rc;
        h.min_key = ~0ull;
    }
    *n_red_out = c->n_red;
    if (c->n_red == 0) return LDW_OK;
    if (min_mi_out) *min_mi_out = key_f64(h.min_key);
    if (!n_pool_out) return LDW_OK;
    if (int rc = build_pool(c, h.min_key)) return rc;
    *n_pool_out = c->n_pool;
    return LDW_OK;
}

int ldw_sr_reduced_fetch(ldw_ctx *c, int64_t capacity, int64_t *row_out, int32_t *a_out, int32_t *b_out, double *MI_out,
                         int32_t *clust_c_out, int32_t *first_clust_out, uint8_t *dup_out, double *srp_out) {
    if (int rc = check_gpu(c)) return rc;
    const int64_t n = c->n_red;
    LDW_REQUIRE(capacity >= n, LDW_ERR_SIZE, "ldw_sr_reduced_fetch: capacity %lld < %lld rows", (long long)capacity, (long long)n);
    if (n == 0) return LDW_OK;
    LDW_REQUIRE(row_out && a_out && b_out && MI_out && clust_c_out && first_clust_out && dup_out && srp_out, LDW_ERR_ARG,
                "ldw_sr_reduced_fetch: null output");
    LDW_REQUIRE(!c->red_from_lr, LDW_ERR_STATE, "ldw_sr_reduced_fetch: the reduced set is the long-range one (ldw_lr_tukey ran last)");
    std::vector<uint32_t> meta((size_t)n);
    if (int rc = c->scratch.reserve((size_t)n * 16)) return rc;
    double *gmi = c->scratch.as<double>();
    int32_t *ga = reinterpret_cast<int32_t *>(gmi + n), *gb = ga + n;
    hipLaunchKernelGGL(k_red_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->red_row.as<int64_t>(), n,
                       c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(), c->sr_mi.as<double>(), ga, gb, gmi);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(a_out, ga, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(b_out, gb, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(MI_out, gmi, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(row_out, c->red_row.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(meta.data(), c->red_meta.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(srp_out, c->red_srp.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < n; ++i) {
        clust_c_out[i] = (int32_t)(meta[i] & 0xFF);
        first_clust_out[i] = (int32_t)((meta[i] >> 8) & 0xFF);
        dup_out[i] = (uint8_t)((meta[i] >> 16) & 1);
    }
    return LDW_OK;
}

int ldw_sr_pool_fetch(ldw_ctx *c, int64_t capacity, int32_t *a_out, int32_t *b_out, double *MI_out) {
    if (int rc = check_gpu(c)) return rc;
    const int64_t n = c->n_pool;
    LDW_REQUIRE(capacity >= n, LDW_ERR_SIZE, "ldw_sr_pool_fetch: capacity %lld < %lld rows", (long long)capacity, (long long)n);
    if (n == 0) return LDW_OK;
    LDW_REQUIRE(a_out && b_out && MI_out, LDW_ERR_ARG, "ldw_sr_pool_fetch: null output");
    LDW_HIP(hipMemcpyAsync(a_out, c->pool_a.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(b_out, c->pool_b.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(MI_out, c->pool_mi.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

// rows of the long-range table kept by ldw_lr_tukey, in table order
int ldw_lr_reduced_fetch(ldw_ctx *c, int64_t capacity, int64_t *row_out, int32_t *a_out, int32_t *b_out, double *MI_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(c->red_from_lr, LDW_ERR_STATE, "ldw_lr_reduced_fetch: call ldw_lr_tukey first");
    const int64_t n = c->n_red;
    LDW_REQUIRE(capacity >= n, LDW_ERR_SIZE, "ldw_lr_reduced_fetch: capacity %lld < %lld rows", (long long)capacity, (long long)n);
    if (n == 0) return LDW_OK;
    LDW_REQUIRE(row_out && a_out && b_out && MI_out, LDW_ERR_ARG, "ldw_lr_reduced_fetch: null output");
    if (int rc = c->scratch.reserve((size_t)n * 16)) return rc;
    double *gmi = c->scratch.as<double>();
    int32_t *ga = reinterpret_cast<int32_t *>(gmi + n), *gb = ga + n;
    hipLaunchKernelGGL(k_red_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->red_row.as<int64_t>(), n,
                       c->lr_a.as<int32_t>(), c->lr_b.as<int32_t>(), c->lr_mi.as<double>(), ga, gb, gmi);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(a_out, ga, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(b_out, gb, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(MI_out, gmi, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(row_out, c->red_row.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_aracne_device(ldw_ctx *c, int64_t capacity, uint8_t *flags_out) {
    if (int rc = check_gpu(c)) return rc;
    const int64_t nr = c->n_red, np = c->n_pool;
    LDW_REQUIRE(capacity >= nr, LDW_ERR_SIZE, "ldw_aracne_device: capacity %lld < %lld links", (long long)capacity, (long long)nr);
    if (nr == 0) return LDW_OK;
    LDW_REQUIRE(flags_out, LDW_ERR_ARG, "ldw_aracne_device: null output");
    LDW_REQUIRE(np > 0, LDW_ERR_STATE, "ldw_aracne_device: no pool (call ldw_sr_pvalues or ldw_lr_tukey first)");
    const int64_t n2 = 2 * np, L = c->L;
    if (int rc = c->ar_key.reserve((size_t)n2 * 8)) return rc;
    if (int rc = c->ar_key2.reserve((size_t)n2 * 8)) return rc;
    if (int rc = c->ar_val.reserve((size_t)n2 * 8)) return rc;
    if (int rc = c->ar_val2.reserve((size_t)n2 * 8)) return rc;
    if (int rc = c->ar_off.reserve((size_t)(L + 2) * 8)) return rc;
    if (int rc = c->ar_flags.reserve((size_t)nr)) return rc;
    const int grid = (int)std::min<int64_t>((np + 255) / 256, 16384);
    hipLaunchKernelGGL(k_ar_edges, dim3(grid), dim3(256), 0, c->stream, c->pool_a.as<int32_t>(), c->pool_b.as<int32_t>(),
                       c->pool_mi.as<double>(), np, c->ar_key.as<uint64_t>(), c->ar_val.as<double>());
    size_t tb = 0;
    LDW_HIP(prim_sort_pairs(nullptr, tb, c->ar_key.as<uint64_t>(), c->ar_key2.as<uint64_t>(), c->ar_val.as<double>(),
                                               c->ar_val2.as<double>(), n2, 0, 64, c->stream));
    if (int rc = c->scratch.reserve(tb)) return rc;
    tb = c->scratch.cap;
    LDW_HIP(prim_sort_pairs(c->scratch.p, tb, c->ar_key.as<uint64_t>(), c->ar_key2.as<uint64_t>(),
                                               c->ar_val.as<double>(), c->ar_val2.as<double>(), n2, 0, 64, c->stream));
    hipLaunchKernelGGL(k_ar_offsets, dim3((unsigned)((L + 1 + 255) / 256)), dim3(256), 0, c->stream, c->ar_key2.as<uint64_t>(), n2, L,
                       c->ar_off.as<int64_t>());
    // the links to check are rows of the short-range table (after ldw_sr_pvalues) or of the long-range one (after ldw_lr_tukey)
    const int32_t *ta = (c->red_from_lr ? c->lr_a : c->sr_a).as<int32_t>(), *tb2 = (c->red_from_lr ? c->lr_b : c->sr_b).as<int32_t>();
    const double *tmi = (c->red_from_lr ? c->lr_mi : c->sr_mi).as<double>();
    hipLaunchKernelGGL(k_ar_check, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, c->stream, c->red_row.as<int64_t>(), nr,
                       ta, tb2, tmi, c->ar_key2.as<uint64_t>(),
                       c->ar_val2.as<double>(), c->ar_off.as<int64_t>(), c->ar_flags.as<uint8_t>());
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(flags_out, c->ar_flags.p, (size_t)nr, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

// ---- r05: the short-range model over ranks (docs/HISTORY.md 7b): the table stays where it was computed ----------------------------------------

int ldw_sr_tail_extract(ldw_ctx *c, int nclust, int32_t S, const double *lower, int64_t *cnt_out, double *mi_out, int64_t capacity, int on_device,
                        int64_t *n_out) {
    if (int rc = sr_ready(c, "ldw_sr_tail_extract")) return rc;
    LDW_REQUIRE(nclust == c->srm_nclust && S == c->srm_S && nclust >= 1 && S >= 1, LDW_ERR_STATE,
                "ldw_sr_tail_extract: nclust/S differ from the last ldw_sr_len_quantiles call");
    LDW_REQUIRE(lower && cnt_out && n_out, LDW_ERR_ARG, "ldw_sr_tail_extract: null argument");
    const int64_t n = c->n_sr, G = (int64_t)nclust * S;
    *n_out = 0;
    for (int64_t k = 0; k < G; ++k) cnt_out[k] = 0;
    if (n == 0) return LDW_OK;
    if (int rc = c->srd_lower.reserve((size_t)G * 8)) return rc;
    if (int rc = c->srd_cur.reserve((size_t)G * 8)) return rc;
    LDW_HIP(hipMemcpyAsync(c->srd_lower.p, lower, (size_t)G * 8, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemsetAsync(c->srd_cur.p, 0, (size_t)G * 8, c->stream));
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 16384);
    const int32_t *sa = c->sr_a.as<int32_t>(), *sb = c->sr_b.as<int32_t>();
    const double *smi = c->sr_mi.as<double>();
    hipLaunchKernelGGL(k_sr_tail<0>, dim3(grid), dim3(256), 0, c->stream, sa, sb, smi, n, c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, c->srm_sr_dist,
                       c->srd_lower.as<double>(), S, nclust, c->srd_cur.as<unsigned long long>(), (double *)nullptr);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(cnt_out, c->srd_cur.p, (size_t)G * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    int64_t total = 0;
    for (int64_t k = 0; k < G; ++k) total += cnt_out[k];
    *n_out = total;
    if (!mi_out || total == 0) return LDW_OK;   // (count only: the caller sizes its buffer and calls again)
    LDW_REQUIRE(capacity >= total, LDW_ERR_SIZE, "ldw_sr_tail_extract: capacity %lld < %lld rows", (long long)capacity, (long long)total);
    std::vector<int64_t> first((size_t)G);
    int64_t run = 0;
    for (int64_t k = 0; k < G; ++k) {
        first[(size_t)k] = run;
        run += cnt_out[k];
    }
    LDW_HIP(hipMemcpyAsync(c->srd_cur.p, first.data(), (size_t)G * 8, hipMemcpyHostToDevice, c->stream));
    double *d_out = mi_out;
    if (!on_device) {
        if (int rc = c->srd_out.reserve((size_t)total * 8)) return rc;
        d_out = c->srd_out.as<double>();
    }
    hipLaunchKernelGGL(k_sr_tail<1>, dim3(grid), dim3(256), 0, c->stream, sa, sb, smi, n, c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, c->srm_sr_dist,
                       c->srd_lower.as<double>(), S, nclust, c->srd_cur.as<unsigned long long>(), d_out);
    LDW_HIP(hipGetLastError());
    if (!on_device) LDW_HIP(hipMemcpyAsync(mi_out, d_out, (size_t)total * 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));   // (`first` is read by the copy above)
    return LDW_OK;
}

int ldw_sr_quantiles_merge(ldw_ctx *c, int nclust, int32_t S, double prob, int n_src, const double *const *mi_src, const int64_t *const *cnt_src,
                           const int64_t *n_total, int on_device, double *q_lo_out, double *q_hi_out, int64_t *violations_out) {
    if (int rc = check_gpu(c)) return rc;
    if (int rc = ldw::join_prepare(c)) return rc;
    LDW_REQUIRE(nclust >= 1 && nclust <= SRM_MAXCL && S >= 1 && S <= 65534, LDW_ERR_ARG, "ldw_sr_quantiles_merge: nclust must be in 1..%d, S in 1..65534", SRM_MAXCL);
    LDW_REQUIRE(prob >= 0 && prob <= 1, LDW_ERR_ARG, "ldw_sr_quantiles_merge: prob outside [0,1]");
    LDW_REQUIRE(n_src >= 1 && mi_src && cnt_src && n_total && q_lo_out && q_hi_out, LDW_ERR_ARG, "ldw_sr_quantiles_merge: null argument");
    const int64_t G = (int64_t)nclust * S;
    const size_t cells = (size_t)G;
    int64_t m = 0;
    std::vector<int64_t> m_src((size_t)n_src, 0);
    for (int r = 0; r < n_src; ++r) {
        LDW_REQUIRE(cnt_src[r], LDW_ERR_ARG, "ldw_sr_quantiles_merge: source %d has no counts", r);
        for (int64_t k = 0; k < G; ++k) {
            LDW_REQUIRE(cnt_src[r][k] >= 0, LDW_ERR_ARG, "ldw_sr_quantiles_merge: negative count");
            m_src[(size_t)r] += cnt_src[r][k];
        }
        LDW_REQUIRE(m_src[(size_t)r] == 0 || mi_src[r], LDW_ERR_ARG, "ldw_sr_quantiles_merge: source %d has no values", r);
        m += m_src[(size_t)r];
    }
    std::vector<double> q(cells * 2, std::nan(""));
    unsigned int viol = 0;
    if (m == 0) {
        for (size_t k = 0; k < cells; ++k) viol += n_total[k] != 0;
    } else {
        LDW_REQUIRE(m < (int64_t)0x7FFFFFFFll * 4, LDW_ERR_SIZE, "ldw_sr_quantiles_merge: too many candidates");
        if (int rc = quant_reserve_sorts(c, m, S, cells)) return rc;
        if (int rc = c->srd_lower.reserve((size_t)G * 8)) return rc;
        if (int rc = c->srd_seg.reserve((size_t)n_src * (size_t)(G + 1) * 8)) return rc;
        if (!on_device)
            if (int rc = c->srd_out.reserve((size_t)m * 8)) return rc;
        LDW_HIP(hipMemcpyAsync(c->srd_lower.p, n_total, (size_t)G * 8, hipMemcpyHostToDevice, c->stream));
        std::vector<int64_t> goff((size_t)n_src * (size_t)(G + 1));
        int64_t base = 0;
        for (int r = 0; r < n_src; ++r) {
            int64_t *go = goff.data() + (size_t)r * (size_t)(G + 1);
            go[0] = 0;
            for (int64_t k = 0; k < G; ++k) go[k + 1] = go[k] + cnt_src[r][k];
            const int64_t mr = m_src[(size_t)r];
            if (mr == 0) continue;
            int64_t *d_go = c->srd_seg.as<int64_t>() + (size_t)r * (size_t)(G + 1);
            LDW_HIP(hipMemcpyAsync(d_go, go, (size_t)(G + 1) * 8, hipMemcpyHostToDevice, c->stream));
            const double *d_mi = mi_src[r];
            if (!on_device) {
                LDW_HIP(hipMemcpyAsync(c->srd_out.as<double>() + base, mi_src[r], (size_t)mr * 8, hipMemcpyHostToDevice, c->stream));
                d_mi = c->srd_out.as<double>() + base;
            }
            hipLaunchKernelGGL(k_tail_unpack, dim3((unsigned)std::min<int64_t>((mr + 255) / 256, 16384)), dim3(256), 0, c->stream, d_mi, mr, d_go, (int)G, nclust,
                               base, c->srm_pack.as<uint32_t>(), c->srm_key.as<uint64_t>());
            LDW_HIP(hipGetLastError());
            base += mr;
        }
        if (int rc = quant_two_sorts(c, m, S, nclust, prob, c->srd_lower.as<int64_t>(), q, nullptr, &viol)) return rc;   // (synchronises: goff may go)
    }
    for (size_t k = 0; k < cells; ++k) {
        q_lo_out[k] = q[2 * k];
        q_hi_out[k] = q[2 * k + 1];
    }
    if (violations_out) *violations_out = (int64_t)viol;
    LDW_REQUIRE(viol == 0 || violations_out, LDW_ERR_STATE, "ldw_sr_quantiles_merge: %u groups whose order statistic lies below the candidates sent", viol);
    return LDW_OK;
}

int ldw_sr_excess_stats_blocks(ldw_ctx *c, int nclust, int32_t S, const double *mean_dist, int64_t nblocks, const int64_t *rows_per_block,
                               double *stats_out) {
    if (int rc = sr_ready(c, "ldw_sr_excess_stats_blocks")) return rc;
    if (int rc = upload_md(c, nclust, S, mean_dist, "ldw_sr_excess_stats_blocks")) return rc;
    LDW_REQUIRE(nblocks >= 0 && (nblocks == 0 || (rows_per_block && stats_out)), LDW_ERR_ARG, "ldw_sr_excess_stats_blocks: null argument");
    constexpr int STRIPS = 64;   // strips per block: fixed, so that a block's sums do not depend on which rank holds it
    std::vector<int64_t> seg;
    std::vector<int64_t> which;
    int64_t run = 0;
    for (int64_t b = 0; b < nblocks; ++b) {
        LDW_REQUIRE(rows_per_block[b] >= 0, LDW_ERR_ARG, "ldw_sr_excess_stats_blocks: negative row count");
        if (rows_per_block[b] > 0) {
            seg.push_back(run);
            seg.push_back(run + rows_per_block[b]);
            which.push_back(b);
        }
        run += rows_per_block[b];
    }
    LDW_REQUIRE(run == c->n_sr, LDW_ERR_ARG, "ldw_sr_excess_stats_blocks: the blocks hold %lld rows, the short-range table %lld", (long long)run, (long long)c->n_sr);
    const size_t per = (size_t)nclust * 5;
    for (size_t k = 0; k < (size_t)nblocks * per; ++k) stats_out[k] = 0.0;
    const int64_t nb = (int64_t)which.size();
    if (nb == 0) return LDW_OK;
    LDW_REQUIRE(nb * STRIPS < (int64_t)1 << 30, LDW_ERR_SIZE, "ldw_sr_excess_stats_blocks: too many blocks");
    const int grid = (int)(nb * STRIPS);
    const size_t pbytes = (size_t)grid * per * 8;
    if (int rc = c->srm_part.reserve(pbytes)) return rc;
    if (int rc = c->srd_seg.reserve(seg.size() * 8)) return rc;
    LDW_HIP(hipMemcpyAsync(c->srd_seg.p, seg.data(), seg.size() * 8, hipMemcpyHostToDevice, c->stream));
    const int64_t n = c->n_sr;
    if (nclust <= 4 && getenv("LDW_SR_STATS_PEEL") == nullptr)
        hipLaunchKernelGGL(k_sr_stats_small<4>, dim3(grid), dim3(256), 0, c->stream, c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(), c->sr_mi.as<double>(), n,
                           c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, c->srm_sr_dist, c->srm_md.as<double>(), S, nclust, c->srm_part.as<double>(),
                           c->srd_seg.as<int64_t>(), STRIPS);
    else
        hipLaunchKernelGGL(k_sr_stats, dim3(grid), dim3(256), (size_t)4 * nclust * 5 * 8, c->stream, c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(),
                           c->sr_mi.as<double>(), n, c->POS.as<int32_t>(), c->paint.as<int32_t>(), c->g, c->srm_sr_dist, c->srm_md.as<double>(), S, nclust,
                           c->srm_part.as<double>(), c->srd_seg.as<int64_t>(), STRIPS);
    LDW_HIP(hipGetLastError());
    std::vector<double> part((size_t)grid * per);
    LDW_HIP(hipMemcpyAsync(part.data(), c->srm_part.p, pbytes, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    for (int64_t k = 0; k < nb; ++k) {   // strips in order: fixed
        double *o = stats_out + (size_t)which[(size_t)k] * per;
        for (int j = 0; j < STRIPS; ++j)
            for (size_t t = 0; t < per; ++t) o[t] += part[((size_t)k * STRIPS + (size_t)j) * per + t];
    }
    return LDW_OK;
}

int ldw_sr_pool_build(ldw_ctx *c, double min_mi, int64_t *n_pool_out) {
    if (int rc = sr_ready(c, "ldw_sr_pool_build")) return rc;
    LDW_REQUIRE(n_pool_out, LDW_ERR_ARG, "ldw_sr_pool_build: null output");
    LDW_REQUIRE(c->srm_S > 0 && c->srm_md.p, LDW_ERR_STATE, "ldw_sr_pool_build: call ldw_sr_pvalues first (the fitted decay is not on the device)");
    *n_pool_out = 0;
    c->n_pool = 0;
    if (std::isnan(min_mi)) return LDW_OK;   // nothing was kept anywhere: no pool
    if (int rc = build_pool(c, (unsigned long long)f64_key(min_mi))) return rc;
    *n_pool_out = c->n_pool;
    return LDW_OK;
}

int ldw_sr_reduced_import(ldw_ctx *c, int64_t n_red, const int32_t *a, const int32_t *b, const double *MI, int64_t n_pool, const int32_t *pool_a,
                          const int32_t *pool_b, const double *pool_MI) {
    return ldw::reduced_import_full(c, n_red, a, b, MI, nullptr, nullptr, n_pool, pool_a, pool_b, pool_MI);
}

}  // extern "C"

namespace ldw {
int reduced_import_full(ldw_ctx *c, int64_t n_red, const int32_t *a, const int32_t *b, const double *MI, const uint32_t *meta, const double *srp, int64_t n_pool,
                        const int32_t *pool_a, const int32_t *pool_b, const double *pool_MI) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(n_red >= 0 && n_pool >= 0 && (n_red == 0 || (a && b && MI)) && (n_pool == 0 || (pool_a && pool_b && pool_MI)), LDW_ERR_ARG,
                "ldw_sr_reduced_import: bad argument");
    for (int64_t i = 0; i < n_red; ++i)
        LDW_REQUIRE(a[i] >= 0 && a[i] < c->L && b[i] >= 0 && b[i] < c->L, LDW_ERR_ARG, "ldw_sr_reduced_import: link %lld has a SNP index outside 0..L-1", (long long)i);
    for (int64_t i = 0; i < n_pool; ++i)
        LDW_REQUIRE(pool_a[i] >= 0 && pool_a[i] < c->L && pool_b[i] >= 0 && pool_b[i] < c->L, LDW_ERR_ARG,
                    "ldw_sr_reduced_import: pool link %lld has a SNP index outside 0..L-1", (long long)i);
    if (int rc = ldw_links_import(c, 0, a, b, MI, n_red, 0)) return rc;   // the kept links ARE the short-range table now (rows 0..n_red-1)
    c->red_from_lr = false;
    const size_t nr = (size_t)std::max<int64_t>(n_red, 1), np = (size_t)std::max<int64_t>(n_pool, 1);
    if (int rc = c->red_row.reserve(nr * 8)) return rc;
    if (int rc = c->red_meta.reserve(nr * 4)) return rc;
    if (int rc = c->red_srp.reserve(nr * 8)) return rc;
    if (int rc = c->pool_a.reserve(np * 4)) return rc;
    if (int rc = c->pool_b.reserve(np * 4)) return rc;
    if (int rc = c->pool_mi.reserve(np * 8)) return rc;
    if (meta && n_red > 0) LDW_HIP(hipMemcpyAsync(c->red_meta.p, meta, (size_t)n_red * 4, hipMemcpyHostToDevice, c->stream));
    else LDW_HIP(hipMemsetAsync(c->red_meta.p, 0, nr * 4, c->stream));
    if (srp && n_red > 0) LDW_HIP(hipMemcpyAsync(c->red_srp.p, srp, (size_t)n_red * 8, hipMemcpyHostToDevice, c->stream));
    else LDW_HIP(hipMemsetAsync(c->red_srp.p, 0, nr * 8, c->stream));
    if (n_red > 0) {
        hipLaunchKernelGGL(k_iota64, dim3((unsigned)((n_red + 255) / 256)), dim3(256), 0, c->stream, c->red_row.as<int64_t>(), n_red);
        LDW_HIP(hipGetLastError());
    }
    if (n_pool > 0) {
        LDW_HIP(hipMemcpyAsync(c->pool_a.p, pool_a, (size_t)n_pool * 4, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemcpyAsync(c->pool_b.p, pool_b, (size_t)n_pool * 4, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemcpyAsync(c->pool_mi.p, pool_MI, (size_t)n_pool * 8, hipMemcpyHostToDevice, c->stream));
    }
    LDW_HIP(hipStreamSynchronize(c->stream));
    c->n_red = n_red;
    c->n_pool = n_pool;
    return LDW_OK;
}

void warm_srp() {   // ldw_ctx_reserve: load this translation unit's code object ahead of its first launch
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_sr_tag));
    (void)hipGetLastError();
}
}  // namespace ldw
