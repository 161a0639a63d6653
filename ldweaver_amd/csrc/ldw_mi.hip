// MI epilogue, link selection and the block drivers (twins of perform_MI_computation_ACGTN,
// R/computePairwiseMI.R:167-386, and of the block loop of perform_MI_computation, :103-116).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <hipcub/hipcub.hpp>

#include "ldw_internal.h"

using namespace ldw;

namespace ldw {

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// len = 0.5*g - abs((pos1 - pos2) %% g - 0.5*g)   (R/computePairwiseMI.R:330; R's floored %%)
__host__ __device__ __forceinline__ double circ_len(double pos1, double pos2, double g) {
    const double x = pos1 - pos2;
    double d = x - floor(x / g) * g;
    if (d < 0) d += g;
    if (d >= g) d -= g;
    return 0.5 * g - fabs(d - 0.5 * g);
}

__device__ __forceinline__ int mi_bucket(double mi) {
    int b = (int)floor(mi * ((double)NBINS / MI_HIST_MAX));
    return b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
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

// one term of src/computeMI.cpp:19 with uq = 1:  pxy/den * log(pxy/(pxpy+RXY+pXrX+pYrY)*den)
__device__ __forceinline__ double mi_term(int64_t nfix, double scale, double pX, double pY, double RXY, double rX,
                                          double rY, double den) {
    const double pxy = (double)nfix * scale + 0.5;
    const double d = ((pX * pY + RXY) + pX * rX) + pY * rY;
    return (pxy / den) * log((pxy / d) * den);
}

// ------------------------------------------------------------------------------------------------
// MI epilogue: one thread per SNP pair; wave = 64 consecutive from-side SNPs at one to-side SNP
// ------------------------------------------------------------------------------------------------
struct EpiArgs {
    const int64_t *G;
    int RFpad;
    const int32_t *idx_f, *lrow_f, *idx_t, *lrow_t;
    int nf, nt;
    const uint32_t *slot_meta;
    const int64_t *slot_pfix;
    const double *r;
    double neff, scale;
    int quirk, lower_only;
    double *MI;
};

__global__ __launch_bounds__(256) void k_mi_epilogue(EpiArgs A) {
    const int a_loc = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b_loc = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (a_loc >= A.nf || b_loc >= A.nt) return;
    if (A.lower_only && a_loc <= b_loc) return;
    const int sa = A.idx_f[a_loc], sb = A.idx_t[b_loc];
    const uint32_t ma = A.slot_meta[sa], mb = A.slot_meta[sb];
    const int na = ma & 7, nb = mb & 7;
    const int64_t ra0 = A.lrow_f[a_loc], rb0 = A.lrow_t[b_loc];

    int64_t pa[5], pb[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        pa[i] = A.slot_pfix[(int64_t)sa * 5 + i];
        pb[i] = A.slot_pfix[(int64_t)sb * 5 + i];
    }
    // joint counts of the row slots, their row / column sums
    int64_t g[4][4], rs[4] = {0, 0, 0, 0}, cs[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t v = 0;
            if (i < na && j < nb) v = A.G[(rb0 + j) * A.RFpad + ra0 + i];
            g[i][j] = v;
            rs[i] += v;
            cs[j] += v;
        }
    // drop-slot marginals
    int64_t pa_drop = 0, pb_drop = 0, dd = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (i == na) pa_drop = pa[i];
        if (i == nb) pb_drop = pb[i];
    }
    dd = pa_drop;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nb) dd -= pb[j] - cs[j];
    (void)pb_drop;

    const double ra = A.r[sa], rb = A.r[sb];
    const double den = A.neff + (ra * rb) * 0.5;  // R/computePairwiseMI.R:260
    double RXY;
    if (A.quirk == LDW_QUIRK_REFERENCE) {
        // rft is nt x nf but read by the linear index of the nf x nt matrix (Q1)
        const int64_t c = (int64_t)a_loc + (int64_t)b_loc * A.nf;
        RXY = (A.r[A.idx_f[c / A.nt]] * A.r[A.idx_t[c % A.nt]]) * 0.25;
    } else {
        RXY = (ra * rb) * 0.25;
    }
    const double rX = 0.5 * ra, rY = 0.5 * rb;

    double mi = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (i <= na && ((ma >> (3 + i)) & 1)) {
            const double pX = (double)pa[i] * A.scale;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (j <= nb && ((mb >> (3 + j)) & 1)) {
                    int64_t nfix;
                    if (i < 4 && j < 4 && i < na && j < nb) nfix = g[i < 4 ? i : 0][j < 4 ? j : 0];
                    else if (i < 4 && i < na) nfix = pa[i] - rs[i < 4 ? i : 0];   // j == nb
                    else if (j < 4 && j < nb) nfix = pb[j] - cs[j < 4 ? j : 0];   // i == na
                    else nfix = dd;
                    const double pY = (double)pb[j] * A.scale;
                    mi += mi_term(nfix, A.scale, pX, pY, RXY, rX, rY, den);
                }
            }
        }
    }
    A.MI[(int64_t)a_loc + (int64_t)b_loc * A.nf] = mi;
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
// selection
// ------------------------------------------------------------------------------------------------
struct SelArgs {
    const double *MI;
    const int32_t *idx_f, *idx_t;
    int nf, nt;
    const int32_t *POS;
    double g, sr_dist;
    int lower_only;
};

// which segment a pair belongs to: 0 = upper (a<b, off-diagonal blocks only), 1 = lower (a>b), -1 = not a pair
__device__ __forceinline__ int pair_seg(int a_loc, int b_loc, int lower_only) {
    if (a_loc == b_loc) return -1;
    if (a_loc > b_loc) return 1;
    return lower_only ? -1 : 0;
}

// per-column short-range counts: colcnt[seg*nt + b]
__global__ __launch_bounds__(256) void k_sr_count(SelArgs S, int32_t *__restrict__ colcnt) {
    const int lane = threadIdx.x & 63;
    const int b_loc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b_loc >= S.nt) return;
    const double pos1 = (double)S.POS[S.idx_t[b_loc]];
    int cu = 0, cl = 0;
    for (int a0 = 0; a0 < S.nf; a0 += 64) {
        const int a_loc = a0 + lane;
        if (a_loc < S.nf) {
            const int seg = pair_seg(a_loc, b_loc, S.lower_only);
            if (seg >= 0) {
                const bool sr = circ_len(pos1, (double)S.POS[S.idx_f[a_loc]], S.g) <= S.sr_dist;
                cu += (sr && seg == 0);
                cl += (sr && seg == 1);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cu += __shfl_xor(cu, off);
        cl += __shfl_xor(cl, off);
    }
    if (lane == 0) {
        colcnt[b_loc] = cu;
        colcnt[S.nt + b_loc] = cl;
    }
}

// exclusive scan of 2*nt column counts (single workgroup); ctr[0] += total (running sr rows), ctr[3] = block total
__global__ __launch_bounds__(1024) void k_scan_cols(const int32_t *__restrict__ colcnt, int n, int64_t *__restrict__ offs,
                                                    int64_t *__restrict__ blk_total) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += colcnt[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int64_t run = 0;
        for (int i = 0; i < 1024; ++i) {
            const int64_t v = part[i];
            part[i] = run;
            run += v;
        }
        *blk_total = run;
    }
    __syncthreads();
    int64_t run = part[t];
    for (int i = lo; i < hi; ++i) {
        offs[i] = run;
        run += colcnt[i];
    }
}

// ordered scatter of the short-range links of one block
__global__ __launch_bounds__(256) void k_sr_scatter(SelArgs S, const int64_t *__restrict__ offs, int64_t base,
                                                    int32_t *__restrict__ out_a, int32_t *__restrict__ out_b,
                                                    double *__restrict__ out_mi) {
    const int lane = threadIdx.x & 63;
    const int b_loc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b_loc >= S.nt) return;
    const int sb = S.idx_t[b_loc];
    const double pos1 = (double)S.POS[sb];
    int64_t wu = base + offs[b_loc], wl = base + offs[S.nt + b_loc];
    for (int a0 = 0; a0 < S.nf; a0 += 64) {
        const int a_loc = a0 + lane;
        int seg = -1;
        bool sr = false;
        int sa = 0;
        if (a_loc < S.nf) {
            seg = pair_seg(a_loc, b_loc, S.lower_only);
            if (seg >= 0) {
                sa = S.idx_f[a_loc];
                sr = circ_len(pos1, (double)S.POS[sa], S.g) <= S.sr_dist;
            }
        }
        const unsigned long long mu = __ballot(sr && seg == 0), ml = __ballot(sr && seg == 1);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (sr) {
            const int64_t dst = seg == 0 ? wu + __popcll(mu & lt) : wl + __popcll(ml & lt);
            out_a[dst] = sa;
            out_b[dst] = sb;
            out_mi[dst] = S.MI[(int64_t)a_loc + (int64_t)b_loc * S.nf];
        }
        wu += __popcll(mu);
        wl += __popcll(ml);
    }
}

// level-1 histogram of the long-range MI values of one block (16 columns per workgroup)
__global__ __launch_bounds__(256) void k_lr_hist(SelArgs S, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int sh[NBINS];
    for (int i = threadIdx.x; i < NBINS; i += 256) sh[i] = 0;
    __syncthreads();
    const int b0 = blockIdx.x * 16;
    for (int bb = 0; bb < 16; ++bb) {
        const int b_loc = b0 + bb;
        if (b_loc >= S.nt) break;
        const double pos1 = (double)S.POS[S.idx_t[b_loc]];
        const double *col = S.MI + (int64_t)b_loc * S.nf;
        for (int a_loc = threadIdx.x; a_loc < S.nf; a_loc += 256) {
            if (pair_seg(a_loc, b_loc, S.lower_only) < 0) continue;
            if (circ_len(pos1, (double)S.POS[S.idx_f[a_loc]], S.g) <= S.sr_dist) continue;
            atomicAdd(&sh[mi_bucket(col[a_loc])], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

struct PickOut {
    long long n;        // number of long-range pairs in the block
    long long lo, hi;   // 1-based ranks of the two order statistics of quantile type 7
    long long n_below;  // pairs in buckets below B
    double index, prob;
    int B;              // first bucket gathered
    int pad;
    unsigned long long n_cand;  // filled by k_lr_gather
    long long n_kept;           // filled by k_lr_thresh
    double disc_thresh;
    long long kstart;
};

// prob and quantile ranks of R/computePairwiseMI.R:352-354 (stats::quantile type 7), then the bucket
// holding rank lo
__global__ void k_pick_bucket(const unsigned long long *__restrict__ hist, double lr_retain, double lr_approx,
                              PickOut *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long n = 0;
    for (int i = 0; i < NBINS; ++i) n += (long long)hist[i];
    PickOut o;
    memset(&o, 0, sizeof(o));
    o.n = n;
    o.B = NBINS;
    o.disc_thresh = nan("");
    if (n > 0) {
        const double dn = (double)n;
        double prob = 1.0 - ((lr_retain * (dn / lr_approx)) / dn);
        if (!(prob > 0.0)) prob = 0.0;
        o.prob = prob;
        o.index = 1.0 + (dn - 1.0) * prob;
        o.lo = (long long)floor(o.index);
        o.hi = (long long)ceil(o.index);
        long long cum = 0;
        for (int i = 0; i < NBINS; ++i) {
            const long long h = (long long)hist[i];
            if (cum + h >= o.lo) {
                o.B = i;
                o.n_below = cum;
                break;
            }
            cum += h;
        }
    }
    *out = o;
}

// gather every long-range pair whose bucket is >= B: (MI key, order key)
__global__ __launch_bounds__(256) void k_lr_gather(SelArgs S, PickOut *__restrict__ pick, uint64_t *__restrict__ ckey,
                                                   uint64_t *__restrict__ cval) {
    const int B = pick->B;
    if (B >= NBINS) return;
    const int b0 = blockIdx.x * 16;
    for (int bb = 0; bb < 16; ++bb) {
        const int b_loc = b0 + bb;
        if (b_loc >= S.nt) break;
        const double pos1 = (double)S.POS[S.idx_t[b_loc]];
        const double *col = S.MI + (int64_t)b_loc * S.nf;
        for (int a_loc = threadIdx.x; a_loc < S.nf; a_loc += 256) {
            const int seg = pair_seg(a_loc, b_loc, S.lower_only);
            if (seg < 0) continue;
            if (circ_len(pos1, (double)S.POS[S.idx_f[a_loc]], S.g) <= S.sr_dist) continue;
            const double mi = col[a_loc];
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
    // first candidate with MI >= qs (binary search on the order-preserving keys)
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

struct BlockGeom {
    int64_t nf = 0, nt = 0;
    int RFpad = 0, RTpad = 0;
    bool diag = false;
};

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

// stage index lists, run GEMM + epilogue for one block; MI lands in ctx->MIblk (column-major nf x nt)
int run_block_mi(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, int quirk,
                 bool lower_only, BlockGeom &geo) {
    if (int rc = ensure_rows(c)) return rc;
    LDW_REQUIRE(nf > 0 && nt > 0, LDW_ERR_ARG, "empty block (nf=%lld nt=%lld)", (long long)nf, (long long)nt);
    LDW_REQUIRE(nf <= 1000000 && nt <= 1000000, LDW_ERR_ARG, "block side too long");
    std::vector<int32_t> vf(from_idx, from_idx + nf), vt(to_idx, to_idx + nt);
    if (int rc = upload_i32(c, c->idx_f, vf)) return rc;
    if (int rc = upload_i32(c, c->idx_t, vt)) return rc;
    if (int rc = c->MIblk.reserve((size_t)nf * nt * 8)) return rc;
    geo.nf = nf;
    geo.nt = nt;
    if (c->engine == LDW_ENGINE_HIST) {
        for (int64_t k = 0; k < nf; ++k)
            LDW_REQUIRE(from_idx[k] >= 0 && from_idx[k] < c->L, LDW_ERR_ARG, "SNP index %d out of range", from_idx[k]);
        for (int64_t k = 0; k < nt; ++k)
            LDW_REQUIRE(to_idx[k] >= 0 && to_idx[k] < c->L, LDW_ERR_ARG, "SNP index %d out of range", to_idx[k]);
        LDW_HIP(hipStreamSynchronize(c->stream));
        LDW_HIP(hipEventRecord(c->ev[0], c->stream));
        LDW_HIP(hipEventRecord(c->ev[1], c->stream));
        if (int rc = launch_hist(c, c->idx_f.as<int32_t>(), (int)nf, c->idx_t.as<int32_t>(), (int)nt,
                                 c->pfix_state.as<int64_t>(), quirk, lower_only ? 1 : 0, c->MIblk.as<double>()))
            return rc;
        LDW_HIP(hipEventRecord(c->ev[2], c->stream));
        return LDW_OK;
    }
    std::vector<int32_t> rl_f, rl_t, lr_f, lr_t;
    int RFpad = 0, RTpad = 0;
    if (int rc = build_side(c, from_idx, nf, rl_f, lr_f, RFpad)) return rc;
    if (int rc = build_side(c, to_idx, nt, rl_t, lr_t, RTpad)) return rc;
    geo.RFpad = RFpad;
    geo.RTpad = RTpad;
    if (int rc = upload_i32(c, c->rowlist_f, rl_f)) return rc;
    if (int rc = upload_i32(c, c->rowlist_t, rl_t)) return rc;
    if (int rc = upload_i32(c, c->lrow_f, lr_f)) return rc;
    if (int rc = upload_i32(c, c->lrow_t, lr_t)) return rc;
    // pageable H2D copies above are complete only after a sync
    LDW_HIP(hipStreamSynchronize(c->stream));
    if (int rc = c->G.reserve((size_t)RFpad * RTpad * 8)) return rc;

    LDW_HIP(hipEventRecord(c->ev[0], c->stream));
    if (int rc = launch_gemm(c, c->rowlist_t.as<int32_t>(), RTpad, c->rowlist_f.as<int32_t>(), RFpad,
                             c->G.as<int64_t>(), c->nlimbs, c->digits.as<int8_t>(), c->M.as<uint8_t>(), c->Npad,
                             lower_only ? 1 : 0, 0))
        return rc;
    LDW_HIP(hipEventRecord(c->ev[1], c->stream));
    EpiArgs A;
    A.G = c->G.as<int64_t>();
    A.RFpad = RFpad;
    A.idx_f = c->idx_f.as<int32_t>();
    A.lrow_f = c->lrow_f.as<int32_t>();
    A.idx_t = c->idx_t.as<int32_t>();
    A.lrow_t = c->lrow_t.as<int32_t>();
    A.nf = (int)nf;
    A.nt = (int)nt;
    A.slot_meta = c->slot_meta.as<uint32_t>();
    A.slot_pfix = c->slot_pfix.as<int64_t>();
    A.r = c->r.as<double>();
    A.neff = c->neff;
    A.scale = std::ldexp(1.0, -c->frac_bits);
    A.quirk = quirk;
    A.lower_only = lower_only ? 1 : 0;
    A.MI = c->MIblk.as<double>();
    dim3 grid((unsigned)((nf + 63) / 64), (unsigned)((nt + 3) / 4));
    LDW_REQUIRE(grid.y <= 65535u, LDW_ERR_ARG, "nt too large for the epilogue grid");
    hipLaunchKernelGGL(k_mi_epilogue, grid, dim3(256), 0, c->stream, A);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipEventRecord(c->ev[2], c->stream));
    return LDW_OK;
}

bool same_list(const int32_t *a, int64_t na, const int32_t *b, int64_t nb) {
    if (na != nb) return false;
    for (int64_t i = 0; i < na; ++i)
        if (a[i] != b[i]) return false;
    return true;
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
    int64_t *blk_sr;     // short-range rows of the current block
    ldw::PickOut *pick;
    int64_t *stats_i;    // [nblocks_cap][3]
    double *stats_d;     // [nblocks_cap]
};

}  // namespace

namespace ldw {

// one block, links appended.  n_lr (host) is an upper bound while running; the exact count lives in lr_count.
static int block_links(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt,
                       const ldw_mi_params *p, int64_t blk_no, int64_t *d_lr_count, int64_t *d_stats_i,
                       double *d_stats_d, ldw::PickOut *d_pick, int64_t *d_blk_sr) {
    const bool diag = same_list(from_idx, nf, to_idx, nt);
    BlockGeom geo;
    if (int rc = run_block_mi(c, from_idx, nf, to_idx, nt, p->quirk_mode, diag, geo)) return rc;
    SelArgs S;
    S.MI = c->MIblk.as<double>();
    S.idx_f = c->idx_f.as<int32_t>();
    S.idx_t = c->idx_t.as<int32_t>();
    S.nf = (int)nf;
    S.nt = (int)nt;
    S.POS = c->POS.as<int32_t>();
    S.g = c->g;
    S.sr_dist = p->sr_dist;
    S.lower_only = diag ? 1 : 0;

    // ---- counts / histogram (no host involvement) ----
    if (int rc = c->colcnt.reserve((size_t)(2 * nt + 4) * 4 + (size_t)(2 * nt) * 8)) return rc;
    int32_t *d_colcnt = c->colcnt.as<int32_t>();
    int64_t *d_offs = reinterpret_cast<int64_t *>(d_colcnt + 2 * ((nt + 1) / 2 * 2));
    hipLaunchKernelGGL(k_sr_count, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, c->stream, S, d_colcnt);
    LDW_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_scan_cols, dim3(1), dim3(1024), 0, c->stream, d_colcnt, (int)(2 * nt), d_offs, d_blk_sr);
    LDW_HIP(hipGetLastError());
    const bool do_lr = !p->sr_only;
    if (do_lr) {
        if (int rc = c->hist.reserve((size_t)NBINS * 8)) return rc;
        LDW_HIP(hipMemsetAsync(c->hist.p, 0, (size_t)NBINS * 8, c->stream));
        hipLaunchKernelGGL(k_lr_hist, dim3((unsigned)((nt + 15) / 16)), dim3(256), 0, c->stream, S,
                           c->hist.as<unsigned long long>());
        LDW_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_pick_bucket, dim3(1), dim3(64), 0, c->stream, c->hist.as<unsigned long long>(),
                           p->lr_retain_links, p->lr_links_approx, d_pick);
        LDW_HIP(hipGetLastError());
        // candidate capacity: every pair of the block in the worst case (all MI in one bucket)
        const size_t cap = (size_t)nf * nt;
        if (int rc = c->cand_key.reserve(cap * 8)) return rc;
        if (int rc = c->cand_val.reserve(cap * 8)) return rc;
        hipLaunchKernelGGL(k_lr_gather, dim3((unsigned)((nt + 15) / 16)), dim3(256), 0, c->stream, S, d_pick,
                           c->cand_key.as<uint64_t>(), c->cand_val.as<uint64_t>());
        LDW_HIP(hipGetLastError());
    } else {
        LDW_HIP(hipMemsetAsync(d_pick, 0, sizeof(ldw::PickOut), c->stream));
    }
    // ---- the one host round trip of the block: sizes ----
    ldw::PickOut h_pick;
    int64_t h_blk_sr = 0;
    LDW_HIP(hipMemcpyAsync(&h_pick, d_pick, sizeof(h_pick), hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(&h_blk_sr, d_blk_sr, 8, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    const int64_t m = (int64_t)h_pick.n_cand;
    const int64_t sr_add = p->keep_sr ? h_blk_sr : 0;
    if (int rc = ensure_links_capacity(c, c->n_sr + sr_add, c->n_lr + m)) return rc;
    if (p->keep_sr && h_blk_sr > 0) {
        hipLaunchKernelGGL(k_sr_scatter, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, c->stream, S, d_offs, c->n_sr,
                           c->sr_a.as<int32_t>(), c->sr_b.as<int32_t>(), c->sr_mi.as<double>());
        LDW_HIP(hipGetLastError());
        c->n_sr += h_blk_sr;
    }
    if (do_lr && m > 0) {
        if (int rc = c->cand_key2.reserve((size_t)m * 8)) return rc;
        if (int rc = c->cand_val2.reserve((size_t)m * 8)) return rc;
        size_t tmp_bytes = 0;
        LDW_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, c->cand_key.as<uint64_t>(),
                                                   c->cand_key2.as<uint64_t>(), c->cand_val.as<uint64_t>(),
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        if (int rc = c->scratch.reserve(tmp_bytes)) return rc;
        LDW_REQUIRE(m < 2147483647LL, LDW_ERR_SIZE, "too many quantile candidates (%lld)", (long long)m);
        LDW_HIP(hipcub::DeviceRadixSort::SortPairs(c->scratch.p, tmp_bytes, c->cand_key.as<uint64_t>(),
                                                   c->cand_key2.as<uint64_t>(), c->cand_val.as<uint64_t>(),
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        hipLaunchKernelGGL(k_lr_thresh, dim3(1), dim3(64), 0, c->stream, c->cand_key2.as<uint64_t>(), d_pick);
        LDW_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_lr_mark, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                           c->cand_key2.as<uint64_t>(), c->cand_val2.as<uint64_t>(), d_pick,
                           c->cand_key.as<uint64_t>(), c->cand_val.as<uint64_t>(), (long long)m);
        LDW_HIP(hipGetLastError());
        LDW_HIP(hipcub::DeviceRadixSort::SortPairs(c->scratch.p, tmp_bytes, c->cand_key.as<uint64_t>(),
                                                   c->cand_key2.as<uint64_t>(), c->cand_val.as<uint64_t>(),
                                                   c->cand_val2.as<uint64_t>(), (int)m, 0, 64, c->stream));
        hipLaunchKernelGGL(k_lr_append, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                           c->cand_key2.as<uint64_t>(), c->cand_val2.as<uint64_t>(), d_pick, c->idx_f.as<int32_t>(),
                           c->idx_t.as<int32_t>(), (int)nf, d_lr_count, c->lr_a.as<int32_t>(), c->lr_b.as<int32_t>(),
                           c->lr_mi.as<double>());
        LDW_HIP(hipGetLastError());
        c->n_lr += m;  // upper bound; exact value is *d_lr_count
    } else if (do_lr) {
        hipLaunchKernelGGL(k_lr_thresh, dim3(1), dim3(64), 0, c->stream, c->cand_key.as<uint64_t>(), d_pick);
        LDW_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_block_done, dim3(1), dim3(64), 0, c->stream, d_pick, d_lr_count, h_blk_sr,
                       d_stats_i + blk_no * 3, d_stats_d + blk_no);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipEventRecord(c->ev[3], c->stream));
    LDW_HIP(hipEventSynchronize(c->ev[3]));
    float t01 = 0, t12 = 0, t23 = 0;
    LDW_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
    LDW_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
    LDW_HIP(hipEventElapsedTime(&t23, c->ev[2], c->ev[3]));
    c->last_ms[0] += t01;
    c->last_ms[1] += t12;
    c->last_ms[2] += t23;
    c->last_ms[3] += t01 + t12 + t23;
    return LDW_OK;
}

}  // namespace ldw

extern "C" {

int ldw_mi_block(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt, int quirk_mode,
                 double *MI_out, int on_device) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(from_idx && to_idx && MI_out, LDW_ERR_ARG, "ldw_mi_block: null argument");
    LDW_REQUIRE(quirk_mode == LDW_QUIRK_REFERENCE || quirk_mode == LDW_QUIRK_INTENDED, LDW_ERR_ARG, "bad quirk mode");
    BlockGeom geo;
    if (int rc = run_block_mi(c, from_idx, nf, to_idx, nt, quirk_mode, false, geo)) return rc;
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
            rc = launch_gemm(c, c->rowlist_t.as<int32_t>(), RTpad, c->rowlist_f.as<int32_t>(), RFpad, c->G.as<int64_t>(),
                             pass == 0 ? 1 : c->nlimbs, pass == 0 ? d_ones.as<int8_t>() : c->digits.as<int8_t>(),
                             c->M.as<uint8_t>(), c->Npad, 0, 0);
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

static void links_layout(ldw_ctx *c, SmallLayout &sl) {
    char *base = c->small.as<char>();
    sl.lr_count = reinterpret_cast<int64_t *>(base);
    sl.blk_sr = reinterpret_cast<int64_t *>(base + 8);
    sl.pick = reinterpret_cast<ldw::PickOut *>(base + 64);
    sl.stats_i = reinterpret_cast<int64_t *>(base + 64 + ((sizeof(ldw::PickOut) + 63) / 64) * 64);
    sl.stats_d = reinterpret_cast<double *>(sl.stats_i + c->blk_capacity * 3);
}

int ldw_links_begin(ldw_ctx *c, int64_t nblocks_capacity) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(nblocks_capacity > 0, LDW_ERR_ARG, "ldw_links_begin: capacity must be positive");
    if (int rc = ensure_rows(c)) return rc;  // uses ctx->small for staging; link bookkeeping takes it over below
    const size_t need = 64 + ((sizeof(ldw::PickOut) + 63) / 64) * 64 + (size_t)nblocks_capacity * 32 + 64;
    if (int rc = c->small.reserve(need)) return rc;
    LDW_HIP(hipMemsetAsync(c->small.p, 0, need, c->stream));
    c->n_sr = 0;
    c->n_lr = 0;
    c->stats.clear();
    c->blk_capacity = nblocks_capacity;
    c->blk_cursor = 0;
    for (int i = 0; i < 4; ++i) c->last_ms[i] = 0;
    return LDW_OK;
}

int ldw_mi_block_links(ldw_ctx *c, const int32_t *from_idx, int64_t nf, const int32_t *to_idx, int64_t nt,
                       const ldw_mi_params *p) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(from_idx && to_idx && p, LDW_ERR_ARG, "ldw_mi_block_links: null argument");
    LDW_REQUIRE(c->blk_capacity > 0 && c->blk_cursor < c->blk_capacity, LDW_ERR_STATE,
                "ldw_mi_block_links: call ldw_links_begin with enough capacity first");
    LDW_REQUIRE(p->sr_only || p->lr_links_approx > 0, LDW_ERR_ARG, "lr_links_approx must be positive");
    LDW_REQUIRE(p->quirk_mode == LDW_QUIRK_REFERENCE || p->quirk_mode == LDW_QUIRK_INTENDED, LDW_ERR_ARG, "bad quirk mode");
    SmallLayout sl;
    links_layout(c, sl);
    if (int rc = block_links(c, from_idx, nf, to_idx, nt, p, c->blk_cursor, sl.lr_count, sl.stats_i, sl.stats_d,
                             sl.pick, sl.blk_sr))
        return rc;
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
    }
    c->blk_capacity = 0;
    return LDW_OK;
}

int ldw_mi_all_pairs(ldw_ctx *c, const int32_t *blocks, int64_t nblocks, const ldw_mi_params *p, int reset) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(blocks && p && nblocks > 0, LDW_ERR_ARG, "ldw_mi_all_pairs: bad argument");
    LDW_REQUIRE(reset, LDW_ERR_ARG, "ldw_mi_all_pairs: appending to earlier calls is not supported (reset must be 1)");
    if (int rc = ldw_links_begin(c, nblocks)) return rc;
    std::vector<int32_t> fi, ti;
    for (int64_t b = 0; b < nblocks; ++b) {
        const int32_t fs = blocks[b * 4 + 0], fe = blocks[b * 4 + 1], ts = blocks[b * 4 + 2], te = blocks[b * 4 + 3];
        LDW_REQUIRE(fs >= 1 && fe >= fs && fe <= c->L && ts >= 1 && te >= ts && te <= c->L, LDW_ERR_ARG,
                    "block %lld = (%d,%d,%d,%d) outside 1..%lld", (long long)b, fs, fe, ts, te, (long long)c->L);
        fi.resize((size_t)(fe - fs + 1));
        ti.resize((size_t)(te - ts + 1));
        for (int32_t k = fs; k <= fe; ++k) fi[k - fs] = k - 1;
        for (int32_t k = ts; k <= te; ++k) ti[k - ts] = k - 1;
        if (int rc = ldw_mi_block_links(c, fi.data(), (int64_t)fi.size(), ti.data(), (int64_t)ti.size(), p)) return rc;
    }
    return ldw_links_end(c);
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
