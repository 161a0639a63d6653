// The histogram formulation of the path (north_star's primary form; reference formulation R/computePairwiseMI.R:390-398 +
// src/computeMI.cpp:19), two kernels:
//
//   LDW_ENGINE_HIST         k_cooc_popc — the joint histogram on BIT PLANES (below, second half of this file): every cell of every
//                           pair's weighted joint table is sum over weight classes of V_class * popcount(x & y & class mask),
//                           tiled through LDS, exact int64; then the same fp64 epilogue as the MFMA engine.
//   LDW_ENGINE_HIST_STATES  k_mi_hist — the first version: byte states staged in LDS, 25 fixed-point sums per pair updated one
//                           sequence at a time.  Kept as the independent on-device cross-check (it shares nothing with the other
//                           engines but the quantised weights) and as the record of what the bit planes buy (233x, docs/HISTORY.md 5.5).
//
// k_mi_hist: the per-pair 5x5 Hamming-weighted joint histogram kernel (VALU + LDS).
//
// The state matrix is tiled into LDS (64 from-side SNPs x 256 sequences, transposed so that the 64 lanes
// of a wave read consecutive bytes); every thread owns one SNP pair at a time and keeps its 25 exact
// fixed-point joint sums in a private LDS column (hist[code][thread], conflict-free 8-byte accesses).
// The MI epilogue then walks the 25 cells in the reference's own order (X outer, Y inner,
// R/computePairwiseMI.R:270-298) with the formula of src/computeMI.cpp:19.
//
// It shares nothing with the MFMA path except the quantised weights, which makes it the on-device
// cross-check of that path and the baseline BASELINE.json asks the GEMM formulation to beat.
#include <cmath>

#include "ldw_internal.h"

using namespace ldw;

namespace ldw {

#ifdef LDW_EXPERIMENTS   // (k_mi_hist, LDW_ENGINE_HIST_STATES: the r01 byte-state kernel, kept as the record and an independent cross-check)
constexpr int HS = 256;  // sequences per LDS chunk
constexpr int HB = 4;    // to-side SNPs handled one after the other by each thread

struct HistArgs {
    const uint8_t *states;
    int64_t Npad, N;
    const int64_t *vfixed;
    const int32_t *idx_f, *idx_t;
    int nf, nt;
    const int64_t *pfix_state;  // [L][5] by state
    const uint8_t *uqe;         // [L][5]
    const double *r;
    double neff, scale;
    int quirk, lower_only;
    double *MI;
};

__global__ __launch_bounds__(256) void k_mi_hist(HistArgs A) {
    __shared__ unsigned long long hist[25][256];
    __shared__ uint8_t sF[HS][64];
    __shared__ uint8_t sT[4][HS];
    __shared__ unsigned long long sV[HS];
    const int tid = threadIdx.x;
    const int al = tid & 63, bl = tid >> 6;
    const int a_loc = blockIdx.x * 64 + al;
    const int a_clamped = a_loc < A.nf ? a_loc : A.nf - 1;
    const int sa = A.idx_f[a_clamped];

    for (int kb = 0; kb < HB; ++kb) {
        const int b_loc = (blockIdx.y * HB + kb) * 4 + bl;
        const int b_clamped = b_loc < A.nt ? b_loc : A.nt - 1;
        const int sb = A.idx_t[b_clamped];
        if ((blockIdx.y * HB + kb) * 4 >= A.nt) break;  // uniform over the workgroup
#pragma unroll
        for (int c = 0; c < 25; ++c) hist[c][tid] = 0ull;
        for (int64_t s0 = 0; s0 < A.N; s0 += HS) {
            __syncthreads();
            // from-tile, transposed: 64 SNPs x 64 dwords
            for (int i = 0; i < 16; ++i) {
                const int idx = tid + 256 * i;
                const int a = idx >> 6, q = idx & 63;
                const int arow = blockIdx.x * 64 + a;
                uint32_t w = 0xFFFFFFFFu;
                if (arow < A.nf && s0 + 4 * q < A.Npad)
                    w = *reinterpret_cast<const uint32_t *>(A.states + (int64_t)A.idx_f[arow] * A.Npad + s0 + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) sF[4 * q + k][a] = (uint8_t)(w >> (8 * k));
            }
            // to-tile: 4 SNPs x 64 dwords (one per thread)
            {
                const int b = tid >> 6, q = tid & 63;
                const int brow = (blockIdx.y * HB + kb) * 4 + b;
                uint32_t w = 0xFFFFFFFFu;
                if (brow < A.nt && s0 + 4 * q < A.Npad)
                    w = *reinterpret_cast<const uint32_t *>(A.states + (int64_t)A.idx_t[brow] * A.Npad + s0 + 4 * q);
                *reinterpret_cast<uint32_t *>(&sT[b][4 * q]) = w;
            }
            sV[tid] = (s0 + tid < A.Npad) ? (unsigned long long)A.vfixed[s0 + tid] : 0ull;
            __syncthreads();
            const int ns = (int)((A.N - s0) < HS ? (A.N - s0) : HS);
            for (int s = 0; s < ns; ++s) {
                const unsigned x = sF[s][al], y = sT[bl][s];
                if (x < 5u && y < 5u) hist[x * 5 + y][tid] += sV[s];
            }
        }
        if (a_loc < A.nf && b_loc < A.nt && !(A.lower_only && a_loc <= b_loc)) {
            const double ra = A.r[sa], rb = A.r[sb];
            const double den = A.neff + (ra * rb) * 0.5;
            double RXY;
            if (A.quirk == LDW_QUIRK_REFERENCE) {
                const int64_t c = (int64_t)a_loc + (int64_t)b_loc * A.nf;
                RXY = (A.r[A.idx_f[c / A.nt]] * A.r[A.idx_t[c % A.nt]]) * 0.25;
            } else {
                RXY = (ra * rb) * 0.25;
            }
            const double rX = 0.5 * ra, rY = 0.5 * rb;
            double mi = 0.0;
            for (int X = 0; X < 5; ++X) {
                if (!A.uqe[(int64_t)sa * 5 + X]) continue;
                const double pX = (double)A.pfix_state[(int64_t)sa * 5 + X] * A.scale;
                for (int Y = 0; Y < 5; ++Y) {
                    if (!A.uqe[(int64_t)sb * 5 + Y]) continue;
                    const double pY = (double)A.pfix_state[(int64_t)sb * 5 + Y] * A.scale;
                    const double pxy = (double)(long long)hist[X * 5 + Y][tid] * A.scale + 0.5;
                    const double d = ((pX * pY + RXY) + pX * rX) + pY * rY;
                    mi += (pxy / den) * log((pxy / d) * den);
                }
            }
            A.MI[(int64_t)a_loc + (int64_t)b_loc * A.nf] = mi;
        }
    }
}

int launch_hist(ldw_ctx *c, const int32_t *idx_f, int nf, const int32_t *idx_t, int nt, const int64_t *pfix_state,
                int quirk, int lower_only, double *MI) {
    HistArgs A;
    A.states = c->states.as<uint8_t>();
    A.Npad = c->Npad;
    A.N = c->N;
    A.vfixed = c->vfixed.as<int64_t>();
    A.idx_f = idx_f;
    A.idx_t = idx_t;
    A.nf = nf;
    A.nt = nt;
    A.pfix_state = pfix_state;
    A.uqe = c->uqe.as<uint8_t>();
    A.r = c->r.as<double>();
    A.neff = c->neff;
    A.scale = std::ldexp(1.0, -c->frac_bits);
    A.quirk = quirk;
    A.lower_only = lower_only;
    A.MI = MI;
    dim3 grid((unsigned)((nf + 63) / 64), (unsigned)((nt + 4 * HB - 1) / (4 * HB)));
    LDW_REQUIRE(grid.y <= 65535u, LDW_ERR_ARG, "nt too large for the histogram grid");
    hipLaunchKernelGGL(k_mi_hist, grid, dim3(256), 0, c->stream, A);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

#endif   // LDW_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// k_cooc_popc: G[t][f] = sum_s V_s [row t has s][row f has s], exact, by popcounts over the weight classes.
//
// Sequences are sorted by weight (ldw_set_weights), so a weight class is a contiguous bit range and the position axis splits into
// (32-bit word, class) segments (PopSeg, built once per weighting: C4 has 57 classes, 160 words, 211 segments).  For one row pair
// and one word: n = popcount(x & y [& segment mask]) accumulates into the class count with the popcount instruction's own add
// operand; where a class ends, the count is folded into three 32-bit sums with the class weight's three 16-bit limbs
// (v_mad_u32_u24, full rate; sum_c n_c limb_c <= N * 65535 < 2^32 for N <= 65535), and the int64 result is assembled once at the
// end: the same integers gemm_bits_kernel<5> produces, so the fp64 epilogue, the link selection and every test are shared.
//
// Workgroup = 256 threads, tile 64 to-rows x 64 from-rows of the block's row lists, thread = 4 x 4 row pairs: per word two
// ds_read_b128 (4 consecutive rows of a word are one 16-byte piece: LDS image [word][row], rows padded to 68 so that the
// transposing staging writes spread over the banks) feed 16 ANDs + 16 popcounts.  K is walked in chunks of 32 words (128 B per row:
// one coalesced line), the segment records are wave-uniform (scalar loads).  Diagonal blocks skip the tiles above the diagonal.
// Bound: VALU issue (2 ops per row pair and word + 3 per row pair and class).
// ------------------------------------------------------------------------------------------------
struct PopSegH {      // == ldw::PopSeg of ldw_apx.h (kept local: this file shares no header with the approximate path)
    uint32_t mask, flush;
    int64_t V;
};

struct CoocArgs {
    const uint64_t *Mbits;
    int64_t KW;                  // 64-bit words per bit row
    const int32_t *rowlist_t, *rowlist_f;
    int RTpad, RFpad, nwords;    // nwords: 32-bit words that hold sequences (<= 2 KW)
    const PopSegH *segs;
    const int32_t *wbeg;         // [nwords + 1] first segment of every word (bit 31: flag of another kernel, masked off)
    int64_t *G;
    int lower_only;
};

constexpr int PC_T = 64;        // rows per side of a workgroup tile
constexpr int PC_WC = 32;       // 32-bit words per staged chunk
constexpr int PC_LD = PC_T + 4; // padded row count of the LDS image [word][row]

__global__ __launch_bounds__(256) void k_cooc_popc(CoocArgs P) {
    __shared__ __attribute__((aligned(16))) uint32_t sT[PC_WC][PC_LD], sF[PC_WC][PC_LD];
    const int tid = threadIdx.x;
    const int ty = blockIdx.y, tx = blockIdx.x;
    if (P.lower_only && tx * PC_T + PC_T - 1 < ty * PC_T) return;   // (same rule as the GEMM: tiles strictly above the diagonal)
    const int ti = tid >> 4, fi = tid & 15;      // 16 x 16 threads, 4 x 4 row pairs each
    uint32_t cnt[4][4], s0[4][4], s1[4][4], s2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cnt[i][j] = s0[i][j] = s1[i][j] = s2[i][j] = 0u;
    // staging roles: 8 threads x 16 B cover the 128 B of one row's chunk; 32 rows per pass, two passes per side
    const int srow = tid >> 3, spiece = tid & 7;
    const uint32_t *rowT[2], *rowF[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        rowT[h] = reinterpret_cast<const uint32_t *>(P.Mbits + (int64_t)P.rowlist_t[ty * PC_T + srow + 32 * h] * P.KW);
        rowF[h] = reinterpret_cast<const uint32_t *>(P.Mbits + (int64_t)P.rowlist_f[tx * PC_T + srow + 32 * h] * P.KW);
    }
    const int nw_alloc = (int)(P.KW * 2);
    for (int w0 = 0; w0 < P.nwords; w0 += PC_WC) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 vt = make_uint4(0, 0, 0, 0), vf = make_uint4(0, 0, 0, 0);
            if (w0 + 4 * spiece < nw_alloc) {
                vt = *reinterpret_cast<const uint4 *>(rowT[h] + w0 + 4 * spiece);
                vf = *reinterpret_cast<const uint4 *>(rowF[h] + w0 + 4 * spiece);
            }
            const int r = srow + 32 * h;
            sT[4 * spiece + 0][r] = vt.x; sT[4 * spiece + 1][r] = vt.y; sT[4 * spiece + 2][r] = vt.z; sT[4 * spiece + 3][r] = vt.w;
            sF[4 * spiece + 0][r] = vf.x; sF[4 * spiece + 1][r] = vf.y; sF[4 * spiece + 2][r] = vf.z; sF[4 * spiece + 3][r] = vf.w;
        }
        __syncthreads();
        const int wn = P.nwords - w0 < PC_WC ? P.nwords - w0 : PC_WC;
        for (int w = 0; w < wn; ++w) {
            const uint4 xv = *reinterpret_cast<const uint4 *>(&sT[w][4 * ti]);
            const uint4 yv = *reinterpret_cast<const uint4 *>(&sF[w][4 * fi]);
            const uint32_t x[4] = {xv.x, xv.y, xv.z, xv.w}, y[4] = {yv.x, yv.y, yv.z, yv.w};
            uint32_t xy[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) xy[i][j] = x[i] & y[j];
            const int sg0 = P.wbeg[w0 + w] & 0x7FFFFFFF, sg1 = P.wbeg[w0 + w + 1] & 0x7FFFFFFF;   // wave-uniform
            for (int sg = sg0; sg < sg1; ++sg) {
                const PopSegH seg = P.segs[sg];
                if (seg.mask == 0xFFFFFFFFu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) cnt[i][j] += (uint32_t)__popc(xy[i][j]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) cnt[i][j] += (uint32_t)__popc(xy[i][j] & seg.mask);
                }
                if (seg.flush) {   // the class ends here: fold its count in with the three 16-bit limbs of its weight
                    const uint32_t l0 = (uint32_t)(seg.V & 0xFFFF), l1 = (uint32_t)((seg.V >> 16) & 0xFFFF), l2 = (uint32_t)((seg.V >> 32) & 0xFFFF);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            s0[i][j] = __umul24(cnt[i][j], l0) + s0[i][j];
                            s1[i][j] = __umul24(cnt[i][j], l1) + s1[i][j];
                            s2[i][j] = __umul24(cnt[i][j], l2) + s2[i][j];
                            cnt[i][j] = 0u;
                        }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int trow = ty * PC_T + 4 * ti + i;
        longlong2 o[2];
        int64_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (int64_t)s0[i][j] + ((int64_t)s1[i][j] << 16) + ((int64_t)s2[i][j] << 32);
        o[0].x = v[0]; o[0].y = v[1]; o[1].x = v[2]; o[1].y = v[3];
        longlong2 *dst = reinterpret_cast<longlong2 *>(P.G + (int64_t)trow * P.RFpad + tx * PC_T + 4 * fi);
        dst[0] = o[0];
        dst[1] = o[1];
    }
}

// exact joint sums of a block by class-wise popcounts: same contract as launch_gemm_bits with all limbs (G[t][f], int64)
int launch_cooc_popc(ldw_ctx *c, const int32_t *rowlist_t, int RTpad, const int32_t *rowlist_f, int RFpad, int64_t *G, int lower_only, hipStream_t st) {
    LDW_REQUIRE(RTpad % PC_T == 0 && RFpad % PC_T == 0, LDW_ERR_ARG, "launch_cooc_popc: row lists must be padded to %d", PC_T);
    LDW_REQUIRE(c->N <= 65535, LDW_ERR_ARG, "the popcount engine's 32-bit limb sums hold at most 65535 sequences (%lld)", (long long)c->N);
    LDW_REQUIRE(c->nlimbs <= 6, LDW_ERR_ARG, "the popcount engine folds weights of at most 48 bits (nlimbs <= 6)");
    LDW_REQUIRE(c->pop_segs.p && c->pop_wbeg.p, LDW_ERR_STATE, "weights not set (no popcount segments)");
    CoocArgs P;
    P.Mbits = c->Mbits.as<uint64_t>();
    P.KW = c->KW;
    P.rowlist_t = rowlist_t;
    P.rowlist_f = rowlist_f;
    P.RTpad = RTpad;
    P.RFpad = RFpad;
    P.nwords = (int)((c->N + 31) / 32);
    P.segs = c->pop_segs.as<PopSegH>();
    P.wbeg = c->pop_wbeg.as<int32_t>();
    P.G = G;
    P.lower_only = lower_only;
    dim3 grid((unsigned)(RFpad / PC_T), (unsigned)(RTpad / PC_T));
    LDW_REQUIRE(grid.y <= 65535u, LDW_ERR_ARG, "too many row tiles for the popcount grid");
    hipLaunchKernelGGL(k_cooc_popc, grid, dim3(256), 0, st ? st : c->stream, P);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

}  // namespace ldw
