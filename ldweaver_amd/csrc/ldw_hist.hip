// LDW_ENGINE_HIST: the per-pair 5x5 Hamming-weighted joint histogram kernel (VALU + LDS).
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

}  // namespace ldw
