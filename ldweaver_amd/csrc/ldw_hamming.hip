// estimate_Hamming_distance_weights (R/performPopulationStuctureCorrection.R:20-81) on the matrix cores.
//
//   shared[i][j] = #{snps a : state(a,i) == state(a,j)} = sum over the 5 states of crossprod(M_X)   (:49-74)
//   hdw[j]       = 1 / (#{i : L - shared[i][j] < thresh} + 1)                                        (:76)
//
// The five sparse x dense crossprods become ONE exact i8 GEMM: a sequence-major one-hot matrix
// H[s][5a+X] (0xFF where sequence s carries state X at SNP a) multiplied with itself over K = 5L with
// unit digits (gemm_limb_kernel<1>), accumulated in int64 over SNP chunks.
#include <algorithm>
#include <cmath>

#include "ldw_internal.h"

using namespace ldw;

namespace ldw {

// states [L][Npad] chunk (SNPs a0 .. a0+nl) -> H[s][5*(a-a0)+X], rows of Kc bytes.  64 SNPs x 64 sequences
// per workgroup through LDS so that both the read (along sequences) and the write (along 5a+X) are
// contiguous.
__global__ __launch_bounds__(256) void k_onehot_T(const uint8_t *__restrict__ states, int64_t Npad, int64_t a0,
                                                  int64_t nl, int64_t N, uint8_t *__restrict__ H, int64_t Kc) {
    __shared__ uint8_t tile[64][65];
    const int64_t la0 = (int64_t)blockIdx.x * 64, s0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {  // i: SNP in tile, tx: sequence
        const int64_t la = la0 + i, s = s0 + tx;
        tile[i][tx] = (la < nl && s < N) ? states[(a0 + la) * Npad + s] : (uint8_t)255;
    }
    __syncthreads();
    // each sequence row of the tile is 320 output bytes; 256 threads cover 64 rows x 320 B in 80 steps of 1 row x 256.. use byte loop
    for (int i = ty; i < 64; i += 4) {  // i: sequence in tile
        const int64_t s = s0 + i;
        if (s >= N) continue;
        uint8_t *dst = H + s * Kc + la0 * 5;
        for (int k = tx; k < 320; k += 64) {
            const int la = k / 5, X = k - la * 5;
            if (la0 + la < nl) dst[k] = tile[la][i] == X ? (uint8_t)0xFF : (uint8_t)0;
        }
    }
}

__global__ __launch_bounds__(256) void k_hdw(const int64_t *__restrict__ G, int ld, int64_t N, int64_t L, int thresh,
                                             double *__restrict__ hdw) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    int cnt = 0;
    for (int64_t i = lane; i < N; i += 64) cnt += ((L - G[j * ld + i]) < (int64_t)thresh) ? 1 : 0;  // shared is symmetric
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) hdw[j] = 1.0 / ((double)cnt + 1.0);
}

__global__ void k_shared_i32(const int64_t *__restrict__ G, int ld, int64_t N, int32_t *__restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < N) out[j * N + i] = (int32_t)G[j * ld + i];
}

}  // namespace ldw

extern "C" int ldw_hamming_weights(ldw_ctx *c, int32_t thresh, double *hdw_out, int32_t *shared_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(c->L > 0, LDW_ERR_STATE, "ldw_hamming_weights: set the alignment first");
    LDW_REQUIRE(hdw_out, LDW_ERR_ARG, "ldw_hamming_weights: hdw_out is null");
    const int64_t L = c->L, N = c->N, Npad = c->Npad;
    const int Rp = (int)((N + TILE - 1) / TILE * TILE);
    // SNP chunk: keep the one-hot matrix below ~2 GiB
    int64_t Lc = std::max<int64_t>(64, (int64_t)((2147483648LL / 5) / std::max<int64_t>(N, 1)) / 64 * 64);
    Lc = std::min<int64_t>(Lc, (L + 63) / 64 * 64);
    const int64_t Kc = (Lc * 5 + KSTEP - 1) / KSTEP * KSTEP;
    ldw::DevBuf H, Gh, ones, rl, dhdw;
    int rc = LDW_OK;
    auto done = [&](int code) {
        H.release(); Gh.release(); ones.release(); rl.release(); dhdw.release();
        return code;
    };
    if ((rc = H.reserve((size_t)(N + 1) * Kc)) || (rc = Gh.reserve((size_t)Rp * Rp * 8)) || (rc = ones.reserve((size_t)Kc)) ||
        (rc = rl.reserve((size_t)Rp * 4)) || (rc = dhdw.reserve((size_t)N * 8)))
        return done(rc);
    std::vector<int32_t> rowlist((size_t)Rp);
    for (int i = 0; i < Rp; ++i) rowlist[i] = i < N ? i : (int32_t)N;  // row N is the zero row
    hipError_t he;
#define HC(x) do { he = (x); if (he != hipSuccess) return done(ldw::hip_fail(he, #x, __FILE__, __LINE__)); } while (0)
    HC(hipMemcpyAsync(rl.p, rowlist.data(), (size_t)Rp * 4, hipMemcpyHostToDevice, c->stream));
    HC(hipMemsetAsync(ones.p, 1, (size_t)Kc, c->stream));
    HC(hipStreamSynchronize(c->stream));
    HC(hipEventRecord(c->ev[0], c->stream));
    int chunk = 0;
    for (int64_t a0 = 0; a0 < L; a0 += Lc, ++chunk) {
        const int64_t nl = std::min(Lc, L - a0);
        HC(hipMemsetAsync(H.p, 0, (size_t)(N + 1) * Kc, c->stream));
        dim3 grid((unsigned)((nl + 63) / 64), (unsigned)((N + 63) / 64));
        hipLaunchKernelGGL(k_onehot_T, grid, dim3(256), 0, c->stream, c->states.as<uint8_t>(), Npad, a0, nl, N,
                           H.as<uint8_t>(), Kc);
        HC(hipGetLastError());
        if ((rc = launch_gemm(c, rl.as<int32_t>(), Rp, rl.as<int32_t>(), Rp, Gh.as<int64_t>(), 1, ones.as<int8_t>(),
                              H.as<uint8_t>(), Kc, 0, chunk > 0)))
            return done(rc);
    }
    HC(hipEventRecord(c->ev[1], c->stream));
    hipLaunchKernelGGL(k_hdw, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, c->stream, Gh.as<int64_t>(), Rp, N, L,
                       (int)thresh, dhdw.as<double>());
    HC(hipGetLastError());
    HC(hipMemcpyAsync(hdw_out, dhdw.p, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    if (shared_out) {
        ldw::DevBuf s32;
        if ((rc = s32.reserve((size_t)N * N * 4))) return done(rc);
        dim3 g2((unsigned)((N + 255) / 256), (unsigned)N);
        hipLaunchKernelGGL(k_shared_i32, g2, dim3(256), 0, c->stream, Gh.as<int64_t>(), Rp, N, s32.as<int32_t>());
        he = hipMemcpyAsync(shared_out, s32.p, (size_t)N * N * 4, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        s32.release();
        if (he != hipSuccess) return done(ldw::hip_fail(he, "shared copy", __FILE__, __LINE__));
    }
    HC(hipStreamSynchronize(c->stream));
    float t = 0;
    HC(hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
    c->last_ms[0] = t;
    c->last_ms[1] = c->last_ms[2] = 0;
    c->last_ms[3] = t;
#undef HC
    return done(LDW_OK);
}
