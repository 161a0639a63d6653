// estimate_Hamming_distance_weights (R/performPopulationStuctureCorrection.R:20-81) on the matrix cores.
//
//   shared[i][j] = #{snps a : state(a,i) == state(a,j)} = sum over the 5 states of crossprod(M_X)   (:49-74)
//   hdw[j]       = 1 / (#{i : L - shared[i][j] < thresh} + 1)                                        (:76)
//
// The five sparse x dense crossprods become ONE exact bit-packed i8 GEMM over far fewer than 5L indicator columns.
// Per SNP let d be its most frequent state and A_i = [x_i != d].  Then
//     [x_i == x_j] = sum_{s != d} [x_i = s][x_j = s] + (1 - A_i)(1 - A_j)
//                  = 1 - A_i - A_j + A_i A_j + sum_{s != d} [x_i = s][x_j = s],
// so with c_i = sum_snps A_i,
//     shared[i][j] = L - c_i - c_j + sum_k w_k B_ki B_kj
// over the columns k = (SNP, minor state) with weight 1, plus one column A per SNP with weight 1; a SNP with a single
// minor state has A == that column, which then simply gets weight 2 (a biallelic SNP is ONE column, a monomorphic SNP
// none).  The GEMM is gemm_bits_kernel<1> (ldw_gemm_bits.hip) on a sequence-major bit matrix with the weights as
// its per-k digit; only tiles on or below the diagonal are computed.  Everything is integer arithmetic: bit-exact.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ldw_internal.h"
#include "ldw_prim.h"

using namespace ldw;

namespace ldw {

// Column bits, SNP-major like Mbits: Hb[k][w] bit i = predicate(states[snp(k)][64 w + i]); info = snp*16 + mode*8 + state,
// mode 0: x == state, mode 1: x != state (and x is a real state, not padding)
// r06: one thread per (column, 64-sequence word) over the FLAT index space — the r01 form gave every column a workgroup of 256 threads for its KW words
// (80 at N = 5 000: 29 of 64 lanes active, 522 k waves, SALU-bound; profiles/r06_pmc_epilogue.json).
__global__ __launch_bounds__(256) void k_hamming_cols(const uint8_t *__restrict__ states, int64_t Npad, const int32_t *__restrict__ info,
                                                      int64_t KW, int64_t KR, uint64_t *__restrict__ Hb) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= KR * KW) return;
    const int64_t k = t / KW, w = t - k * KW;
    const int32_t inf = info[k];
    const int64_t snp = inf >> 4;
    const uint32_t st = (uint32_t)(inf & 7);
    const bool neq = (inf >> 3) & 1;
    const uint4 *src = reinterpret_cast<const uint4 *>(states + snp * Npad + w * 64);
    uint64_t bits = 0;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const uint4 v4 = src[q4];
        const uint32_t xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t v = (xs[q] >> (8 * b)) & 0xFFu;
                const bool on = neq ? (v != st && v < 5u) : (v == st);
                bits |= (uint64_t)on << (16 * q4 + 4 * q + b);
            }
        }
    }
    Hb[t] = bits;
}

// r06: the column list on the DEVICE (the host loop over L SNPs + the download of the state counts + the uploads of the list were 2.8 of the call's 6.3 ms at
// config 4: profiles/r06_hamming_host_timing.txt).  k_ham_ncols: columns of every SNP — none for a monomorphic SNP, one for a biallelic SNP (the minor state,
// weight 2: it is its own "not the major state" column), otherwise one per minor state + the A column; an exclusive scan gives every SNP its first column;
// k_ham_fill writes the same records the host loop wrote, in the same order.
__device__ __forceinline__ int ham_drop(const int32_t *cnt, int &present) {
    int drop = -1;
    present = 0;
#pragma unroll
    for (int x = 0; x < 5; ++x)
        if (cnt[x] > 0) {
            ++present;
            if (drop < 0 || cnt[x] > cnt[drop]) drop = x;
        }
    return drop;
}
__global__ __launch_bounds__(256) void k_ham_ncols(const int32_t *__restrict__ counts, int64_t L, int32_t *__restrict__ ncol) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a > L) return;
    int n = 0;
    if (a < L) {
        int cnt[5], present;
#pragma unroll
        for (int x = 0; x < 5; ++x) cnt[x] = counts[a * 5 + x];
        (void)ham_drop(cnt, present);
        n = present <= 1 ? 0 : (present == 2 ? 1 : present);
    }
    ncol[a] = n;   // (ncol[L] = 0: the scan's last entry is the total)
}
__global__ __launch_bounds__(256) void k_ham_fill(const int32_t *__restrict__ counts, int64_t L, const int32_t *__restrict__ off, int32_t *__restrict__ info,
                                                  int8_t *__restrict__ digits, unsigned long long *__restrict__ umask) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= L) return;
    int cnt[5], present;
#pragma unroll
    for (int x = 0; x < 5; ++x) cnt[x] = counts[a * 5 + x];
    const int drop = ham_drop(cnt, present);
    if (present <= 1) return;
    const bool single = present == 2;
    int64_t k = off[a];
    for (int x = 0; x < 5; ++x)
        if (x != drop && cnt[x] > 0) {
            info[k] = (int32_t)(a * 16 + x);
            digits[k] = single ? 2 : 1;
            if (single) atomicOr(&umask[k >> 6], 1ull << (k & 63));
            ++k;
        }
    if (!single) {
        info[k] = (int32_t)(a * 16 + 8 + drop);
        digits[k] = 1;
        atomicOr(&umask[k >> 6], 1ull << (k & 63));
    }
}

// 64 x 64 bit-tile transpose: Hb[KR][KW] (columns x sequence words) -> T[KW*64][KWr] (sequences x column words).
// One wave per tile; ballot b collects bit b of every lane's word = the output word of sequence 64 w + b.
__global__ __launch_bounds__(256) void k_bits_transpose(const uint64_t *__restrict__ Hb, int64_t KR, int64_t KW,
                                                        uint64_t *__restrict__ T, int64_t KWr) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), w = blockIdx.y;
    if (v >= KWr) return;
    const int64_t k = v * 64 + lane;
    const uint64_t x = k < KR ? Hb[k * KW + w] : 0ull;
    uint64_t out = 0;
#pragma unroll 8
    for (int b = 0; b < 64; ++b) {
        const uint64_t m = __ballot((x >> b) & 1ull);
        if (lane == b) out = m;
    }
    T[(w * 64 + lane) * KWr + v] = out;
}

// c[i] = number of SNPs at which sequence i does not carry the SNP's most frequent state
__global__ __launch_bounds__(256) void k_seq_minor_count(const uint64_t *__restrict__ T, const uint64_t *__restrict__ umask,
                                                         int64_t rows, int64_t KWr, int32_t *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    int c = 0;
    for (int64_t v = lane; v < KWr; v += 64) c += __popcll(T[i * KWr + v] & umask[v]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) cnt[i] = c;
}

// shared(i, j) from the lower-triangular G (element (t, f) with t <= f is always inside a computed tile)
__device__ __forceinline__ int64_t shared_ij(const int64_t *__restrict__ G, int ld, const int32_t *__restrict__ cnt, int64_t L,
                                             int64_t i, int64_t j) {
    const int64_t t = i < j ? i : j, f = i < j ? j : i;
    return L - cnt[i] - cnt[j] + G[t * ld + f];
}

__global__ __launch_bounds__(256) void k_hdw(const int64_t *__restrict__ G, int ld, const int32_t *__restrict__ cnt, int64_t N,
                                             int64_t L, int thresh, double *__restrict__ hdw) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    int n = 0;
    for (int64_t i = lane; i < N; i += 64) n += ((L - shared_ij(G, ld, cnt, L, i, j)) < (int64_t)thresh) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if (lane == 0) hdw[j] = 1.0 / ((double)n + 1.0);
}

// Sharded variant: the workgroup rows t of a strip of the lower-triangular G against every f >= t.  Each unordered pair
// {t, f} lives in exactly one strip (the one that holds min(t, f)), so the strips' counts add up to k_hdw's n.
__global__ __launch_bounds__(256) void k_hdw_strip(const int64_t *__restrict__ G, int ld, const int32_t *__restrict__ cnt, int64_t N, int64_t L,
                                                   int thresh, int64_t t0, int64_t t1, int32_t *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t t = t0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= t1 || t >= N) return;
    int n = 0;
    for (int64_t f = t + lane; f < N; f += 64) {
        if ((L - shared_ij(G, ld, cnt, L, t, f)) < (int64_t)thresh) {
            ++n;
            if (f != t) atomicAdd(&counts[f], 1);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if (lane == 0 && n) atomicAdd(&counts[t], n);
}

__global__ void k_shared_i32(const int64_t *__restrict__ G, int ld, const int32_t *__restrict__ cnt, int64_t N, int64_t L,
                             int32_t *__restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < N) out[j * N + i] = (int32_t)shared_ij(G, ld, cnt, L, i, j);
}

}  // namespace ldw

// tile0 < 0: the whole matrix -> hdw_out (and shared_out); else the strip of 128-sequence row tiles [tile0, tile1) -> counts_out
static int hamming_impl(ldw_ctx *c, int32_t thresh, double *hdw_out, int32_t *shared_out, int tile0, int tile1, int64_t *counts_out) {
    if (int rc = check_gpu(c)) return rc;
    const auto wall0 = std::chrono::steady_clock::now();
    static const bool host_timing = getenv("LDW_HOST_TIMING") != nullptr;
    double t_last = 0;
    auto lap = [&](const char *what) {
        if (!host_timing) return;
        const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        fprintf(stderr, "[ldw host] hamming: %-28s %7.3f ms (+%.3f)\n", what, t, t - t_last);
        t_last = t;
    };
    LDW_REQUIRE(c->L > 0, LDW_ERR_STATE, "ldw_hamming_weights: set the alignment first");
    const bool strip = tile0 >= 0;
    LDW_REQUIRE(strip ? counts_out != nullptr : hdw_out != nullptr, LDW_ERR_ARG, "ldw_hamming_weights: output is null");
    const int64_t L = c->L, N = c->N, Npad = c->Npad, KW = c->KW;
    const int Rp = (int)Npad;  // sequences padded to the GEMM tile (Npad is a multiple of 128)
    LDW_REQUIRE(!strip || (tile0 < tile1 && tile1 <= Rp / ldw::TILE), LDW_ERR_ARG, "ldw_hamming_counts: tile range %d..%d outside 0..%d", tile0,
                tile1, Rp / ldw::TILE);
    ldw::DevBuf info, Hb, T, dig, um, Gh, rl, scnt, dhdw, tmp;
    int rc = LDW_OK;
    auto done = [&](int code) {
        // (the stream has been drained on every path that reaches this with work queued; a failed launch has queued nothing behind it)
        const bool drained = hipStreamSynchronize(c->stream) == hipSuccess;
        if (drained) {
            ldw::DrainedScope quiet;   // no device-wide synchronisation per released block (other contexts of the device may be in the middle of a pass)
            for (ldw::DevBuf *b : {&info, &Hb, &T, &dig, &um, &Gh, &rl, &scnt, &dhdw, &tmp}) b->release();
        } else {
            for (ldw::DevBuf *b : {&info, &Hb, &T, &dig, &um, &Gh, &rl, &scnt, &dhdw, &tmp}) b->release();
        }
        return code;
    };
    hipError_t he;
#define HC(x) do { he = (x); if (he != hipSuccess) return done(ldw::hip_fail(he, #x, __FILE__, __LINE__)); } while (0)
    HC(hipEventRecord(c->ev[0], c->stream));
    // per-SNP state counts -> which state is dropped, which columns exist: on the device (k_ham_ncols, a prefix sum, k_ham_fill); the host learns the column
    // count alone (one 4-byte copy) to size the bit matrices
    LDW_REQUIRE(L < (1ll << 27), LDW_ERR_ARG, "ldw_hamming_weights: too many SNPs");
    auto done2 = [&](int code) { return done(code); };
    if ((rc = ldw::launch_state_counts(c))) return done(rc);
    size_t scan_bytes = 0;
    HC(ldw::prim_exclusive_sum<int32_t>(nullptr, scan_bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)L + 1, c->stream));
    const size_t o_off = ((size_t)(L + 1) * 4 + 255) / 256 * 256, o_scan = 2 * o_off;
    if ((rc = tmp.reserve(o_scan + scan_bytes + 256))) return done(rc);
    int32_t *d_ncol = tmp.as<int32_t>(), *d_off = reinterpret_cast<int32_t *>(tmp.as<char>() + o_off);
    hipLaunchKernelGGL(k_ham_ncols, dim3((unsigned)((L + 1 + 255) / 256)), dim3(256), 0, c->stream, c->counts.as<int32_t>(), L, d_ncol);
    he = hipGetLastError();
    if (he == hipSuccess) he = ldw::prim_exclusive_sum<int32_t>(tmp.as<char>() + o_scan, scan_bytes, d_ncol, d_off, (size_t)L + 1, c->stream);
    int32_t kr32 = 0;
    if (he == hipSuccess) he = hipMemcpyAsync(&kr32, d_off + L, 4, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) return done2(ldw::hip_fail(he, "column count of the Hamming GEMM", __FILE__, __LINE__));
    lap("column count on the host");
    const int64_t KR = (int64_t)kr32;
    int64_t KWr = (KR + 63) / 64;
    KWr = std::max<int64_t>(2, (KWr + 1) / 2 * 2);  // the GEMM loads word pairs
    const int64_t Kpad = KWr * 64;
    if ((rc = info.reserve((size_t)std::max<int64_t>(KR, 1) * 4)) || (rc = Hb.reserve((size_t)std::max<int64_t>(KR, 1) * KW * 8)) ||
        (rc = T.reserve((size_t)Rp * KWr * 8)) || (rc = dig.reserve((size_t)Kpad)) || (rc = um.reserve((size_t)KWr * 8)) ||
        (rc = Gh.reserve((size_t)Rp * Rp * 8)) || (rc = rl.reserve((size_t)Rp * 4)) || (rc = scnt.reserve((size_t)Rp * 4)) ||
        (rc = dhdw.reserve((size_t)N * 8)))
        return done2(rc);
    lap("device buffers");
    he = hipMemsetAsync(dig.p, 0, (size_t)Kpad, c->stream);
    if (he == hipSuccess) he = hipMemsetAsync(um.p, 0, (size_t)KWr * 8, c->stream);
    if (he != hipSuccess) return done2(ldw::hip_fail(he, "zeroing the column weights", __FILE__, __LINE__));
    if (KR > 0) {
        hipLaunchKernelGGL(k_ham_fill, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, c->stream, c->counts.as<int32_t>(), L, d_off, info.as<int32_t>(), dig.as<int8_t>(),
                           um.as<unsigned long long>());
        he = hipGetLastError();
        if (he != hipSuccess) return done2(ldw::hip_fail(he, "k_ham_fill", __FILE__, __LINE__));
    }
    std::vector<int32_t> rowlist((size_t)Rp);
    for (int i = 0; i < Rp; ++i) rowlist[i] = i;
    HC(hipMemcpyAsync(rl.p, rowlist.data(), (size_t)Rp * 4, hipMemcpyHostToDevice, c->stream));
    lap("uploads queued");
    HC(hipEventRecord(c->ev[4], c->stream));   // (r06: the kernels in front of the GEMM bracketed on their own — ev[0] also sees the host work above)
    if (KR > 0) {
        if (KR * KW >= ((int64_t)1 << 39)) {
            ldw::set_error("ldw_hamming_weights: %lld columns x %lld words exceed the launch grid", (long long)KR, (long long)KW);
            return done(LDW_ERR_SIZE);
        }
        hipLaunchKernelGGL(k_hamming_cols, dim3((unsigned)((KR * KW + 255) / 256)), dim3(256), 0, c->stream, c->states.as<uint8_t>(), Npad, info.as<int32_t>(),
                           KW, KR, Hb.as<uint64_t>());
        HC(hipGetLastError());
    }
    hipLaunchKernelGGL(k_bits_transpose, dim3((unsigned)((KWr + 3) / 4), (unsigned)KW), dim3(256), 0, c->stream, Hb.as<uint64_t>(), KR, KW,
                       T.as<uint64_t>(), KWr);
    HC(hipGetLastError());
    hipLaunchKernelGGL(k_seq_minor_count, dim3((unsigned)((Rp + 3) / 4)), dim3(256), 0, c->stream, T.as<uint64_t>(), um.as<uint64_t>(),
                       (int64_t)Rp, KWr, scnt.as<int32_t>());
    HC(hipGetLastError());
    HC(hipEventRecord(c->ev[2], c->stream));
    if ((rc = launch_gemm_bits(c, T.as<uint64_t>(), KWr, rl.as<int32_t>(), Rp, rl.as<int32_t>(), Rp, Gh.as<int64_t>(), 1, dig.as<int8_t>(), 1,
                               nullptr, strip ? tile0 : 0, strip ? tile1 : -1)))
        return done(rc);
    HC(hipEventRecord(c->ev[1], c->stream));
    if (strip) {
        ldw::DevBuf dcnt;
        if ((rc = dcnt.reserve((size_t)N * 4))) return done(rc);
        const int64_t t0 = (int64_t)tile0 * ldw::TILE, t1 = std::min<int64_t>((int64_t)tile1 * ldw::TILE, N);
        std::vector<int32_t> hcnt((size_t)N, 0);
        he = hipMemsetAsync(dcnt.p, 0, (size_t)N * 4, c->stream);
        if (he == hipSuccess && t1 > t0) {
            hipLaunchKernelGGL(k_hdw_strip, dim3((unsigned)((t1 - t0 + 3) / 4)), dim3(256), 0, c->stream, Gh.as<int64_t>(), Rp, scnt.as<int32_t>(), N, L,
                               (int)thresh, t0, t1, dcnt.as<int32_t>());
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipMemcpyAsync(hcnt.data(), dcnt.p, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        dcnt.release();
        if (he != hipSuccess) return done(ldw::hip_fail(he, "strip counts", __FILE__, __LINE__));
        for (int64_t i = 0; i < N; ++i) counts_out[i] = hcnt[(size_t)i];
        return done(LDW_OK);
    }
    hipLaunchKernelGGL(k_hdw, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, c->stream, Gh.as<int64_t>(), Rp, scnt.as<int32_t>(), N, L,
                       (int)thresh, dhdw.as<double>());
    HC(hipGetLastError());
    HC(hipEventRecord(c->ev[3], c->stream));
    HC(hipMemcpyAsync(hdw_out, dhdw.p, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    if (shared_out) {
        ldw::DevBuf s32;
        if ((rc = s32.reserve((size_t)N * N * 4))) return done(rc);
        dim3 g2((unsigned)((N + 255) / 256), (unsigned)N);
        hipLaunchKernelGGL(k_shared_i32, g2, dim3(256), 0, c->stream, Gh.as<int64_t>(), Rp, scnt.as<int32_t>(), N, L, s32.as<int32_t>());
        he = hipMemcpyAsync(shared_out, s32.p, (size_t)N * N * 4, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        s32.release();
        if (he != hipSuccess) return done(ldw::hip_fail(he, "shared copy", __FILE__, __LINE__));
    }
    lap("kernels queued");
    HC(hipStreamSynchronize(c->stream));
    lap("stream drained");
    float t = 0, tg = 0;
    HC(hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
    HC(hipEventElapsedTime(&tg, c->ev[2], c->ev[1]));
    float tp = 0, tpre = 0;
    HC(hipEventElapsedTime(&tp, c->ev[1], c->ev[3]));
    HC(hipEventElapsedTime(&tpre, c->ev[4], c->ev[2]));
    c->last_ms[0] = tg;          // the GEMM alone
    c->last_ms[1] = tp;          // r06: the N x N neighbour count (k_hdw)
    c->last_ms[2] = 0;
    c->last_ms[3] = t;           // counts + column bits + transpose + GEMM
    // r06: what the stage moved, for its roofline on the bench line (ldw_hamming_stats).  Algorithmic bytes of the kernels around the
    // GEMM: k_hamming_cols reads its SNP's state row once per COLUMN and writes the column's bits, k_bits_transpose reads and writes the bit matrix,
    // k_seq_minor_count reads it once more; k_hdw reads the lower triangle of G twice (every (i, j) from row min(i, j)) and writes N doubles.
    c->ham_stat[0] = (double)KR;
    c->ham_stat[1] = (double)Kpad;
    c->ham_stat[2] = (double)tpre;
    c->ham_stat[3] = (double)tg;
    c->ham_stat[4] = (double)tp;
    c->ham_stat[5] = (double)KR * (double)Npad + 2.0 * (double)KR * (double)KW * 8.0 + 2.0 * (double)Rp * (double)KWr * 8.0;
    c->ham_stat[6] = (double)N * (double)N * 8.0 + (double)N * 8.0;
    c->ham_stat[7] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
#undef HC
    return done(LDW_OK);
}

extern "C" int ldw_hamming_weights(ldw_ctx *c, int32_t thresh, double *hdw_out, int32_t *shared_out) {
    return hamming_impl(c, thresh, hdw_out, shared_out, -1, -1, nullptr);
}

extern "C" int ldw_hamming_stats(ldw_ctx *c, double out[8]) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_hamming_stats: null argument");
    for (int k = 0; k < 8; ++k) out[k] = c->ham_stat[k];
    return LDW_OK;
}

extern "C" int ldw_hamming_counts(ldw_ctx *c, int32_t thresh, int32_t tile0, int32_t tile1, int64_t *counts_out) {
    LDW_REQUIRE(tile0 >= 0, LDW_ERR_ARG, "ldw_hamming_counts: tile0 must be >= 0");
    return hamming_impl(c, thresh, nullptr, nullptr, tile0, tile1, counts_out);
}
