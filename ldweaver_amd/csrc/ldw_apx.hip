// Approximate-GEMM path of the speculative blocks: see ldw_apx.h for the formulation.
//   prepare_apx_weights  host: dual digits a, b and block exponents of the weights, popcount segments of the weight classes
//   k_pack_panel         per block side: bit rows of the row list, transposed to [macro step][row][2 words]
//   gemm_apx_kernel      one int8 MFMA pass, both operands masked digits -> int32 approximate joint sums (screen input)
//   k_pair_sums / k_pair_mi  exact joint sums of the listed candidate pairs by class-wise popcounts, fp64 MI, emission
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ldw_apx.h"
#include "ldw_dev.h"

using namespace ldw;

namespace ldw {

// ------------------------------------------------------------------------------------------------
// host: weights -> (a, b, e) and popcount segments
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int64_t APX_PROD_CAP = 12000;
struct ProdTable {
    std::vector<int32_t> val;
    std::vector<uint8_t> a, b;
    ProdTable() {
        std::vector<std::pair<int32_t, std::pair<uint8_t, uint8_t>>> all;
        for (int x = 0; x <= 127; ++x)
            for (int y = x; y <= 127; ++y) all.push_back({x * y, {(uint8_t)x, (uint8_t)y}});
        std::sort(all.begin(), all.end());
        for (auto &e : all)
            if (val.empty() || val.back() != e.first) {
                val.push_back(e.first);
                a.push_back(e.second.first);
                b.push_back(e.second.second);
            }
    }
    // index of the product nearest to t (0 <= t <= 16129)
    size_t nearest(double t) const {
        size_t hi = std::lower_bound(val.begin(), val.end(), (int32_t)std::ceil(t)) - val.begin();
        if (hi >= val.size()) hi = val.size() - 1;
        size_t lo = hi > 0 ? hi - 1 : 0;
        return (t - (double)val[lo]) <= ((double)val[hi] - t) ? lo : hi;
    }
};
}  // namespace

int prepare_apx_weights(ldw_ctx *c) {
    static const ProdTable PT;
    const int64_t N = c->N, Npad = c->Npad;
    const int M2 = (int)(Npad / 128);
    c->apx_ok = false;
    std::vector<int64_t> Vp((size_t)Npad, 0);
    for (int64_t p = 0; p < N; ++p) Vp[(size_t)p] = c->h_vfixed[(size_t)c->h_seq_perm[(size_t)p]];
    // block exponents, one per MFMA k-step of 32 consecutive positions (r03; r02 had one per macro step of 128, which put weights
    // 1/128 .. 1 of an alignment with N distinct weights under ONE exponent: delta 6e-3, path off): smallest non-decreasing e(k)
    // with ceil(Vmax(k) / 2^e) <= APX_PROD_CAP.  Products of two 7-bit digits are dense up to ~12000 (worst relative half-gap
    // 1.0e-3 over [6000, 12000], 1.2e-3 over [3000, 12000]) and sparse above (15750 -> 15875: 4e-3), so the largest weight of a
    // step is mapped below the sparse region
    const int S4 = 4 * M2;
    std::vector<int32_t> em((size_t)S4, 0), sh((size_t)S4, 0);
    std::vector<uint8_t> da((size_t)Npad, 0), db((size_t)Npad, 0);
    double delta = 0;
    int transitions = 0;
    // gran = 4: one exponent per macro step of 128 positions (the faster kernel variant); gran = 1: one per k-step.  The coarse form
    // is kept whenever it is tight enough (delta <= 1.5e-3: the screen's margin grows with delta)
    auto assign = [&](int gran) {
        std::fill(sh.begin(), sh.end(), 0);
        std::fill(da.begin(), da.end(), 0);
        std::fill(db.begin(), db.end(), 0);
        c->h_vapx.assign((size_t)Npad, 0);
        int e_prev = 0;
        transitions = 0;
        for (int k0 = 0; k0 < S4; k0 += gran) {
            int64_t vmax = 0;
            for (int q = 0; q < 32 * gran; ++q) vmax = std::max(vmax, Vp[(size_t)k0 * 32 + q]);
            int e = e_prev;
            while (e < 62 && ((vmax + ((int64_t)1 << e) - 1) >> e) > APX_PROD_CAP) ++e;
            for (int k = k0; k < k0 + gran; ++k) em[(size_t)k] = e;
            if (k0 > 0 && e > e_prev) {
                sh[(size_t)k0] = std::min(e - e_prev, 31);
                ++transitions;
            }
            e_prev = e;
        }
        delta = 0;
        for (int64_t p = 0; p < N; ++p) {
            const int64_t V = Vp[(size_t)p];
            if (V <= 0) continue;
            const int e = em[(size_t)(p / 32)];
            const size_t k = PT.nearest(std::ldexp((double)V, -e));
            da[(size_t)p] = PT.a[k];
            db[(size_t)p] = PT.b[k];
            const int64_t Va = (int64_t)PT.val[k] << e;
            c->h_vapx[(size_t)c->h_seq_perm[(size_t)p]] = Va;
            delta = std::max(delta, std::fabs((double)(Va - V)) / (double)V);
        }
    };
    static const int force_gran = [] { const char *e = exp_env("LDW_APX_GRAN"); return e ? atoi(e) : 0; }();   // 1 / 4: force fine / coarse (A/B)
    assign(force_gran == 1 ? 1 : 4);
    c->apx_fine = force_gran == 1;
    if (force_gran == 0 && delta > 1.5e-3) {
        assign(1);
        c->apx_fine = true;
    }
    c->apx_delta = delta;
    c->apx_e_last = S4 > 0 ? em[(size_t)S4 - 1] : 0;
    c->apx_transitions = transitions;
    // units of 2^e_last a GEMM entry can have lost to the truncations: < 1 unit of 2^e(m) at every transition into macro step m
    c->apx_lost_units = 0;
    for (int k = 1; k < S4; ++k)
        if (sh[(size_t)k] > 0) c->apx_lost_units += std::ldexp(1.0, em[(size_t)k] - c->apx_e_last);
    // (experiments build, a PRICING switch: LDW_APX_EXTRA_UNITS=x treats every GEMM entry as up to x WEIGHT units low on top of that — what an absolute
    // slack per entry, e.g. of a contraction over compressed clone groups (docs/HISTORY.md 10), would cost the screen in listed pairs; results stay exact)
    if (const char *xs = exp_env("LDW_APX_EXTRA_UNITS")) {
        const double x = atof(xs);
        if (x > 0) c->apx_lost_units += x / std::ldexp(1.0, c->apx_e_last - c->frac_bits);
    }
    // weight classes = runs of equal V along the positions; segments = (32-bit word, class) intersections
    std::vector<PopSeg> segs;
    std::vector<int32_t> wbeg((size_t)(Npad / 32) + 1, 0);
    std::vector<std::vector<PopSeg>> per_word((size_t)(Npad / 32));
    int n_classes = 0;
    for (int64_t p0 = 0; p0 < N;) {
        int64_t p1 = p0 + 1;
        while (p1 < N && Vp[(size_t)p1] == Vp[(size_t)p0]) ++p1;
        if (Vp[(size_t)p0] > 0) {
            ++n_classes;
            for (int64_t w = p0 / 32; w <= (p1 - 1) / 32; ++w) {
                const int64_t lo = std::max<int64_t>(p0, w * 32) - w * 32, hi = std::min<int64_t>(p1, w * 32 + 32) - w * 32;   // bits [lo, hi)
                const uint32_t mask = (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
                per_word[(size_t)w].push_back(PopSeg{mask, w == (p1 - 1) / 32 ? 1u : 0u, Vp[(size_t)p0]});
            }
        }
        p0 = p1;
    }
    for (size_t w = 0; w < per_word.size(); ++w) {
        wbeg[w] = (int32_t)segs.size();
        // bit 31: the word is ONE full segment that does not end its class (k_units_pop reads no segment record for it)
        if (per_word[w].size() == 1 && per_word[w][0].mask == 0xFFFFFFFFu && !per_word[w][0].flush) wbeg[w] |= (int32_t)0x80000000;
        for (auto &s : per_word[w]) segs.push_back(s);
    }
    wbeg[per_word.size()] = (int32_t)segs.size();
    c->n_pop_segs = (int)segs.size();
    c->n_classes = n_classes;
    if (segs.empty()) segs.push_back(PopSeg{0, 0, 0});
    if (int rc = c->dig_a.reserve((size_t)Npad)) return rc;
    if (int rc = c->dig_b.reserve((size_t)Npad)) return rc;
    if (int rc = c->apx_shift.reserve((size_t)std::max(S4, 4) * 4)) return rc;
    if (int rc = c->pop_segs.reserve(segs.size() * sizeof(PopSeg))) return rc;
    if (int rc = c->pop_wbeg.reserve(wbeg.size() * 4)) return rc;
    {   // r05: the fixed-point weight of every POSITION (0 in the padding) for the bit-walking pair sums of weightings with many classes (k_pair_sums_bits)
        std::vector<int64_t> vpos((size_t)Npad, 0);
        for (int64_t q = 0; q < N; ++q) vpos[(size_t)q] = Vp[(size_t)q];
        if (int rc = c->pop_vpos.reserve((size_t)Npad * 8)) return rc;
        LDW_HIP(hipMemcpyAsync(c->pop_vpos.p, vpos.data(), (size_t)Npad * 8, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));   // (vpos is on this frame)
    }
    LDW_HIP(hipMemcpyAsync(c->dig_a.p, da.data(), (size_t)Npad, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->dig_b.p, db.data(), (size_t)Npad, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->apx_shift.p, sh.data(), (size_t)S4 * 4, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->pop_segs.p, segs.data(), segs.size() * sizeof(PopSeg), hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->pop_wbeg.p, wbeg.data(), wbeg.size() * 4, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    // The path pays when the approximation is tight enough for the screen to dismiss almost everything (its margin grows with
    // delta); the digit arrays of the GEMM must fit in LDS.  Any weights qualify: with many distinct values the popcount sums of
    // the listed pairs walk more segments per word, which is still cheap for the few pairs that are listed.
    // (r03 also required the popcount segment tables of k_pair_sums to fit 60 000 B of LDS, which switched the path off for ~3700 and more
    // distinct weights; the kernel now reads larger tables from global memory)
    c->apx_ok = delta <= 4e-3 && Npad <= 30720 && M2 > 0;
    {
        char why[160];
        if (c->apx_ok) snprintf(why, sizeof(why), c->apx_fine ? "ok (block exponents per 32 positions)" : "ok");
        else if (!(delta <= 4e-3)) snprintf(why, sizeof(why), "delta %.3g > 4e-03 (dual-digit weights too coarse)", delta);
        else if (Npad > 30720) snprintf(why, sizeof(why), "Npad %lld > 30720 (digit arrays exceed the GEMM's LDS)", (long long)Npad);
        else snprintf(why, sizeof(why), "no macro step (Npad %lld)", (long long)Npad);
        c->apx_gate = why;
    }
    return LDW_OK;
}

// ------------------------------------------------------------------------------------------------
// k_pack_panel: the 128 bits of macro step m of row-list position r as ONE 16-byte piece panel[m][r] (consecutive rows are
// consecutive pieces: the GEMM's loads are contiguous runs).  interleave != 0 (gemm_apx_kernel, one block exponent per k-step): the eight
// 16-bit groups g of the macro step are stored as word 0 = groups (0, 2, 4, 6), word 1 = groups (1, 3, 5, 7), so that the lane half fh of an
// MFMA k-step kk finds ITS sixteen positions 32 kk + 16 fh .. + 15 — a k-step covers 32 CONSECUTIVE positions, one block exponent —
// at bits 16 kk of word fh.  interleave == 0 (gemm_apx_lds_kernel; gemm_apx_kernel with one exponent per macro step): the words as they are.
// scaled != 0 (r06, gemm_apx_kernel): every word goes out as FOUR dwords, dword kk = the word's bytes 2 kk and 2 kk + 1 each multiplied by 8 in
// a 16-bit field — the byte offsets of the two entries of the kernel's 8-byte expansion table that k-step kk looks up: the kernel gets an offset
// with ONE VALU instruction (and / shift) instead of two (extract, then scale); piece ((m Rpad + r) 2 + word) of 16 bytes, a panel twice the size.
// ------------------------------------------------------------------------------------------------
// blockIdx.y = 1: the second row list of the launch (rowlist2 / Rpad2 / panel2: the to side of an off-diagonal block — both panels in ONE launch).
__global__ __launch_bounds__(256) void k_pack_panel(const uint64_t *__restrict__ Mbits, int64_t KW, const int32_t *__restrict__ rowlist,
                                                    int Rpad, int M2, uint64_t *__restrict__ panel, int interleave, int scaled,
                                                    const int32_t *__restrict__ rowlist2 = nullptr, int Rpad2 = 0, uint64_t *__restrict__ panel2 = nullptr) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (blockIdx.y == 1) {
        rowlist = rowlist2;
        Rpad = Rpad2;
        panel = panel2;
    }
    const int r = blockIdx.x * 64 + (threadIdx.x & 63);
    if (r >= Rpad) return;
    const u64x2 *src = reinterpret_cast<const u64x2 *>(Mbits + (int64_t)rowlist[r] * KW);
    u64x2 *dst = reinterpret_cast<u64x2 *>(panel);
    u32x4 *dst4 = reinterpret_cast<u32x4 *>(panel);
    for (int m = threadIdx.x >> 6; m < M2; m += 4) {
        u64x2 v = src[m];
        if (interleave) {
            const unsigned long long x = v[0], y = v[1];
            // groups of x: g0 g1 g2 g3 (16 bits each, g0 lowest), of y: g4 g5 g6 g7
            const unsigned long long ex = (x & 0xFFFFull) | ((x >> 16) & 0xFFFF0000ull) | ((y & 0xFFFFull) << 32) | ((y << 16) & 0xFFFF000000000000ull);
            const unsigned long long ox = ((x >> 16) & 0xFFFFull) | ((x >> 32) & 0xFFFF0000ull) | (((y >> 16) & 0xFFFFull) << 32) | (y & 0xFFFF000000000000ull);
            v[0] = ex;   // g0 g2 g4 g6
            v[1] = ox;   // g1 g3 g5 g7
        }
        if (scaled) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned long long w = v[h];
                u32x4 o;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    o[kk] = (((unsigned int)(w >> (16 * kk)) & 0xFFu) << 3) | (((unsigned int)(w >> (16 * kk + 8)) & 0xFFu) << 19);
                dst4[((int64_t)m * Rpad + r) * 2 + h] = o;
            }
        } else {
            dst[(int64_t)m * Rpad + r] = v;
        }
    }
}

static bool apx_kernel_is_lds() {
    // "reg" (default): operands expanded in registers per wave; "lds": expansion shared through LDS (r03 experiment: correct, and
    // 28 % slower — 0.659 vs 0.515 ms per C4 launch — because the fragment reads + table reads + tile writes make it LDS-bound)
    static const bool lds = [] {
        const char *e = exp_env("LDW_APX_KERNEL");
        return e && e[0] == 'l';
    }();
    return lds;
}

// the register-expansion kernel (gemm_apx_kernel: the production one) reads the SCALED panel, the experiment kernels the plain words
static bool apx_kernel_is_reg() {
    static const bool reg = [] {
        const char *e = exp_env("LDW_APX_KERNEL");
        return !(e && (e[0] == 'l' || e[0] == 'p'));
    }();
    return reg;
}

int launch_pack_panel(ldw_ctx *c, const int32_t *rowlist, int Rpad, uint64_t *panel, hipStream_t st, const int32_t *rowlist2, int Rpad2, uint64_t *panel2) {
    const int rmax = (rowlist2 && Rpad2 > Rpad) ? Rpad2 : Rpad;
    hipLaunchKernelGGL(k_pack_panel, dim3((unsigned)((rmax + 63) / 64), rowlist2 ? 2u : 1u), dim3(256), 0, st, c->Mbits.as<uint64_t>(), c->KW, rowlist, Rpad,
                       (int)(c->KW / 2), panel, (c->apx_fine && !apx_kernel_is_lds()) ? 1 : 0, apx_kernel_is_reg() ? 1 : 0, rowlist2, Rpad2, panel2);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

// ------------------------------------------------------------------------------------------------
// gemm_apx_kernel<MT, NT>: G'[t][f] = sum_p [row t has bit p][row f has bit p] a_p b_p 2^(e(p) - e_last), truncated at the
// exponent transitions (each loses < 1 unit).  256 threads = 4 independent waves (2 x 2), each with a tile of MT x NT MFMA
// tiles of 32 x 32 x 32 int8 (default 4 x 2 = 128 to-side x 64 from-side rows: 128 accumulator registers, everything in
// VGPRs so that the shift at an exponent transition is plain VALU, two waves per SIMD; with 4 x 4 tiles the accumulators
// have to live in AGPRs and hipcc 7.2 spills around the shift).  No LDS staging of the operand bits and no barrier in the
// K loop: every lane reads the 64-bit word of ITS row and lane half straight from the packed panel (consecutive rows are
// consecutive 16-B pieces: coalesced), one macro step ahead; LDS only holds the 0xFF expansion table and the two digit
// arrays.  Per MFMA k-step (32 positions) a lane expands 16 bits of each of its MT + NT rows through the table (2 look-ups
// each) and masks them with the 16 digits of its lane half: 6 (MT + NT) VALU + 2 (MT + NT) ds_read_b64 for MT NT MFMAs.
// ------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// Wave tile of the pruned, table-fused launches (the default path's): APX_MT x 2 MFMA tiles of 32 x 32 = 128 to-rows x 64 from-rows, two waves per SIMD.
// (Measured and NOT kept, r02-r05 — numbers in docs/HISTORY.md and profiles/r05_*: 2 x 2 at four waves per SIMD, 3 x 2 at three, 3 x 3, 5 x 2; the table
// index as one SDWA instruction; the table reads one k-step ahead; a 16-way replicated table.  Their code left this file in r06: git history has it.)
constexpr int APX_MT = 4, APX_WPS = 2;

// the 16 expanded bytes (0xFF / 0x00) of the 16 bits of k-step KK of a 64-bit panel word: two look-ups in the 256-entry byte -> 8 x 0xFF table (LDS)
template <int KK>
__device__ __forceinline__ v4i expand16(const uint64_t *lutFF, uint64_t w) {
    typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
    const u64x2v q = {lutFF[(w >> (16 * KK)) & 0xFFu], lutFF[(w >> (16 * KK + 8)) & 0xFFu]};
    return __builtin_bit_cast(v4i, q);
}
// the same from a piece of the SCALED panel (k_pack_panel): dword KK holds the two table offsets, in bytes, as 16-bit fields
typedef unsigned int apx_u32x4 __attribute__((ext_vector_type(4)));
template <int KK>
__device__ __forceinline__ v4i expand16s(const uint8_t *lut_bytes, const apx_u32x4 &w) {
    typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
    const unsigned int d = w[KK];
    const u64x2v q = {*reinterpret_cast<const uint64_t *>(lut_bytes + (d & 0xFFFFu)), *reinterpret_cast<const uint64_t *>(lut_bytes + (d >> 16))};
    return __builtin_bit_cast(v4i, q);
}
constexpr int APX_LUT_BYTES = 2048;

// Epilogue of a wave tile (MT x NT MFMA tiles at wave-tile coordinates ty, tx): the threshold-table test per region of 32 to-rows x
// the wave's from-rows (P.fuse) and the store of the regions that are not clean.  s_bt: 256 wave-private LDS bytes.
template <int MT, int NT>
__device__ __forceinline__ void apx_gemm_epilogue(const ApxGemmArgs &P, v16i (&acc)[MT][NT], int ty, int tx, int lane, const int2 *s_tab, uint8_t *s_bt) {
    constexpr int TH = 32 * MT, TWd = 32 * NT;
    const int frow = lane & 31, fh = lane >> 5;
    int bf[NT];
    if (P.fuse) {
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = (int)P.bin_f[tx * TWd + 32 * j + frow];
        // one coalesced load of the wave tile's TH to-side bins, then LDS byte reads at static offsets (64 separate global byte
        // loads per lane made the epilogue cost more than the screen saved)
        for (int r = lane; r < TH; r += 64) s_bt[r] = P.bin_t[ty * TH + r];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);   // (wave-private LDS: the writes of this wave are done before its reads)
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        bool store = true;
        if (P.fuse) {   // region = to-rows [ty TH + 32 i, + 32) x the wave's from-rows (NT * 32 = 64 when NT = 2)
            bool ok = true;
#pragma unroll
            for (int j = 0; j < NT; ++j) ok = ok && bf[j] != 255;
            if (P.sr_mask && P.sr_mask[(int64_t)((ty * TH) / 128) * (P.RFpad / 64) + tx] != 0) ok = false;   // (the band mask's tiles are 128 x 64; TWd = 64)
            int bt[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                bt[e] = (int)s_bt[32 * i + (e & 3) + 8 * (e >> 2) + 4 * fh];
                ok = ok && bt[e] != 255;
            }
            unsigned int fails = 0;   // bit e * NT + j: entry outside its thresholds
            const bool eligible = __ballot(!ok) == 0ull;
            if (eligible) {   // (a region with a row of another kind of SNP is stored without the 32 look-ups per lane)
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int2 th = s_tab[bt[e] * P.tab_nb + bf[j]];
                        const int n = acc[i][j][e];
                        const bool in = n > th.x && n < th.y;
                        ok = ok && in;
                        if (!in) fails |= 1u << (e * NT + j);
                    }
            }
            if (NT == 2 && P.maybe && eligible && __ballot(!ok) != 0ull) {
                // r04: the few entries outside their thresholds go to the maybe list, the region counts as clean
                const unsigned int mine = (unsigned int)__popc(fails);
                unsigned int incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                const unsigned int tot = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
                if (tot <= (unsigned int)APX_MAYBE_MAX) {
                    unsigned int base = 0;
                    if (lane == 0) base = atomicAdd(P.maybe_n, tot);
                    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                    unsigned int pos = base + incl - mine;
                    unsigned int f = fails;
                    while (f) {
                        const int b = __builtin_ctz(f);
                        f &= f - 1;
                        const int e = b / NT, j = b % NT;
                        if (pos < P.maybe_cap) {
                            ApxMaybe m;
                            m.trow = (uint32_t)(ty * TH + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * fh);
                            m.fcol = (uint32_t)(tx * TWd + 32 * j + frow);
                            // (dynamic index into the accumulator tile: a select chain over the 32 candidates, only on this rare path)
                            int v = 0;
#pragma unroll
                            for (int ee = 0; ee < 16; ++ee)
#pragma unroll
                                for (int jj = 0; jj < NT; ++jj) v = (ee == e && jj == j) ? acc[i][jj][ee] : v;
                            m.n = v;
                            P.maybe[pos] = m;
                        }
                        ++pos;
                    }
                    ok = true;   // handled: clean
                }
            }
#ifdef LDW_ABLATE_GEMM_STORE   // timing ablation only (wrong results): what the G' stores of the table-eligible regions that are not clean cost
            const bool clean = __ballot(bf[0] == 255 || bt[0] == 255) == 0ull || __ballot(!ok) == 0ull;
#else
            const bool clean = __ballot(!ok) == 0ull;
#endif
            if (lane == 0) {
#pragma unroll
                for (int h = 0; h < (NT * 32) / 64; ++h) P.clean[(int64_t)((ty * TH + 32 * i) / 32) * (P.RFpad / 64) + (tx * TWd) / 64 + h] = clean ? 1 : 0;
            }
            store = !clean;
        }
        if (store) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int fcol = tx * TWd + 32 * j + frow;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int trow = ty * TH + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    if ((128 % TH == 0 || trow < P.RTpad) && (64 % TWd == 0 || fcol < P.RFpad)) P.G[(int64_t)trow * P.RFpad + fcol] = acc[i][j][e];
                }
            }
        }
    }
}

// Tile pruning (ApxGemmArgs::skip_ctr).  k_apx_live_tiles: which wave tiles (128 to-rows x 64 from-rows) does the approximate GEMM have
// to compute?  A tile is pruned when EVERY pair in it is dismissed whatever its joint count:
//   (a) all rows carry bins of the threshold table and the rectangle [min, max] x [min, max] of their bins holds only unconditional
//       entries (-1, INT_MAX) — the epilogue's test n' > L && n' < H would pass whatever the GEMM computed.  The rows of a block are
//       ordered by their marginal (prep_block), so a tile spans a few bins per side (rectangles of more than 64 entries: computed);
//   (b) the rows of each side are all of ONE kind (2 or 3 flagged states) and one side is dead versus the other's kind (k_snp_sup).
// One workgroup per 128-row group of the to side, one THREAD per tile: it folds the bins / flags of its 64 from-rows and of the
// group's 128 to-rows (16-byte loads).  Pruned tiles are flagged clean where that applies and counted; the others are appended to
// the list the GEMM runs over — in order within the group, ONE atomic per workgroup for the place of its run (a wave per tile with
// an atomic each took 130 us for the 16.6k tiles of a C4 block; this takes a few).
struct TileSummary {
    int bmin, bmax;
    unsigned fand, forr;
};
__device__ __forceinline__ TileSummary apx_fold_rows(const uint8_t *__restrict__ bins, const uint8_t *__restrict__ flags, int n) {
    TileSummary S{255, 0, 0xFFu, 0u};
    for (int i = 0; i < n; i += 16) {
        const uint4 b = *reinterpret_cast<const uint4 *>(bins + i);
        const unsigned w[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int v = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
            S.bmin = v < S.bmin ? v : S.bmin;
            S.bmax = v > S.bmax ? v : S.bmax;
        }
        if (flags) {
            const uint4 f = *reinterpret_cast<const uint4 *>(flags + i);
            unsigned a = f.x & f.y & f.z & f.w, o = f.x | f.y | f.z | f.w;
            a &= a >> 16; a &= a >> 8;
            o |= o >> 16; o |= o >> 8;
            S.fand &= a & 0xFFu;
            S.forr |= o & 0xFFu;
        }
    }
    return S;
}

__global__ __launch_bounds__(256) void k_apx_live_tiles(ApxGemmArgs P) {
    constexpr int MT = APX_MT, NT = 2, TH = 32 * MT, TWd = 32 * NT;
    __shared__ unsigned int s_wave[4], s_base;
    const int ty = (int)blockIdx.x, ntx = P.RFpad / TWd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TileSummary T = apx_fold_rows(P.bin_t + (int64_t)ty * TH, P.rflag_t ? P.rflag_t + (int64_t)ty * TH : nullptr, TH);
    unsigned int n_pruned = 0;
    for (int tx0 = 0; tx0 < ntx; tx0 += 256) {
        const int tx = tx0 + (int)threadIdx.x;
        // (a diagonal block: the tiles above the diagonal of row positions are not part of the GEMM at all — their regions keep the zero the host set)
        const bool valid = tx < ntx && !(P.lower_only && tx * TWd + TWd - 1 < ty * TH);
        bool pruned = false, all_binned = false;
        if (valid && !(P.sr_mask && P.sr_mask[(int64_t)((ty * TH) / 128) * (P.RFpad / 64) + tx] != 0)) {
            const TileSummary F = apx_fold_rows(P.bin_f + (int64_t)tx * TWd, P.rflag_f ? P.rflag_f + (int64_t)tx * TWd : nullptr, TWd);
            all_binned = T.bmax < P.tab_nb && F.bmax < P.tab_nb;
            if (P.rflag_t) {   // the wider tables: rows of ONE kind per side, one side dead versus the other's kind (k_snp_sup)
                const unsigned kt = T.fand & PF_KIND, kf = F.fand & PF_KIND;
                pruned = kt >= 2u && kt == (T.forr & PF_KIND) && kf >= 2u && kf == (F.forr & PF_KIND) &&
                         ((T.fand & (kf == 2u ? PF_DEAD2 : PF_DEAD3)) != 0u || (F.fand & (kt == 2u ? PF_DEAD2 : PF_DEAD3)) != 0u);
            }
            if (!pruned && all_binned && (T.bmax - T.bmin + 1) * (F.bmax - F.bmin + 1) <= 64) {   // the table: only unconditional entries
                bool ok = true;
                for (int bt = T.bmin; bt <= T.bmax; ++bt)
                    for (int bf = F.bmin; bf <= F.bmax; ++bf) {
                        const int2 th = P.tab[bt * P.tab_nb + bf];
                        ok = ok && th.x < 0 && th.y == 2147483647;
                    }
                pruned = ok;
            }
        }
        if (pruned && all_binned) {
            // (clean flags speak of biallelic x biallelic regions only: k_mi_screen maps a region to a tile and 32 column slots by row
            // position, which holds there; the other pruned tiles are found again by the screen from the SNPs' flags)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < (NT * 32) / 64; ++h) P.clean[(int64_t)((ty * TH + 32 * i) / 32) * (P.RFpad / 64) + (tx * TWd) / 64 + h] = 1;
        }
        const bool live = valid && !pruned;
        const unsigned long long lm = __ballot(live);
        n_pruned += (unsigned int)__popcll(__ballot(valid && pruned));
        if (lane == 0) s_wave[wave] = (unsigned int)__popcll(lm);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            s_base = tot ? atomicAdd(P.n_live, tot) : 0u;
        }
        __syncthreads();
        unsigned int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (live) P.tile_list[off + (unsigned int)__popcll(lm & ((1ull << lane) - 1ull))] = (uint32_t)(ty * ntx + tx);
        __syncthreads();
    }
    if (lane == 0 && n_pruned) atomicAdd(P.skip_ctr, (unsigned long long)n_pruned);
}

template <int MT, int NT, bool FINE, int WPS = 2>
__global__ __launch_bounds__(256, WPS) void gemm_apx_kernel(ApxGemmArgs P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *lutFF = reinterpret_cast<uint64_t *>(smem);           // [256]: byte of bits -> 8 bytes of 0xFF / 0x00
    uint8_t *sA = smem + APX_LUT_BYTES, *sB = sA + (size_t)P.M2 * 128;       // digits by position
    int2 *s_tab = reinterpret_cast<int2 *>(sB + (size_t)P.M2 * 128);  // threshold table (P.fuse)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int TH = 32 * MT, TWd = 32 * NT;
    int ty = 2 * blockIdx.y + (wave >> 1), tx = 2 * blockIdx.x + (wave & 1);
    bool outside;
    if (P.tile_list) {
        // tile pruning (k_apx_live_tiles): the grid is 1-D over the list of the tiles that are left, four per workgroup; it is sized for
        // the worst case, so the workgroups beyond the list leave at once, before anything is staged
        const unsigned int n_live = *P.n_live;
        if (blockIdx.x * 4u >= n_live) return;
        const unsigned int k = blockIdx.x * 4u + (unsigned int)wave;
        outside = k >= n_live;
        const int t = outside ? 0 : (int)P.tile_list[k];
        const int ntx = P.RFpad / TWd;
        ty = t / ntx;
        tx = t - ty * ntx;
    } else {
        outside = ty * TH >= P.RTpad || tx * TWd >= P.RFpad || (P.lower_only && tx * TWd + TWd - 1 < ty * TH);
    }
    if (P.fuse)
        for (int i = tid; i < P.tab_nb * P.tab_nb; i += 256) s_tab[i] = P.tab[i];
    {
        uint64_t e = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) e |= ((tid >> k) & 1) ? (0xFFull << (8 * k)) : 0ull;
        lutFF[tid] = e;
        const int n16 = P.M2 * 8;   // 16-byte pieces per digit array
        for (int i = tid; i < n16; i += 256) {
            reinterpret_cast<uint4 *>(sA)[i] = reinterpret_cast<const uint4 *>(P.dig_a)[i];
            reinterpret_cast<uint4 *>(sB)[i] = reinterpret_cast<const uint4 *>(P.dig_b)[i];
        }
    }
    __syncthreads();
    if (outside) return;
    const int frow = lane & 31, fh = lane >> 5;
    const int64_t sta = (int64_t)P.RTpad * 2, stb = (int64_t)P.RFpad * 2;
    {
        const apx_u32x4 *pa[MT], *pb[NT];   // (pieces of the scaled panel: k_pack_panel)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int rt = ty * TH + 32 * i + frow;
            if (128 % TH != 0 && rt >= P.RTpad) rt = P.RTpad - 1;   // (only the 96-row tiles of the r05 experiment can run past RTpad, a multiple of 128)
            pa[i] = reinterpret_cast<const apx_u32x4 *>(P.panel_t) + ((int64_t)rt * 2 + fh);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            int rf = tx * TWd + 32 * i + frow;
            if (64 % TWd != 0 && rf >= P.RFpad) rf = P.RFpad - 1;   // (only the 96-column tiles of the r05 experiments can run past RFpad, a multiple of 64)
            pb[i] = reinterpret_cast<const apx_u32x4 *>(P.panel_f) + ((int64_t)rf * 2 + fh);
        }
        v16i acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
        apx_u32x4 wa[MT], wb[NT], na[MT], nb[NT];
        const uint8_t *lut_bytes = smem;
#pragma unroll
        for (int i = 0; i < MT; ++i) na[i] = pa[i][0];
#pragma unroll
        for (int i = 0; i < NT; ++i) nb[i] = pb[i][0];
        for (int m = 0; m < P.M2; ++m) {
#pragma unroll
            for (int i = 0; i < MT; ++i) wa[i] = na[i];
#pragma unroll
            for (int i = 0; i < NT; ++i) wb[i] = nb[i];
            if (m + 1 < P.M2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) na[i] = pa[i][(int64_t)(m + 1) * sta];
#pragma unroll
                for (int i = 0; i < NT; ++i) nb[i] = pb[i][(int64_t)(m + 1) * stb];
            }
            // FINE: one block exponent per k-step; k-step kk = positions 128 m + 32 kk .. + 31 (k_pack_panel interleaves the panel words
            // accordingly), lane half fh holds 16 of them.  Coarse (the weights' dynamic range within 128 positions is small: clonal
            // data): one exponent per macro step, k-step kk = bits 16 kk of the plain words 0 and 1 — 4 % faster (0.517 vs 0.538 ms).
            const int4 sh4 = reinterpret_cast<const int4 *>(P.shift)[m];   // (wave-uniform: one scalar load per macro step)
            const int shk[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const uint8_t *dA = sA + m * 128 + fh * (FINE ? 16 : 64), *dB = sB + m * 128 + fh * (FINE ? 16 : 64);
            auto kstep = [&](auto KKc) {
                constexpr int kk = decltype(KKc)::value;
                const int sh = shk[kk];
                if ((FINE || kk == 0) && sh) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[i][j][e] = (int)((unsigned)acc[i][j][e] >> sh);
                }
                const v4i da = *reinterpret_cast<const v4i *>(dA + (FINE ? 32 : 16) * kk);
                const v4i db = *reinterpret_cast<const v4i *>(dB + (FINE ? 32 : 16) * kk);
                v4i fa[MT], fb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = expand16s<kk>(lut_bytes, wa[i]) & da;
#pragma unroll
                for (int i = 0; i < NT; ++i) fb[i] = expand16s<kk>(lut_bytes, wb[i]) & db;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
            };
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            kstep(std::integral_constant<int, 2>{});
            kstep(std::integral_constant<int, 3>{});
        }
        apx_gemm_epilogue<MT, NT>(P, acc, ty, tx, lane, s_tab, reinterpret_cast<uint8_t *>(s_tab + 64 * 64) + wave * 256);
    }
}
#ifdef LDW_EXPERIMENTS
// ------------------------------------------------------------------------------------------------
// gemm_apx_pipe_kernel (r03 experiment, LDW_APX_KERNEL=pipe; coarse exponents only): the register-expansion kernel with the expansion
// of k-step s + 1 SOFTWARE-PIPELINED under the MFMAs of k-step s, at a dependency distance of three MFMA slots: slot k of a step
// issues MFMA k, the index arithmetic and the two table reads of fragment k of the NEXT step (k < 6), and the four ANDs of the
// fragment whose reads were issued three slots earlier.  Two fragment sets alternate (4 k-steps per macro step: no copies).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_apx_pipe_kernel(ApxGemmArgs P) {
    constexpr int MT = 4, NT = 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *lutFF = reinterpret_cast<uint64_t *>(smem);
    uint8_t *sA = smem + 2048, *sB = sA + (size_t)P.M2 * 128;
    int2 *s_tab = reinterpret_cast<int2 *>(sB + (size_t)P.M2 * 128);
    const int tid = threadIdx.x;
    if (P.fuse)
        for (int i = tid; i < P.tab_nb * P.tab_nb; i += 256) s_tab[i] = P.tab[i];
    {
        uint64_t e = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) e |= ((tid >> k) & 1) ? (0xFFull << (8 * k)) : 0ull;
        lutFF[tid] = e;
        const int n16 = P.M2 * 8;
        for (int i = tid; i < n16; i += 256) {
            reinterpret_cast<uint4 *>(sA)[i] = reinterpret_cast<const uint4 *>(P.dig_a)[i];
            reinterpret_cast<uint4 *>(sB)[i] = reinterpret_cast<const uint4 *>(P.dig_b)[i];
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int ty = 2 * blockIdx.y + (wave >> 1), tx = 2 * blockIdx.x + (wave & 1);
    constexpr int TH = 32 * MT, TWd = 32 * NT;
    if (ty * TH >= P.RTpad || tx * TWd >= P.RFpad) return;
    if (P.lower_only && tx * TWd + TWd - 1 < ty * TH) return;
    const int frow = lane & 31, fh = lane >> 5;
    const uint64_t *pw[6];   // rows of this lane: 0..3 to side, 4..5 from side
#pragma unroll
    for (int i = 0; i < MT; ++i) pw[i] = P.panel_t + ((int64_t)(ty * TH + 32 * i + frow) * 2 + fh);
#pragma unroll
    for (int i = 0; i < NT; ++i) pw[MT + i] = P.panel_f + ((int64_t)(tx * TWd + 32 * i + frow) * 2 + fh);
    const int64_t st6[6] = {(int64_t)P.RTpad * 2, (int64_t)P.RTpad * 2, (int64_t)P.RTpad * 2, (int64_t)P.RTpad * 2, (int64_t)P.RFpad * 2, (int64_t)P.RFpad * 2};
    v16i acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
    uint64_t wcur[6], wnxt[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        wcur[r] = pw[r][0];
        wnxt[r] = P.M2 > 1 ? pw[r][st6[r]] : 0ull;
    }
    // fragment sets X (even k-steps) and Y (odd k-steps); index 0..3 to side, 4..5 from side
    v4i fX[6], fY[6];
    auto lut2 = [&](uint64_t w, int kk) -> v4i {
        const u64x2v q = {lutFF[(w >> (16 * kk)) & 0xFFu], lutFF[(w >> (16 * kk + 8)) & 0xFFu]};
        return __builtin_bit_cast(v4i, q);
    };
    const uint8_t *dAl = sA + fh * 64, *dBl = sB + fh * 64;
    {   // prologue: fragments of k-step 0 into X
        const v4i da = *reinterpret_cast<const v4i *>(dAl), db = *reinterpret_cast<const v4i *>(dBl);
#pragma unroll
        for (int r = 0; r < 6; ++r) fX[r] = lut2(wcur[r], 0) & (r < MT ? da : db);
    }
    // fragment order of the expansion = the order in which the MFMAs of a step first need them
    constexpr int ORD[6] = {4, 0, 1, 5, 2, 3};                    // B0 A0 A1 B1 A2 A3
    constexpr int MI_[8] = {0, 1, 0, 1, 2, 2, 3, 3}, MJ_[8] = {0, 0, 1, 1, 0, 1, 0, 1};
    for (int m = 0; m < P.M2; ++m) {
        const int sh = P.shift[4 * m];
        if (sh) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = (int)((unsigned)acc[i][j][e] >> sh);
        }
        const bool more = m + 1 < P.M2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            v4i(&cur)[6] = (kk & 1) ? fY : fX;
            v4i(&nxt)[6] = (kk & 1) ? fX : fY;
            const int kn = (kk + 1) & 3;                            // k-step within its macro step of the NEXT step
            const bool wrap = kk == 3;                              // the next step belongs to macro step m + 1
            const int mo = wrap ? (m + 1) * 128 : m * 128;
            const bool have_next = !wrap || more;
            v4i dna = {0, 0, 0, 0}, dnb = {0, 0, 0, 0};
            if (have_next) {
                dna = *reinterpret_cast<const v4i *>(dAl + mo + 16 * kn);
                dnb = *reinterpret_cast<const v4i *>(dBl + mo + 16 * kn);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[MI_[k]][MJ_[k]] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur[MI_[k]], cur[MT + MJ_[k]], acc[MI_[k]][MJ_[k]], 0, 0, 0);
                if (k < 6) {
                    const int r = ORD[k];
                    nxt[r] = lut2(wrap ? wnxt[r] : wcur[r], kn);
                }
                if (k >= 3 && k < 7) {
                    const int r = ORD[k - 3];
                    nxt[r] = nxt[r] & (r < MT ? dna : dnb);
                }
                if (k == 7) {
                    nxt[ORD[4]] = nxt[ORD[4]] & dna;
                    nxt[ORD[5]] = nxt[ORD[5]] & dna;
                }
#ifdef LDW_PIPE_SGB   // pinning the interleave (MFMA, 2 table reads, 8 VALU per slot): 0.572 ms; without it (the compiler's own order): see docs/HISTORY.md 5.1c
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
#endif
            }
        }
        // panel words: the next macro step's become current, the one after that is requested
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            wcur[r] = wnxt[r];
            if (m + 2 < P.M2) wnxt[r] = pw[r][(int64_t)(m + 2) * st6[r]];
        }
    }
    apx_gemm_epilogue<MT, NT>(P, acc, ty, tx, lane, s_tab, reinterpret_cast<uint8_t *>(s_tab + 64 * 64) + wave * 256);
}

// ------------------------------------------------------------------------------------------------
// gemm_apx_lds_kernel: the same contraction with the operand expansion SHARED through LDS (r03).
//
// gemm_apx_kernel expands every fragment in the wave that consumes it: 6 expansions (48 VALU, 12 table reads) per 8 MFMAs, which
// keeps the VALU issue port ~88 % busy at the full MFMA rate even on paper — it runs at 0.51 of the int8 peak.  Here a
// workgroup of 8 waves (2 along the to side x 4 along the from side, 256 x 256 rows, wave tile 128 x 64 as before) expands
// each operand row ONCE per 64 positions: thread t owns row t & 255 of side t >> 8, reads the row's 64-bit panel word, turns it
// into four digit-masked 16-byte fragments (table look-ups as before) and writes them to an LDS byte tile; every wave then feeds
// its MFMAs with ds_read_b128.  Per 64 positions a wave does 4 expansions (32 VALU, 8 table reads, 4 ds_write_b128) and
// 12 ds_read_b128 for 16 MFMAs: 2 VALU per MFMA instead of 6.  Two LDS buffers, one workgroup barrier per 64 positions (512
// MFMA cycles).  The byte tile is [row][64 B] with the 16-byte slot XOR-swizzled by (row >> 2) & 3: the 16 lanes that share
// an LDS pass (rows r .. r + 15, same slot) then cover all 64 banks once, for the writes and for the fragment reads.
// LDS: 2 x 32 KB tiles + 2 KB table + the digit arrays (+ 33 KB threshold table when the epilogue applies it): one workgroup
// of 8 waves per CU, two waves per SIMD as before.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void gemm_apx_lds_kernel(ApxGemmArgs P) {
    constexpr int MT = 4, NT = 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *lutFF = reinterpret_cast<uint64_t *>(smem);                       // [256]
    uint8_t *sA = smem + 2048, *sB = sA + (size_t)P.M2 * 128;                   // digits by position
    uint8_t *tile0 = sB + (size_t)P.M2 * 128;                                   // 2 buffers x (256 to-rows + 256 from-rows) x 64 B
    int2 *s_tab = reinterpret_cast<int2 *>(tile0 + 2 * 32768);                  // threshold table (P.fuse)
    const int tid = threadIdx.x;
    if (P.fuse)
        for (int i = tid; i < P.tab_nb * P.tab_nb; i += 512) s_tab[i] = P.tab[i];
    if (tid < 256) {
        uint64_t e = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) e |= ((tid >> k) & 1) ? (0xFFull << (8 * k)) : 0ull;
        lutFF[tid] = e;
    }
    {
        const int n16 = P.M2 * 8;
        for (int i = tid; i < n16; i += 512) {
            reinterpret_cast<uint4 *>(sA)[i] = reinterpret_cast<const uint4 *>(P.dig_a)[i];
            reinterpret_cast<uint4 *>(sB)[i] = reinterpret_cast<const uint4 *>(P.dig_b)[i];
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 2, wx = wave & 3;
    const int ty = 2 * blockIdx.y + wy, tx = 4 * blockIdx.x + wx;      // wave-tile coordinates (128 to-rows, 64 from-rows)
    constexpr int TH = 128, TWd = 64;
    // the whole 256 x 256 tile above the diagonal: nothing to do (uniform over the workgroup: no barrier is skipped by a part of it)
    if (P.lower_only && (int)(4 * blockIdx.x + 3) * TWd + TWd - 1 < (int)(2 * blockIdx.y) * TH) return;
    const bool wave_live = ty * TH < P.RTpad && tx * TWd < P.RFpad && !(P.lower_only && tx * TWd + TWd - 1 < ty * TH);
    // ---- expansion role: row er of side eside ----
    const int eside = tid >> 8, er = tid & 255;
    const int grow = eside == 0 ? (int)(2 * blockIdx.y) * TH + er : (int)(4 * blockIdx.x) * TWd + er;   // row in the side's row list
    const int gmax = eside == 0 ? P.RTpad : P.RFpad;
    const bool erow_ok = grow < gmax;
    const uint64_t *prow = (eside == 0 ? P.panel_t : P.panel_f) + (int64_t)(erow_ok ? grow : 0) * 2;
    const int64_t pst = (int64_t)gmax * 2;     // words per macro step in this side's panel
    const uint8_t *dig = eside == 0 ? sA : sB;
    const int wsw = (er >> 2) & 3;             // the row's slot swizzle
    uint8_t *wdst = tile0 + eside * 16384 + er * 64;
    typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
    // The expansion of a chunk is split around the MFMAs of the chunk being computed: the eight table reads and the digits are
    // REQUESTED before the first k-step (exp_issue), masked and written behind it (exp_finish), so that the LDS round trip and
    // the write pass run under MFMAs instead of in a phase of their own (all eight waves leave a barrier together: a phase
    // without MFMAs is a phase in which the matrix pipe of every SIMD idles).
    u64x2v xq[4];
    v4i xdg[4];
    auto exp_issue = [&](uint64_t w, int chunk) {
        const uint8_t *d = dig + chunk * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xq[g][0] = lutFF[(w >> (16 * g)) & 0xFFu];
            xq[g][1] = lutFF[(w >> (16 * g + 8)) & 0xFFu];
            xdg[g] = *reinterpret_cast<const v4i *>(d + 16 * g);
        }
    };
    auto exp_finish = [&](uint8_t *buf_base) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<v4i *>(buf_base + 16 * (g ^ wsw)) = __builtin_bit_cast(v4i, xq[g]) & xdg[g];
    };
    // ---- consumer role ----
    const int frow = lane & 31, fh = lane >> 5;
    int offA[MT], offB[NT];      // byte offset of this lane's row in the A / B half of a buffer, and its swizzle
    int swA[MT], swB[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wy * TH + 32 * i + frow;
        offA[i] = r * 64;
        swA[i] = (r >> 2) & 3;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wx * TWd + 32 * j + frow;
        offB[j] = 16384 + r * 64;
        swB[j] = (r >> 2) & 3;
    }
    v16i acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    // one macro step (128 positions = two chunks) per iteration; the panel words of macro step m + 2 are requested at the top of
    // iteration m and first touched at its bottom, two chunks of MFMAs later
    u64x2v wcur = {0ull, 0ull}, wnext = {0ull, 0ull}, wfar = {0ull, 0ull};
    if (erow_ok) wcur = *reinterpret_cast<const u64x2v *>(prow);
    if (erow_ok && P.M2 > 1) wnext = *reinterpret_cast<const u64x2v *>(prow + pst);
    __syncthreads();            // table, digits
    exp_issue(wcur[0], 0);
    exp_finish(wdst);
    __syncthreads();
    auto kstep = [&](const uint8_t *cur, int ks) {
        v4i fa[MT], fb[NT];
        const int slot = 2 * ks + fh;
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const v4i *>(cur + offA[i] + 16 * (slot ^ swA[i]));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const v4i *>(cur + offB[j] + 16 * (slot ^ swB[j]));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    for (int m = 0; m < P.M2; ++m) {
        if (erow_ok && m + 2 < P.M2) wfar = *reinterpret_cast<const u64x2v *>(prow + (int64_t)(m + 2) * pst);
        // ---- chunk 2m (buffer 0); chunk 2m + 1 is expanded into buffer 1 meanwhile ----
        const int4 sh4 = reinterpret_cast<const int4 *>(P.shift)[m];   // one exponent per k-step of 32 positions
        auto rescale = [&](int sh) {
            if (sh && wave_live) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = (int)((unsigned)acc[i][j][e] >> sh);
            }
        };
        exp_issue(wcur[1], 2 * m + 1);
        rescale(sh4.x);
        if (wave_live) kstep(tile0, 0);
        exp_finish(wdst + 32768);
        rescale(sh4.y);
        if (wave_live) kstep(tile0, 1);
        __syncthreads();
        // ---- chunk 2m + 1 (buffer 1); the first chunk of the next macro step goes into buffer 0 ----
        if (m + 1 < P.M2) exp_issue(wnext[0], 2 * m + 2);
        rescale(sh4.z);
        if (wave_live) kstep(tile0 + 32768, 0);
        if (m + 1 < P.M2) exp_finish(wdst);
        rescale(sh4.w);
        if (wave_live) kstep(tile0 + 32768, 1);
        __syncthreads();
        wcur = wnext;
        wnext = wfar;
    }
    if (!wave_live) return;
    apx_gemm_epilogue<MT, NT>(P, acc, ty, tx, lane, s_tab, reinterpret_cast<uint8_t *>(s_tab + 64 * 64) + wave * 256);
}

#endif   // LDW_EXPERIMENTS

int launch_apx_live_tiles(ldw_ctx *c, const ApxGemmArgs &P, hipStream_t st) {
    LDW_REQUIRE(P.fuse && P.skip_ctr && P.tile_list && P.n_live && P.RTpad % 128 == 0 && P.RFpad % 64 == 0 && P.tab && P.tab_nb == 64,
                LDW_ERR_ARG, "launch_apx_live_tiles: bad pruning arguments");
    hipLaunchKernelGGL(k_apx_live_tiles, dim3((unsigned)(P.RTpad / (32 * APX_MT))), dim3(256), 0, st, P);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

int launch_gemm_apx(ldw_ctx *c, const ApxGemmArgs &P, hipStream_t st) {
    LDW_REQUIRE(P.RTpad % APX_TW == 0 && P.RFpad % APX_TW == 0 && P.M2 > 0, LDW_ERR_ARG, "launch_gemm_apx: padding violated (RT %d RF %d M2 %d)", P.RTpad,
                P.RFpad, P.M2);
    const size_t lds = APX_LUT_BYTES + (size_t)P.M2 * 256 + (P.fuse ? (size_t)P.tab_nb * P.tab_nb * 8 + 1024 : 0);
    LDW_REQUIRE(lds <= 65536, LDW_ERR_ARG, "launch_gemm_apx: %d positions do not fit the LDS digit arrays", P.M2 * 128);
    LDW_REQUIRE(!P.fuse || (P.tab_nb == 64 && P.bin_t && P.bin_f && P.tab && P.clean), LDW_ERR_ARG, "launch_gemm_apx: bad table arguments");
    static const int tile = [] {
        const char *e = exp_env("LDW_APX_TILE");   // tuning: wave tile in MFMA tiles, to side x from side (default 4 x 2)
        return e ? atoi(e) : 42;
    }();
#ifdef LDW_EXPERIMENTS
    const int kern = apx_kernel_is_lds() ? 1 : 0;
    static const bool pipe = [] { const char *e = exp_env("LDW_APX_KERNEL"); return e && e[0] == 'p'; }();
    if (pipe && !P.fine && tile == 42) {
        const int ntx = P.RFpad / 64, nty = P.RTpad / 128;
        hipLaunchKernelGGL(gemm_apx_pipe_kernel, dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P);
        LDW_HIP(hipGetLastError());
        int64_t waves = 0;
        for (int ty = 0; ty < nty; ++ty) {
            if (!P.lower_only) waves += ntx;
            else for (int tx = 0; tx < ntx; ++tx) waves += (tx * 64 + 63 < ty * 128) ? 0 : 1;
        }
        c->gemm_stat[0] += 1;
        if (P.fuse) c->gemm_stat[5] += 1;
        c->gemm_stat[1] += 2.0 * (double)waves * 128 * 64 * ((double)P.M2 * 128.0);
        return LDW_OK;
    }
    const size_t lds2 = 2048 + (size_t)P.M2 * 256 + 2 * 32768 + (P.fuse ? (size_t)P.tab_nb * P.tab_nb * 8 + 8 * 256 : 0);
    if (kern == 1 && lds2 <= 160 * 1024 && tile == 42) {
        static bool attr_set = false;
        if (!attr_set) {
            LDW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_apx_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        const int ntx = P.RFpad / 64, nty = P.RTpad / 128;
        hipLaunchKernelGGL(gemm_apx_lds_kernel, dim3((unsigned)((ntx + 3) / 4), (unsigned)((nty + 1) / 2)), dim3(512), lds2, st, P);
        LDW_HIP(hipGetLastError());
        int64_t waves = 0;
        for (int ty = 0; ty < nty; ++ty) {
            if (!P.lower_only) waves += ntx;
            else for (int tx = 0; tx < ntx; ++tx) waves += (tx * 64 + 63 < ty * 128) ? 0 : 1;
        }
        c->gemm_stat[0] += 1;
        if (P.fuse) c->gemm_stat[5] += 1;
        c->gemm_stat[1] += 2.0 * (double)waves * 128 * 64 * ((double)P.M2 * 128.0);
        return LDW_OK;
    }
#endif   // LDW_EXPERIMENTS
#define LDW_APX_LAUNCH(MTv, NTv)                                                                                              \
    {                                                                                                                         \
        const int ntx = P.RFpad / (32 * NTv), nty = P.RTpad / (32 * MTv);                                                     \
        if (P.fine) hipLaunchKernelGGL((gemm_apx_kernel<MTv, NTv, true>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P); \
        else hipLaunchKernelGGL((gemm_apx_kernel<MTv, NTv, false>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P); \
    }
    if (P.skip_ctr) {
        LDW_REQUIRE(P.fuse && P.tile_list && P.n_live && P.RFpad % 64 == 0, LDW_ERR_ARG, "launch_gemm_apx: bad pruning arguments");
        const int tiles = (P.RTpad / (32 * APX_MT)) * (P.RFpad / 64), g = (tiles + 3) / 4;   // (the list is made by launch_apx_live_tiles)
        if (P.fine) hipLaunchKernelGGL((gemm_apx_kernel<APX_MT, 2, true, APX_WPS>), dim3((unsigned)g), dim3(256), lds, st, P);
        else hipLaunchKernelGGL((gemm_apx_kernel<APX_MT, 2, false, APX_WPS>), dim3((unsigned)g), dim3(256), lds, st, P);
    }
#ifdef LDW_EXPERIMENTS
    else if (tile == 22 && !P.fuse) LDW_APX_LAUNCH(2, 2)          // (the table epilogue assumes 64 from-rows per wave: NT = 2)
    else if (tile == 33 && !P.fuse) {   // r05: 3 x 3 MFMA tiles per wave (96 x 96: 6 fragments per 9 MFMAs instead of 6 per 8), two waves per SIMD
        const int ntx = (P.RFpad + 95) / 96, nty = (P.RTpad + 95) / 96;
        if (P.fine) hipLaunchKernelGGL((gemm_apx_kernel<3, 3, true, 2>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P);
        else hipLaunchKernelGGL((gemm_apx_kernel<3, 3, false, 2>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P);
    }
    else if (tile == 224 && !P.fuse) {   // r05: 2 x 2 tiles built for FOUR waves per SIMD (<= 128 VGPRs)
        const int ntx = P.RFpad / 64, nty = P.RTpad / 64;
        if (P.fine) hipLaunchKernelGGL((gemm_apx_kernel<2, 2, true, 4>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P);
        else hipLaunchKernelGGL((gemm_apx_kernel<2, 2, false, 4>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P);
    }
    else if (tile == 24 && !P.fuse) LDW_APX_LAUNCH(2, 4)
    else if (tile == 32 && !P.fuse) {   // r05: 3 x 2 MFMA tiles per wave = 96 accumulators, built for THREE waves per SIMD (<= 170 VGPRs)
        const int ntx = P.RFpad / 64, nty = (P.RTpad + 95) / 96;
        if (P.fine) hipLaunchKernelGGL((gemm_apx_kernel<3, 2, true, 3>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P);
        else hipLaunchKernelGGL((gemm_apx_kernel<3, 2, false, 3>), dim3((unsigned)((ntx + 1) / 2), (unsigned)((nty + 1) / 2)), dim3(256), lds, st, P);
    }
#endif
    else LDW_APX_LAUNCH(4, 2)
#undef LDW_APX_LAUNCH
    LDW_HIP(hipGetLastError());
    {   // executed work (ldw_gemm_stats): waves that do not leave at once, each 2 * rows_t * rows_f * K int8 operations
        const int tl = P.fuse ? 42 : tile;
        const int MTv = (P.fuse && P.skip_ctr) ? APX_MT : (tl == 22 || tl == 24 || tl == 224 ? 2 : (tl == 32 || tl == 33 ? 3 : 4)), NTv = tl == 24 ? 4 : (tl == 33 ? 3 : 2), TH = 32 * MTv, TWd = 32 * NTv;
        int64_t waves = 0;
        for (int ty = 0; ty * TH < P.RTpad; ++ty) {
            const int ntx = P.RFpad / TWd;
            if (!P.lower_only) waves += ntx;
            else for (int tx = 0; tx < ntx; ++tx) waves += (tx * TWd + TWd - 1 < ty * TH) ? 0 : 1;
        }
        c->gemm_stat[0] += 1;
        if (P.fuse) c->gemm_stat[5] += 1;
        c->gemm_stat[1] += 2.0 * (double)waves * TH * TWd * ((double)P.M2 * 128.0);
        if (P.fuse && P.skip_ctr) {   // the tiles the kernel prunes are taken off again when the counter is read (ldw_gemm_stats, ldw_links_end)
            c->apx_ops_per_wave = 2.0 * TH * TWd * ((double)P.M2 * 128.0);
            c->apx_waves_total += waves;
        }
    }
    return LDW_OK;
}

// ------------------------------------------------------------------------------------------------
// The long-range candidates listed pair by pair (units without a short-range pair), in two kernels:
//   k_pair_sums<CA, CB>  one WAVE per pair, lane = 32-bit word of the bit rows (coalesced 256-B reads of the pair's CA + CB rows
//                        from the row-major bit matrix).  sum_s V_s x_s y_s = sum_words sum_segments V_class popcount(x & y & mask):
//                        every lane multiplies the popcounts of ITS words by the class weight at once (one 64-bit multiply-add
//                        per segment) and a wave reduction adds the lanes up: no per-class flush, no serial walk over the row.
//                        The CA * CB exact sums of pair i of list `sub` go to sums[(sub * cap + i) * 16 + j * 4 + i].
//   k_pair_mi<NA, NB>    one LANE per pair: the sums -> fp64 MI -> emission (candidate list + histogram).
// One pair in a few thousand gets here.
// ------------------------------------------------------------------------------------------------
struct PairArgs {
    const uint64_t *Mbits;
    int64_t KW;
    int nwords, path, nseg;      // 32-bit words per row; segment records
    int seg_in_lds;              // the segment tables fit the workgroup's LDS (else they are read from global memory)
    const PopSeg *segs;
    const int32_t *wbeg, *row0;
    int32_t zero_row;
    int64_t *sums;
    EpiArgs A;
    unsigned long long *ghist;
};

// CA x CB = 4 x 4 is the list of the pairs with a SNP of three or four indicator rows: nearly all of them 3 x 1 or 3 x 2, so the products with the
// absent rows (na, nb are wave-uniform: scalar branches) are skipped — r03 ran all sixteen against rows of zeros (k_pair_sums<true>: 1.0 ms per C4 pass)
template <int CA, int CB>
__device__ __forceinline__ void pair_sums_body(const PairArgs &P, int path, const int32_t *s_wbeg, const PopSeg *s_segs) {
    constexpr bool GENB = CA == 4 && CB == 4;
    const EpiArgs &A = P.A;
    const int sub = path * PAIR_SHARDS + (int)blockIdx.y;
    unsigned int n = A.pl_n[sub];
    n = n > A.pl_cap ? A.pl_cap : n;
    const int lane = threadIdx.x & 63;
    const PairEnt *list = A.pl_pairs + (int64_t)sub * A.pl_cap;
    // the segments of THIS lane's first words are the same for every pair: keep two per word in registers (a word holds more
    // than two only where several tiny weight classes meet: those go through the table)
    constexpr int KW = 3, KS = 2;
    uint32_t smask[KW][KS];
    int64_t sV[KW][KS];
    bool more[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int w = lane + 64 * k;
        int s0 = 0, s1 = 0;
        if (w < P.nwords) {
            s0 = s_wbeg[w] & 0x7FFFFFFF;
            s1 = s_wbeg[w + 1] & 0x7FFFFFFF;
        }
        more[k] = s1 - s0 > KS;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const bool on = s0 + q < s1 && !more[k];
            smask[k][q] = on ? s_segs[s0 + q].mask : 0u;
            sV[k][q] = on ? s_segs[s0 + q].V : 0;
        }
    }
    for (unsigned int idx = blockIdx.x * 4u + (threadIdx.x >> 6); idx < n; idx += gridDim.x * 4u) {
        const PairEnt e = list[idx];
        const uint32_t era = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.ra), erb = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.rb);
        const int r0a = (int)(era & 0x1FFFFFFFu), na = (int)(era >> 29), r0b = (int)(erb & 0x1FFFFFFFu), nb = (int)(erb >> 29);
        const uint32_t *fr[CA], *tr[CB];
#pragma unroll
        for (int i = 0; i < CA; ++i) fr[i] = reinterpret_cast<const uint32_t *>(P.Mbits + (int64_t)(i < na ? r0a + i : P.zero_row) * P.KW);
#pragma unroll
        for (int j = 0; j < CB; ++j) tr[j] = reinterpret_cast<const uint32_t *>(P.Mbits + (int64_t)(j < nb ? r0b + j : P.zero_row) * P.KW);
        int64_t s[CB][CA];
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int i = 0; i < CA; ++i) s[j][i] = 0;
        // the first KW * 64 words: all loads of the pair are issued at once, the segments come from registers
        uint32_t f[KW][CA], tt[KW][CB];
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int w = lane + 64 * k;
            const bool in = w < P.nwords;
#pragma unroll
            for (int i = 0; i < CA; ++i) f[k][i] = (in && (!GENB || i < na)) ? fr[i][w] : 0u;
#pragma unroll
            for (int j = 0; j < CB; ++j) tt[k][j] = (in && (!GENB || j < nb)) ? tr[j][w] : 0u;
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            if (more[k]) {
                const int w = lane + 64 * k;
                const int s0 = s_wbeg[w] & 0x7FFFFFFF, s1 = s_wbeg[w + 1] & 0x7FFFFFFF;
                for (int sg = s0; sg < s1; ++sg) {
                    const PopSeg seg = s_segs[sg];
#pragma unroll
                    for (int j = 0; j < CB; ++j)
#pragma unroll
                        for (int i = 0; i < CA; ++i)
                            if (!GENB || (i < na && j < nb)) s[j][i] += (int64_t)((uint64_t)__popc(f[k][i] & tt[k][j] & seg.mask) * (uint64_t)seg.V);
                }
            } else {
#pragma unroll
                for (int q2 = 0; q2 < KS; ++q2)
#pragma unroll
                    for (int j = 0; j < CB; ++j)
#pragma unroll
                        for (int i = 0; i < CA; ++i)
                            if (!GENB || (i < na && j < nb)) s[j][i] += (int64_t)((uint64_t)__popc(f[k][i] & tt[k][j] & smask[k][q2]) * (uint64_t)sV[k][q2]);
            }
        }
        for (int w = lane + 64 * KW; w < P.nwords; w += 64) {   // longer rows (more than 6144 sequences): the rest through the table
            uint32_t f2[CA], t2[CB];
#pragma unroll
            for (int i = 0; i < CA; ++i) f2[i] = (!GENB || i < na) ? fr[i][w] : 0u;
#pragma unroll
            for (int j = 0; j < CB; ++j) t2[j] = (!GENB || j < nb) ? tr[j][w] : 0u;
            const int s0 = s_wbeg[w] & 0x7FFFFFFF, s1 = s_wbeg[w + 1] & 0x7FFFFFFF;
            for (int sg = s0; sg < s1; ++sg) {
                const PopSeg seg = s_segs[sg];
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int i = 0; i < CA; ++i)
                        if (!GENB || (i < na && j < nb)) s[j][i] += (int64_t)((uint64_t)__popc(f2[i] & t2[j] & seg.mask) * (uint64_t)seg.V);
            }
        }
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                int64_t v = s[j][i];
                if (!GENB || (i < na && j < nb)) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                }
                if (lane == 0) P.sums[((int64_t)sub * A.pl_cap + idx) * 16 + j * 4 + i] = v;
            }
    }
}

// the four straight-line lists in one launch (blockIdx.z = path), the predicated one (16 sums per pair: many more registers)
// in another.  The segment tables are staged in LDS once per workgroup.
template <bool GEN>
__global__ __launch_bounds__(256) void k_pair_sums(PairArgs P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pop_smem[];
    int32_t *s_wbeg = reinterpret_cast<int32_t *>(pop_smem);
    PopSeg *s_segs = reinterpret_cast<PopSeg *>(pop_smem + (((size_t)P.nwords + 4) * 4 + 15) / 16 * 16);
    const int path = GEN ? 4 : (int)blockIdx.z;
    if (blockIdx.x * 4u >= P.A.pl_n[path * PAIR_SHARDS + (int)blockIdx.y]) return;   // nothing for this workgroup
    if (P.seg_in_lds) {
        for (int i = threadIdx.x; i < P.nwords + 1; i += 256) s_wbeg[i] = P.wbeg[i];
        for (int i = threadIdx.x; i < P.nseg; i += 256) s_segs[i] = P.segs[i];
        __syncthreads();
    } else {   // r04: thousands of weight classes (N distinct weights): the tables stay in global memory (L2-resident: every wave reads the same records)
        s_wbeg = const_cast<int32_t *>(P.wbeg);
        s_segs = const_cast<PopSeg *>(P.segs);
    }
    if constexpr (GEN) {
        pair_sums_body<4, 4>(P, 4, s_wbeg, s_segs);
    } else {
        switch (path) {
            case 0: pair_sums_body<1, 1>(P, 0, s_wbeg, s_segs); break;
            case 1: pair_sums_body<2, 1>(P, 1, s_wbeg, s_segs); break;
            case 2: pair_sums_body<1, 2>(P, 2, s_wbeg, s_segs); break;
            default: pair_sums_body<2, 2>(P, 3, s_wbeg, s_segs); break;
        }
    }
}

// r05: the same exact sums for weightings with MANY classes (every sequence its own weight: hdw = 1 / (k + 1) with all k distinct, the bench's
// adversarial "distinct weights" case).  The class-wise form above then walks one segment per POSITION — 32 records per word, read from global
// memory because 5 000 records do not fit LDS: 1.37 ms per launch, 38 ms of an 85 ms pass, bound by the L2.  Here a lane walks the SET BITS of
// its words instead (two SNPs' minor states co-occur in a few sequences per word) and adds the weight of each position from a table in LDS,
// laid out [bit][word] so that the lanes of a wave — different words, any bit — never share a bank (stride a multiple of 32 words).
// Same integers (a sum of the same int64 terms in another order), so the tables downstream are bit-identical.
template <int CA, int CB>
__device__ __forceinline__ void pair_sums_bits_body(const PairArgs &P, int path, const int64_t *sVT, int stride) {
    constexpr bool GENB = CA == 4 && CB == 4;
    const EpiArgs &A = P.A;
    const int sub = path * PAIR_SHARDS + (int)blockIdx.y;
    unsigned int n = A.pl_n[sub];
    n = n > A.pl_cap ? A.pl_cap : n;
    const int lane = threadIdx.x & 63;
    const PairEnt *list = A.pl_pairs + (int64_t)sub * A.pl_cap;
    for (unsigned int idx = blockIdx.x * 4u + (threadIdx.x >> 6); idx < n; idx += gridDim.x * 4u) {
        const PairEnt e = list[idx];
        const uint32_t era = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.ra), erb = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.rb);
        const int r0a = (int)(era & 0x1FFFFFFFu), na = (int)(era >> 29), r0b = (int)(erb & 0x1FFFFFFFu), nb = (int)(erb >> 29);
        const uint32_t *fr[CA], *tr[CB];
#pragma unroll
        for (int i = 0; i < CA; ++i) fr[i] = reinterpret_cast<const uint32_t *>(P.Mbits + (int64_t)(i < na ? r0a + i : P.zero_row) * P.KW);
#pragma unroll
        for (int j = 0; j < CB; ++j) tr[j] = reinterpret_cast<const uint32_t *>(P.Mbits + (int64_t)(j < nb ? r0b + j : P.zero_row) * P.KW);
        int64_t s[CB][CA];
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int i = 0; i < CA; ++i) s[j][i] = 0;
        for (int w = lane; w < P.nwords; w += 64) {
            uint32_t f[CA], t[CB], uf = 0u, ut = 0u;
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                f[i] = (!GENB || i < na) ? fr[i][w] : 0u;
                uf |= f[i];
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                t[j] = (!GENB || j < nb) ? tr[j][w] : 0u;
                ut |= t[j];
            }
            uint32_t u = uf & ut;   // positions where SOME row of a and some row of b are set
            while (u) {
                const int b = __builtin_ctz(u);
                u &= u - 1u;
                const int64_t v = sVT[b * stride + w];
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int i = 0; i < CA; ++i)
                        if (CA * CB == 1 || (((f[i] & t[j]) >> b) & 1u)) s[j][i] += v;
            }
        }
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                int64_t v = s[j][i];
                if (!GENB || (i < na && j < nb)) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                }
                if (lane == 0) P.sums[((int64_t)sub * A.pl_cap + idx) * 16 + j * 4 + i] = v;
            }
    }
}

template <bool GEN>
__global__ __launch_bounds__(256) void k_pair_sums_bits(PairArgs P, const int64_t *__restrict__ vpos, int stride) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pop_smem[];
    int64_t *sVT = reinterpret_cast<int64_t *>(pop_smem);
    const int path = GEN ? 4 : (int)blockIdx.z;
    if (blockIdx.x * 4u >= P.A.pl_n[path * PAIR_SHARDS + (int)blockIdx.y]) return;   // nothing for this workgroup
    for (int i = threadIdx.x; i < 32 * stride; i += 256) {
        const int b = i / stride, w = i - b * stride;
        sVT[i] = w < P.nwords ? vpos[32 * w + b] : 0;
    }
    __syncthreads();
    if constexpr (GEN) {
        pair_sums_bits_body<4, 4>(P, 4, sVT, stride);
    } else {
        switch (path) {
            case 0: pair_sums_bits_body<1, 1>(P, 0, sVT, stride); break;
            case 1: pair_sums_bits_body<2, 1>(P, 1, sVT, stride); break;
            case 2: pair_sums_bits_body<1, 2>(P, 2, sVT, stride); break;
            default: pair_sums_bits_body<2, 2>(P, 3, sVT, stride); break;
        }
    }
}

// Emission of the listed pairs.  Nearly all of them ARE candidates (they sit in the few buckets just above the guess), so one
// returning atomic on the candidate counter plus one histogram atomic per pair made the kernel a queue at two or three
// addresses of the L2 (100 us for 7e4 pairs; 14 us with the atomics compiled out): a wave bumps the counter ONCE for all its
// candidates, and the bucket counts go through a workgroup-local window of PAIR_HWIN buckets in LDS, flushed at the end.
constexpr int PAIR_HWIN = 128;

template <int NA, int NB>   // NA = 0: any slot counts (predicated code)
__device__ __forceinline__ void pair_mi_body(const PairArgs &P, int path, unsigned int *s_hist) {
    constexpr bool GEN = NA == 0;
    const EpiArgs &A = P.A;
    const EmitArgs &E = A.E;
    const int sub = path * PAIR_SHARDS + (int)blockIdx.y;
    unsigned int n = A.pl_n[sub];
    n = n > A.pl_cap ? A.pl_cap : n;
    const bool square = A.nf == A.nt;
    const PairEnt *list = A.pl_pairs + (int64_t)sub * A.pl_cap;
    const int lane = (int)(threadIdx.x & 63);
    // wave-uniform trip count: every lane stays in the loop, so the ballots below see the whole wave
    for (unsigned int base = blockIdx.x * 256u + (threadIdx.x & ~63u); base < n; base += gridDim.x * 256u) {
        const unsigned int idx = base + (unsigned int)lane;
        bool valid = idx < n;
        uint32_t t = 0, q = 0;
        if (valid) {
            t = list[idx].t;
            q = list[idx].q;
        }
        const RowPack &RP = A.rowpack[t];
        valid = valid && RP.a_loc >= 0;
        bool cand = false;
        int bk = 0, seg = -1, a_loc = 0, b_loc = 0, sg = 0;
        double mi = 0.0;
        if (valid) {
            const RowSide R = RP.R;
            a_loc = RP.a_loc;
            const ColMeta M = A.colpack[q];
            b_loc = M.bl;
            sg = A.span ? M.ci.pad[0] : 0;   // a span: the reference block (segment) of the column; b_loc is local to it
            const int64_t *sp = P.sums + ((int64_t)sub * A.pl_cap + idx) * 16;
            if constexpr (GEN) {
                mi = pair_mi<4, 4>(A, R, M, a_loc, b_loc, square, gacc_plain(sp, 1, 4));
            } else {
                mi = pair_mi_full<NA, NB>(A, R, M, a_loc, b_loc, square, gacc_plain(sp, 1, 4));
            }
            seg = pair_seg(a_loc, b_loc, E.lower_only);
            if (seg >= 0 && E.any_sr && col_is_sr(M.ci, a_loc)) {
                emit_pair_spec(E, M.ci, a_loc, b_loc, R.sa, M.sb, mi, P.ghist);   // (a short-range pair is never listed; kept for exactness)
            } else if (seg >= 0 && E.do_lr && mi >= E.spec_lo) {
                bk = mi_bucket(mi);
                cand = bk >= E.spec_B;
            }
        }
        if (A.span) {
            // every reference block of the span keeps its own candidate list and histogram (the long-range filter is per block:
            // R/computePairwiseMI.R:352-358); the lanes of a wave hold columns of any segment (the to side is ordered by weight)
            for (int k = 0; k < A.span; ++k) {
                const bool mine = cand && sg == k;
                const unsigned long long mk = __ballot(mine);
                if (mk == 0ull) continue;
                const int leader = __builtin_ctzll(mk);
                unsigned long long pos = 0;
                if (lane == leader) pos = atomicAdd(A.sseg[k].n_cand, (unsigned long long)__popcll(mk));
                pos = __shfl(pos, leader);
                if (mine) {
                    pos += (unsigned long long)__popcll(mk & ((1ull << lane) - 1ull));
                    A.sseg[k].ckey[pos] = f64_key(mi);
                    A.sseg[k].cval[pos] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)E.nf);
                    const int w = bk - (E.spec_B > 0 ? E.spec_B : 0);
                    if (w < PAIR_HWIN) atomicAdd(&s_hist[k * PAIR_HWIN + w], 1u);
                    else atomicAdd(&A.sseg[k].ghist[bk], 1ull);
                }
            }
            continue;
        }
        const unsigned long long mk = __ballot(cand);
        if (mk == 0ull) continue;
        const int leader = __builtin_ctzll(mk);
        unsigned long long pos = 0;
        if (lane == leader) pos = atomicAdd(E.n_cand, (unsigned long long)__popcll(mk));
        pos = __shfl(pos, leader);
        if (cand) {
            pos += (unsigned long long)__popcll(mk & ((1ull << lane) - 1ull));
            E.ckey[pos] = f64_key(mi);
            E.cval[pos] = ((uint64_t)seg << 62) | ((uint64_t)a_loc + (uint64_t)b_loc * (uint64_t)E.nf);
            const int w = bk - (E.spec_B > 0 ? E.spec_B : 0);
            if (w < PAIR_HWIN) atomicAdd(&s_hist[w], 1u);
            else atomicAdd(&P.ghist[bk], 1ull);
        }
    }
}

__global__ __launch_bounds__(256) void k_pair_mi(PairArgs P) {
    __shared__ unsigned int s_hist[LDW_SPAN_MAX * PAIR_HWIN];   // one bucket window per segment of a span (segment 0: an ordinary block)
    for (int i = threadIdx.x; i < LDW_SPAN_MAX * PAIR_HWIN; i += 256) s_hist[i] = 0u;
    __syncthreads();
    switch (blockIdx.z) {
        case 0: pair_mi_body<1, 1>(P, 0, s_hist); break;
        case 1: pair_mi_body<2, 1>(P, 1, s_hist); break;
        case 2: pair_mi_body<1, 2>(P, 2, s_hist); break;
        case 3: pair_mi_body<2, 2>(P, 3, s_hist); break;
        default: pair_mi_body<0, 0>(P, 4, s_hist); break;
    }
    __syncthreads();
    const int hb0 = P.A.E.spec_B > 0 ? P.A.E.spec_B : 0;
    if (P.A.span) {
        for (int i = threadIdx.x; i < P.A.span * PAIR_HWIN; i += 256) {
            const int k = i / PAIR_HWIN, w = i - k * PAIR_HWIN;
            if (s_hist[i] != 0u && hb0 + w < NBINS) atomicAdd(&P.A.sseg[k].ghist[hb0 + w], (unsigned long long)s_hist[i]);
        }
        return;
    }
    if (threadIdx.x < PAIR_HWIN && s_hist[threadIdx.x] != 0u && hb0 + (int)threadIdx.x < NBINS)
        atomicAdd(&P.ghist[hb0 + threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

int launch_pairs_exact(ldw_ctx *c, const EpiArgs &A, unsigned long long *ghist, int64_t *sums, hipStream_t st) {
    PairArgs P;
    memset(&P, 0, sizeof(P));
    P.Mbits = c->Mbits.as<uint64_t>();
    P.KW = c->KW;
    P.nwords = (int)(c->KW * 2);
    P.nseg = c->n_pop_segs;
    P.segs = c->pop_segs.as<PopSeg>();
    P.wbeg = c->pop_wbeg.as<int32_t>();
    P.row0 = c->row0.as<int32_t>();
    P.zero_row = (int32_t)c->R;
    P.sums = sums;
    P.A = A;
    P.ghist = ghist;
    // 1024 waves per list stride over its pairs (one wave per pair), then one lane per pair for the fp64 value
    size_t lds = (((size_t)P.nwords + 4) * 4 + 15) / 16 * 16 + (size_t)P.nseg * sizeof(PopSeg);
    P.seg_in_lds = lds <= 60000 ? 1 : 0;
    if (!P.seg_in_lds) lds = 0;
    // r05: many weight classes (the segment tables do not fit LDS) -> walk the set bits against a per-position weight table in LDS instead, as long as
    // THAT fits (256 B per word of a row: 40 KB at N = 5120, 80 KB at 10 240; beyond 160 KB — N > 20 480 — the class-wise kernel reads its tables from global memory)
    const int stride = (P.nwords + 31) / 32 * 32;
    const size_t lds_bits = (size_t)32 * stride * 8;
    const bool bits_off = getenv("LDW_NO_PAIR_BITS") != nullptr;   // (read per call: A/B, and the test runs both forms)
    // (which form: the class-wise one costs a popcount per SEGMENT — 1.3 segments per word with C4's 57 clonal classes —, the bit walk an LDS read per
    // co-occurring position; from ~8 segments per word on, i.e. weightings whose classes are a few sequences each, the bits are fewer than the segments)
    const bool many_classes = !P.seg_in_lds || P.nseg >= 8 * P.nwords;
    if (many_classes && !bits_off && lds_bits <= 160 * 1024 && c->pop_vpos.p) {
        if (lds_bits > 64 * 1024 && !c->pair_bits_attr) {
            LDW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pair_sums_bits<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            LDW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pair_sums_bits<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            c->pair_bits_attr = true;
        }
        hipLaunchKernelGGL(k_pair_sums_bits<false>, dim3(256, PAIR_SHARDS, 4), dim3(256), lds_bits, st, P, c->pop_vpos.as<int64_t>(), stride);
        hipLaunchKernelGGL(k_pair_sums_bits<true>, dim3(256, PAIR_SHARDS, 1), dim3(256), lds_bits, st, P, c->pop_vpos.as<int64_t>(), stride);
    } else {
        // (grid: r06 — 512 / 1024 workgroups per list instead of 256: 67 -> 61 and 34 -> 23 us per launch; 128 / 64 / 32: 96 / 150 / 236 us.  A 1-D grid whose
        // waves share the pairs of all lists evenly was built too: no better (64 / 42 us) — the long launches are not a matter of balance.  LDW_PAIR_GRID: A/B)
        static const unsigned gx_env = [] { const char *e = getenv("LDW_PAIR_GRID"); return e ? (unsigned)atoi(e) : 0u; }();
        hipLaunchKernelGGL(k_pair_sums<false>, dim3(gx_env ? gx_env : 512u, PAIR_SHARDS, 4), dim3(256), lds, st, P);
        hipLaunchKernelGGL(k_pair_sums<true>, dim3(gx_env ? gx_env : 1024u, PAIR_SHARDS, 1), dim3(256), lds, st, P);
    }
    hipLaunchKernelGGL(k_pair_mi, dim3(64, PAIR_SHARDS, PAIR_PATHS), dim3(256), 0, st, P);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

}  // namespace ldw

namespace ldw {
void warm_apx() {   // ldw_ctx_reserve: load this translation unit's code object ahead of its first launch
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_pack_panel));
    (void)hipGetLastError();
}
}  // namespace ldw
