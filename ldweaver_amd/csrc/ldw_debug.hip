// Test hooks (r06): the engine's BOUNDS as functions.  The default path dismisses pairs on hand-derived bounds (BOUNDS.md); each of them is
// reachable here WITHOUT running a pass, on caller-made inputs, so that tests/test_bounds.py can brute-force it against the MI formula of
// src/computeMI.cpp:19 — random and extremal joint tables, quirk Q1's RXY != r_a r_b / 4, the engine's own constants:
//   ldw_debug_apx_params    the dual-digit weights V' of the current weighting and every constant the approximate screen's bound is built from
//   ldw_debug_rows          SNP -> indicator rows (which state each row stands for)
//   ldw_debug_apx_gemm      gemm_apx_kernel on caller-chosen rows: the int32 sums G' (truncation bound apx_lost_units)
//   ldw_debug_screen_bound  full_cells_screen / pair_screen_generic (fp32; APX and exact-limb forms) and full_cells_mi (fp64) on caller-made tables
// Nothing here is on the product path; the kernels are the product's own device functions (ldw_epi.h), instantiated once more.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ldw_apx.h"
#include "ldw_dev.h"

using namespace ldw;

namespace ldw {

struct DbgArrays {
    const int64_t *g;      // [n][4][4]: joint sums of the indicator rows, g[case][j][i] (slot i of the from-side SNP, slot j of the to-side SNP)
    const int32_t *g32;    // the same as int32 (approximate path)
    const int64_t *pa, *pb;   // [n][5] integer marginals by slot (units of the sums)
    const float *pX, *pY;     // [n][5] weighted marginals by slot (weight units)
    const double *rr;         // [n][3]: r_a, r_b, RXY
    const uint32_t *mk;       // [n][2]: slot meta of the two SNPs (rows | uqe flags << 3), generic kinds only
    float *out;
    double *out64;
    int64_t n;
};

template <int NA, int NB>
__device__ __forceinline__ void dbg_sides(const DbgArrays &D, int64_t k, RowSide &R, ColMeta &M) {
    R.ra = D.rr[k * 3 + 0];
    M.rb = D.rr[k * 3 + 1];
    R.rta = M.rq = 0.0;
    R.ra0 = 0;
    M.rb0 = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        R.pa[i] = D.pa[k * 5 + i];
        M.pb[i] = D.pb[k * 5 + i];
        R.pXf[i] = D.pX[k * 5 + i];
        M.pYf[i] = D.pY[k * 5 + i];
        R.pXd[i] = (double)R.pXf[i];
        M.pYd[i] = (double)M.pYf[i];
    }
    R.na = NA;
    R.ma = (uint32_t)NA | (((2u << NA) - 1u) << 3);
    M.mb = (uint32_t)NB | (((2u << NB) - 1u) << 3);
}

// kind 0: the approximate path's multi-cell bound, exactly as screen_cols_apx evaluates it (full_cells32 -> full_cells_screen<.., true>);
// kind 2: the exact-limb screen (full_cells -> full_cells_screen<.., false>: the value is an fp32 MI, no bound terms)
template <int NA, int NB, int KIND>
__global__ __launch_bounds__(256) void k_dbg_full(EpiArgs A, DbgArrays D) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= D.n) return;
    RowSide R;
    ColMeta M;
    dbg_sides<NA, NB>(D, k, R, M);
    const double rxy = D.rr[k * 3 + 2];
    if (KIND == 0) {
        int raw[NB][NA];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < NA; ++i) raw[j][i] = D.g32[k * 16 + j * 4 + i];
        FullCells32<NA, NB> C;
        full_cells32<NA, NB>(R, M, raw, C);
        D.out[k] = full_cells_screen<NA, NB, true>(A, R, M, rxy, C);
    } else {
        const GAcc Ga = gacc_plain(D.g + k * 16, 1, 4);
        FullCells<NA, NB> C;
        full_cells<NA, NB>(R, M, Ga, C);
        if (KIND == 2) D.out[k] = full_cells_screen<NA, NB, false>(A, R, M, rxy, C);
        else {   // KIND 4: the fp64 evaluation of the emitted value (full_cells_mi) — marginals in double from the integer ones, as load_row_side makes them
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                R.pXd[i] = (double)R.pa[i] * A.scale;
                M.pYd[i] = (double)M.pb[i] * A.scale;
            }
            D.out64[k] = full_cells_mi<NA, NB>(A, R, M, rxy, C);
        }
    }
}

// kinds 1 / 3: the predicated screens of SNPs with any slot counts and unflagged slots (pair_screen_generic<APX>)
template <bool APX>
__global__ __launch_bounds__(256) void k_dbg_generic(EpiArgs A, DbgArrays D) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= D.n) return;
    RowSide R;
    ColMeta M;
    dbg_sides<1, 1>(D, k, R, M);
    R.ma = D.mk[k * 2 + 0];
    M.mb = D.mk[k * 2 + 1];
    R.na = (int)(R.ma & 7);
    GAcc Ga = gacc_plain(APX ? reinterpret_cast<const int64_t *>(D.g32 + k * 16) : D.g + k * 16, 1, 4);
    Ga.w32 = APX ? 1 : 0;
    D.out[k] = pair_screen_generic<APX>(A, R, M, D.rr[k * 3 + 2], Ga);
}

}  // namespace ldw

extern "C" {

int ldw_debug_apx_params(ldw_ctx *c, double out[20], int64_t *vfixed_out, int64_t *vapx_out, int64_t capacity) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(out, LDW_ERR_ARG, "ldw_debug_apx_params: null output");
    LDW_REQUIRE(c->have_weights, LDW_ERR_STATE, "ldw_debug_apx_params: no weights (ldw_set_weights)");
    EmitArgs E;
    memset(&E, 0, sizeof(E));
    apx_screen_params(c, E);
    out[0] = c->frac_bits;
    out[1] = c->apx_e_last;
    out[2] = c->apx_delta;
    out[3] = c->apx_lost_units;
    out[4] = (double)c->total_fixed;
    out[5] = c->neff;
    out[6] = E.apx_EG;
    out[7] = E.apx_dfac;
    out[8] = E.apx_s1;
    out[9] = E.apx_c1;
    out[10] = E.apx_W;
    out[11] = E.apx_unit;
    out[12] = E.scr_scale;
    {   // the exact-limb screen reads the top 31 bits of a sum (make_emit_args, all limbs in the block-wide GEMM)
        int bits = 0;
        while (bits < 62 && (c->total_fixed >> bits) != 0) ++bits;
        const int shift = bits > 31 ? bits - 31 : 0;
        out[13] = shift;
        out[14] = std::ldexp(1.0, shift - c->frac_bits);
    }
    out[15] = (c->apx_ok ? 1 : 0) | (c->apx_fine ? 2 : 0);
    out[16] = c->lo_abs_sum;   // mixed-precision path: sum_s |V_lo,s| 2^-F and the margin it adds to the screen (lo_bound, ldw_mi_items.inc)
    {
        const double den = c->neff > 1.0 ? c->neff : 1.0;
        out[17] = c->lo_abs_sum * (2.0 * std::log(den + 12.5) + 3.0) / den;
    }
    out[18] = E.apx_MU;
    out[19] = c->nlimbs;
    if (vfixed_out || vapx_out) {
        LDW_REQUIRE(capacity >= c->N, LDW_ERR_SIZE, "ldw_debug_apx_params: capacity %lld < N %lld", (long long)capacity, (long long)c->N);
        for (int64_t s = 0; s < c->N; ++s) {
            if (vfixed_out) vfixed_out[s] = c->h_vfixed[(size_t)s];
            if (vapx_out) vapx_out[s] = (size_t)s < c->h_vapx.size() ? c->h_vapx[(size_t)s] : 0;
        }
    }
    return LDW_OK;
}

int ldw_debug_rows(ldw_ctx *c, int32_t *row0_out, uint32_t *slot_meta_out, int64_t capacity) {
    if (int rc = check_gpu(c)) return rc;
    if (int rc = ensure_rows(c)) return rc;
    LDW_REQUIRE(row0_out && slot_meta_out && capacity >= c->L + 1, LDW_ERR_ARG, "ldw_debug_rows: need L + 1 = %lld entries", (long long)c->L + 1);
    memcpy(row0_out, c->h_row0.data(), (size_t)(c->L + 1) * 4);
    memcpy(slot_meta_out, c->h_slot_meta.data(), (size_t)c->L * 4);
    return LDW_OK;
}

int ldw_debug_apx_gemm(ldw_ctx *c, const int32_t *rows_t, int nrt, const int32_t *rows_f, int nrf, int32_t *out) {
    if (int rc = check_gpu(c)) return rc;
    if (int rc = join_prepare(c)) return rc;
    if (int rc = ensure_rows(c)) return rc;
    LDW_REQUIRE(rows_t && rows_f && out && nrt > 0 && nrf > 0 && nrt <= 8192 && nrf <= 8192, LDW_ERR_ARG, "ldw_debug_apx_gemm: bad argument");
    LDW_REQUIRE(c->apx_ok, LDW_ERR_STATE, "ldw_debug_apx_gemm: the weights do not allow the approximate path (%s)", c->apx_gate.c_str());
    for (int k = 0; k < nrt; ++k) LDW_REQUIRE(rows_t[k] >= 0 && rows_t[k] <= c->R, LDW_ERR_ARG, "ldw_debug_apx_gemm: to-side row %d = %d outside 0..R", k, rows_t[k]);
    for (int k = 0; k < nrf; ++k) LDW_REQUIRE(rows_f[k] >= 0 && rows_f[k] <= c->R, LDW_ERR_ARG, "ldw_debug_apx_gemm: from-side row %d = %d outside 0..R", k, rows_f[k]);
    const int RTpad = (nrt + APX_TW - 1) / APX_TW * APX_TW, RFpad = (nrf + APX_TW - 1) / APX_TW * APX_TW, M2 = (int)(c->KW / 2);
    std::vector<int32_t> rl((size_t)RTpad + RFpad, (int32_t)c->R);   // padding -> the all-zero row R
    std::copy(rows_t, rows_t + nrt, rl.begin());
    std::copy(rows_f, rows_f + nrf, rl.begin() + RTpad);
    DevBuf d_rl, d_pt, d_pf, d_G;
    int rc = LDW_OK;
    if ((rc = d_rl.reserve(rl.size() * 4)) || (rc = d_pt.reserve((size_t)M2 * RTpad * 32)) || (rc = d_pf.reserve((size_t)M2 * RFpad * 32)) ||
        (rc = d_G.reserve((size_t)RTpad * RFpad * 4))) {
        d_rl.release(); d_pt.release(); d_pf.release(); d_G.release();
        return rc;
    }
    auto body = [&]() -> int {
        LDW_HIP(hipMemcpyAsync(d_rl.p, rl.data(), rl.size() * 4, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemsetAsync(d_G.p, 0xFF, (size_t)RTpad * RFpad * 4, c->stream));   // (every entry must be WRITTEN by the kernel)
        if (int r2 = launch_pack_panel(c, d_rl.as<int32_t>() + RTpad, RFpad, d_pf.as<uint64_t>(), c->stream, d_rl.as<int32_t>(), RTpad, d_pt.as<uint64_t>())) return r2;
        ApxGemmArgs P;
        memset(&P, 0, sizeof(P));
        P.panel_t = d_pt.as<uint64_t>();
        P.panel_f = d_pf.as<uint64_t>();
        P.RTpad = RTpad;
        P.RFpad = RFpad;
        P.M2 = M2;
        P.dig_a = c->dig_a.as<uint8_t>();
        P.dig_b = c->dig_b.as<uint8_t>();
        P.shift = c->apx_shift.as<int32_t>();
        P.G = d_G.as<int32_t>();
        P.fine = c->apx_fine ? 1 : 0;
        if (int r2 = launch_gemm_apx(c, P, c->stream)) return r2;
        std::vector<int32_t> h((size_t)RTpad * RFpad);
        LDW_HIP(hipMemcpyAsync(h.data(), d_G.p, h.size() * 4, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        for (int t = 0; t < nrt; ++t) memcpy(out + (size_t)t * nrf, h.data() + (size_t)t * RFpad, (size_t)nrf * 4);
        return LDW_OK;
    };
    rc = body();
    (void)hipStreamSynchronize(c->stream);
    d_rl.release(); d_pt.release(); d_pf.release(); d_G.release();
    return rc;
}

int ldw_debug_screen_bound(ldw_ctx *c, int kind, int na, int nb, int64_t n, const int64_t *g, const int64_t *pa, const int64_t *pb, const float *pX, const float *pY,
                           const double *rr, const uint32_t *masks, const double params[20], float *out, double *out64) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(n > 0 && n < ((int64_t)1 << 28) && g && pa && pb && pX && pY && rr && params, LDW_ERR_ARG, "ldw_debug_screen_bound: bad argument");
    LDW_REQUIRE(kind >= 0 && kind <= 4 && ((kind == 4) ? out64 != nullptr : out != nullptr), LDW_ERR_ARG, "ldw_debug_screen_bound: kind %d / missing output", kind);
    const bool generic = kind == 1 || kind == 3;
    LDW_REQUIRE(generic ? masks != nullptr : ((na == 1 || na == 2) && (nb == 1 || nb == 2)), LDW_ERR_ARG, "ldw_debug_screen_bound: kinds 0 / 2 / 4 take na, nb in {1, 2}; kinds 1 / 3 take masks");
    const bool apx = kind == 0 || kind == 1;
    std::vector<int32_t> g32;
    if (apx) {
        g32.resize((size_t)n * 16);
        for (size_t k = 0; k < g32.size(); ++k) {
            LDW_REQUIRE(g[k] >= -2147483647LL && g[k] <= 2147483647LL, LDW_ERR_ARG, "ldw_debug_screen_bound: entry %zu does not fit int32", k);
            g32[k] = (int32_t)g[k];
        }
    }
    DevBuf B;
    const size_t sz_g = (size_t)n * 16 * 8, sz_g32 = (size_t)n * 16 * 4, sz_p = (size_t)n * 5 * 8, sz_f = (size_t)n * 5 * 4, sz_r = (size_t)n * 3 * 8, sz_m = (size_t)n * 2 * 4,
                 sz_o = (size_t)n * 4, sz_o64 = (size_t)n * 8;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_g = 0, o_g32 = o_g + al(sz_g), o_pa = o_g32 + al(sz_g32), o_pb = o_pa + al(sz_p), o_pX = o_pb + al(sz_p), o_pY = o_pX + al(sz_f), o_rr = o_pY + al(sz_f),
                 o_m = o_rr + al(sz_r), o_o = o_m + al(sz_m), o_o64 = o_o + al(sz_o), total = o_o64 + al(sz_o64);
    if (int rc = B.reserve(total)) return rc;
    char *base = B.as<char>();
    auto body = [&]() -> int {
        LDW_HIP(hipMemcpyAsync(base + o_g, g, sz_g, hipMemcpyHostToDevice, c->stream));
        if (apx) LDW_HIP(hipMemcpyAsync(base + o_g32, g32.data(), sz_g32, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemcpyAsync(base + o_pa, pa, sz_p, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemcpyAsync(base + o_pb, pb, sz_p, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemcpyAsync(base + o_pX, pX, sz_f, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemcpyAsync(base + o_pY, pY, sz_f, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipMemcpyAsync(base + o_rr, rr, sz_r, hipMemcpyHostToDevice, c->stream));
        if (masks) LDW_HIP(hipMemcpyAsync(base + o_m, masks, sz_m, hipMemcpyHostToDevice, c->stream));
        EpiArgs A;
        memset(&A, 0, sizeof(A));
        A.neff = params[5];
        A.scale = std::ldexp(1.0, -(int)params[0]);
        A.quirk = LDW_QUIRK_INTENDED;
        A.E.apx = apx ? 1 : 0;
        A.E.apx_EG = (float)params[6];
        A.E.apx_dfac = (float)params[7];
        A.E.apx_s1 = (float)params[8];
        A.E.apx_c1 = (float)params[9];
        A.E.apx_W = params[10];
        A.E.apx_unit = params[11];
        A.E.apx_MU = (float)params[18];
        A.E.scr_shift = apx ? 0 : (int)params[13];
        A.E.scr_scale = (float)(apx ? params[12] : params[14]);
        DbgArrays D;
        D.g = reinterpret_cast<const int64_t *>(base + o_g);
        D.g32 = reinterpret_cast<const int32_t *>(base + o_g32);
        D.pa = reinterpret_cast<const int64_t *>(base + o_pa);
        D.pb = reinterpret_cast<const int64_t *>(base + o_pb);
        D.pX = reinterpret_cast<const float *>(base + o_pX);
        D.pY = reinterpret_cast<const float *>(base + o_pY);
        D.rr = reinterpret_cast<const double *>(base + o_rr);
        D.mk = reinterpret_cast<const uint32_t *>(base + o_m);
        D.out = reinterpret_cast<float *>(base + o_o);
        D.out64 = reinterpret_cast<double *>(base + o_o64);
        D.n = n;
        const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
#define LDW_DBG_FULL(K)                                                                                   \
    {                                                                                                     \
        if (na == 1 && nb == 1) hipLaunchKernelGGL((k_dbg_full<1, 1, K>), grid, blk, 0, c->stream, A, D); \
        else if (na == 1) hipLaunchKernelGGL((k_dbg_full<1, 2, K>), grid, blk, 0, c->stream, A, D);       \
        else if (nb == 1) hipLaunchKernelGGL((k_dbg_full<2, 1, K>), grid, blk, 0, c->stream, A, D);       \
        else hipLaunchKernelGGL((k_dbg_full<2, 2, K>), grid, blk, 0, c->stream, A, D);                    \
    }
        if (kind == 0) LDW_DBG_FULL(0)
        else if (kind == 2) LDW_DBG_FULL(2)
        else if (kind == 4) LDW_DBG_FULL(4)
        else if (kind == 1) hipLaunchKernelGGL(k_dbg_generic<true>, grid, blk, 0, c->stream, A, D);
        else hipLaunchKernelGGL(k_dbg_generic<false>, grid, blk, 0, c->stream, A, D);
#undef LDW_DBG_FULL
        LDW_HIP(hipGetLastError());
        if (kind == 4) LDW_HIP(hipMemcpyAsync(out64, base + o_o64, sz_o64, hipMemcpyDeviceToHost, c->stream));
        else LDW_HIP(hipMemcpyAsync(out, base + o_o, sz_o, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
        return LDW_OK;
    };
    const int rc = body();
    (void)hipStreamSynchronize(c->stream);
    B.release();
    return rc;
}

}  // extern "C"
