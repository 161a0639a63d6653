// Host-side natives of the path: ARACNE (R/io_functions.R:101-164) and the four small helpers it is
// built from in the reference (src/computeMI.cpp:25-77, src/fintersect.cpp:6-32).  Exact comparisons
// only; no floating-point arithmetic.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "ldw_internal.h"


// ------------------------------------------------------------------------------------------------
// R's sample(n, size) for a given seed: set.seed(seed) (initial scrambling of src/main/RNG.c: 50 steps of the LCG 69069 s + 1, then one
// word per seed slot, dummy[0] = 624), Mersenne-Twister with R's fixup into (0, 1), R_unif_index by rejection on bits drawn 16 at a time
// (sample.kind = "Rejection", R >= 3.6), and the partial Fisher-Yates of do_sample.  The 10 % SNP subset of perform_MI_computation
// (R/computePairwiseMI.R:94-97: set.seed(1988); sample(nsnp, 0.1 nsnp)) is the only random draw on the path; the Python statement of the
// same stream (rcompat.RRandom, checked against R's own output for seeds 42 and 123) took 20 ms for C4 and 100 ms for C5.
// ------------------------------------------------------------------------------------------------
namespace {
struct RMersenne {
    static constexpr int N = 624, M = 397;
    uint32_t mt[N];
    int mti;
    explicit RMersenne(uint32_t seed) {
        uint32_t s = seed;
        for (int j = 0; j < 50; ++j) s = 69069u * s + 1u;
        s = 69069u * s + 1u;   // dummy[0], overwritten by the position word
        for (int j = 0; j < N; ++j) {
            s = 69069u * s + 1u;
            mt[j] = s;
        }
        mti = N;
    }
    double unif_rand() {
        if (mti >= N) {
            static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
            int kk;
            for (kk = 0; kk < N - M; ++kk) {
                const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + M] ^ (y >> 1) ^ mag01[y & 1u];
            }
            for (; kk < N - 1; ++kk) {
                const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ mag01[y & 1u];
            }
            const uint32_t y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ mag01[y & 1u];
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        const double v = (double)y * 2.3283064365386963e-10;
        if (v <= 0.0) return 0.5 * 2.328306437080797e-10;
        if (1.0 - v <= 0.0) return 1.0 - 0.5 * 2.328306437080797e-10;
        return v;
    }
    int64_t unif_index(double dn) {
        if (dn <= 0) return 0;
        const int bits = (int)std::ceil(std::log2(dn));
        for (;;) {
            uint64_t v = 0;
            for (int n = 0; n <= bits; n += 16) v = 65536u * v + (uint64_t)std::floor(unif_rand() * 65536.0);
            if (bits < 64) v &= ((uint64_t)1 << bits) - 1u;
            if ((double)v < dn) return (int64_t)v;
        }
    }
};
}  // namespace

extern "C" {

int ldw_compare_to_row(const double *x, int64_t nr, int64_t nc, const double *y, int64_t ny, uint8_t *ret) {
    LDW_REQUIRE(x && y && ret && nr >= 0 && nc >= 0 && ny >= 0, LDW_ERR_ARG, "ldw_compare_to_row: bad argument");
    for (int64_t j = 0; j < nr; ++j) {
        uint8_t hit = 0;
        for (int64_t k = 0; k < nc && !hit; ++k)
            for (int64_t m = 0; m < ny; ++m)
                if (x[j + k * nr] == y[m]) {
                    hit = 1;
                    break;
                }
        ret[j] = hit;
    }
    return LDW_OK;
}

int ldw_vec_pos_match(const double *x, int64_t nx, const double *y, int64_t ny, double *ret) {
    LDW_REQUIRE(x && y && ret && nx >= 0 && ny >= 0, LDW_ERR_ARG, "ldw_vec_pos_match: bad argument");
    for (int64_t i = 0; i < nx; ++i) {
        ret[i] = 0;
        for (int64_t j = 0; j < ny; ++j)
            if (y[j] == x[i]) {
                ret[i] = (double)(j + 1);
                break;
            }
    }
    return LDW_OK;
}

int ldw_compare_triplet(const double *MI0X, const double *MI0Z, int64_t n, double MI0, int *ret) {
    LDW_REQUIRE(MI0X && MI0Z && ret && n >= 0, LDW_ERR_ARG, "ldw_compare_triplet: bad argument");
    int ok = 1;
    for (int64_t i = 0; i < n; ++i)
        if (MI0 < MI0X[i] && MI0 < MI0Z[i]) {
            ok = 0;
            break;
        }
    *ret = ok;
    return LDW_OK;
}

int ldw_fast_intersect(const int32_t *A, int64_t na, const int32_t *B, int64_t nb, int32_t *out, int64_t *n_out) {
    LDW_REQUIRE(A && B && out && n_out && na >= 0 && nb >= 0, LDW_ERR_ARG, "ldw_fast_intersect: bad argument");
    std::vector<int32_t> a(A, A + na), b(B, B + nb);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    int64_t i = 0, j = 0, n = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) ++i;
        else if (a[i] > b[j]) ++j;
        else {
            out[n++] = a[i];
            ++i;
            ++j;
        }
    }
    *n_out = n;
    return LDW_OK;
}

// runARACNE.  For each link to check (X,Z): the neighbours of X and of Z in the full link set are
// intersected; for a common neighbour Y the MI of the FIRST full-set link joining X and Y (resp. Z and Y)
// is used (that is what `.vecPosMatch` returns in the reference); the link is indirect (flag 0) iff some Y
// has MI(X,Z) < MI(X,Y) and MI(X,Z) < MI(Z,Y).  Duplicated links in the full set cannot change the
// boolean (they repeat the same first-occurrence MI), so adjacency lists are de-duplicated here.  Like
// `.fast_intersect`, neighbour positions are compared after conversion to int.
int ldw_aracne(ldw_ctx * /*ctx*/, const double *chk_pos1, const double *chk_pos2, const double *chk_MI, int64_t n_chk,
               const double *full_pos1, const double *full_pos2, const double *full_MI, int64_t n_full,
               uint8_t *flags_out) {
    LDW_REQUIRE(n_chk >= 0 && n_full >= 0, LDW_ERR_ARG, "ldw_aracne: negative length");
    if (n_chk == 0) return LDW_OK;
    LDW_REQUIRE(chk_pos1 && chk_pos2 && chk_MI && flags_out, LDW_ERR_ARG, "ldw_aracne: null argument");
    LDW_REQUIRE(n_full == 0 || (full_pos1 && full_pos2 && full_MI), LDW_ERR_ARG, "ldw_aracne: null argument");
    struct Nb {
        int32_t y;     // neighbour position as int
        int64_t idx;   // first full-set link joining the node and y
    };
    std::unordered_map<double, std::vector<Nb>> adj;
    adj.reserve((size_t)n_full);
    for (int64_t k = 0; k < n_full; ++k) {
        const double p1 = full_pos1[k], p2 = full_pos2[k];
        // a node's neighbour list holds the endpoints different from the node itself (matX[matX != pX])
        if (p2 != p1) {
            adj[p1].push_back({(int32_t)p2, k});
            adj[p2].push_back({(int32_t)p1, k});
        }
    }
    for (auto &kv : adj) {
        auto &v = kv.second;
        std::stable_sort(v.begin(), v.end(), [](const Nb &a, const Nb &b) { return a.y < b.y; });
        // keep the first occurrence (smallest link index: pushes were in link order and the sort is stable)
        size_t w = 0;
        for (size_t i = 0; i < v.size(); ++i)
            if (w == 0 || v[w - 1].y != v[i].y) v[w++] = v[i];
        v.resize(w);
    }
    static const std::vector<Nb> empty;
    for (int64_t i = 0; i < n_chk; ++i) {
        flags_out[i] = 1;  // un-checkable links stay TRUE
        auto ix = adj.find(chk_pos1[i]);
        auto iz = adj.find(chk_pos2[i]);
        if (ix == adj.end() || iz == adj.end()) continue;
        const auto &vx = ix->second, &vz = iz->second;
        const double mi0 = chk_MI[i];
        size_t a = 0, b = 0;
        while (a < vx.size() && b < vz.size()) {
            if (vx[a].y < vz[b].y) ++a;
            else if (vx[a].y > vz[b].y) ++b;
            else {
                if (mi0 < full_MI[vx[a].idx] && mi0 < full_MI[vz[b].idx]) {
                    flags_out[i] = 0;
                    break;
                }
                ++a;
                ++b;
            }
        }
    }
    return LDW_OK;
}

int ldw_r_sample(uint32_t seed, int64_t n, int64_t size, int64_t *out) {
    LDW_REQUIRE(out && n >= 0 && size >= 0 && size <= n, LDW_ERR_ARG, "ldw_r_sample: bad argument (n %lld, size %lld)", (long long)n, (long long)size);
    RMersenne rng(seed);
    if (n > 10000000 && size <= n / 2) {
        // R >= 3.6: sample.int switches to do_sample2 (useHash = n > 1e7 && !replace && is.null(prob) && size <= n/2): every element is drawn
        // with R_unif_index(n) + 1 and re-drawn while it repeats an earlier one (at most 100 draws per element, as in R).  Only WHICH values
        // count as duplicates matters, not R's hash function: an open-addressing set of the values drawn so far.
        size_t cap = 16;
        while (cap < (size_t)size * 2 + 2) cap <<= 1;
        std::vector<int64_t> set(cap, 0);   // 0 = empty (values are 1-based)
        auto insert = [&](int64_t v) -> bool {   // false: already present
            size_t h = (size_t)((uint64_t)v * 0x9E3779B97F4A7C15ull) & (cap - 1);
            while (set[h] != 0) {
                if (set[h] == v) return false;
                h = (h + 1) & (cap - 1);
            }
            set[h] = v;
            return true;
        };
        for (int64_t i = 0; i < size; ++i) {
            int64_t v = 0;
            for (int j = 0; j < 100; ++j) {
                v = rng.unif_index((double)n) + 1;
                if (insert(v)) break;
            }
            out[i] = v;
        }
        return LDW_OK;
    }
    std::vector<int64_t> pool((size_t)n);
    for (int64_t i = 0; i < n; ++i) pool[(size_t)i] = i + 1;
    int64_t left = n;
    for (int64_t i = 0; i < size; ++i) {
        const int64_t j = rng.unif_index((double)left);
        out[i] = pool[(size_t)j];
        pool[(size_t)j] = pool[(size_t)--left];
    }
    return LDW_OK;
}

}  // extern "C"
