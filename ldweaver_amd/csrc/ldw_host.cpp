// Host-side natives of the path: ARACNE (R/io_functions.R:101-164) and the four small helpers it is
// built from in the reference (src/computeMI.cpp:25-77, src/fintersect.cpp:6-32).  Exact comparisons
// only; no floating-point arithmetic.
#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "ldw_internal.h"

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

}  // extern "C"
