/*
 * R-side binding of libldweaver_amd.so: the `.Call` entry points a maintainer would add to LDWeaver's src/
 * (registered next to the existing table in src/RcppExports.cpp:154-172).  Plain R C API (no Rcpp needed).
 * NOT compiled in this repository: R headers are absent from the build image.  See INTEGRATION.md.
 *
 *   R CMD SHLIB ldweaver_amd_shim.c -I<repo>/include -L<repo>/ldweaver_amd -lldweaver_amd
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "ldweaver_amd.h"

/* r05: one context per device of options(ldwamd.devices = 0:7) (ldwamd_set_devices); g_ctxs[0] is the context the tables end up in and
 * every single-context entry point works on.  Without the option: one context on device 0, as before. */
#define LDWAMD_MAX_DEV 64
static ldw_ctx *g_ctxs[LDWAMD_MAX_DEV];
static int g_nctx = 0;
static int g_rows_stay = 0;   /* options(ldwamd.sr_rows_stay): see ldwamd_mi_all_pairs */
#define g_ctx (g_ctxs[0])

static ldw_ctx *ctx_or_stop(void) {
    if (g_nctx == 0) {
        if (ldw_ctx_create(0, &g_ctxs[0]) != LDW_OK) error("ldweaver_amd: %s", ldw_last_error());
        g_nctx = 1;
    }
    return g_ctx;
}
#define CHK(call) do { if ((call) != LDW_OK) error("ldweaver_amd: %s", ldw_last_error()); } while (0)

/* devices: INTSXP of HIP device ids, one context each (existing contexts are destroyed first: call it before the alignment is set) */
SEXP ldwamd_set_devices(SEXP devices) {
    const R_xlen_t n = XLENGTH(devices);
    if (n < 1 || n > LDWAMD_MAX_DEV) error("ldweaver_amd: 1..%d devices", LDWAMD_MAX_DEV);
    for (int k = 0; k < g_nctx; ++k) { ldw_ctx_destroy(g_ctxs[k]); g_ctxs[k] = NULL; }
    g_nctx = 0;
    for (R_xlen_t k = 0; k < n; ++k) {
        if (ldw_ctx_create(INTEGER(devices)[k], &g_ctxs[k]) != LDW_OK) error("ldweaver_amd: device %d: %s", INTEGER(devices)[k], ldw_last_error());
        g_nctx = (int)k + 1;
    }
    return ScalarInteger(g_nctx);
}

/* ldwamd_release(): destroy every context and give the library's pooled device blocks and host buffers back (ldw_host_trim): between jobs of a long R
 * session, or before another package allocates on the same GPUs.  Returns the bytes released. */
SEXP ldwamd_release(void) {
    int64_t n = 0;
    for (int k = 0; k < g_nctx; ++k) { ldw_ctx_destroy(g_ctxs[k]); g_ctxs[k] = NULL; }
    g_nctx = 0;
    CHK(ldw_host_trim(NULL, &n));
    return ScalarReal((double)n);
}

/* .ACGTN2num(nv, cv, ncores): nv REALSXP 5 x L mutated in place, returns R_NilValue (src/RcppExports.cpp:16-25) */
SEXP ldwamd_ACGTN2num(SEXP nv, SEXP cv, SEXP ncores) {
    const R_xlen_t L = XLENGTH(cv);
    char *ref = (char *)R_alloc((size_t)L + 1, 1);
    for (R_xlen_t c = 0; c < L; ++c) ref[c] = CHAR(STRING_ELT(cv, c))[0];  /* as<char>(cv[c]) */
    CHK(ldw_acgtn2num(ctx_or_stop(), REAL(nv), ref, (int64_t)L, asInteger(ncores)));
    return R_NilValue;
}

/* states: RAWSXP L x N in ROW-major order (built by ldwamd_states_from_snpdat in the .R file) */
SEXP ldwamd_set_alignment(SEXP states, SEXP L, SEXP N) {
    ctx_or_stop();
    for (int k = 0; k < g_nctx; ++k) CHK(ldw_set_alignment(g_ctxs[k], RAW(states), (int64_t)asReal(L), (int64_t)asReal(N), 0));   /* replicated: <= 5 GB of 288 per GPU */
    return R_NilValue;
}

/* r04: per-block buffers of the all-pairs loop sized and pinned on a side thread (call after ldwamd_set_alignment; optional) */
SEXP ldwamd_ctx_reserve(SEXP L, SEXP N, SEXP max_blk_sz) {
    ctx_or_stop();
    for (int k = 0; k < g_nctx; ++k) CHK(ldw_ctx_reserve(g_ctxs[k], (int64_t)asReal(L), (int64_t)asReal(N), (int64_t)asReal(max_blk_sz)));
    return R_NilValue;
}

/* r04: spans on / off and their length (execution option: results do not depend on it) */
SEXP ldwamd_set_span(SEXP on, SEXP max_blocks) {
    CHK(ldw_set_span(ctx_or_stop(), asInteger(on), asInteger(max_blocks)));
    return R_NilValue;
}

/* estimate_Hamming_distance_weights core: thresh = as.integer(nsnp*threshold) computed in R */
SEXP ldwamd_hamming_weights(SEXP thresh, SEXP N) {
    SEXP out = PROTECT(allocVector(REALSXP, (R_xlen_t)asReal(N)));
    ctx_or_stop();
    if (g_nctx > 1) CHK(ldw_hamming_weights_multi(g_ctxs, g_nctx, asInteger(thresh), REAL(out)));   /* one strip of the comparison per device, same integers */
    else CHK(ldw_hamming_weights(g_ctx, asInteger(thresh), REAL(out), NULL));
    UNPROTECT(1);
    return out;
}

SEXP ldwamd_set_weights(SEXP hdw) {
    ctx_or_stop();
    for (int k = 0; k < g_nctx; ++k) CHK(ldw_set_weights(g_ctxs[k], REAL(hdw), (int64_t)XLENGTH(hdw), 0));
    return R_NilValue;
}

/* r: REALSXP[L]; uqe: RAWSXP L x 5 row-major; POS, paint: INTSXP[L]; g: scalar */
SEXP ldwamd_set_snp_meta(SEXP r, SEXP uqe, SEXP POS, SEXP paint, SEXP g) {
    ctx_or_stop();
    for (int k = 0; k < g_nctx; ++k) CHK(ldw_set_snp_meta(g_ctxs[k], REAL(r), RAW(uqe), INTEGER(POS), INTEGER(paint), asReal(g)));
    return R_NilValue;
}

SEXP ldwamd_set_sr_rows_stay(SEXP on) {
    g_rows_stay = asLogical(on) == TRUE;
    return ScalarLogical(g_rows_stay);
}

/* blocks: INTSXP 4 x nb (column-major == [nb][4] row-major); returns list(sr = list(a, b, MI), lr = ..., stats) */
SEXP ldwamd_mi_all_pairs(SEXP blocks, SEXP sr_dist, SEXP lr_retain, SEXP lr_approx, SEXP sr_only, SEXP quirk) {
    ldw_ctx *c = ctx_or_stop();
    const int64_t nb = XLENGTH(blocks) / 4;
    ldw_mi_params p;
    memset(&p, 0, sizeof(p));
    p.sr_dist = asReal(sr_dist);
    p.lr_retain_links = asReal(lr_retain);
    p.lr_links_approx = asReal(lr_approx);
    p.sr_only = asLogical(sr_only);
    p.quirk_mode = asInteger(quirk);
    p.keep_sr = 1;
    /* r05: the block loop of R/computePairwiseMI.R:103-116 over every device of options(ldwamd.devices): dealt, run and gathered into
     * context 0 inside the library; what follows reads context 0 exactly as after a single-device pass.  options(ldwamd.sr_rows_stay = TRUE):
     * only the long-range table is gathered, every device keeps the short-range rows it computed (the returned sr list is EMPTY) and
     * mergeNsort_sr_links_device runs the model over the devices (ldw_sr_*_multi) */
    const int rows_stay = g_rows_stay && g_nctx > 1 && !p.sr_only;
    if (rows_stay) p.flags |= LDW_MI_SR_ROWS_STAY;
    if (g_nctx > 1) CHK(ldw_mi_all_pairs_multi(g_ctxs, g_nctx, INTEGER(blocks), nb, &p, NULL, NULL));
    else CHK(ldw_mi_all_pairs(c, INTEGER(blocks), nb, &p, 1));
    SEXP res = PROTECT(allocVector(VECSXP, 3));
    for (int which = 0; which < 2; ++which) {
        int64_t n = 0;
        CHK(ldw_links_count(c, which, &n));
        if (which == 0 && rows_stay) n = 0;   /* (context 0 holds its own share only: nothing of it goes to R) */
        SEXP a = PROTECT(allocVector(INTSXP, (R_xlen_t)n)), b = PROTECT(allocVector(INTSXP, (R_xlen_t)n));
        SEXP mi = PROTECT(allocVector(REALSXP, (R_xlen_t)n));
        if (n > 0) CHK(ldw_links_fetch(c, which, INTEGER(a), INTEGER(b), REAL(mi), n, 0));
        SEXP t = PROTECT(allocVector(VECSXP, 3));
        SET_VECTOR_ELT(t, 0, a); SET_VECTOR_ELT(t, 1, b); SET_VECTOR_ELT(t, 2, mi);
        SET_VECTOR_ELT(res, which, t);
        UNPROTECT(4);
    }
    SEXP thr = PROTECT(allocVector(REALSXP, (R_xlen_t)nb));
    CHK(ldw_block_stats(c, nb, NULL, NULL, NULL, REAL(thr)));
    SET_VECTOR_ELT(res, 2, thr);
    UNPROTECT(2);
    return res;
}

/* runARACNE core (R/io_functions.R:101-164): logical vector */
SEXP ldwamd_aracne(SEXP cp1, SEXP cp2, SEXP cmi, SEXP fp1, SEXP fp2, SEXP fmi) {
    const R_xlen_t n = XLENGTH(cp1);
    SEXP out = PROTECT(allocVector(LGLSXP, n));
    unsigned char *flags = (unsigned char *)R_alloc((size_t)n + 1, 1);
    CHK(ldw_aracne(NULL, REAL(cp1), REAL(cp2), REAL(cmi), (int64_t)n, REAL(fp1), REAL(fp2), REAL(fmi), (int64_t)XLENGTH(fp1), flags));
    for (R_xlen_t i = 0; i < n; ++i) LOGICAL(out)[i] = flags[i];
    UNPROTECT(1);
    return out;
}

/* ---- mergeNsort_sr_links / runARACNE on the device-resident sr table (R/computePairwiseMI.R:400-495) ---- */
/* list(q_lo, q_hi, n): nclust x S matrices stored row-major (read them with matrix(., ncol = S, byrow = TRUE)) */
SEXP ldwamd_sr_len_quantiles(SEXP nclust, SEXP sr_dist, SEXP prob) {
    ldw_ctx *c = ctx_or_stop();
    const int nc = asInteger(nclust);
    const int32_t S = (int32_t)ceil(asReal(sr_dist)) - 1;
    SEXP res = PROTECT(allocVector(VECSXP, 3));
    SEXP lo = PROTECT(allocVector(REALSXP, (R_xlen_t)nc * S)), hi = PROTECT(allocVector(REALSXP, (R_xlen_t)nc * S));
    int64_t *n = (int64_t *)R_alloc((size_t)nc * S, sizeof(int64_t));
    (void)c;   /* (the _multi forms are the one-table calls on context 0 unless the last pass left the rows on their devices) */
    CHK(ldw_sr_len_quantiles_multi(g_ctxs, g_nctx, nc, asReal(sr_dist), asReal(prob), S, REAL(lo), REAL(hi), n));
    SEXP nn = PROTECT(allocVector(REALSXP, (R_xlen_t)nc * S));
    for (R_xlen_t i = 0; i < (R_xlen_t)nc * S; ++i) REAL(nn)[i] = (double)n[i];
    SET_VECTOR_ELT(res, 0, lo); SET_VECTOR_ELT(res, 1, hi); SET_VECTOR_ELT(res, 2, nn);
    UNPROTECT(4);
    return res;
}

/* mean_dist: nclust x S row-major doubles (NA beyond the fitted lens) -> nclust x 5 row-major sufficient statistics */
SEXP ldwamd_sr_excess_stats(SEXP nclust, SEXP mean_dist) {
    ldw_ctx *c = ctx_or_stop();
    const int nc = asInteger(nclust);
    SEXP out = PROTECT(allocVector(REALSXP, (R_xlen_t)nc * 5));
    (void)c;
    CHK(ldw_sr_excess_stats_multi(g_ctxs, g_nctx, nc, (int32_t)(XLENGTH(mean_dist) / nc), REAL(mean_dist), REAL(out)));
    UNPROTECT(1);
    return out;
}

/* shape: nclust x 3 row-major (shape1, shape2, lbeta).  Returns the kept links with their ARACNE flags:
 * list(row, a, b, MI, clust_c, first_clust, dup, srp_max, ARACNE) in no particular order. */
SEXP ldwamd_sr_pvalues_aracne(SEXP nclust, SEXP mean_dist, SEXP shape, SEXP srp_cutoff, SEXP run_aracne) {
    ldw_ctx *c = ctx_or_stop();
    const int nc = asInteger(nclust);
    int64_t n_red = 0, n_pool = 0;
    double min_mi = 0;
    CHK(ldw_sr_pvalues_multi(g_ctxs, g_nctx, nc, (int32_t)(XLENGTH(mean_dist) / nc), REAL(mean_dist), REAL(shape), asReal(srp_cutoff), &n_red, &n_pool, &min_mi));
    const R_xlen_t n = (R_xlen_t)n_red;
    SEXP res = PROTECT(allocVector(VECSXP, 9));
    SEXP row = PROTECT(allocVector(REALSXP, n)), a = PROTECT(allocVector(INTSXP, n)), b = PROTECT(allocVector(INTSXP, n));
    SEXP mi = PROTECT(allocVector(REALSXP, n)), cc = PROTECT(allocVector(INTSXP, n)), fc = PROTECT(allocVector(INTSXP, n));
    SEXP dup = PROTECT(allocVector(LGLSXP, n)), srp = PROTECT(allocVector(REALSXP, n)), ar = PROTECT(allocVector(REALSXP, n));
    int64_t *r64 = (int64_t *)R_alloc((size_t)n + 1, sizeof(int64_t));
    unsigned char *d8 = (unsigned char *)R_alloc((size_t)n + 1, 1), *f8 = (unsigned char *)R_alloc((size_t)n + 1, 1);
    CHK(ldw_sr_reduced_fetch(c, n_red, r64, INTEGER(a), INTEGER(b), REAL(mi), INTEGER(cc), INTEGER(fc), d8, REAL(srp)));
    memset(f8, 1, (size_t)n + 1);
    if (asLogical(run_aracne)) CHK(ldw_aracne_device(c, n_red, f8));
    for (R_xlen_t i = 0; i < n; ++i) {
        REAL(row)[i] = (double)r64[i] + 1;
        LOGICAL(dup)[i] = d8[i];
        REAL(ar)[i] = f8[i];
    }
    SEXP parts[9] = {row, a, b, mi, cc, fc, dup, srp, ar};
    for (int k = 0; k < 9; ++k) SET_VECTOR_ELT(res, k, parts[k]);
    UNPROTECT(10);
    return res;
}

/* analyse_long_range_links (R/lr_analyser.R:72-111) on the device-resident tables:
 * list(row, a, b, MI, ARACNE, q13, thresholds, fallback) with the outlier links in lr-table order. */
SEXP ldwamd_lr_tukey_aracne(SEXP min_links, SEXP sr_a, SEXP sr_b, SEXP sr_mi) {
    /* sr_a / sr_b (INTSXP, 0-based SNP index of pos2 / pos1) and sr_mi (REALSXP): the rows of sr_links.tsv (sr_links_red) */
    ldw_ctx *c = ctx_or_stop();
    double q13[2], thr[2];
    int fallback = 0;
    int64_t n_red = 0, n_pool = 0;
    if (XLENGTH(sr_a) != XLENGTH(sr_mi) || XLENGTH(sr_b) != XLENGTH(sr_mi)) error("ldweaver_amd: ragged short-range table");
    CHK(ldw_lr_tukey(c, (int64_t)asReal(min_links), INTEGER(sr_a), INTEGER(sr_b), REAL(sr_mi), (int64_t)XLENGTH(sr_mi), q13, thr,
                     &fallback, &n_red, &n_pool));
    const R_xlen_t n = (R_xlen_t)n_red;
    SEXP res = PROTECT(allocVector(VECSXP, 8));
    SEXP row = PROTECT(allocVector(REALSXP, n)), a = PROTECT(allocVector(INTSXP, n)), b = PROTECT(allocVector(INTSXP, n));
    SEXP mi = PROTECT(allocVector(REALSXP, n)), ar = PROTECT(allocVector(REALSXP, n));
    SEXP q = PROTECT(allocVector(REALSXP, 2)), t = PROTECT(allocVector(REALSXP, 2)), fb = PROTECT(ScalarLogical(fallback));
    int64_t *r64 = (int64_t *)R_alloc((size_t)n + 1, sizeof(int64_t));
    unsigned char *f8 = (unsigned char *)R_alloc((size_t)n + 1, 1);
    CHK(ldw_lr_reduced_fetch(c, n_red, r64, INTEGER(a), INTEGER(b), REAL(mi)));
    memset(f8, 1, (size_t)n + 1);
    if (n > 0) CHK(ldw_aracne_device(c, n_red, f8));
    for (R_xlen_t i = 0; i < n; ++i) {
        REAL(row)[i] = (double)r64[i] + 1;
        REAL(ar)[i] = f8[i];
    }
    REAL(q)[0] = q13[0]; REAL(q)[1] = q13[1]; REAL(t)[0] = thr[0]; REAL(t)[1] = thr[1];
    SEXP parts[8] = {row, a, b, mi, ar, q, t, fb};
    for (int k = 0; k < 8; ++k) SET_VECTOR_ELT(res, k, parts[k]);
    UNPROTECT(9);
    return res;
}

/* genomewide_LDMap (R/LDSummaryPlot.R:55-109): the reduced, log10-scaled, rescaled B x B matrix `htm`;
 * attributes n_pos, reducer.  reducer = 0: the reference's default; from = to = 0: genome-wide. */
SEXP ldwamd_ldmap(SEXP reducer, SEXP from, SEXP to) {
    ldw_ctx *c = ctx_or_stop();
    int64_t n_pos = 0;
    int32_t r = 0, B = 0;
    CHK(ldw_ldmap(c, asInteger(reducer), asInteger(from), asInteger(to), &n_pos, &r, &B, NULL, 0));
    SEXP htm = PROTECT(allocMatrix(REALSXP, B, B));
    CHK(ldw_ldmap(c, asInteger(reducer), asInteger(from), asInteger(to), &n_pos, &r, &B, REAL(htm), (int64_t)B * B));
    setAttrib(htm, install("n_pos"), ScalarReal((double)n_pos));
    setAttrib(htm, install("reducer"), ScalarInteger(r));
    UNPROTECT(1);
    return htm;
}

/* ---- native-level twins of the reference's own .Call table (src/RcppExports.cpp:154-160): same symbol names, arity, argument
 * types and return values, so that LDWeaver's UNCHANGED R code (.fastHadamard R/computePairwiseMI.R:396, .compareToRow :374,
 * .vecPosMatch / .compareTriplet / .fast_intersect R/io_functions.R:125-155, .ACGTN2num R/computePairwiseMI.R:256-259) binds to
 * this library when the shim is built as the package's native library (or loaded in its place). ---- */

/* void fastHadamard(MIt, den, uq_t, pxy_t, pxpy_t, RXY, pXrX, pYrY, ncores): MIt updated IN PLACE over the linear index
 * (src/computeMI.cpp:11-21); ncores accepted and ignored */
SEXP _LDWeaver_fastHadamard(SEXP MIt, SEXP den, SEXP uq_t, SEXP pxy_t, SEXP pxpy_t, SEXP RXY, SEXP pXrX, SEXP pYrY, SEXP ncores) {
    const R_xlen_t n = XLENGTH(MIt);
    SEXP ops[7] = {den, uq_t, pxy_t, pxpy_t, RXY, pXrX, pYrY};
    for (int k = 0; k < 7; ++k)
        if (TYPEOF(ops[k]) != REALSXP || XLENGTH(ops[k]) < n) error("ldweaver_amd: .fastHadamard operand %d is not a numeric matrix of MIt's size", k + 2);
    if (TYPEOF(MIt) != REALSXP) error("ldweaver_amd: .fastHadamard: MIt must be a numeric matrix");
    (void)ncores;
    CHK(ldw_fast_hadamard(ctx_or_stop(), REAL(MIt), REAL(den), REAL(uq_t), REAL(pxy_t), REAL(pxpy_t), REAL(RXY), REAL(pXrX), REAL(pYrY),
                          (int64_t)n, 0));
    return R_NilValue;
}

/* void ACGTN2num(nv, cv, ncores) (src/ACGTN2num_parallel.cpp:10-43) */
SEXP _LDWeaver_ACGTN2num(SEXP nv, SEXP cv, SEXP ncores) { return ldwamd_ACGTN2num(nv, cv, ncores); }

/* LogicalVector compareToRow(NumericMatrix x, NumericVector y) (src/computeMI.cpp:25-41) */
SEXP _LDWeaver_compareToRow(SEXP x, SEXP y) {
    if (TYPEOF(x) != REALSXP || TYPEOF(y) != REALSXP) error("ldweaver_amd: .compareToRow takes a numeric matrix and a numeric vector");
    SEXP dim = getAttrib(x, R_DimSymbol);
    if (dim == R_NilValue || LENGTH(dim) != 2) error("ldweaver_amd: .compareToRow: x must be a matrix");
    const int64_t nr = INTEGER(dim)[0], nc = INTEGER(dim)[1];
    unsigned char *f8 = (unsigned char *)R_alloc((size_t)nr + 1, 1);
    CHK(ldw_compare_to_row(REAL(x), nr, nc, REAL(y), (int64_t)XLENGTH(y), f8));
    SEXP out = PROTECT(allocVector(LGLSXP, (R_xlen_t)nr));
    for (int64_t j = 0; j < nr; ++j) LOGICAL(out)[j] = f8[j] ? TRUE : FALSE;
    UNPROTECT(1);
    return out;
}

/* NumericVector vecPosMatch(NumericVector x, NumericVector y) (src/computeMI.cpp:44-59) */
SEXP _LDWeaver_vecPosMatch(SEXP x, SEXP y) {
    if (TYPEOF(x) != REALSXP || TYPEOF(y) != REALSXP) error("ldweaver_amd: .vecPosMatch takes numeric vectors");
    SEXP out = PROTECT(allocVector(REALSXP, XLENGTH(x)));
    CHK(ldw_vec_pos_match(REAL(x), (int64_t)XLENGTH(x), REAL(y), (int64_t)XLENGTH(y), REAL(out)));
    UNPROTECT(1);
    return out;
}

/* bool compareTriplet(NumericVector MI0X, NumericVector MI0Z, double MI0) (src/computeMI.cpp:63-77) */
SEXP _LDWeaver_compareTriplet(SEXP MI0X, SEXP MI0Z, SEXP MI0) {
    if (TYPEOF(MI0X) != REALSXP || TYPEOF(MI0Z) != REALSXP || XLENGTH(MI0X) != XLENGTH(MI0Z)) error("ldweaver_amd: .compareTriplet: MI0X and MI0Z must be numeric vectors of one length");
    int r = 0;
    CHK(ldw_compare_triplet(REAL(MI0X), REAL(MI0Z), (int64_t)XLENGTH(MI0X), asReal(MI0), &r));
    return ScalarLogical(r);
}

/* std::vector<int> fast_intersect(std::vector<int> A, std::vector<int> B) (src/fintersect.cpp:6-32); Rcpp coerces numeric input
 * to int the same way (truncation) */
SEXP _LDWeaver_fast_intersect(SEXP A, SEXP B) {
    SEXP a = PROTECT(coerceVector(A, INTSXP)), b = PROTECT(coerceVector(B, INTSXP));
    const int64_t na = XLENGTH(a), nb = XLENGTH(b), cap = na < nb ? na : nb;
    int32_t *buf = (int32_t *)R_alloc((size_t)cap + 1, sizeof(int32_t));
    int64_t n = 0;
    CHK(ldw_fast_intersect(INTEGER(a), na, INTEGER(b), nb, buf, &n));
    SEXP out = PROTECT(allocVector(INTSXP, (R_xlen_t)n));
    if (n > 0) memcpy(INTEGER(out), buf, (size_t)n * sizeof(int32_t));
    UNPROTECT(3);
    return out;
}

/* lr_links.tsv / raw sr rows straight from the device-resident table by the library's threaded writer: the bytes write.table(x, file,
 * append = T, quote = F, row.names = F, col.names = F, sep = '\t') produces for the same rows (R/computePairwiseMI.R:362).  which: 0 sr, 1 lr.
 * Returns the number of rows written. */
SEXP ldwamd_write_links_tsv(SEXP which, SEXP path) {
    int64_t rows = 0, bytes = 0;
    CHK(ldw_write_links_tsv(ctx_or_stop(), asInteger(which), CHAR(STRING_ELT(path, 0)), 1, 0, &rows, &bytes));
    return ScalarReal((double)rows);
}

/* r04: the same table written BESIDE the next calls — _begin fetches it from the device and returns, host threads format and write it while
 * R runs mergeNsort_sr_links (which does not touch the long-range table: R/computePairwiseMI.R:119-126); _end waits, returns the rows written */
SEXP ldwamd_write_links_tsv_begin(SEXP which, SEXP path) {
    CHK(ldw_write_links_tsv_begin(ctx_or_stop(), asInteger(which), CHAR(STRING_ELT(path, 0)), 1, 0));
    return R_NilValue;
}
SEXP ldwamd_write_links_tsv_end(void) {
    int64_t rows = 0, bytes = 0;
    CHK(ldw_write_links_tsv_end(ctx_or_stop(), &rows, &bytes));
    return ScalarReal((double)rows);
}

/* r05: lr_links.tsv appended WHILE ldwamd_mi_all_pairs runs, item by item, like the reference's per-block write.table(append = T)
 * (R/computePairwiseMI.R:362): _begin before the pass (single device only: the multi-device gather reorders rows), _end after it — also on
 * error: what the pass finished is on disk.  _end returns c(rows, blocks whose rows are in the file). */
SEXP ldwamd_lr_stream_begin(SEXP path) {
    ctx_or_stop();
    if (g_nctx > 1) error("ldweaver_amd: lr_links.tsv streaming needs a single device (options(ldwamd.devices) has %d)", g_nctx);
    CHK(ldw_lr_stream_begin(g_ctx, CHAR(STRING_ELT(path, 0)), 1, 0));
    return R_NilValue;
}
SEXP ldwamd_lr_stream_end(void) {
    int64_t rows = 0, bytes = 0, blocks = 0;
    CHK(ldw_lr_stream_end(ctx_or_stop(), &rows, &bytes, &blocks));
    SEXP out = PROTECT(allocVector(REALSXP, 2));
    REAL(out)[0] = (double)rows;
    REAL(out)[1] = (double)blocks;
    UNPROTECT(1);
    return out;
}

/* any numeric data.frame's columns (INTSXP / REALSXP / LGLSXP-as-int) by the same writer; cols: a list of equally long vectors */
SEXP ldwamd_write_table_tsv(SEXP cols, SEXP path) {
    const int nc = (int)XLENGTH(cols);
    if (nc <= 0 || nc > 64) error("ldweaver_amd: 1..64 columns expected");
    const void *ptr[64];
    int32_t kind[64];
    const R_xlen_t n = XLENGTH(VECTOR_ELT(cols, 0));
    for (int k = 0; k < nc; ++k) {
        SEXP v = VECTOR_ELT(cols, k);
        if (XLENGTH(v) != n) error("ldweaver_amd: columns of different lengths");
        if (TYPEOF(v) == REALSXP) { kind[k] = LDW_COL_DOUBLE; ptr[k] = REAL(v); }
        else if (TYPEOF(v) == INTSXP) { kind[k] = LDW_COL_INT32; ptr[k] = INTEGER(v); }   /* (NA_integer_ is not handled: the link frames hold none) */
        else error("ldweaver_amd: column %d is neither integer nor double", k + 1);
    }
    int64_t bytes = 0;
    CHK(ldw_write_table_tsv(CHAR(STRING_ELT(path, 0)), 1, (int64_t)n, nc, kind, ptr, 0, &bytes));
    return ScalarReal((double)bytes);
}

static const R_CallMethodDef CallEntries[] = {
    {"_LDWeaver_ACGTN2num", (DL_FUNC)&_LDWeaver_ACGTN2num, 3},
    {"_LDWeaver_fastHadamard", (DL_FUNC)&_LDWeaver_fastHadamard, 9},
    {"_LDWeaver_compareToRow", (DL_FUNC)&_LDWeaver_compareToRow, 2},
    {"_LDWeaver_vecPosMatch", (DL_FUNC)&_LDWeaver_vecPosMatch, 2},
    {"_LDWeaver_compareTriplet", (DL_FUNC)&_LDWeaver_compareTriplet, 3},
    {"_LDWeaver_fast_intersect", (DL_FUNC)&_LDWeaver_fast_intersect, 2},
    {"ldwamd_lr_tukey_aracne", (DL_FUNC)&ldwamd_lr_tukey_aracne, 4},
    {"ldwamd_ldmap", (DL_FUNC)&ldwamd_ldmap, 3},
    {"ldwamd_set_sr_rows_stay", (DL_FUNC)&ldwamd_set_sr_rows_stay, 1},
    {"ldwamd_sr_len_quantiles", (DL_FUNC)&ldwamd_sr_len_quantiles, 3},
    {"ldwamd_sr_excess_stats", (DL_FUNC)&ldwamd_sr_excess_stats, 2},
    {"ldwamd_sr_pvalues_aracne", (DL_FUNC)&ldwamd_sr_pvalues_aracne, 5},
    {"ldwamd_ACGTN2num", (DL_FUNC)&ldwamd_ACGTN2num, 3},
    {"ldwamd_set_alignment", (DL_FUNC)&ldwamd_set_alignment, 3},
    {"ldwamd_ctx_reserve", (DL_FUNC)&ldwamd_ctx_reserve, 3},
    {"ldwamd_set_span", (DL_FUNC)&ldwamd_set_span, 2},
    {"ldwamd_set_devices", (DL_FUNC)&ldwamd_set_devices, 1},
    {"ldwamd_release", (DL_FUNC)&ldwamd_release, 0},
    {"ldwamd_hamming_weights", (DL_FUNC)&ldwamd_hamming_weights, 2},
    {"ldwamd_set_weights", (DL_FUNC)&ldwamd_set_weights, 1},
    {"ldwamd_set_snp_meta", (DL_FUNC)&ldwamd_set_snp_meta, 5},
    {"ldwamd_mi_all_pairs", (DL_FUNC)&ldwamd_mi_all_pairs, 6},
    {"ldwamd_aracne", (DL_FUNC)&ldwamd_aracne, 6},
    {"ldwamd_write_links_tsv", (DL_FUNC)&ldwamd_write_links_tsv, 2},
    {"ldwamd_write_links_tsv_begin", (DL_FUNC)&ldwamd_write_links_tsv_begin, 2},
    {"ldwamd_write_links_tsv_end", (DL_FUNC)&ldwamd_write_links_tsv_end, 0},
    {"ldwamd_lr_stream_begin", (DL_FUNC)&ldwamd_lr_stream_begin, 1},
    {"ldwamd_lr_stream_end", (DL_FUNC)&ldwamd_lr_stream_end, 0},
    {"ldwamd_write_table_tsv", (DL_FUNC)&ldwamd_write_table_tsv, 2},
    {NULL, NULL, 0}};

void R_init_ldweaver_amd_shim(DllInfo *dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
