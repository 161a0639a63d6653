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
#include <stdint.h>
#include <string.h>

#include "ldweaver_amd.h"

static ldw_ctx *g_ctx = NULL;

static ldw_ctx *ctx_or_stop(void) {
    if (!g_ctx && ldw_ctx_create(0, &g_ctx) != LDW_OK) error("ldweaver_amd: %s", ldw_last_error());
    return g_ctx;
}
#define CHK(call) do { if ((call) != LDW_OK) error("ldweaver_amd: %s", ldw_last_error()); } while (0)

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
    CHK(ldw_set_alignment(ctx_or_stop(), RAW(states), (int64_t)asReal(L), (int64_t)asReal(N), 0));
    return R_NilValue;
}

/* estimate_Hamming_distance_weights core: thresh = as.integer(nsnp*threshold) computed in R */
SEXP ldwamd_hamming_weights(SEXP thresh, SEXP N) {
    SEXP out = PROTECT(allocVector(REALSXP, (R_xlen_t)asReal(N)));
    CHK(ldw_hamming_weights(ctx_or_stop(), asInteger(thresh), REAL(out), NULL));
    UNPROTECT(1);
    return out;
}

SEXP ldwamd_set_weights(SEXP hdw) {
    CHK(ldw_set_weights(ctx_or_stop(), REAL(hdw), (int64_t)XLENGTH(hdw), 0));
    return R_NilValue;
}

/* r: REALSXP[L]; uqe: RAWSXP L x 5 row-major; POS, paint: INTSXP[L]; g: scalar */
SEXP ldwamd_set_snp_meta(SEXP r, SEXP uqe, SEXP POS, SEXP paint, SEXP g) {
    CHK(ldw_set_snp_meta(ctx_or_stop(), REAL(r), RAW(uqe), INTEGER(POS), INTEGER(paint), asReal(g)));
    return R_NilValue;
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
    CHK(ldw_mi_all_pairs(c, INTEGER(blocks), nb, &p, 1));
    SEXP res = PROTECT(allocVector(VECSXP, 3));
    for (int which = 0; which < 2; ++which) {
        int64_t n = 0;
        CHK(ldw_links_count(c, which, &n));
        SEXP a = PROTECT(allocVector(INTSXP, (R_xlen_t)n)), b = PROTECT(allocVector(INTSXP, (R_xlen_t)n));
        SEXP mi = PROTECT(allocVector(REALSXP, (R_xlen_t)n));
        CHK(ldw_links_fetch(c, which, INTEGER(a), INTEGER(b), REAL(mi), n, 0));
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

static const R_CallMethodDef CallEntries[] = {
    {"ldwamd_ACGTN2num", (DL_FUNC)&ldwamd_ACGTN2num, 3},
    {"ldwamd_set_alignment", (DL_FUNC)&ldwamd_set_alignment, 3},
    {"ldwamd_hamming_weights", (DL_FUNC)&ldwamd_hamming_weights, 2},
    {"ldwamd_set_weights", (DL_FUNC)&ldwamd_set_weights, 1},
    {"ldwamd_set_snp_meta", (DL_FUNC)&ldwamd_set_snp_meta, 5},
    {"ldwamd_mi_all_pairs", (DL_FUNC)&ldwamd_mi_all_pairs, 6},
    {"ldwamd_aracne", (DL_FUNC)&ldwamd_aracne, 6},
    {NULL, NULL, 0}};

void R_init_ldweaver_amd_shim(DllInfo *dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
