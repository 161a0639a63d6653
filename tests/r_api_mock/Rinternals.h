/* Declaration-only stand-in for the part of R's public C API that r_shim/ldweaver_amd_shim.c uses — test infrastructure, NOT R: it lets
 * tests/test_cabi_and_host.py type-check the shim with gcc (the image has no R) and list the ldw_* symbols it needs.  Names and argument
 * types follow "Writing R Extensions" (the API is the interface a .Call shim is written against); nothing here has a definition. */
#ifndef LDW_TEST_R_API_MOCK_RINTERNALS_H
#define LDW_TEST_R_API_MOCK_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned char Rbyte;
typedef enum { FALSE = 0, TRUE } Rboolean;
typedef unsigned int SEXPTYPE;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define RAWSXP 24
extern SEXP R_NilValue, R_DimSymbol, R_NamesSymbol;
int TYPEOF(SEXP x);
int LENGTH(SEXP x);
R_xlen_t XLENGTH(SEXP x);
double *REAL(SEXP x);
int *INTEGER(SEXP x);
int *LOGICAL(SEXP x);
Rbyte *RAW(SEXP x);
const char *CHAR(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_coerceVector(SEXP x, SEXPTYPE type);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int Rf_asLogical(SEXP x);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_ScalarReal(double v);
SEXP Rf_ScalarLogical(int v);
SEXP Rf_install(const char *name);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP v);
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP Rf_mkChar(const char *s);
SEXP Rf_mkString(const char *s);
Rboolean Rf_isNull(SEXP x);
void Rf_error(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void Rf_warning(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
char *R_alloc(size_t n, int size);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
#define allocVector Rf_allocVector
#define allocMatrix Rf_allocMatrix
#define coerceVector Rf_coerceVector
#define asInteger Rf_asInteger
#define asReal Rf_asReal
#define asLogical Rf_asLogical
#define ScalarInteger Rf_ScalarInteger
#define ScalarReal Rf_ScalarReal
#define ScalarLogical Rf_ScalarLogical
#define install Rf_install
#define setAttrib Rf_setAttrib
#define getAttrib Rf_getAttrib
#define mkChar Rf_mkChar
#define mkString Rf_mkString
#define isNull Rf_isNull
#define error Rf_error
#define warning Rf_warning
#endif
