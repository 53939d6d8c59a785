/* A stand-in R runtime for EXECUTING integration/r/src/machisplin_shim.c where R is not installed (the build image, the GPU
 * box): the small, documented subset of R's C API the shim uses ("Writing R Extensions" 5.9-5.13; declarations in
 * integration/r/check/Rinternals.h), implemented just far enough that every mhsr_* entry point can be called with real
 * REALSXP / INTSXP / VECSXP / EXTPTRSXP objects, that Rf_error() unwinds (longjmp) to the caller the way R's does, and that
 * external-pointer finalizers run.  TEST INFRASTRUCTURE ONLY (tests/test_r_shim_exec.py drives it through ctypes); nothing
 * here is R's code and nothing here ships.  What it cannot show: R's own garbage collector, PROTECT discipline (a no-op
 * here) and terra's objects. */
#include <limits.h>
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <R.h>
#include <Rinternals.h>

struct SEXPREC {
    SEXPTYPE type;
    R_xlen_t len;
    void *data;                 /* double[] / int[] / SEXP[] / char* */
    int nrow, ncol, is_matrix;
    void *ext;                  /* EXTPTRSXP address */
    R_CFinalizer_t fin;
};

static struct SEXPREC nil_rec = { NILSXP, 0, NULL, 0, 0, 0, NULL, NULL };
SEXP R_NilValue = &nil_rec;
double R_NaN, R_NaReal;
int R_NaInt = INT_MIN;

static jmp_buf *active_jmp = NULL;
static char errmsg[1024] = "", warnmsg[1024] = "";
static int finalizers_run = 0, n_protected = 0;
static void **ralloc_list = NULL;
static size_t ralloc_n = 0, ralloc_cap = 0;

__attribute__((constructor)) static void rstub_boot(void) {
    union { double d; uint64_t u; } na;
    na.u = 0x7FF00000000007A2ULL;            /* R's NA_real_: a NaN whose low word is 1954 */
    R_NaReal = na.d;
    R_NaN = NAN;
}

int R_IsNA(double x) {
    union { double d; uint64_t u; } v;
    v.d = x;
    return isnan(x) && (uint32_t)(v.u & 0xFFFFFFFFu) == 1954u;
}
int R_IsNaN(double x) { return isnan(x) && !R_IsNA(x); }
int R_finite(double x) { return isfinite(x); }

static SEXP new_rec(SEXPTYPE type, R_xlen_t n) {
    SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
    size_t esz = type == REALSXP ? sizeof(double) : (type == INTSXP || type == LGLSXP) ? sizeof(int) : type == VECSXP ? sizeof(SEXP) : 1;
    s->type = type; s->len = n;
    s->data = calloc((size_t)(n > 0 ? n : 1), esz);
    if (type == VECSXP) for (R_xlen_t i = 0; i < n; ++i) ((SEXP *)s->data)[i] = R_NilValue;
    return s;
}

double *REAL(SEXP x) { if (x->type != REALSXP) Rf_error("REAL() can only be applied to a 'numeric', not type %u", x->type); return (double *)x->data; }
int *INTEGER(SEXP x) { if (x->type != INTSXP && x->type != LGLSXP) Rf_error("INTEGER() can only be applied to a 'integer', not type %u", x->type); return (int *)x->data; }
int *LOGICAL(SEXP x) { return INTEGER(x); }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) {
    if (x->type != VECSXP) Rf_error("VECTOR_ELT() can only be applied to a 'list', not type %u", x->type);
    if (i < 0 || i >= x->len) Rf_error("VECTOR_ELT: index out of range");
    return ((SEXP *)x->data)[i];
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) {
    if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("SET_VECTOR_ELT: bad list or index");
    ((SEXP *)x->data)[i] = v;
    return v;
}
int TYPEOF(SEXP x) { return (int)x->type; }
SEXP Rf_protect(SEXP s) { ++n_protected; return s; }
void Rf_unprotect(int n) { n_protected -= n; }
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) { return new_rec(type, n); }
SEXP Rf_allocMatrix(SEXPTYPE type, int nr, int nc) {
    SEXP s = new_rec(type, (R_xlen_t)nr * nc);
    s->nrow = nr; s->ncol = nc; s->is_matrix = 1;
    return s;
}
int Rf_asInteger(SEXP x) {
    if (x->len < 1) return R_NaInt;
    if (x->type == INTSXP || x->type == LGLSXP) return ((int *)x->data)[0];
    if (x->type == REALSXP) { double v = ((double *)x->data)[0]; return isnan(v) ? R_NaInt : (int)v; }
    return R_NaInt;
}
double Rf_asReal(SEXP x) {
    if (x->len < 1) return R_NaReal;
    if (x->type == REALSXP) return ((double *)x->data)[0];
    if (x->type == INTSXP || x->type == LGLSXP) { int v = ((int *)x->data)[0]; return v == R_NaInt ? R_NaReal : (double)v; }
    return R_NaReal;
}
int Rf_length(SEXP x) { return (int)x->len; }
R_xlen_t Rf_xlength(SEXP x) { return x->len; }
int Rf_nrows(SEXP x) { return x->is_matrix ? x->nrow : (int)x->len; }
int Rf_ncols(SEXP x) { return x->is_matrix ? x->ncol : 1; }
Rboolean Rf_isNull(SEXP x) { return x->type == NILSXP ? TRUE : FALSE; }
Rboolean Rf_isReal(SEXP x) { return x->type == REALSXP ? TRUE : FALSE; }
Rboolean Rf_isMatrix(SEXP x) { return x->is_matrix ? TRUE : FALSE; }
SEXP Rf_ScalarInteger(int v) { SEXP s = new_rec(INTSXP, 1); ((int *)s->data)[0] = v; return s; }
SEXP Rf_ScalarReal(double v) { SEXP s = new_rec(REALSXP, 1); ((double *)s->data)[0] = v; return s; }
SEXP Rf_mkString(const char *c) { SEXP s = new_rec(STRSXP, 1); free(s->data); s->data = strdup(c); return s; }

void Rf_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(errmsg, sizeof(errmsg), fmt, ap);
    va_end(ap);
    if (active_jmp) longjmp(*active_jmp, 1);
    fprintf(stderr, "rstub: Rf_error outside rstub_call: %s\n", errmsg);
    abort();
}
void Rf_warning(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(warnmsg, sizeof(warnmsg), fmt, ap);
    va_end(ap);
}
char *R_alloc(size_t n, int size) {          /* transient storage: reclaimed when the .Call returns */
    void *p = calloc(n ? n : 1, (size_t)(size > 0 ? size : 1));
    if (ralloc_n == ralloc_cap) { ralloc_cap = ralloc_cap ? 2 * ralloc_cap : 64; ralloc_list = (void **)realloc(ralloc_list, ralloc_cap * sizeof(void *)); }
    ralloc_list[ralloc_n++] = p;
    return (char *)p;
}
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot) { (void)tag; (void)prot; SEXP s = new_rec(EXTPTRSXP, 0); s->ext = p; return s; }
void *R_ExternalPtrAddr(SEXP s) { if (s->type != EXTPTRSXP) Rf_error("R_ExternalPtrAddr: argument of type %u is not an external pointer", s->type); return s->ext; }
void R_ClearExternalPtr(SEXP s) { s->ext = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) { (void)onexit; s->fin = fun; }
void R_CheckUserInterrupt(void) {}

/* ------------------------------------------------------------------ what the test drives ---------- */
SEXP rstub_real(const double *v, R_xlen_t n) { SEXP s = new_rec(REALSXP, n); if (n) memcpy(s->data, v, sizeof(double) * (size_t)n); return s; }
SEXP rstub_real_matrix(const double *v, int nr, int nc) { SEXP s = Rf_allocMatrix(REALSXP, nr, nc); if (nr && nc) memcpy(s->data, v, sizeof(double) * (size_t)nr * nc); return s; }
SEXP rstub_int(const int *v, R_xlen_t n) { SEXP s = new_rec(INTSXP, n); if (n) memcpy(s->data, v, sizeof(int) * (size_t)n); return s; }
SEXP rstub_int_matrix(const int *v, int nr, int nc) { SEXP s = Rf_allocMatrix(INTSXP, nr, nc); if (nr && nc) memcpy(s->data, v, sizeof(int) * (size_t)nr * nc); return s; }
SEXP rstub_list(R_xlen_t n) { return new_rec(VECSXP, n); }
void rstub_list_set(SEXP l, R_xlen_t i, SEXP v) { ((SEXP *)l->data)[i] = v; }
SEXP rstub_list_get(SEXP l, R_xlen_t i) { return ((SEXP *)l->data)[i]; }
SEXP rstub_null(void) { return R_NilValue; }
double rstub_na_real(void) { return R_NaReal; }
int rstub_type(SEXP s) { return (int)s->type; }
R_xlen_t rstub_len(SEXP s) { return s->len; }
int rstub_nrow(SEXP s) { return s->nrow; }
int rstub_ncol(SEXP s) { return s->ncol; }
void *rstub_data(SEXP s) { return s->data; }
void *rstub_extptr(SEXP s) { return s->ext; }
const char *rstub_last_error(void) { return errmsg; }
int rstub_finalizers_run(void) { return finalizers_run; }
int rstub_protect_balance(void) { return n_protected; }

/* what R's garbage collector does to an unreachable object: the finalizer of an external pointer, then the memory.
 * Lists release their elements. */
void rstub_release(SEXP s) {
    if (!s || s == R_NilValue) return;
    if (s->type == VECSXP) for (R_xlen_t i = 0; i < s->len; ++i) rstub_release(((SEXP *)s->data)[i]);
    if (s->type == EXTPTRSXP && s->fin) { s->fin(s); ++finalizers_run; }
    free(s->data);
    free(s);
}

/* .Call(fn, args...): Rf_error() inside unwinds to here (NULL is returned, rstub_last_error() has the message);
 * R_alloc storage is reclaimed either way */
typedef SEXP (*fn0)(void);
SEXP rstub_call(void *fn, int nargs, SEXP *a) {
    jmp_buf jb;
    SEXP out = NULL;
    errmsg[0] = 0;
    n_protected = 0;
    active_jmp = &jb;
    if (setjmp(jb) == 0) {
        switch (nargs) {
            case 0: out = ((fn0)fn)(); break;
            case 1: out = ((SEXP (*)(SEXP))fn)(a[0]); break;
            case 2: out = ((SEXP (*)(SEXP, SEXP))fn)(a[0], a[1]); break;
            case 3: out = ((SEXP (*)(SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2]); break;
            case 4: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3]); break;
            case 5: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4]); break;
            case 6: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
            case 7: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
            case 8: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
            case 9: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
            case 10: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
            case 11: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
            default: snprintf(errmsg, sizeof(errmsg), "rstub_call: %d arguments not supported", nargs); out = NULL;
        }
    } else {
        out = NULL;
    }
    active_jmp = NULL;
    for (size_t i = 0; i < ralloc_n; ++i) free(ralloc_list[i]);
    ralloc_n = 0;
    return out;
}
