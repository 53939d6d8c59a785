/* NOT R's header.  Declarations of the small, documented subset of R's C API that machisplin_shim.c uses ("Writing R
 * Extensions", sections 5.9-5.13), so that the shim can be put through a C compiler for a syntax / prototype check in an
 * image that has no R (tests/test_r_shim_compiles.py: gcc -fsyntax-only).  Building the real package uses R's own
 * <Rinternals.h>; nothing here is linked or shipped. */
#ifndef MHS_CHECK_RINTERNALS_H
#define MHS_CHECK_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
extern SEXP R_NilValue;
extern double R_NaN, R_NaReal;
double *REAL(SEXP x);
int *INTEGER(SEXP x);
int *LOGICAL(SEXP x);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
int TYPEOF(SEXP x);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_length(SEXP);
R_xlen_t Rf_xlength(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
Rboolean Rf_isNull(SEXP);
Rboolean Rf_isReal(SEXP);
Rboolean Rf_isMatrix(SEXP);
extern int R_NaInt;
#define NA_INTEGER R_NaInt
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_mkString(const char *);
void Rf_error(const char *, ...) __attribute__((noreturn));
void Rf_warning(const char *, ...);
char *R_alloc(size_t, int);
typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);
void R_CheckUserInterrupt(void);
typedef struct _DllInfo DllInfo;     /* R_ext/Rdynload.h */
#endif
