/* NOT R's header: see Rinternals.h in this directory. */
#ifndef MHS_CHECK_R_H
#define MHS_CHECK_R_H
#include <stdlib.h>
#include <string.h>
#include <math.h>
/* R_ext/Arith.h */
int R_IsNA(double);
int R_IsNaN(double);
int R_finite(double);
#define ISNA(x) R_IsNA(x)
#define ISNAN(x) (isnan(x) != 0)
#define R_FINITE(x) R_finite(x)
#endif
