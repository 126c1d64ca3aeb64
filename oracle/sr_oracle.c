/*
 * sr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see sr_oracle_body.h).
 * Builds liboracle.so exporting sr_oracle_*_f32 (reference-faithful fp32) and
 * sr_oracle_*_f64 (double-precision arbitration).  Build: `make -C oracle`.
 */
#include <math.h>
#include <stdlib.h>

#define REAL float
#define OUT_T float
#define SUFFIX _f32
#define FLOOR floorf
#define SQRT sqrtf
#include "sr_oracle_body.h"
#undef REAL
#undef OUT_T
#undef SUFFIX
#undef FLOOR
#undef SQRT

#define REAL double
#define OUT_T double
#define SUFFIX _f64
#define FLOOR floor
#define SQRT sqrt
#include "sr_oracle_body.h"

#ifdef _OPENMP
#include <omp.h>
int sr_oracle_num_threads(void) { return omp_get_max_threads(); }
void sr_oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
int sr_oracle_num_threads(void) { return 1; }
void sr_oracle_set_threads(int n) { (void)n; }
#endif
