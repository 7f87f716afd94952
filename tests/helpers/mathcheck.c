/* mathcheck.c -- test helper: the product's scalar math (gmix_amd/csrc/gmx_math.h, host
 * compile) against the machine's libm, which is what the reference links
 * (mixer/sigmoid.cpp:5).  Built by tests/test_math.py with gcc -O2 -ffp-contract=off -fopenmp. */
#include <math.h>
#include <stdint.h>
#include "../../gmix_amd/csrc/gmx_math.h"

/* Compare gmx_expf with libm expf for every float whose bit pattern lies in [lo, hi]
 * (NaN results compare equal when both are NaN).  Returns the mismatch count; the first
 * few offending bit patterns go to bad[0..nbad). */
uint64_t gmx_check_expf_range(uint64_t lo, uint64_t hi, uint32_t* bad, int nbad) {
  uint64_t mism = 0;
#pragma omp parallel for reduction(+ : mism) schedule(static)
  for (uint64_t u = lo; u <= hi; ++u) {
    float x = gmx_u2f((uint32_t)u);
    float a = gmx_expf(x), b = expf(x);
    int same = (gmx_f2u(a) == gmx_f2u(b)) || (a != a && b != b);
    if (!same) {
      uint64_t k;
#pragma omp atomic capture
      k = mism++;
      if ((int)k < nbad) bad[k] = (uint32_t)u;
    }
  }
  return mism;
}

/* Same for the whole squash: 1/(1+expf(-p)) against the libm form. */
uint64_t gmx_check_logistic_range(uint64_t lo, uint64_t hi, uint32_t* bad, int nbad) {
  uint64_t mism = 0;
#pragma omp parallel for reduction(+ : mism) schedule(static)
  for (uint64_t u = lo; u <= hi; ++u) {
    float x = gmx_u2f((uint32_t)u);
    float a = gmx_logistic(x), b = 1 / (1 + expf(-x));
    int same = (gmx_f2u(a) == gmx_f2u(b)) || (a != a && b != b);
    if (!same) {
      uint64_t k;
#pragma omp atomic capture
      k = mism++;
      if ((int)k < nbad) bad[k] = (uint32_t)u;
    }
  }
  return mism;
}

void gmx_host_logistic_array(const float* x, float* y, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) y[i] = gmx_logistic(x[i]);
}
void gmx_host_squash_array(const float* x, float* y, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) y[i] = gmx_squash_clamp(x[i]);
}


/* The LSTM's functions against libm: logf, expm1f (the core of tanhf), tanhf. */
#define GMX_CHECK_RANGE(NAME, OURS, LIBM)                                              \
  uint64_t NAME(uint64_t lo, uint64_t hi, uint32_t* bad, int nbad) {                    \
    uint64_t mism = 0;                                                                  \
    _Pragma("omp parallel for reduction(+ : mism) schedule(static)")                    \
    for (uint64_t u = lo; u <= hi; ++u) {                                               \
      float x = gmx_u2f((uint32_t)u);                                                   \
      float a = OURS(x), b = LIBM(x);                                                   \
      int same = (gmx_f2u(a) == gmx_f2u(b)) || (a != a && b != b);                      \
      if (!same) {                                                                      \
        uint64_t k;                                                                     \
        _Pragma("omp atomic capture")                                                   \
        k = mism++;                                                                     \
        if ((int)k < nbad) bad[k] = (uint32_t)u;                                        \
      }                                                                                 \
    }                                                                                   \
    return mism;                                                                        \
  }
GMX_CHECK_RANGE(gmx_check_logf_range, gmx_logf, logf)
GMX_CHECK_RANGE(gmx_check_expm1f_range, gmx_expm1f, expm1f)
GMX_CHECK_RANGE(gmx_check_tanhf_range, gmx_tanhf, tanhf)
