/* rangecheck.c -- host twin of gmx_math_range_kernel (gmix_amd/csrc/gmx_kernels.hip): folds
 * gmx_expf / gmx_logistic / gmx_squash_clamp over a range of float bit patterns into
 * {xor-fold, sum} so that device and host can be compared over all 2^32 inputs. */
#include <stdint.h>
#include "../../gmix_amd/csrc/gmx_math.h"

void gmx_host_math_range(uint64_t lo, uint64_t count, int what, unsigned long long out[2]) {
  unsigned long long x = 0, s = 0;
#pragma omp parallel for reduction(^ : x) reduction(+ : s) schedule(static)
  for (uint64_t i = 0; i < count; ++i) {
    const uint32_t u = (uint32_t)(lo + i);
    const float v = gmx_u2f(u);
    float r = what == 0 ? gmx_expf(v) : (what == 1 ? gmx_logistic(v) : gmx_squash_clamp(v));
    uint32_t rb = gmx_f2u(r);
    if (r != r) rb = 0x7fc00000u;
    x ^= (unsigned long long)rb * 0x9E3779B97F4A7C15ull + u;
    s += rb;
  }
  out[0] = x;
  out[1] = s;
}
