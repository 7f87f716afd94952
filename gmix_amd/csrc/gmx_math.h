// gmx_math.h -- scalar arithmetic of the mixer path that has to be bit-reproducible on
// host cores and on gfx950 alike: expf, the logistic squash and the final clamp.
//
// The reference squashes with libm: Sigmoid::Logistic(p) = 1 / (1 + expf(-p))
// (mixer/sigmoid.cpp:5; callers mixer.cpp:113-122, predictor.cpp:369).  libm is a
// third-party dependency outside /root/reference (glibc 2.35 in the oracle's container),
// so its published algorithm is restated here: glibc's expf since 2.27 is the Arm
// Optimized Routines single-precision exp (sysdeps/ieee754/flt-32/e_expf.c with
// math_config.h EXP2F_TABLE_BITS = 5): z = x*N/ln2 in double, k = round(z) via the
// 0x1.8p52 shift, r = z - k, 2^(k/N) from a 32-entry table, cubic in r, one final
// rounding to float.  On every x86-64 CPU with FMA+AVX2 the dynamic linker selects the
// build of that source compiled with -mfma (__expf_fma), in which the compiler contracted
// five of the multiply-adds; the contraction pattern below (gmx_fma calls) is that
// build's.  tests/test_math.py compares gmx_expf with the machine's libm over ALL 2^32
// float inputs on the host, and tests/test_gpu_math.py compares device with host over the
// same set, so "same float as the reference's Logistic" is a tested fact, not a hope.
//
// The special cases (|x| >= 88, infinities, NaN) are applied as selects after the main
// path instead of branches before it: same results, and no divergent branch in the middle
// of a kernel's basic block.  The 2^(i/32) table can be supplied by the caller so that the
// kernels read it from LDS (a lookup in global memory would sit in the same in-order
// vmcnt queue as the prefetched weight rows and drain it on every bit).
//
// Nothing here may be contracted or re-associated by the compiler: build with
// -ffp-contract=off (Makefile) -- the fused operations are spelled out as gmx_fma().
#ifndef GMX_MATH_H_
#define GMX_MATH_H_

#include <stdint.h>
#include <string.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GMX_HD __host__ __device__ __forceinline__
#else
#define GMX_HD static inline
#endif

// 2^(i/32) as raw doubles with (i << 47) subtracted, so that adding (k << 47) to entry
// k%32 yields 2^(k/32) for any integer k (e_expf.c: "t += ki << (52 - EXP2F_TABLE_BITS)").
// The entries are the correctly rounded values of the definition; tests pin them.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const uint64_t gmx_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

GMX_HD double gmx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

GMX_HD uint32_t gmx_f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
GMX_HD float gmx_u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
GMX_HD uint64_t gmx_d2u(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}
GMX_HD double gmx_u2d(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

// expf as glibc 2.27+ computes it on FMA-capable x86-64 (see file header); `tab` is
// gmx_exp2f_tab or a copy of it.
GMX_HD float gmx_expf_tab(float x, const uint64_t* tab) {
  const double kShift = 0x1.8p+52;
  const double kInvLn2N = 0x1.71547652b82fep+0 * 32;
  const double kC0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
  const double kC1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
  const double kC2 = 0x1.62e42ff0c52d6p-1 / 32;
  const uint32_t ux = gmx_f2u(x);
  const uint32_t abstop = (ux >> 20) & 0x7ff;
  // main path (meaningful for |x| < ~104; harmless garbage beyond, overridden below)
  const double xd = (double)x;
  double kd = gmx_fma(kInvLn2N, xd, kShift);    // round(x*N/ln2) in the low mantissa bits
  const uint64_t ki = gmx_d2u(kd);
  kd -= kShift;
  const double r = gmx_fma(kInvLn2N, xd, -kd);  // x*N/ln2 - k, one rounding
  const uint64_t t = tab[ki & 31] + (ki << 47);
  const double s = gmx_u2d(t);
  const double z = gmx_fma(kC0, r, kC1);
  const double r2 = r * r;
  double y = gmx_fma(kC2, r, 1.0);
  y = gmx_fma(z, r2, y);
  y = y * s;
  float res = (float)y;
  // special cases of e_expf.c, in its order of precedence (last assignment wins here)
  const bool big = abstop >= 0x42b;                               // |x| >= 88 or NaN
  res = (big && x < -0x1.9d1d9ep6f) ? gmx_u2f(1u) : res;          // 0x1.4p-75f squared = 2^-149
  res = (big && x < -0x1.9fe368p6f) ? 0.0f : res;                 // underflow
  res = (big && x > 0x1.62e42ep6f) ? gmx_u2f(0x7f800000u) : res;  // overflow
  res = (abstop >= 0x7f8) ? x + x : res;                          // +-inf, NaN
  res = (ux == 0xff800000u) ? 0.0f : res;                         // exp(-inf)
  return res;
}

GMX_HD float gmx_expf(float x) { return gmx_expf_tab(x, gmx_exp2f_tab); }

// Sigmoid::Logistic (mixer/sigmoid.cpp:5): float add and IEEE float divide.
GMX_HD float gmx_logistic_tab(float p, const uint64_t* tab) {
  return 1.0f / (1.0f + gmx_expf_tab(-p, tab));
}
GMX_HD float gmx_logistic(float p) { return gmx_logistic_tab(p, gmx_exp2f_tab); }

// Final squash of Predictor::Predict (predictor.cpp:369-375): clamp to [1e-4f, 1-1e-4f].
GMX_HD float gmx_clamp_prob(float prob) {
  const float eps = 0.0001f;
  return prob < eps ? eps : (prob > 1.0f - eps ? 1.0f - eps : prob);
}
GMX_HD float gmx_squash_clamp(float out) { return gmx_clamp_prob(gmx_logistic(out)); }

#endif  // GMX_MATH_H_
