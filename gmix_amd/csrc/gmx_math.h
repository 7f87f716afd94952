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
// gmx_exp2f_tab or a copy of it.  First the main path alone: expf(x) for |x| < 88 (e_expf.c takes
// none of its special cases there).
GMX_HD float gmx_expf_main(float x, const uint64_t* tab) {
  const double kShift = 0x1.8p+52;
  const double kInvLn2N = 0x1.71547652b82fep+0 * 32;
  const double kC0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
  const double kC1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
  const double kC2 = 0x1.62e42ff0c52d6p-1 / 32;
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
  return (float)y;
}

GMX_HD float gmx_expf_tab(float x, const uint64_t* tab) {
  const uint32_t ux = gmx_f2u(x);
  const uint32_t abstop = (ux >> 20) & 0x7ff;
  // main path (meaningful for |x| < ~104; harmless garbage beyond, overridden below)
  float res = gmx_expf_main(x, tab);
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

// ---- wave-level short ways (gfx950 only; used by gmx_stock.hip) ---------------------------------
#if defined(__HIPCC__)
// Sigmoid::Logistic of the wave's outputs (mixer.cpp:113-122) and the row-age ratio (1.0 * row.steps) / max_steps_
// of mixer.cpp:112, together and in one basic block: neither needs the other, and the ratio's double-precision
// chain fills the wait for expf's table entry.  The wave issues every instruction at full price, so both take
// the short way whenever EVERY lane allows it (anything else: the general expressions, same results):
//  * |p| < 64: none of e_expf.c's special cases applies (they start at 88), and d = 1 + e lies in [1, 2^93),
//    where the IEEE division 1.0f / d as the compiler expands it (2 x v_div_scale, v_rcp, Newton steps,
//    v_div_fmas, v_div_fixup) scales nothing and fixes nothing up: the same v_rcp / fma sequence without those steps;
//  * both counters below 2^32: the conversions to double are exact from the low words, and the double
//    division's scaling / fix-up steps are identities for integers of that size (0 / b is +0 either way;
//    max_steps_ is never 0: it starts at 1 and only grows, mixer.cpp:8, :126-128).
// gmx_debug_math_range (what = 3, 4) compares the short ways with the general ones over all floats / over
// 2^32 counter pairs on the device.
__device__ __forceinline__ bool gmx_wave_short_math(float p, uint64_t rs, uint64_t ms) {
  const uint32_t abstop = (gmx_f2u(p) >> 20) & 0x7ffu;
  return __ballot(abstop >= 0x428u || (uint32_t)((rs | ms) >> 32) != 0u) == 0;
}
__device__ __forceinline__ float gmx_logistic_short(float p, const uint64_t* tab) {
  const float d = 1.0f + gmx_expf_main(-p, tab);
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float e1 = __builtin_fmaf(-d, r1, 1.0f);  // the quotient starts as 1.0f * r1
  const float q1 = __builtin_fmaf(e1, r1, r1);
  const float e2 = __builtin_fmaf(-d, q1, 1.0f);
  return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ double gmx_row_age_short(uint64_t rs, uint64_t ms) {
  const double a = (double)(uint32_t)rs, b = (double)(uint32_t)ms;
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = a * r;
  e = __builtin_fma(-b, q, a);
  return __builtin_fma(e, r, q);
}
__device__ __forceinline__ float gmx_wave_logistic_and_age(float p, const uint64_t* tab, uint64_t rs, uint64_t ms,
                                                        double& age) {
  if (gmx_wave_short_math(p, rs, ms)) {
    age = gmx_row_age_short(rs, ms);
    return gmx_logistic_short(p, tab);
  }
  age = (double)rs / (double)ms;
  return gmx_logistic_tab(p, tab);
}

// The same when the caller already knows whether every lane that matters has |p| < 64 (gmx_stock_kernel's plain
// build learns it from the test it makes for non-finite outputs anyway; lanes that do not matter may hold anything).
__device__ __forceinline__ float gmx_wave_logistic_and_age_hint(float p, const uint64_t* tab, uint64_t rs, uint64_t ms,
                                                             bool p_small, double& age) {
  if (p_small && __ballot((uint32_t)((rs | ms) >> 32) != 0u) == 0) {
    age = gmx_row_age_short(rs, ms);
    return gmx_logistic_short(p, tab);
  }
  age = (double)rs / (double)ms;
  return gmx_logistic_tab(p, tab);
}

// Either alone (the per-bit kernels learn in a later command than they predict).
__device__ __forceinline__ float gmx_wave_logistic(float p, const uint64_t* tab) {
  double unused;
  return gmx_wave_logistic_and_age(p, tab, 0, 1, unused);
}
__device__ __forceinline__ double gmx_wave_row_age(uint64_t rs, uint64_t ms) {
  if (__ballot((uint32_t)((rs | ms) >> 32) != 0u) == 0) return gmx_row_age_short(rs, ms);
  return (double)rs / (double)ms;
}

#endif  // __HIPCC__

// Final squash of Predictor::Predict (predictor.cpp:369-375): clamp to [1e-4f, 1-1e-4f].
GMX_HD float gmx_clamp_prob(float prob) {
  const float eps = 0.0001f;
  return prob < eps ? eps : (prob > 1.0f - eps ? 1.0f - eps : prob);
}
GMX_HD float gmx_squash_clamp(float out) { return gmx_clamp_prob(gmx_logistic(out)); }

// ---- logf, expm1f, tanhf: what the LSTM byte model needs (models/lstm-layer.cpp:208, :215;
// Sigmoid::Logit, mixer/sigmoid.cpp:7-13) ------------------------------------------------------
//
// logf: glibc >= 2.28 (sysdeps/ieee754/flt-32/e_logf.c, Arm Optimized Routines): x = 2^k z with
// z in [0x1.66p-1, 0x1.66p0) split in 16 intervals; r = z*invc - 1 from a table of (1/c, log c),
// cubic in r, all in double, one final rounding.  As with expf the FMA-contracted build
// (__logf_fma) is what an FMA-capable x86-64 runs; the contractions are spelled out.  The table
// is the published one (logf_data.c); tests/test_math.py compares gmx_logf with the machine's
// libm over all 2^32 inputs.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const uint64_t gmx_logf_tab[16][2] = {  // {invc, logc} as raw doubles
    {0x3ff661ec79f8f3beull, 0xbfd57bf7808caadeull}, {0x3ff571ed4aaf883dull, 0xbfd2bef0a7c06ddbull},
    {0x3ff49539f0f010b0ull, 0xbfd01eae7f513a67ull}, {0x3ff3c995b0b80385ull, 0xbfcb31d8a68224e9ull},
    {0x3ff30d190c8864a5ull, 0xbfc6574f0ac07758ull}, {0x3ff25e227b0b8ea0ull, 0xbfc1aa2bc79c8100ull},
    {0x3ff1bb4a4a1a343full, 0xbfba4e76ce8c0e5eull}, {0x3ff12358f08ae5baull, 0xbfb1973c5a611cccull},
    {0x3ff0953f419900a7ull, 0xbfa252f438e10c1eull}, {0x3ff0000000000000ull, 0x0000000000000000ull},
    {0x3fee608cfd9a47acull, 0x3faaa5aa5df25984ull}, {0x3feca4b31f026aa0ull, 0x3fbc5e53aa362eb4ull},
    {0x3feb2036576afce6ull, 0x3fc526e57720db08ull}, {0x3fe9c2d163a1aa2dull, 0x3fcbc2860d224770ull},
    {0x3fe886e6037841edull, 0x3fd1058bc8a07ee1ull}, {0x3fe767dcf5534862ull, 0x3fd4043057b6ee09ull}};

GMX_HD float gmx_logf(float x) {
  const double kLn2 = 0x1.62e42fefa39efp-1;
  const double kA0 = -0x1.00ea348b88334p-2, kA1 = 0x1.5575b0be00b6ap-2, kA2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = gmx_f2u(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2u == 0) return gmx_u2f(0xff800000u);                                  // log(+-0) = -inf
    if (ix == 0x7f800000u) return x;                                                // log(inf) = inf
    if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return gmx_u2f(0x7fc00000u) + (x != x ? x : 0.0f);  // NaN
    ix = gmx_f2u(x * 0x1p23f);                                                      // subnormal: normalise
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const int32_t k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = gmx_u2d(gmx_logf_tab[i][0]), logc = gmx_u2d(gmx_logf_tab[i][1]);
  const double z = (double)gmx_u2f(iz);
  const double r = gmx_fma(z, invc, -1.0);
  const double y0 = gmx_fma((double)k, kLn2, logc);
  const double r2 = r * r;
  double y = gmx_fma(kA1, r, kA2);
  y = gmx_fma(kA0, r2, y);
  y = gmx_fma(y, r2, y0 + r);
  return (float)y;
}

// Sigmoid::Logit (mixer/sigmoid.cpp:7-13): the comparisons and the clamp constants are double
// literals; the division and 1 - p are float.
GMX_HD float gmx_logit(float p) {
  if ((double)p < 0.0001) p = (float)0.0001;
  else if ((double)p > 0.9999) p = (float)0.9999;
  return gmx_logf(p / (1 - p));
}

// expm1f: glibc's fdlibm port (sysdeps/ieee754/flt-32/s_expm1f.c), plain float arithmetic,
// no fused operations (there is no FMA build of it).
GMX_HD float gmx_expm1f(float x) {
  const float one = 1.0f, huge = 1.0e+30f, tiny = 1.0e-30f;
  const float o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
              invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
              Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  float y, hi, lo, c = 0.0f, t, e, hxs, hfx, r1;
  int32_t k;
  uint32_t hx = gmx_f2u(x);
  const uint32_t xsb = hx & 0x80000000u;
  hx &= 0x7fffffffu;
  if (hx >= 0x4195b844u) {        // |x| >= 27 ln2
    if (hx >= 0x42b17218u) {      // |x| >= 88.72
      if (hx > 0x7f800000u) return x + x;
      if (hx == 0x7f800000u) return xsb == 0 ? x : -1.0f;
      if (x > o_threshold) return huge * huge;
    }
    if (xsb != 0) return tiny - one;
  }
  if (hx > 0x3eb17218u) {         // |x| > 0.5 ln2
    if (hx < 0x3F851592u) {       // |x| < 1.5 ln2
      if (xsb == 0) {
        hi = x - ln2_hi;
        lo = ln2_lo;
        k = 1;
      } else {
        hi = x + ln2_hi;
        lo = -ln2_lo;
        k = -1;
      }
    } else {
      k = (int32_t)(invln2 * x + (xsb == 0 ? 0.5f : -0.5f));
      t = (float)k;
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x33000000u) {  // |x| < 2^-25
    t = huge + x;
    return x - (t - (huge + x));
  } else {
    k = 0;
  }
  hfx = 0.5f * x;
  hxs = x * hfx;
  r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  t = 3.0f - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0f - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5f * (x - e) - 0.5f;
  if (k == 1) {
    if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
    return one + 2.0f * (x - e);
  }
  if (k <= -2 || k > 56) {
    y = one - (e - x);
    y = gmx_u2f(gmx_f2u(y) + ((uint32_t)k << 23));
    return y - one;
  }
  if (k < 23) {
    t = gmx_u2f(0x3f800000u - (0x1000000u >> k));  // 1 - 2^-k
    y = t - (e - x);
    y = gmx_u2f(gmx_f2u(y) + ((uint32_t)k << 23));
  } else {
    t = gmx_u2f((uint32_t)(0x7f - k) << 23);       // 2^-k
    y = x - (e + t);
    y += one;
    y = gmx_u2f(gmx_f2u(y) + ((uint32_t)k << 23));
  }
  return y;
}

// tanhf: glibc's fdlibm port (sysdeps/ieee754/flt-32/s_tanhf.c) on top of expm1f.
GMX_HD float gmx_tanhf(float x) {
  const float one = 1.0f, two = 2.0f, tiny = 1.0e-30f;
  float t, z;
  const uint32_t jx = gmx_f2u(x);
  const uint32_t ix = jx & 0x7fffffffu;
  if (ix >= 0x7f800000u) return (jx >> 31) == 0 ? one / x + one : one / x - one;
  if (ix < 0x41b00000u) {  // |x| < 22
    if (ix == 0) return x;
    if (ix < 0x24000000u) return x * (one + x);  // |x| < 2^-55
    const float ax = gmx_u2f(ix);
    if (ix >= 0x3f800000u) {
      t = gmx_expm1f(two * ax);
      z = one - two / (t + two);
    } else {
      t = gmx_expm1f(-two * ax);
      z = -t / (t + two);
    }
  } else {
    z = one - tiny;
  }
  return (jx >> 31) == 0 ? z : -z;
}

#endif  // GMX_MATH_H_
