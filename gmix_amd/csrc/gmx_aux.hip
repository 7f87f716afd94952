// gmx_aux.hip -- small gfx950 kernels around the hot path: the synthetic record generator
// of the benchmark, the device-side math probes of the parity tests, bank initialisation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmx_internal.h"
#include "gmx_math.h"
#include "gmx_step_dev.h"

// ---------------------------------------------------------------------------------------
// Synthetic record generator (BASELINE.json configs[1]; definition in oracle/gmx_synth.h,
// restated here for the device): one thread walks one stream's xorshift64 sequence.
// ---------------------------------------------------------------------------------------
struct GmxSynthArgs {
  float* pred;        // [S][rec_stride][n_pad]
  uint32_t* mask;     // [S][rec_stride][mask_words] or null
  uint32_t* ctx;      // [S][rec_stride][m]
  uint8_t* bits;      // [S][rec_stride]
  uint64_t* rng;      // [S] persistent xorshift state
  uint64_t* tcount;   // [S] bits generated so far (for the every-8th-bit context modes)
  float* pstate;      // [S][n_pad] persistent prediction slots
  uint32_t* cstate;   // [S][m] persistent contexts
  uint64_t rec_stride, n_bits, seed;
  int32_t n, n_pad, m, mask_words, n_streams, restart, ctx_mode, bit_mode;
  uint32_t ctx_mod, zero_mod;
};

// bit j set = gate context j is redrawn every bit in ctx_mode 4/5 (GMX_SYNTH_BITLEVEL)
constexpr uint64_t kSynthBitLevel = (1ull << 2) | (1ull << 11) | (1ull << 26) | (1ull << 29);

__device__ __forceinline__ uint32_t gmx_xs64(uint64_t& s) {
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  return (uint32_t)(s >> 11);
}

__global__ void __launch_bounds__(64) gmx_synth_kernel(const GmxSynthArgs a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_streams) return;
  float* pst = a.pstate + (uint64_t)s * a.n_pad;
  uint32_t* cst = a.cstate + (uint64_t)s * a.m;
  uint64_t st, tc;
  if (a.restart) {
    st = a.seed + (uint64_t)s * 0x9E3779B97F4A7C15ull;
    if (st == 0) st = 0x9E3779B97F4A7C15ull;
    tc = 0;
    for (int i = 0; i < a.n_pad; ++i) pst[i] = 0.f;
    for (int j = 0; j < a.m; ++j) cst[j] = 0u;
  } else {
    st = a.rng[s];
    tc = a.tcount[s];
  }
  const uint32_t cmod = a.ctx_mod ? a.ctx_mod : 1u;
  for (uint64_t t = 0; t < a.n_bits; ++t) {
    float* pr = a.pred + ((uint64_t)s * a.rec_stride + t) * a.n_pad;
    uint32_t* mk = a.mask ? a.mask + ((uint64_t)s * a.rec_stride + t) * a.mask_words : nullptr;
    uint32_t* cx = a.ctx + ((uint64_t)s * a.rec_stride + t) * a.m;
    uint32_t mword = 0;
    for (int i = 0; i < a.n; ++i) {
      bool act = false;
      bool silent = false;
      if (a.zero_mod) {
        const uint32_t dr = gmx_xs64(st);
        silent = (dr % a.zero_mod) == 0;
      }
      if (!silent) {
        const float x = (float)((int)(gmx_xs64(st) % 2001u) - 1000) / 250.0f;
        pst[i] = x;
        act = (x != 0.0f);
      }
      pr[i] = pst[i];
      if (act) mword |= 1u << (i & 31);
      if ((i & 31) == 31 || i == a.n - 1) {
        if (mk) mk[i >> 5] = mword;
        mword = 0;
      }
    }
    for (int i = a.n; i < a.n_pad; ++i) pr[i] = 0.f;
    const bool redraw = (a.ctx_mode < 2) || ((tc & 7u) == 0);
    if (redraw) {
      for (int j = 0; j < a.m; ++j) {
        uint32_t c = gmx_xs64(st);
        if (a.ctx_mode & 1) c %= cmod;
        cst[j] = c;
      }
    } else if (a.ctx_mode >= 4) {
      for (int j = 0; j < a.m && j < 64; ++j) {
        if (!((kSynthBitLevel >> j) & 1)) continue;
        uint32_t c = gmx_xs64(st);
        if (a.ctx_mode & 1) c %= cmod;
        cst[j] = c;
      }
    }
    for (int j = 0; j < a.m; ++j) cx[j] = cst[j];
    ++tc;
    const uint32_t r = gmx_xs64(st);
    uint32_t bit = r & 1u;
    if (a.bit_mode == 1) bit = (uint32_t)((pst[0] > 0.0f) ^ ((r & 7u) == 0));
    a.bits[(uint64_t)s * a.rec_stride + t] = (uint8_t)bit;
  }
  a.rng[s] = st;
  a.tcount[s] = tc;
}

extern "C" hipError_t gmx_launch_synth_kernel(const GmxSynthArgs* args, hipStream_t stream) {
  const int blocks = (args->n_streams + 63) / 64;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_synth_kernel, dim3(blocks), dim3(64), 0, stream, *args);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Synthetic records for the Indirect models (definition in oracle/gmx_ind_synth.h, restated
// here for the device): byte-structured contexts, one thread per stream.
// ---------------------------------------------------------------------------------------
struct GmxIndSynthArgs {
  uint32_t* ctx;      // [S][rec_stride][k]
  uint32_t* bc;       // [S][rec_stride]
  uint8_t* bits;      // [S][rec_stride]
  uint64_t* rng;      // [S]
  uint32_t* recent;   // [S] recent_bits
  uint32_t* cstate;   // [S][k] contexts of the current byte
  uint64_t rec_stride, n_bits, seed;
  int32_t k, n_streams, restart;
  uint32_t ctx_mod[4];
};

__global__ void __launch_bounds__(64) gmx_ind_synth_kernel(const GmxIndSynthArgs a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_streams) return;
  uint32_t* cst = a.cstate + (uint64_t)s * a.k;
  uint64_t st;
  uint32_t recent;
  if (a.restart) {
    st = a.seed + (uint64_t)s * 0x9E3779B97F4A7C15ull;
    if (st == 0) st = 0x9E3779B97F4A7C15ull;
    recent = 1;
    for (int j = 0; j < a.k; ++j) cst[j] = 0u;
  } else {
    st = a.rng[s];
    recent = a.recent[s];
  }
  for (uint64_t t = 0; t < a.n_bits; ++t) {
    uint32_t* cx = a.ctx + ((uint64_t)s * a.rec_stride + t) * a.k;
    if (recent == 1) {  // byte boundary: every context is redrawn
      for (int j = 0; j < a.k; ++j) {
        uint32_t c = gmx_xs64(st);
        const uint32_t m = a.ctx_mod[j & 3];
        if (m) c %= m;
        cst[j] = c;
      }
    }
    for (int j = 0; j < a.k; ++j) cx[j] = cst[j];
    const uint32_t bcx = recent - 1;
    a.bc[(uint64_t)s * a.rec_stride + t] = bcx;
    const uint32_t r = gmx_xs64(st);
    const uint32_t bit = ((cst[0] ^ (bcx * 7u)) & 1u) ^ (uint32_t)((r % 5u) == 0);
    a.bits[(uint64_t)s * a.rec_stride + t] = (uint8_t)bit;
    recent = recent * 2 + bit;
    if (recent >= 256) recent = 1;
  }
  a.rng[s] = st;
  a.recent[s] = recent;
}

extern "C" hipError_t gmx_launch_ind_synth_kernel(const GmxIndSynthArgs* args, hipStream_t stream) {
  const int blocks = (args->n_streams + 63) / 64;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_ind_synth_kernel, dim3(blocks), dim3(64), 0, stream, *args);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Device-side math probes for the parity tests (the same gmx_math.h the kernels use).
// ---------------------------------------------------------------------------------------
__global__ void gmx_math_probe_kernel(const float* x, float* y, uint64_t n, int what) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  y[i] = what == 0 ? gmx_expf(v) : (what == 1 ? gmx_logistic(v) : gmx_squash_clamp(v));
}

// Compare device gmx_expf / gmx_logistic over a whole range of float bit patterns against a
// host-computed table is too slow over PCIe; instead the device checksums its results and
// the host checksums its own: out[0] = xor-fold, out[1] = sum of the result bit patterns.
__global__ void gmx_math_range_kernel(uint64_t lo, uint64_t count, int what,
                                      unsigned long long* out) {
  unsigned long long x = 0, sacc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t u = (uint32_t)(lo + i);
    const float v = gmx_u2f(u);
    float r;
    if (what == 3) {
      // the wave-level logistic of gmx_stock.hip (short way where a whole wave has |v| < 64): the caller
      // compares the checksums with those of what == 1
      r = gmx_wave_logistic(v, gmx_exp2f_tab);
    } else if (what == 4) {
      // the short row-age division against the general one on a pair of counters below 2^32 made from i:
      // small pairs exhaustively (i < 2^24: 4096 x 4096), then hashed ones with the edges mixed in; the
      // checksums count the differing bit patterns
      const uint64_t i64 = lo + i;
      uint32_t ca, cb;
      if (i64 < (1ull << 24)) {
        ca = (uint32_t)(i64 & 4095u);
        cb = (uint32_t)(i64 >> 12) + 1u;
      } else {
        uint64_t h = i64 * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        cb = (uint32_t)h >> ((h >> 59) & 31u);
        if (cb == 0) cb = 0xffffffffu;
        ca = (uint32_t)(h >> 32) >> ((h >> 54) & 31u);
        if ((i64 & 7u) == 0) ca = ca % cb;          // the reference's case: row.steps <= max_steps_
        if ((i64 & 1023u) == 1) ca = cb;
        if ((i64 & 1023u) == 2) ca = cb - 1u;
      }
      const double q_short = gmx_row_age_short(ca, cb), q_gen = (double)ca / (double)cb;
      const bool bad = gmx_d2u(q_short) != gmx_d2u(q_gen);
      x ^= bad ? i64 + 1 : 0;
      sacc += bad ? 1 : 0;
      continue;
    } else {
      r = what == 0 ? gmx_expf(v) : (what == 1 ? gmx_logistic(v) : gmx_squash_clamp(v));
    }
    uint32_t rb = gmx_f2u(r);
    if (r != r) rb = 0x7fc00000u;  // all NaNs alike
    x ^= (unsigned long long)rb * 0x9E3779B97F4A7C15ull + u;
    sacc += rb;
  }
  atomicXor(&out[0], x);
  atomicAdd(&out[1], sacc);
}

extern "C" hipError_t gmx_launch_math_probe(const float* x, float* y, uint64_t n, int what,
                                            hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     x, y, n, what);
  return hipGetLastError();
}

extern "C" hipError_t gmx_launch_math_range(uint64_t lo, uint64_t count, int what,
                                            unsigned long long* out, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_math_range_kernel, dim3(2048), dim3(256), 0, stream, lo, count, what, out);
  return hipGetLastError();
}

// Constructed state of every Mixer: steps_ = 0, max_steps_ = 1, contexts_seen_ = 0
// (mixer.cpp:8-9, mixer.h:38); the tables themselves are zero-filled by the host side.
__global__ void gmx_init_scal_kernel(uint8_t* banks, uint64_t bank_bytes, uint64_t scal_off, int m,
                                     int n_streams) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * n_streams) return;
  const int s = i / m, j = i % m;
  uint64_t* sc = (uint64_t*)(banks + (uint64_t)s * bank_bytes + scal_off) + 3 * j;
  sc[0] = 0;
  sc[1] = 1;
  sc[2] = 0;
}

extern "C" hipError_t gmx_launch_init_scal(uint8_t* banks, uint64_t bank_bytes, uint64_t scal_off,
                                           int m, int n_streams, hipStream_t stream) {
  const int n = m * n_streams;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_init_scal_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, banks,
                     bank_bytes, scal_off, m, n_streams);
  return hipGetLastError();
}

// ---- the bit-count factor of the learning-rate decay, for many different bit counts ---------
// float(0.9 / pow(1e-7 * steps_ + 0.8, 0.8)) (mixer.cpp:111) is a function of the bit count only;
// streams that stand at the same count share one table row, which the host fills with the libm the
// reference calls.  When a launch covers streams at MANY different counts that loop would dwarf
// the kernel, so the table is made here -- and stays the reference's, bit for bit, without
// restating glibc's pow: the value needed is a FLOAT.  This device's pow and glibc's both lie
// within a few ulp (of double) of the true value; if every double within 64 ulp of ours rounds to
// the same float, that float is glibc's too.  The rare entry where it does not (a double within
// 2^-46 of a float rounding boundary: about one in 2^22) is reported and the host computes it.
struct GmxDecayArgs {
  const uint64_t* steps0;  // [U] bit count of row u at t = 0
  float* table;            // [U][T]
  uint32_t* amb;           // [0]: count of unsettled entries, [1 .. cap]: their flat indices
  uint32_t amb_cap, U;
  uint64_t T;
};

__global__ void __launch_bounds__(256) gmx_decay_kernel(const GmxDecayArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= (uint64_t)a.U * a.T) return;
  const uint64_t steps = a.steps0[i / a.T] + i % a.T;
  const double x = 0.0000001 * (double)steps + 0.8;
  const double d = 0.9 / pow(x, 0.8);
  const double margin = d * 0x1p-46;
  const float f = (float)d, lo = (float)(d - margin), hi = (float)(d + margin);
  a.table[i] = f;
  if (lo != hi) {
    const uint32_t k = atomicAdd(&a.amb[0], 1u);
    if (k < a.amb_cap) a.amb[1 + k] = (uint32_t)i;
  }
}

extern "C" hipError_t gmx_launch_decay_kernel(const GmxDecayArgs* args, hipStream_t stream) {
  (void)hipGetLastError();
  const uint64_t n = (uint64_t)args->U * args->T;
  hipLaunchKernelGGL(gmx_decay_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *args);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------
// 16-byte words from one place to another -- the lock-step chain's record upload (gmx_chainstep.inc): pinned host
// memory, read across the link by the device itself, to device memory.  A kernel node instead of a memcpy node in
// the step's hipGraph: the graph's copies of 50-200 KB cost 9-11 us each, this one 2-4.
__global__ void __launch_bounds__(256) gmx_copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint64_t n16) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
// The head of a lock-step step as a launch of its own (chains whose first kernel is not the Indirect models'): block s
// brings stream s's inputs in.
__global__ void __launch_bounds__(64) gmx_step_upload_kernel(const GmxStepUpload u) {
  gmx_step_upload(u, (int)blockIdx.x, (int)threadIdx.x);
}
extern "C" hipError_t gmx_launch_step_upload(const GmxStepUpload* u, int n_streams, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_step_upload_kernel, dim3(n_streams), dim3(64), 0, stream, *u);
  return hipGetLastError();
}
extern "C" hipError_t gmx_launch_copy16(void* dst, const void* src, uint64_t n16, hipStream_t stream) {
  (void)hipGetLastError();
  if (n16 == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n16 + 255) / 256 < 256 ? (n16 + 255) / 256 : 256);
  hipLaunchKernelGGL(gmx_copy16_kernel, dim3(blocks), dim3(256), 0, stream, (uint4*)dst, (const uint4*)src, n16);
  return hipGetLastError();
}
