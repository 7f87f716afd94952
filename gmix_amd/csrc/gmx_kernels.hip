// gmx_kernels.hip -- gfx950 kernels of the mixer hot path (hand-written HIP, CDNA4 only).
//
// What runs here is the body of Mixer::Predict and Mixer::Learn (mixer/mixer.cpp:51-176) for
// all mixers of a bank, bit after bit, with the reference's exact fp32 operation order:
// every dot product is one strict left-to-right chain of (multiply, then add), never fused,
// never re-associated; the update is w -= update * x with two roundings.  Parallelism comes
// from what the reference leaves independent:
//   * lanes of a wave  = the mixers of one layer (24 layer-0 chains advance in lock step,
//     then the layer-0 cascade, then the 8 layer-1 chains, then the final mixer);
//   * lanes of a wave  = the elements of one weight row for the loads, the update and the
//     stores (each row is a contiguous, 128-byte aligned run in HBM: fully coalesced);
//   * waves            = independent byte-streams (one bank each), one wave per stream.
// Rows are staged through LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip) into
// a two-slot, tag-checked row cache per mixer, so that the rows of bit t+1 (known as soon as
// its contexts are) stream in while bit t is being computed, and a row that is used again
// on the next bit -- the common case for the small gate tables of the stock topology --
// never leaves the CU.  The transposed (lane = mixer) reads of that image are
// bank-conflict-free because consecutive mixers' rows are pitch = stride + 4 floats apart.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile); the parity of the
// whole file rests on no contraction and IEEE divide, both asserted by tests on the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmx_internal.h"
#include "gmx_math.h"

typedef __attribute__((address_space(3))) void gmx_lds_void;
typedef const __attribute__((address_space(1))) void gmx_glb_void;

// One LDS-DMA piece: every active lane moves 16 bytes from its own global address to
// LDS[ldst + 16*lane] (ldst must be wave-uniform: it travels in M0).
__device__ __forceinline__ void gmx_glds16(const void* gsrc, float* ldst) {
  __builtin_amdgcn_global_load_lds((gmx_glb_void*)gsrc, (gmx_lds_void*)ldst, 16, 0, 0);
}

__device__ __forceinline__ float gmx_readlane_f(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// ---------------------------------------------------------------------------------------
// General bank kernel: any topology gmx_group_create accepts.  One wave = one stream.
// ---------------------------------------------------------------------------------------
template <bool HAS_MASK>
__global__ void __launch_bounds__(64)
gmx_bank_kernel(const GmxTopoDev* __restrict__ tp, const GmxRunArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int rec = a.rec_base + (int)blockIdx.x;   // stream index inside the record arrays
  const int s = a.stream_base + (int)blockIdx.x;  // bank index
  const int N = tp->n, NPAD = tp->n_pad, NS = tp->n_skip, M = tp->m, L0 = tp->l0, L1 = tp->l1;
  const int MW = tp->mask_words;
  const bool has_final = tp->has_final != 0;
  const uint64_t T = a.T;
  if (T == 0) return;
  uint8_t* const bank = a.banks + (uint64_t)s * tp->bank_bytes;

  float* const in0 = lds + tp->lds_in0;
  const uint32_t in0_sz = tp->in0_sz;
  float* const o1s = lds + tp->lds_o1;
  float* const skipv = lds + tp->lds_skip;
  uint32_t* const m_row = (uint32_t*)(lds + tp->lds_misc);
  uint32_t* const m_dst = m_row + 64;
  float* const m_upd = (float*)(m_row + 128);
  uint32_t* const m_flag = m_row + 192;

  const bool is_mx = lane < M;
  const GmxMixerDev d = tp->mx[is_mx ? lane : 0];
  const bool is_l0 = lane < L0;
  const bool is_l1 = lane >= L0 && lane < L0 + L1;
  const bool is_fin = has_final && lane == L0 + L1;

  const uint64_t RS = a.rec_stride;
  const float* const pred_s = a.pred + (uint64_t)rec * RS * NPAD;
  const uint32_t* const mask_s = HAS_MASK ? a.mask + (uint64_t)rec * RS * MW : nullptr;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS * M;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  const float* const dec_s = a.decay + (uint64_t)a.decay_idx[blockIdx.x] * T;
  float* const p_s = a.p_out + (uint64_t)rec * RS;
  float* const oa_s = a.out_all ? a.out_all + (uint64_t)rec * RS * M : nullptr;
  float* const latch_s = a.latch_out + (uint64_t)s * M;

  // Per-mixer scalars: Mixer::steps_, max_steps_, contexts_seen_ (mixer.h:33-38).
  uint64_t* const scal = (uint64_t*)(bank + tp->scal_off) + 3 * lane;
  uint64_t steps = 0, max_steps = 1, seen_cnt = 0;
  if (is_mx) {
    steps = scal[0];
    max_steps = scal[1];
    seen_cnt = scal[2];
  }
  uint64_t* const rs_tab = (uint64_t*)(bank + d.rs_off);

  // Two-slot row cache of this lane's mixer: tags (row index) and MixerData::steps of the
  // rows held; `cur` is the slot of the row in use for the current bit.
  uint32_t tag0 = 0xffffffffu, tag1 = 0xffffffffu;
  uint64_t rs0 = 0, rs1 = 0;
  uint32_t cur = 0, xb = 0;

  // _n: the bit being prefetched; _c: the bit being computed.
  uint32_t ctx_nn = is_mx ? ctx_s[lane] : 0;
  uint32_t row_n = 0, row_c = 0, mask_n = ~0u, mask_c = ~0u, bit_n = 0, bit_c = 0;
  float dec_n = 0.f, dec_c = 0.f;
  uint64_t rs_ld = 0;
  bool need = false;

  for (uint64_t t = 0; t <= T; ++t) {
    // ================= prefetch bit t (runs one iteration ahead of its compute) ==========
    if (t < T) {
      row_n = ctx_nn % d.table_size;  // FindMixerData: context_ % table size (mixer.cpp:32)
      const uint32_t tag_c = cur ? tag1 : tag0, tag_o = cur ? tag0 : tag1;
      need = is_mx && row_n != tag_c && row_n != tag_o;
      if (is_mx) {
        m_row[lane] = row_n;
        m_dst[lane] = d.lds_off + (cur ? 0u : d.pitch);  // the slot that is not in use
      }
      if (need) rs_ld = rs_tab[row_n];
      uint64_t nm = __ballot(need);
      while (nm) {
        const int m = __builtin_amdgcn_readfirstlane((int)__ffsll((unsigned long long)nm) - 1);
        nm &= nm - 1;
        const uint32_t row = m_row[m], dst = m_dst[m];
        const uint32_t stride = tp->mx[m].stride;
        const uint8_t* g = bank + tp->mx[m].w_off + (uint64_t)row * stride * 4u;
        for (uint32_t c0 = 0; c0 < stride; c0 += 256) {
          const uint32_t c = c0 + (uint32_t)lane * 4u;
          if (c < stride) gmx_glds16(g + (uint64_t)c * 4u, lds + dst + c0);
        }
      }
      {
        const float* g = pred_s + t * (uint64_t)NPAD;
        float* dstx = in0 + (xb ^ 1u) * in0_sz;
        for (uint32_t c0 = 0; c0 < (uint32_t)NPAD; c0 += 256) {
          const uint32_t c = c0 + (uint32_t)lane * 4u;
          if (c < (uint32_t)NPAD) gmx_glds16(g + c, dstx + c0);
        }
      }
      if (HAS_MASK) mask_n = (lane < MW) ? mask_s[t * (uint64_t)MW + lane] : 0u;
      bit_n = bits_s[t];
      dec_n = dec_s[t];
      ctx_nn = (is_mx && t + 1 < T) ? ctx_s[(t + 1) * (uint64_t)M + lane] : 0u;
    }
    if (t == 0) {
      // nothing to compute yet: fall through to the commit below
    } else {
      // ================= compute bit t-1 ================================================
      const uint64_t tc = t - 1;
      float* const xin = in0 + xb * in0_sz;
      const uint32_t slot = d.lds_off + (cur ? d.pitch : 0u);
      const float* const wrow = lds + slot;
      const uint64_t rs_c = cur ? rs1 : rs0;
      // An unseen row is "no row": output 0, nothing accumulated (mixer.cpp:52-55).
      const bool seen = is_mx && rs_c != 0;

      if (lane < NS) skipv[lane] = xin[tp->skip_idx[lane]];  // raw, possibly stale (mixer.cpp:76-79)
      if (HAS_MASK) {
        // Only active_models are visited (mixer.cpp:57-59): silent slots contribute nothing.
        for (uint32_t c = (uint32_t)lane * 4u; c < (uint32_t)NPAD; c += 256) {
          float4 v = *(float4*)(xin + c);
          const uint32_t w = (uint32_t)__shfl((int)mask_c, (int)(c >> 5));
          const uint32_t b = w >> (c & 31u);
          v.x = (b & 1u) ? v.x : 0.f;
          v.y = (b & 2u) ? v.y : 0.f;
          v.z = (b & 4u) ? v.z : 0.f;
          v.w = (b & 8u) ? v.w : 0.f;
          *(float4*)(xin + c) = v;
        }
      }

      float acc = 0.f;
      if (a.mode & GMX_MODE_PREDICT) {
        // ---- layer 0, inputs 0..N-1: 24 chains side by side (mixer.cpp:56-59) ----------
        if (is_l0 && seen) {
          int j = 0;
#pragma unroll 4
          for (; j + 4 <= N; j += 4) {
            const float4 xv = *(const float4*)(xin + j);
            const float4 wv = *(const float4*)(wrow + j);
            acc = acc + xv.x * wv.x;
            acc = acc + xv.y * wv.y;
            acc = acc + xv.z * wv.z;
            acc = acc + xv.w * wv.w;
          }
          for (; j < N; ++j) acc = acc + xin[j] * wrow[j];
        }
        // ---- layer-0 cascade: mixer k adds outputs 0..k-1 in order (mixer.cpp:60-64) ----
        for (int i = 0; i + 1 < L0; ++i) {
          const float o = gmx_readlane_f(acc, i);
          if (is_l0 && lane > i && seen) acc = acc + o * wrow[N + i];
        }
        if (is_l0) xin[N + lane] = acc;  // mixer_layer0_outputs[k] (mixer.cpp:104)
        // ---- layer 1 and final: the 24 layer-0 outputs first (mixer.cpp:66-68, 82-84) ----
        if ((is_l1 || is_fin) && seen) {
          for (int i = 0; i < L0; ++i) acc = acc + xin[N + i] * wrow[i];
        }
        // ---- layer-1 cascade, then each mixer's skip inputs (mixer.cpp:69-80) -----------
        for (int i = 0; i < L1; ++i) {
          if (is_l1 && lane - L0 == i && seen) {
            for (int k = 0; k < NS; ++k) acc = acc + skipv[k] * wrow[L0 + i + k];
          }
          const float o = gmx_readlane_f(acc, L0 + i);
          if (is_l1 && lane - L0 > i && seen) acc = acc + o * wrow[L0 + i];
        }
        if (is_l1) o1s[lane - L0] = acc;  // mixer_layer1_outputs (mixer.cpp:102)
        // ---- final mixer: layer-1 outputs, skip inputs (mixer.cpp:85-97) ----------------
        if (is_fin && seen) {
          for (int i = 0; i < L1; ++i) acc = acc + o1s[i] * wrow[L0 + i];
          for (int k = 0; k < NS; ++k) acc = acc + skipv[k] * wrow[L0 + L1 + k];
        }
      } else {
        // Learn-only call of the per-bit surface: outputs were latched by the forward call.
        acc = is_mx ? latch_s[lane] : 0.f;
        if (is_l0) xin[N + lane] = acc;
        if (is_l1) o1s[lane - L0] = acc;
      }

      if (is_mx) {
        if (oa_s) oa_s[tc * (uint64_t)M + lane] = acc;
        if (a.mode & GMX_MODE_LATCH) latch_s[lane] = acc;
      }
      // Final squash + clamp of Predictor::Predict on the last mixer's logit (predictor.cpp:369-375).
      if (lane == M - 1) p_s[tc] = gmx_squash_clamp(acc);

      if (a.mode & GMX_MODE_LEARN) {
        // ================= Mixer::Learn (mixer.cpp:108-176) =============================
        if (is_mx) {
          // decay = float(0.9 / pow(1e-7*steps_+0.8, 0.8))   <- host, dec_c (mixer.cpp:111)
          // decay *= 1.5 - (1.0*data->steps)/max_steps_       (mixer.cpp:112, in double)
          const double dd = (double)dec_c * (1.5 - ((double)rs_c) / (double)max_steps);
          const float decay = (float)dd;
          const float p = gmx_logistic(acc);
          const float upd = decay * d.lr * (p - (float)bit_c);  // (mixer.cpp:123)
          ++steps;
          const uint64_t rs_new = rs_c + 1;
          if (rs_new > max_steps) max_steps = rs_new;
          if (rs_c == 0) ++seen_cnt;  // FindOrCreateMixerData (mixer.cpp:44-46)
          if (cur) rs1 = rs_new; else rs0 = rs_new;
          rs_tab[row_c] = rs_new;
          m_upd[lane] = upd;
          m_flag[lane] = ((rs_new & 1023u) == 0) ? 1u : 0u;  // shrink this visit (mixer.cpp:173)
          m_dst[lane] = slot;
          m_row[lane] = row_c;
        }
        const float shrink = 1.0f - 3.0e-6f;
        // layer-0 rows: 4 weights per lane, inputs are in0 = [x | layer-0 outputs]
        for (int m = 0; m < L0; ++m) {
          const float u = m_upd[m];
          const bool shr = m_flag[m] != 0;
          const uint32_t so = m_dst[m];
          const uint32_t ws = (uint32_t)(N + m);
          const uint32_t stride = tp->mx[m].stride;
          uint8_t* g = bank + tp->mx[m].w_off + (uint64_t)m_row[m] * stride * 4u;
          for (uint32_t c = (uint32_t)lane * 4u; c < ws; c += 256) {
            float4 w = *(float4*)(lds + so + c);
            float4 x = *(const float4*)(xin + c);
            x.y = (c + 1 < ws) ? x.y : 0.f;
            x.z = (c + 2 < ws) ? x.z : 0.f;
            x.w = (c + 3 < ws) ? x.w : 0.f;
            w.x = w.x - u * x.x;
            w.y = w.y - u * x.y;
            w.z = w.z - u * x.z;
            w.w = w.w - u * x.w;
            if (shr) {
              w.x *= shrink;
              w.y *= shrink;
              w.z *= shrink;
              w.w *= shrink;
            }
            *(float4*)(lds + so + c) = w;
            *(float4*)(g + (uint64_t)c * 4u) = w;
          }
        }
        // layer-1 and final rows (<= 64 weights): one weight per lane
        for (int m = L0; m < M; ++m) {
          const float u = m_upd[m];
          const bool shr = m_flag[m] != 0;
          const uint32_t so = m_dst[m];
          const int ws = (int)tp->mx[m].weight_size;
          const int casc = (tp->mx[m].layer == 1) ? (m - L0) : L1;  // own-layer inputs
          uint8_t* g = bank + tp->mx[m].w_off + (uint64_t)m_row[m] * tp->mx[m].stride * 4u;
          for (int c = lane; c < ws; c += 64) {
            float x;
            if (c < L0) x = xin[N + c];
            else if (c < L0 + casc) x = o1s[c - L0];
            else x = skipv[c - L0 - casc];
            float w = lds[so + c];
            w = w - u * x;
            if (shr) w *= shrink;
            lds[so + c] = w;
            *(float*)(g + (uint64_t)c * 4u) = w;
          }
        }
      }
    }
    // ================= commit the prefetch issued above ==================================
    if (t < T) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (need) {
        if (cur) { tag0 = row_n; rs0 = rs_ld; } else { tag1 = row_n; rs1 = rs_ld; }
      }
      const uint32_t tcur = cur ? tag1 : tag0;
      if (is_mx && row_n != tcur) cur ^= 1u;
      xb ^= 1u;
      row_c = row_n;
      mask_c = mask_n;
      bit_c = bit_n;
      dec_c = dec_n;
    }
  }
  if (is_mx && (a.mode & GMX_MODE_LEARN)) {
    scal[0] = steps;
    scal[1] = max_steps;
    scal[2] = seen_cnt;
  }
}

template __global__ void gmx_bank_kernel<false>(const GmxTopoDev*, const GmxRunArgs);
template __global__ void gmx_bank_kernel<true>(const GmxTopoDev*, const GmxRunArgs);

extern "C" hipError_t gmx_launch_bank_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args,
                                             int n_streams, unsigned lds_bytes, int has_mask,
                                             hipStream_t stream) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call is not this launch's
  if (has_mask)
    hipLaunchKernelGGL(gmx_bank_kernel<true>, dim3(n_streams), dim3(64), lds_bytes, stream, tp_dev,
                       *args);
  else
    hipLaunchKernelGGL(gmx_bank_kernel<false>, dim3(n_streams), dim3(64), lds_bytes, stream, tp_dev,
                       *args);
  return hipGetLastError();
}

extern "C" hipError_t gmx_bank_kernel_set_lds(unsigned lds_bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)gmx_bank_kernel<true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)gmx_bank_kernel<false>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

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
    float r = what == 0 ? gmx_expf(v) : (what == 1 ? gmx_logistic(v) : gmx_squash_clamp(v));
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
