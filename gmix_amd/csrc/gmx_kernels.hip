// gmx_kernels.hip -- the general mixer-bank kernel for gfx950 (hand-written HIP, CDNA4 only):
// any topology gmx_group_create accepts, in particular the stock 24/8/1 bank of
// Predictor::AddMixers (predictor.cpp:251-358).
//
// What runs here is the body of Mixer::Predict and Mixer::Learn (mixer/mixer.cpp:51-176) for
// all mixers of a bank, bit after bit, with the reference's exact fp32 operation order:
// every dot product is one strict left-to-right chain of (multiply, then add), never fused,
// never re-associated; the update is w -= update * x with two roundings.  Parallelism comes
// from what the reference leaves independent:
//   * lanes of a wave  = the mixers of one layer (24 layer-0 chains advance in lock step,
//     then the layer-0 cascade, then the 8 layer-1 chains, then the final mixer); values that
//     cross mixers (cascade outputs, per-mixer update factors) travel by v_readlane /
//     ds_bpermute, never through memory;
//   * lanes of a wave  = the elements of weight rows for the loads, the update and the
//     stores (each row is a contiguous, 128-byte aligned run in HBM: fully coalesced; rows of
//     <= 128 floats are updated two at a time, 32 lanes each);
//   * waves            = independent byte-streams (one bank each), one wave per stream.
// Rows are staged through LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip) into
// a two-slot, tag-checked row cache per mixer, so that the rows of bit t+1 (known as soon as
// its contexts are) stream in while bit t is being computed, and a row that is used again
// on the next bit -- the common case for the small gate tables of the stock topology --
// never leaves the CU.  The transposed (lane = mixer) reads of that image are
// bank-conflict-free because consecutive mixers' rows are pitch = stride + 4 floats apart.
//
// All vector-memory instructions of the bit loop are issued from inline asm: hipcc's own
// s_waitcnt placement would wait for the stores of bit t before touching the prefetched
// data of bit t+1 (everything shares one in-order vmcnt queue).  Here one counted
// s_waitcnt vmcnt(#stores of the bit) per bit waits for exactly the prefetch loads, which
// are older than that bit's stores.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile); the parity of the
// whole file rests on no contraction and IEEE divide, both asserted by tests on the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmx_internal.h"
#include "gmx_math.h"

typedef float gmx_f4 __attribute__((ext_vector_type(4)));

// ---- hand-issued vector memory (see file header) -----------------------------------------
__device__ __forceinline__ uint32_t gmx_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// One LDS-DMA piece: every active lane moves 16 bytes from its own global address to
// LDS[lds_byte_off + 16*lane] (the LDS base travels in M0, saved and restored around it).
__device__ __forceinline__ void gmx_dma16(const void* gsrc, uint32_t lds_byte_off) {
  uint32_t saved;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(gsrc), "s"(lds_byte_off)
      : "memory");
}
// Same with a scalar 64-bit base and a 32-bit per-lane byte offset.
__device__ __forceinline__ void gmx_dma16s(uint64_t sbase, uint32_t voff, uint32_t lds_byte_off) {
  uint32_t saved;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(saved)
      : "s"(sbase), "v"(voff), "s"(lds_byte_off)
      : "memory");
}
__device__ __forceinline__ void gmx_vld8(uint64_t& d, const uint64_t* p) {
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gmx_vld4(uint32_t& d, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gmx_vld1(uint32_t& d, const uint8_t* p) {
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gmx_vst16(void* p, const gmx_f4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
// Store from the lanes of `mask` only (mask non-empty, all lanes active on entry): the exec
// juggling stays inside the asm so that the compiler sees straight-line code.
__device__ __forceinline__ void gmx_vst16_m(void* p, const gmx_f4& v, uint64_t mask) {
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\t"
      "global_store_dwordx4 %1, %2, off\n\ts_mov_b64 exec, %0"
      : "=&s"(saved)
      : "v"(p), "v"(v), "s"(mask)
      : "memory");
}
__device__ __forceinline__ void gmx_vst8(void* p, uint64_t v) {
  asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void gmx_vst4(void* p, float v) {
  asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void gmx_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate).
__device__ __forceinline__ void gmx_wait_vm(int n) {
  switch (n) {
#define GMX_C(k) case k: gmx_vmcnt<k>(); break;
    GMX_C(1) GMX_C(2) GMX_C(3) GMX_C(4) GMX_C(5) GMX_C(6) GMX_C(7) GMX_C(8) GMX_C(9) GMX_C(10)
    GMX_C(11) GMX_C(12) GMX_C(13) GMX_C(14) GMX_C(15) GMX_C(16) GMX_C(17) GMX_C(18) GMX_C(19)
    GMX_C(20) GMX_C(21) GMX_C(22) GMX_C(23) GMX_C(24) GMX_C(25) GMX_C(26) GMX_C(27) GMX_C(28)
    GMX_C(29) GMX_C(30) GMX_C(31) GMX_C(32) GMX_C(33) GMX_C(34) GMX_C(35) GMX_C(36) GMX_C(37)
    GMX_C(38) GMX_C(39) GMX_C(40) GMX_C(41) GMX_C(42) GMX_C(43) GMX_C(44) GMX_C(45) GMX_C(46)
    GMX_C(47) GMX_C(48) GMX_C(49) GMX_C(50) GMX_C(51) GMX_C(52) GMX_C(53) GMX_C(54) GMX_C(55)
    GMX_C(56) GMX_C(57) GMX_C(58) GMX_C(59) GMX_C(60) GMX_C(61) GMX_C(62) GMX_C(63)
#undef GMX_C
    default: gmx_vmcnt<0>(); break;
  }
}

__device__ __forceinline__ float gmx_readlane_f(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ uint32_t gmx_readlane_u(uint32_t v, int src_lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}
__device__ __forceinline__ uint32_t gmx_bperm_u(uint32_t v, int src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

constexpr int GMX_CH = 8;  // cascade / small-layer weights are preloaded 8 at a time

// TL0 / TL1 / TNS / TFIN: layer sizes, skip count and "has a final mixer" as compile-time
// constants (loops unroll, guards fold); TLPR / TCH0: lanes per row and 4*TLPR-float chunks per
// row of the layer-0 update pass (a function of n_inputs) -- or -1 for the run-time version.
template <bool HAS_MASK, int TL0, int TL1, int TNS, int TFIN, int TLPR, int TCH0>
__global__ void __launch_bounds__(64)
gmx_bank_kernel(const GmxTopoDev* __restrict__ tp, const GmxRunArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int rec = a.rec_base + (int)blockIdx.x;   // stream index inside the record arrays
  const int s = a.stream_base + (int)blockIdx.x;  // bank index
  const int N = tp->n, NPAD = tp->n_pad;
  const int L0 = TL0 >= 0 ? TL0 : tp->l0, L1 = TL1 >= 0 ? TL1 : tp->l1;
  const int NS = TNS >= 0 ? TNS : tp->n_skip;
  const bool has_final = TFIN >= 0 ? (TFIN != 0) : (tp->has_final != 0);
  const int M = L0 + L1 + (has_final ? 1 : 0);
  const int MW = tp->mask_words;
  const uint64_t T = a.T_list ? a.T_list[blockIdx.x] : a.T;  // (wave-uniform: one stream per block)
  if (T == 0) return;
  uint8_t* const bank = a.banks + (uint64_t)s * tp->bank_bytes;
  const bool do_predict = (a.mode & GMX_MODE_PREDICT) != 0;
  const bool do_learn = (a.mode & GMX_MODE_LEARN) != 0;
  const bool do_latch = (a.mode & GMX_MODE_LATCH) != 0;

  float* const in0 = lds + tp->lds_in0;  // [2][in0_sz]: [x (n) | layer-0 outputs (l0)]
  const uint32_t in0_sz = tp->in0_sz;
  float* const o1s = lds + tp->lds_o1;
  float* const skipv = lds + tp->lds_skip;
  // expf's 2^(i/32) table in LDS: its lookups use lgkmcnt, not the vmcnt queue.
  uint64_t* const s_tab = (uint64_t*)(lds + tp->lds_misc);
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];

  const bool is_mx = lane < M;
  const GmxMixerDev d = tp->mx[is_mx ? lane : 0];
  const bool is_l0 = lane < L0;
  const bool is_l1 = lane >= L0 && lane < L0 + L1;
  const bool is_fin = has_final && lane == L0 + L1;
  int skip_idx = 0;
  if (lane < NS) skip_idx = tp->skip_idx[lane];

  const uint64_t RS = a.rec_stride;
  const float* const pred_s = a.pred + (uint64_t)rec * RS * NPAD;
  const uint32_t* const mask_s = HAS_MASK ? a.mask + (uint64_t)rec * RS * MW : nullptr;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS * M;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  const float* const dec_s = a.decay + (uint64_t)a.decay_idx[blockIdx.x] * a.T;
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
  uint8_t* const rs_base = bank + d.rs_off;  // row r's step counter: rs_base + r * d.rs_pitch (gmx_internal.h)
  uint8_t* const w_tab = bank + d.w_off;

  // Geometry of the layer-0 update pass: rows of <= 64 / <= 128 floats are processed 4 / 2
  // at a time (16 / 32 lanes each), longer rows one at a time in 256-float chunks.
  const uint32_t stride0 = tp->mx[L0 - 1].stride;  // the longest layer-0 row
  const int LPR = TLPR > 0 ? TLPR : (stride0 <= 64 ? 16 : (stride0 <= 128 ? 32 : 64));
  const int RP = 64 / LPR;
  const int chunks0 = TCH0 > 0 ? TCH0 : (int)((stride0 + 4u * LPR - 1) / (4u * LPR));
  const int sub = lane / LPR, lr_ = lane % LPR;
  // Vector-memory instructions issued per bit AFTER the prefetch loads (all stores): the
  // count the commit wait leaves outstanding.  Every one of them has at least one active
  // lane on every path, so none is ever skipped.
  int n_st = 1 + (oa_s ? 1 : 0) + (do_latch ? 1 : 0);
  if (do_learn) {
    n_st += 1 + (M - L0);
    for (int m0 = 0; m0 < L0; m0 += RP) {
      const int ws_max = N + (m0 + RP - 1 < L0 ? m0 + RP - 1 : L0 - 1);
      for (int ch = 0; ch < chunks0; ++ch)
        if (ch * 4 * LPR < ws_max) ++n_st;  // a chunk past every row of the pass has no lanes
    }
  }

  // Two-slot row cache of this lane's mixer: tags (row index) and MixerData::steps of the
  // rows held; `cur` is the slot of the row in use for the current bit.
  uint32_t tag0 = 0xffffffffu, tag1 = 0xffffffffu;
  uint64_t rs0 = 0, rs1 = 0;
  uint32_t cur = 0, xb = 0;

  // _n: the bit being prefetched; _c: the bit being computed.
  uint32_t ctx_nn = is_mx ? ctx_s[lane] : 0;
  uint32_t row_n = 0, row_c = 0, mask_n = ~0u, mask_c = ~0u, bit_n = 0, bit_c = 0;
  uint32_t dec_n = 0, dec_c = 0;
  uint64_t rs_ld = 0;
  bool need = false;

  float acc_last = 0.f;
  for (uint64_t t = 0; t <= T; ++t) {
    // ================= prefetch bit t (runs one iteration ahead of its compute) ==========
    if (t < T) {
      row_n = ctx_nn % d.table_size;  // FindMixerData: context_ % table size (mixer.cpp:32)
      const uint32_t tag_c = cur ? tag1 : tag0, tag_o = cur ? tag0 : tag1;
      need = is_mx && row_n != tag_c && row_n != tag_o;
      const uint32_t dst = d.lds_off + (cur ? 0u : d.pitch);  // the slot that is not in use
      if (need) gmx_vld8(rs_ld, (const uint64_t*)(rs_base + (uint64_t)row_n * d.rs_pitch));
      uint64_t nm = __ballot(need);
      const uint32_t lds_base = gmx_lds_addr(lds);
      // every lane works out where its own mixer's row lives and where it goes; the issue
      // loop then only moves those three words into scalar registers
      const uint64_t gsrc = (uint64_t)w_tab + (uint64_t)row_n * (d.stride * 4u);
      const uint32_t gs_lo = (uint32_t)gsrc, gs_hi = (uint32_t)(gsrc >> 32);
      const uint32_t ldst = lds_base + dst * 4u;
      const uint32_t lane16 = (uint32_t)lane * 16u;
      while (nm) {
        const int m = __builtin_amdgcn_readfirstlane((int)__ffsll((unsigned long long)nm) - 1);
        nm &= nm - 1;
        const uint64_t sb = ((uint64_t)gmx_readlane_u(gs_hi, m) << 32) | gmx_readlane_u(gs_lo, m);
        const uint32_t ld = gmx_readlane_u(ldst, m);
        const uint32_t nbytes = gmx_readlane_u(d.stride, m) * 4u;
        for (uint32_t b0 = 0; b0 < nbytes; b0 += 1024)
          if (b0 + lane16 < nbytes) gmx_dma16s(sb + b0, lane16, ld + b0);
      }
      {
        const uint64_t g = (uint64_t)(pred_s + t * (uint64_t)NPAD);
        const uint32_t dstx = lds_base + (tp->lds_in0 + (xb ^ 1u) * in0_sz) * 4u;
        const uint32_t nbytes = (uint32_t)NPAD * 4u;
        for (uint32_t b0 = 0; b0 < nbytes; b0 += 1024)
          if (b0 + lane16 < nbytes) gmx_dma16s(g + b0, lane16, dstx + b0);
      }
      if (HAS_MASK) {
        mask_n = 0u;
        if (lane < MW) gmx_vld4(mask_n, mask_s + t * (uint64_t)MW + lane);
      }
      gmx_vld1(bit_n, bits_s + t);
      gmx_vld4(dec_n, dec_s + t);
      ctx_nn = 0u;
      if (is_mx && t + 1 < T) gmx_vld4(ctx_nn, ctx_s + (t + 1) * (uint64_t)M + lane);
    }
    if (t > 0) {
      // ================= compute bit t-1 ================================================
      const uint64_t tc = t - 1;
      float* const xin = in0 + xb * in0_sz;
      const uint32_t slot = d.lds_off + (cur ? d.pitch : 0u);
      float* const wrow = lds + slot;
      const uint64_t rs_c = cur ? rs1 : rs0;
      // An unseen row is "no row": output 0, nothing accumulated (mixer.cpp:52-55).
      const bool seen = is_mx && rs_c != 0;

      if (lane < NS) skipv[lane] = xin[skip_idx];  // raw, possibly stale (mixer.cpp:76-79)
      if (HAS_MASK) {
        // Only active_models are visited (mixer.cpp:57-59): silent slots contribute nothing.
        for (uint32_t c = (uint32_t)lane * 4u; c < (uint32_t)NPAD; c += 256) {
          float4 v = *(float4*)(xin + c);
          const uint32_t w = gmx_bperm_u(mask_c, (int)(c >> 5));
          const uint32_t b = w >> (c & 31u);
          v.x = (b & 1u) ? v.x : 0.f;
          v.y = (b & 2u) ? v.y : 0.f;
          v.z = (b & 4u) ? v.z : 0.f;
          v.w = (b & 8u) ? v.w : 0.f;
          *(float4*)(xin + c) = v;
        }
      }

      float acc = 0.f;
      if (do_predict) {
        // ---- layer 0, inputs 0..N-1: 24 chains side by side (mixer.cpp:56-59) ----------
        if (is_l0 && seen) {
          int j = 0;
#pragma unroll 2
          for (; j + 8 <= N; j += 8) {
            const float4 xa = *(const float4*)(xin + j), xb4 = *(const float4*)(xin + j + 4);
            const float4 wa = *(const float4*)(wrow + j), wb = *(const float4*)(wrow + j + 4);
            acc = acc + xa.x * wa.x;
            acc = acc + xa.y * wa.y;
            acc = acc + xa.z * wa.z;
            acc = acc + xa.w * wa.w;
            acc = acc + xb4.x * wb.x;
            acc = acc + xb4.y * wb.y;
            acc = acc + xb4.z * wb.z;
            acc = acc + xb4.w * wb.w;
          }
          for (; j < N; ++j) acc = acc + xin[j] * wrow[j];
        }
        // ---- layer-0 cascade: mixer k adds outputs 0..k-1 in order (mixer.cpp:60-64) ----
#pragma unroll
        for (int i0 = 0; i0 + 1 < L0; i0 += GMX_CH) {
          float wt[GMX_CH];
#pragma unroll
          for (int j = 0; j < GMX_CH; ++j) wt[j] = (is_l0 && i0 + j + 1 < L0) ? wrow[N + i0 + j] : 0.f;
#pragma unroll
          for (int j = 0; j < GMX_CH; ++j) {
            const int i = i0 + j;
            if (i + 1 < L0) {
              const float o = gmx_readlane_f(acc, i);
              if (is_l0 && lane > i && seen) acc = acc + o * wt[j];
            }
          }
        }
        const float out0 = acc;          // mixer_layer0_outputs[k] in lane k (mixer.cpp:104)
        if (is_l0) xin[N + lane] = acc;  // the update pass reads them as inputs
        // ---- layer 1 and final: the layer-0 outputs first (mixer.cpp:66-68, 82-84) -------
        const bool up = (is_l1 || is_fin) && seen;
#pragma unroll
        for (int i0 = 0; i0 < L0; i0 += GMX_CH) {
          float wt[GMX_CH];
#pragma unroll
          for (int j = 0; j < GMX_CH; ++j) wt[j] = (up && i0 + j < L0) ? wrow[i0 + j] : 0.f;
#pragma unroll
          for (int j = 0; j < GMX_CH; ++j) {
            const int i = i0 + j;
            if (i < L0) {
              const float o = gmx_readlane_f(out0, i);
              if (up) acc = acc + o * wt[j];
            }
          }
        }
        // ---- layer-1 cascade, then each mixer's skip inputs (mixer.cpp:69-80) -----------
        {
          float wsk[GMX_MAX_SKIP];
#pragma unroll
          for (int k = 0; k < GMX_MAX_SKIP; ++k)
            wsk[k] = (is_l1 && k < NS) ? wrow[L0 + (lane - L0) + k] : 0.f;
#pragma unroll
          for (int i0 = 0; i0 < L1; i0 += GMX_CH) {
            float wt[GMX_CH];
#pragma unroll
            for (int j = 0; j < GMX_CH; ++j) wt[j] = (is_l1 && i0 + j < L1) ? wrow[L0 + i0 + j] : 0.f;
#pragma unroll
            for (int j = 0; j < GMX_CH; ++j) {
              const int i = i0 + j;
              if (i < L1) {
                if (is_l1 && lane - L0 == i && seen) {
#pragma unroll
                  for (int k = 0; k < GMX_MAX_SKIP; ++k)
                    if (k < NS) acc = acc + skipv[k] * wsk[k];
                }
                const float o = gmx_readlane_f(acc, L0 + i);
                if (is_l1 && lane - L0 > i && seen) acc = acc + o * wt[j];
              }
            }
          }
        }
        const float out1 = acc;           // mixer_layer1_outputs in lanes L0.. (mixer.cpp:102)
        if (is_l1) o1s[lane - L0] = acc;  // inputs of the update pass
        // ---- final mixer: layer-1 outputs, skip inputs (mixer.cpp:85-97) ----------------
        if (has_final) {
          const bool fin = is_fin && seen;
#pragma unroll
          for (int i0 = 0; i0 < L1; i0 += GMX_CH) {
            float wt[GMX_CH];
#pragma unroll
            for (int j = 0; j < GMX_CH; ++j) wt[j] = (fin && i0 + j < L1) ? wrow[L0 + i0 + j] : 0.f;
#pragma unroll
            for (int j = 0; j < GMX_CH; ++j) {
              const int i = i0 + j;
              if (i < L1) {
                const float o = gmx_readlane_f(out1, L0 + i);
                if (fin) acc = acc + o * wt[j];
              }
            }
          }
          if (fin)
#pragma unroll
            for (int k = 0; k < NS; ++k) acc = acc + skipv[k] * wrow[L0 + L1 + k];
        }
      } else {
        // Learn-only call of the per-bit surface: outputs were latched by the forward call.
        acc = is_mx ? latch_s[lane] : 0.f;
        if (is_l0) xin[N + lane] = acc;
        if (is_l1) o1s[lane - L0] = acc;
      }

      // Sigmoid::Logistic of every mixer's own output: the last mixer's is what
      // Predictor::Predict returns after clamping (predictor.cpp:369-375), all of them feed
      // Mixer::Learn (mixer.cpp:113-122).
      const float pl = gmx_logistic_tab(acc, s_tab);
      acc_last = acc;
      if (lane == M - 1) gmx_vst4(p_s + tc, gmx_clamp_prob(pl));
      if (oa_s && is_mx) gmx_vst4(oa_s + tc * (uint64_t)M + lane, acc);
      if (do_latch && is_mx) gmx_vst4(latch_s + lane, acc);

      if (do_learn) {
        // ================= Mixer::Learn (mixer.cpp:108-176) =============================
        // decay = float(0.9 / pow(1e-7*steps_+0.8, 0.8))   <- host, dec_c (mixer.cpp:111)
        // decay *= 1.5 - (1.0*data->steps)/max_steps_       (mixer.cpp:112, in double)
        const double dd = (double)__uint_as_float(dec_c) * (1.5 - ((double)rs_c) / (double)max_steps);
        const float decay = (float)dd;
        const float upd = decay * d.lr * (pl - (float)bit_c);  // (mixer.cpp:123)
        const uint64_t rs_new = rs_c + 1;
        const uint32_t shr = ((rs_new & 1023u) == 0) ? 1u : 0u;  // shrink this visit (mixer.cpp:173)
        if (is_mx) {
          ++steps;
          if (rs_new > max_steps) max_steps = rs_new;
          if (rs_c == 0) ++seen_cnt;  // FindOrCreateMixerData (mixer.cpp:44-46)
          if (cur) rs1 = rs_new; else rs0 = rs_new;
          gmx_vst8((uint64_t*)(rs_base + (uint64_t)row_c * d.rs_pitch), rs_new);
        }
        const float shrink = 1.0f - 3.0e-6f;
        const uint64_t grow = (uint64_t)w_tab + (uint64_t)row_c * d.stride * 4u;  // this mixer's row in HBM
        const uint32_t grow_lo = (uint32_t)grow, grow_hi = (uint32_t)(grow >> 32);
        const uint32_t upd_u = __float_as_uint(upd);
        // ---- layer-0 rows: RP rows per pass, 4 weights per lane; inputs in0 = [x | out0] ----
        if (LPR < 64 && 4u * (uint32_t)LPR <= stride0) {
          // Rows of exactly 64 / 128 floats: no branches, so the passes overlap each other's LDS
          // latency.  Every lane of a row's LPR lanes runs the update; lanes past
          // weight_size see x = 0 (their stored weights are the zero padding and stay zero),
          // lanes past the stored row only skip the HBM store (exec mask inside the asm).
          const uint32_t c = 4u * (uint32_t)lr_;
          // (a folded step counter lives in the row's last quad: that quad is never stored as weights)
          const uint64_t st_mask = __ballot(c < stride0 - (d.rs_folded == 1u ? 4u : 0u));
#pragma unroll
          for (int m0 = 0; m0 < L0; m0 += RP) {
            const int m = m0 + sub;
            const int mm = m < L0 ? m : m0;  // an odd tail repeats row m0: same values, same address
            const float u = __uint_as_float(gmx_bperm_u(upd_u, mm));
            const uint32_t so = gmx_bperm_u(slot, mm);
            const bool sh = gmx_bperm_u(shr, mm) != 0;
            uint8_t* g = (uint8_t*)(((uint64_t)gmx_bperm_u(grow_hi, mm) << 32) | gmx_bperm_u(grow_lo, mm));
            const uint32_t ws = (uint32_t)(N + mm);
            float4* wp = (float4*)__builtin_assume_aligned(lds + so + c, 16);
            float4 w = *wp;
            float4 x = *(const float4*)__builtin_assume_aligned(xin + c, 16);
            x.x = (c + 0 < ws) ? x.x : 0.f;
            x.y = (c + 1 < ws) ? x.y : 0.f;
            x.z = (c + 2 < ws) ? x.z : 0.f;
            x.w = (c + 3 < ws) ? x.w : 0.f;
            const float scl = sh ? shrink : 1.0f;  // * 1.0f is exact
            w.x = (w.x - u * x.x) * scl;
            w.y = (w.y - u * x.y) * scl;
            w.z = (w.z - u * x.z) * scl;
            w.w = (w.w - u * x.w) * scl;
            *wp = w;
            gmx_f4 wv;
            wv.x = w.x; wv.y = w.y; wv.z = w.z; wv.w = w.w;
            gmx_vst16_m(g + (uint64_t)c * 4u, wv, st_mask);
          }
        } else
#pragma unroll
        for (int m0 = 0; m0 < L0; m0 += RP) {
          const int m = m0 + sub;
          const bool valid = m < L0;
          const int mm = valid ? m : m0;
          const float u = __uint_as_float(gmx_bperm_u(upd_u, mm));
          const uint32_t so = gmx_bperm_u(slot, mm);
          const bool sh = gmx_bperm_u(shr, mm) != 0;
          uint8_t* g = (uint8_t*)(((uint64_t)gmx_bperm_u(grow_hi, mm) << 32) | gmx_bperm_u(grow_lo, mm));
          const uint32_t ws = (uint32_t)(N + mm);
#pragma unroll
          for (int ch = 0; ch < chunks0; ++ch) {
            const uint32_t c = (uint32_t)(ch * 4 * LPR + 4 * lr_);
            if (valid && c < ws) {
              float4 w = *(float4*)(lds + so + c);
              float4 x = *(const float4*)(xin + c);
              x.y = (c + 1 < ws) ? x.y : 0.f;
              x.z = (c + 2 < ws) ? x.z : 0.f;
              x.w = (c + 3 < ws) ? x.w : 0.f;
              w.x = w.x - u * x.x;
              w.y = w.y - u * x.y;
              w.z = w.z - u * x.z;
              w.w = w.w - u * x.w;
              if (sh) {
                w.x *= shrink;
                w.y *= shrink;
                w.z *= shrink;
                w.w *= shrink;
              }
              *(float4*)(lds + so + c) = w;
              gmx_f4 wv;
              wv.x = w.x; wv.y = w.y; wv.z = w.z; wv.w = w.w;
              gmx_vst16(g + (uint64_t)c * 4u, wv);
            }
          }
        }
        // ---- layer-1 and final rows (<= 64 weights): one weight per lane ------------------
        // Weight c of such a row multiplies, in order: the layer-0 outputs (c < L0), the
        // layer-1 outputs before it (own cascade; all of them for the final mixer), then the
        // skip inputs -- and output c of layers 0/1 is exactly what lane c holds in `acc`.
        float sk[GMX_MAX_SKIP];
#pragma unroll
        for (int k = 0; k < GMX_MAX_SKIP; ++k) sk[k] = (k < NS) ? skipv[k] : 0.f;
#pragma unroll
        for (int m = L0; m < M; ++m) {
          const float u = gmx_readlane_f(upd, m);
          const bool sh = gmx_readlane_u(shr, m) != 0;
          const uint32_t so = gmx_readlane_u(slot, m);
          const int casc = (m < L0 + L1) ? (m - L0) : L1;  // own-layer inputs
          const int ws = L0 + casc + NS;                   // weight_size_ (mixer.cpp:21-25)
          uint8_t* g = (uint8_t*)(((uint64_t)gmx_readlane_u(grow_hi, m) << 32) | gmx_readlane_u(grow_lo, m));
          const int c = lane;
          float x = acc;
#pragma unroll
          for (int k = 0; k < GMX_MAX_SKIP; ++k)
            if (k < NS) x = (c == L0 + casc + k) ? sk[k] : x;
          x = (c < ws) ? x : 0.f;  // stored rows are 64 floats: the padding stays zero
          float w = lds[so + c];
          w = w - u * x;
          w = sh ? w * shrink : w;
          lds[so + c] = w;
          // (a folded step counter -- the last quad, or floats 36/37 behind the weights -- is never stored as weights)
          const bool is_rs = d.rs_folded == 1u ? c >= 60 : (d.rs_folded == 2u && (c == 36 || c == 37));
          if (!is_rs) gmx_vst4(g + (uint64_t)c * 4u, w);
        }
      }
    }
    // ================= commit the prefetch issued above ==================================
    if (t < T) {
      // Everything older than this bit's stores has landed (at t == 0 there are no stores).
      gmx_wait_vm(t == 0 ? 0 : n_st);
      asm volatile("" : "+v"(rs_ld), "+v"(mask_n), "+v"(bit_n), "+v"(dec_n), "+v"(ctx_nn));
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
  if (a.out_last && is_mx) a.out_last[(uint64_t)rec * M + lane] = acc_last;  // the last bit's outputs
  gmx_vmcnt<0>();
  if (is_mx && do_learn) {
    scal[0] = steps;
    scal[1] = max_steps;
    scal[2] = seen_cnt;
  }
}

typedef void (*gmx_bank_fn)(const GmxTopoDev*, const GmxRunArgs);

// Shape dispatch: the stock 24/8/1 layer sizes with one skip input (Predictor::AddMixers,
// predictor.cpp:251-358) get the fully unrolled build, everything else the run-time one.
// stride0 = stored length (floats) of the longest layer-0 row: <= 128 -> two rows per update
// pass (the stock 90-input bank), 257..512 -> one row in two 256-float chunks (256 inputs).
static gmx_bank_fn gmx_pick_bank_kernel(int l0, int l1, int ns, int fin, unsigned stride0,
                                        int has_mask) {
  if (l0 == 24 && l1 == 8 && ns == 1 && fin == 1) {
    if (stride0 > 64 && stride0 <= 128)
      return has_mask ? gmx_bank_kernel<true, 24, 8, 1, 1, 32, 1> : gmx_bank_kernel<false, 24, 8, 1, 1, 32, 1>;
    if (stride0 > 256 && stride0 <= 512)
      return has_mask ? gmx_bank_kernel<true, 24, 8, 1, 1, 64, 2> : gmx_bank_kernel<false, 24, 8, 1, 1, 64, 2>;
  }
  return has_mask ? gmx_bank_kernel<true, -1, -1, -1, -1, -1, -1>
                  : gmx_bank_kernel<false, -1, -1, -1, -1, -1, -1>;
}

extern "C" hipError_t gmx_launch_bank_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args,
                                             int n_streams, unsigned lds_bytes, int has_mask,
                                             int l0, int l1, int ns, int fin, unsigned stride0,
                                             hipStream_t stream) {
  (void)hipGetLastError();  // a stale error of an unrelated earlier call is not this launch's
  gmx_bank_fn fn = gmx_pick_bank_kernel(l0, l1, ns, fin, stride0, has_mask);
  hipLaunchKernelGGL(fn, dim3(n_streams), dim3(64), lds_bytes, stream, tp_dev, *args);
  return hipGetLastError();
}

extern "C" hipError_t gmx_bank_kernel_set_lds(unsigned lds_bytes) {
  gmx_bank_fn fns[] = {gmx_bank_kernel<true, 24, 8, 1, 1, 32, 1>, gmx_bank_kernel<false, 24, 8, 1, 1, 32, 1>,
                       gmx_bank_kernel<true, 24, 8, 1, 1, 64, 2>, gmx_bank_kernel<false, 24, 8, 1, 1, 64, 2>,
                       gmx_bank_kernel<true, -1, -1, -1, -1, -1, -1>,
                       gmx_bank_kernel<false, -1, -1, -1, -1, -1, -1>};
  for (gmx_bank_fn f : fns) {
    hipError_t e = hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
