// gmx_wide.hip -- the 256-input 24/8/1 bank (BASELINE.json configs[2]/[3]: "256 inputs, 3 GLN
// layers") with its rows resident in registers.
//
// Per coded bit a stream reads and writes 24 layer-0 rows of 256+k weights, 8 layer-1 rows of
// 25+k and the final row of 33 (mixer.cpp:17-26): 54.6 KB of algorithmic traffic for 26.7 kflop --
// HBM-bound like every mixer shape.  The general kernel (gmx_kernels.hip) stages those rows in
// LDS (65 KB per stream), which leaves two waves per CU: half the SIMDs idle, every wave exposed to
// the full memory latency.  Here a row lives in the registers of the lanes that use it:
//
//   * one wave = one stream.  Layer-0 mixer m is the lane PAIR (m, m+32): lane m holds weights
//     [0, 144) of its 288-float stored row, lane m+32 holds [144, 288) = inputs 144..255, then the
//     cascade weights (outputs of layer-0 mixers 0..m-1, mixer.cpp:60-64), then zero padding.
//     36 global_load_dwordx4 per lane bring a row in, 36 stores take it out -- and only when the
//     gate context selects a different row (tag check, write-back on replacement: a row that is
//     used again on the next bit never leaves the registers).
//   * the strict left-to-right sum of a layer-0 mixer (mixer.cpp:56-59) runs as two phases of the
//     same wave: lanes 0..23 add inputs 0..143, hand their partial sum to lane+32 (ds_bpermute),
//     lanes 32..55 continue with inputs 144..255 and the cascade.  Each phase advances all 24
//     mixers in lock step, one multiply and one add per input, the inputs broadcast from LDS.
//     The two halves of a row cost no extra time, only the registers of otherwise idle lanes:
//     144 weights per lane instead of 288, so two waves fit a SIMD and hide each other's latency.
//   * layer-1 mixers are lanes 24..31, the final mixer is lane 56; their rows (<= 33 weights)
//     use the first 9 of the 36 register quads.  Layer outputs cross lanes by v_readlane.
//   * the update w -= update * x (mixer.cpp:129-172) runs on both halves at once with exactly the
//     inputs the forward pass used; weights past weight_size see x = 0 and stay zero.
//   * the next bit's contexts, inputs, mask, bit and decay factor are requested behind this bit's
//     row traffic (inline-asm loads) and collected by one s_waitcnt at the end of the bit.
// Same floats as the general kernel and the oracle (tests/test_gpu_wide.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmx_internal.h"
#include "gmx_math.h"

typedef float gmx_f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kL0 = 24, kL1 = 8, kM = kL0 + kL1 + 1;
constexpr int kQS = 9;         // quads of a layer-1 / final row that can hold weights (<= 33 of 64 stored floats)
constexpr int kFinLane = 56;
constexpr int kPitchS = 4 * kQS + 4;

// The shape: N inputs, layer-0 rows stored as 2 * HALF floats (weight sizes N .. N+23, zero padded),
// HALF of them per lane of a pair.  256 inputs: HALF 144; the reference's 90 inputs: HALF 64.
template <int N, int HALF>
struct WideShape {
  static constexpr int kN = N, kHalf = HALF;
  static constexpr int kNPad = (N + 3) / 4 * 4;   // floats of a prediction record
  static constexpr int kMW = (N + 31) / 32;       // words of an active mask
  static constexpr int kQ = HALF / 4;             // register quads per lane
  static constexpr int kUp = N - HALF;            // inputs the upper half faces; its cascade weights follow
  static constexpr int kQU = (kUp + 3) / 4;       // quads of the upper half that face inputs (the last one maybe partly)
  // Staging image of the rows in LDS (floats): 48 layer-0 halves `kPitch` apart -- HALF floats + 4,
  // so that the 16 lanes of a ds_read_b128 group, each reading its own row at the same quad, hit
  // 16 different bank quads -- then the 9 small rows `kPitchS` apart.
  static constexpr int kPitch = HALF + 4;
  static constexpr int kStageSmall = 2 * kL0 * kPitch;
  static constexpr int kStageFloats = kStageSmall + (kL1 + 1) * kPitchS;
  static_assert(HALF % 4 == 0 && HALF <= N && N - HALF + kL0 - 1 <= HALF, "layer-0 rows must split into two lanes");
  static_assert((kPitch / 4) % 2 == 1, "odd quad pitch keeps the transposed reads conflict-free");
  // staging slot of lane h's row piece, its length in 16-byte lanes (0: lane h holds no row)
  static constexpr int stage_off(int h) {
    return h < kL0 ? h * kPitch
           : h < 32 ? kStageSmall + (h - kL0) * kPitchS
           : h < 32 + kL0 ? (kL0 + h - 32) * kPitch
           : h == kFinLane ? kStageSmall + kL1 * kPitchS : 0;
  }
  // Lower halves are weights throughout; upper half k holds kUp input weights and k cascade
  // weights, the rest of its stored 16-byte pieces is zero padding in HBM and in the registers alike
  // (never loaded, never stored, never anything but zero).
  static constexpr int stage_lanes(int h) {
    return h < kL0 ? kQ : (h >= 32 && h < 32 + kL0) ? (kUp + (h - 32) + 3) / 4 : ((h < 32 || h == kFinLane) ? kQS : 0);
  }
};

__device__ __forceinline__ void wide_ld16(gmx_f4& d, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void wide_ld4(uint32_t& d, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void wide_ld1(uint32_t& d, const void* p) {
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ uint32_t wide_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// One coalesced piece of a row, HBM -> LDS without a VGPR round trip: the lanes of `mask` move 16
// bytes each from sbase + voff to LDS[lds_byte + 16 * lane] (LDS base in M0).  An empty mask issues
// nothing.  Called only where all 64 lanes of the wave are active (uniform control flow, blocks of
// one full wave), so exec goes back to all ones instead of being saved; M0 is saved and restored.
__device__ __forceinline__ void wide_dma16(uint64_t sbase, uint32_t voff, uint32_t lds_byte, uint64_t mask) {
  uint32_t sm0;
  asm volatile(
      "s_mov_b64 exec, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %1 nt\n\ts_mov_b32 m0, %0\n\ts_mov_b64 exec, -1"
      : "=&s"(sm0)
      : "s"(sbase), "v"(voff), "s"(mask), "s"(lds_byte)
      : "memory");
}
// ... and back: the lanes of `mask` store 16 bytes each to sbase + voff.
__device__ __forceinline__ void wide_st16(uint64_t sbase, uint32_t voff, const gmx_f4& v, uint64_t mask) {
  asm volatile(
      "s_mov_b64 exec, %3\n\t"
      "global_store_dwordx4 %1, %2, %0 nt\n\ts_mov_b64 exec, -1"
      :
      : "s"(sbase), "v"(voff), "v"(v), "s"(mask)
      : "memory");
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float el(const gmx_f4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

}  // namespace

template <int N, int HALF, bool HAS_MASK>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
gmx_wide_kernel(const GmxTopoDev* __restrict__ tp, const GmxRunArgs a) {
  using SH = WideShape<N, HALF>;
  constexpr int kHalf = SH::kHalf, kNPad = SH::kNPad, kMW = SH::kMW, kQ = SH::kQ, kUp = SH::kUp,
                kQU = SH::kQU, kPitch = SH::kPitch, kStageSmall = SH::kStageSmall;
  __shared__ __attribute__((aligned(16))) float xin[kNPad + 4];
  __shared__ __attribute__((aligned(16))) float stage[SH::kStageFloats];
  __shared__ uint64_t s_tab[32];
  const int lane = threadIdx.x;
  const int rec = a.rec_base + (int)blockIdx.x;
  const int s = a.stream_base + (int)blockIdx.x;
  const uint64_t T = a.T;
  if (T == 0) return;
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];
  // the staging image starts out zero: the padding pieces of a row are not fetched, so what the
  // owner reads there must be the zeros the write-back path put (or these)
  for (int i = 4 * lane; i < SH::kStageFloats; i += 256) *(gmx_f4*)(stage + i) = gmx_f4{0.f, 0.f, 0.f, 0.f};
  const bool do_learn = (a.mode & GMX_MODE_LEARN) != 0;
  uint8_t* const bank = a.banks + (uint64_t)s * tp->bank_bytes;

  // ---- who this lane is ------------------------------------------------------------------
  const int half = lane >> 5, li = lane & 31;
  const bool is_l0 = li < kL0;                   // both halves of a layer-0 pair
  const bool is_l1 = half == 0 && li >= kL0;     // lanes 24..31
  const bool is_fin = lane == kFinLane;
  const bool act = is_l0 || is_l1 || is_fin;
  const bool owner = act && !(is_l0 && half);    // writes the row's step counter and the mixer's scalars
  const int mxi = is_fin ? kM - 1 : (act ? li : 0);
  const int k1 = li - kL0;                       // output index of a layer-1 lane
  const GmxMixerDev d = tp->mx[mxi];
  const int skip_idx = tp->skip_idx[0];
  const bool all_pow2 = __ballot(act && (d.table_size & (d.table_size - 1u)) != 0) == 0;
  uint64_t* const scal = (uint64_t*)(bank + tp->scal_off) + 3 * mxi;
  uint64_t steps = 0, max_steps = 1, seen_cnt = 0;
  if (act) {
    steps = scal[0];
    max_steps = scal[1];
    seen_cnt = scal[2];
  }
  // Row r's step counter (gmx_internal.h): both shapes this kernel serves keep it inside the row's padding (the
  // reference's own in the row's last 8 bytes, the 256-input banks right behind the weights; d.rs_off says
  // where), one row length apart -- known at compile time here, which matters: this kernel sits at 256
  // VGPRs and a run-time pitch cost it 20 %.
  constexpr bool kFolded = true;
  uint64_t* const rs_tab = (uint64_t*)(bank + d.rs_off);
  uint8_t* const w_tab = bank + d.w_off + ((is_l0 && half) ? kHalf * 4u : 0u);
  const uint32_t row_bytes = d.stride * 4u;

  const uint64_t RS = a.rec_stride;
  const float* const pred_s = a.pred + (uint64_t)rec * RS * kNPad;
  const uint32_t* const mask_s = HAS_MASK ? a.mask + (uint64_t)rec * RS * kMW : nullptr;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS * kM;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  const float* const dec_s = a.decay + (uint64_t)a.decay_idx[blockIdx.x] * T;
  float* const p_s = a.p_out + (uint64_t)rec * RS;
  float* const oa_s = a.out_all ? a.out_all + (uint64_t)rec * RS * kM : nullptr;

  // ---- the resident row ---------------------------------------------------------------------
  gmx_f4 w[kQ];
#pragma unroll
  for (int q = 0; q < kQ; ++q) w[q] = gmx_f4{0.f, 0.f, 0.f, 0.f};
  uint32_t tag = 0xffffffffu;
  uint64_t rs = 0;      // MixerData::steps of the resident row (long-term-memory.h:29)
  bool dirty = false;

  // Rows travel between HBM and the registers through the staging image, so that every global
  // access is a coalesced run of 16-byte lanes (lane-private 16-byte accesses cost the texture
  // path four tag look-ups per 64-byte line, and four partial writes: measured, it was 70% busy).
  float* const my_stage = stage + (is_l0 ? (half * kL0 + li) * kPitch
                                                        : kStageSmall + (is_fin ? kL1 : (li - kL0)) * kPitchS);
  const uint32_t stage_base = wide_lds_addr(stage);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto evict = [&](bool ev) {
    const uint64_t em = __ballot(ev);
    if (em == 0) return;
    const uint64_t dst = (uint64_t)(w_tab + (uint64_t)tag * row_bytes);
    const uint32_t dlo = (uint32_t)dst, dhi = (uint32_t)(dst >> 32);
    if (ev) {
#pragma unroll
      for (int q = 0; q < kQS; ++q) *(gmx_f4*)(my_stage + 4 * q) = w[q];
      if (owner) *(kFolded ? (uint64_t*)((uint8_t*)rs_tab + (uint64_t)tag * row_bytes) : rs_tab + tag) = rs;
    }
    if (ev && is_l0) {
#pragma unroll
      for (int q = kQS; q < kQ; ++q) *(gmx_f4*)(my_stage + 4 * q) = w[q];
    }
#pragma unroll
    for (int h = 0; h <= kFinLane; ++h) {
      if (SH::stage_lanes(h) == 0) continue;
      const uint64_t m = ((em >> h) & 1u) ? ((1ull << SH::stage_lanes(h)) - 1ull) : 0ull;
      const uint64_t sb = ((uint64_t)readlane_u(dhi, h) << 32) | readlane_u(dlo, h);
      const gmx_f4 v = *(const gmx_f4*)(stage + SH::stage_off(h) + 4 * (lane < SH::stage_lanes(h) ? lane : 0));
      wide_st16(sb, lane16, v, m);
    }
  };
  // the rows of the lanes in `nm` (non-empty) from HBM into the staging image
  auto fetch = [&](uint64_t nm, uint64_t src) {
    const uint32_t slo = (uint32_t)src, shi = (uint32_t)(src >> 32);
#pragma unroll
    for (int h = 0; h <= kFinLane; ++h) {
      if (SH::stage_lanes(h) == 0) continue;
      const uint64_t m = ((nm >> h) & 1u) ? ((1ull << SH::stage_lanes(h)) - 1ull) : 0ull;
      const uint64_t sb = ((uint64_t)readlane_u(shi, h) << 32) | readlane_u(slo, h);
      wide_dma16(sb, lane16, stage_base + (uint32_t)SH::stage_off(h) * 4u, m);
    }
  };

  // ---- prefetched record fields (_n: of the bit about to be computed) --------------------
  uint32_t ctx_n = 0, mask_n = ~0u, bit_n = 0, dec_n = 0;
  gmx_f4 x_n = gmx_f4{0.f, 0.f, 0.f, 0.f};
  auto request = [&](uint64_t t) {
    const uint64_t tt = t < T ? t : T - 1;  // past the end: a harmless re-read of the last record
    wide_ld4(ctx_n, ctx_s + tt * (uint64_t)kM + mxi);
    wide_ld16(x_n, pred_s + tt * (uint64_t)kNPad + 4 * (4 * lane < kNPad ? lane : 0));
    if (HAS_MASK) wide_ld4(mask_n, mask_s + tt * (uint64_t)kMW + (lane < kMW ? lane : 0));
    wide_ld1(bit_n, bits_s + tt);
    wide_ld4(dec_n, dec_s + tt);
  };
  // The wait that releases a request sits in the SAME loop iteration as the request (at its end):
  // across the back edge the compiler may copy the destination registers before the data is in.
  auto landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ctx_n), "+v"(x_n), "+v"(mask_n), "+v"(bit_n), "+v"(dec_n));
  };
  request(0);
  landed();

  for (uint64_t t = 0; t < T; ++t) {
    const uint32_t ctx = ctx_n, mword = mask_n, bit = bit_n & 1u;
    const float dec = __uint_as_float(dec_n);
    const gmx_f4 xv = x_n;

    // ---- FindMixerData (mixer.cpp:29-37): replace the resident row if the context moved ----
    const uint32_t row = all_pow2 ? (ctx & (d.table_size - 1u)) : (ctx % d.table_size);
    const bool need = act && row != tag;
    evict(need && dirty);
    const uint64_t nm = __ballot(need);
    if (nm) {
      // the staging image is free again once the write-back has read it (its ds_reads are done:
      // their data went into the stores above)
      fetch(nm, (uint64_t)(w_tab + (uint64_t)row * row_bytes));
      if (need) rs = *(kFolded ? (const uint64_t*)((const uint8_t*)rs_tab + (uint64_t)row * row_bytes) : rs_tab + row);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs) : : "memory");
      if (need) {
#pragma unroll
        for (int q = 0; q < kQS; ++q) w[q] = *(const gmx_f4*)(my_stage + 4 * q);
        tag = row;
        dirty = false;
      }
      if (need && is_l0) {
#pragma unroll
        for (int q = kQS; q < kQ; ++q) w[q] = *(const gmx_f4*)(my_stage + 4 * q);
      }
    }
    request(t + 1);  // a whole bit ahead of its use

    // ---- the blackboard of this bit into LDS ----------------------------------------------
    const bool has_x = 4 * lane < kNPad;
    if (has_x) *(gmx_f4*)(xin + 4 * lane) = xv;
    const float skip = xin[skip_idx];  // raw, possibly stale (mixer.cpp:76-79)
    if (HAS_MASK) {
      // only active_models are visited (mixer.cpp:57-59): silent slots contribute nothing
      const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((has_x ? (lane >> 3) : 0) << 2, (int)mword);
      const uint32_t b = word >> ((4u * (uint32_t)lane) & 31u);
      gmx_f4 v = xv;
      v.x = (b & 1u) ? v.x : 0.f;
      v.y = (b & 2u) ? v.y : 0.f;
      v.z = (b & 4u) ? v.z : 0.f;
      v.w = (b & 8u) ? v.w : 0.f;
      if (has_x) *(gmx_f4*)(xin + 4 * lane) = v;
    }
    const bool seen = act && rs != 0;  // an unseen row is "no row": output 0 (mixer.cpp:52-55)

    // ---- layer 0, inputs 0..143 in lanes 0..23 (mixer.cpp:56-59) -------------------------
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      const gmx_f4 x = *(const gmx_f4*)(xin + 4 * q);
      acc = acc + x.x * w[q].x;
      acc = acc + x.y * w[q].y;
      acc = acc + x.z * w[q].z;
      acc = acc + x.w * w[q].w;
    }
    // ---- ... handed to lanes 32..55, which go on with inputs 144..255 ---------------------
    acc = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 31)) << 2, __float_as_int(acc)));
#pragma unroll
    for (int q = 0; q < kQU; ++q) {
      const gmx_f4 x = *(const gmx_f4*)(xin + kHalf + 4 * q);
      if (4 * q + 0 < kUp) acc = acc + x.x * w[q].x;
      if (4 * q + 1 < kUp) acc = acc + x.y * w[q].y;
      if (4 * q + 2 < kUp) acc = acc + x.z * w[q].z;
      if (4 * q + 3 < kUp) acc = acc + x.w * w[q].w;
    }
    acc = seen ? acc : 0.f;
    // ---- layer-0 cascade: mixer k adds outputs 0..k-1 in order (mixer.cpp:60-64) -----------
    float o0[kL0];
    const bool up0 = is_l0 && half && seen;
#pragma unroll
    for (int i = 0; i < kL0; ++i) {
      o0[i] = readlane_f(acc, 32 + i);
      if (i + 1 < kL0) {
        const float wt = el(w[(kUp + i) / 4], (kUp + i) % 4);
        acc = (up0 && li > i) ? acc + o0[i] * wt : acc;
      }
    }
    // ---- layers 1 and 2: the layer-0 outputs first (mixer.cpp:66-68, 82-84) ----------------
    const bool up1 = (is_l1 || is_fin) && seen;
    float a1 = 0.f;
#pragma unroll
    for (int i = 0; i < kL0; ++i) a1 = a1 + o0[i] * el(w[i / 4], i % 4);
    a1 = up1 ? a1 : 0.f;
    // layer-1 cascade, each mixer's skip input when its turn comes (mixer.cpp:69-80); the final
    // mixer takes every layer-1 output, then the skip input (mixer.cpp:85-97)
    float o1[kL1];
#pragma unroll
    for (int i = 0; i < kL1; ++i) {
      const float wt = el(w[(kL0 + i) / 4], (kL0 + i) % 4);
      a1 = (is_l1 && k1 == i && seen) ? a1 + skip * wt : a1;
      o1[i] = readlane_f(a1, kL0 + i);
      a1 = (((is_l1 && k1 > i) || is_fin) && seen) ? a1 + o1[i] * wt : a1;
    }
    a1 = (is_fin && seen) ? a1 + skip * w[(kL0 + kL1) / 4].x : a1;
    static_assert((kL0 + kL1) % 4 == 0, "the final mixer's skip weight is element 0 of its quad");

    // every lane's own mixer output (both halves of a layer-0 pair hold it)
    const float own0 = __int_as_float(__builtin_amdgcn_ds_bpermute((32 + (lane & 31)) << 2, __float_as_int(acc)));
    const float own = is_l0 ? own0 : a1;
    // Sigmoid::Logistic of it: the final mixer's is Predictor::Predict's result after clamping
    // (predictor.cpp:369-375), all of them feed Mixer::Learn (mixer.cpp:113-122)
    const float pl = gmx_logistic_tab(own, s_tab);
    if (is_fin) p_s[t] = gmx_clamp_prob(pl);
    if (oa_s && owner) oa_s[t * (uint64_t)kM + mxi] = own;

    if (do_learn) {
      // ---- Mixer::Learn (mixer.cpp:108-176) --------------------------------------------------
      const double dd = (double)dec * (1.5 - ((double)rs) / (double)max_steps);  // mixer.cpp:112
      const float decay = (float)dd;
      const float upd = decay * d.lr * (pl - (float)bit);  // mixer.cpp:123
      const uint64_t rs_new = rs + 1;
      const float scl = ((rs_new & 1023u) == 0) ? (1.0f - 3.0e-6f) : 1.0f;  // mixer.cpp:173-175; * 1.0f is exact
      if (act) {
        ++steps;
        if (rs_new > max_steps) max_steps = rs_new;
        if (rs == 0) ++seen_cnt;  // FindOrCreateMixerData (mixer.cpp:44-46)
        rs = rs_new;
        dirty = true;             // row and counter go back to HBM when the row is replaced
      }
      // w -= update * x over the segments Predict walked (mixer.cpp:129-172).  Lower halves and
      // small rows: quads 0..35 of [x | nothing]; upper halves: inputs 144..255, then the
      // cascade inputs (outputs of the layer-0 mixers before this one), then padding.
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        // lower halves: inputs 4q .. 4q+3 (all < HALF <= N); upper halves: local index j faces
        // input HALF + j for j < kUp, then the cascade input j - kUp (an output of a layer-0
        // mixer before this one), then padding
        const gmx_f4 xl = *(const gmx_f4*)(xin + 4 * q);
        gmx_f4 xh = gmx_f4{0.f, 0.f, 0.f, 0.f};
        if (4 * q < kUp) xh = *(const gmx_f4*)(xin + kHalf + 4 * q);
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 4 * q + e;
          const int i = j - kUp;  // cascade input index of an upper half
          c[e] = j < kUp ? el(xh, e) : ((i + 1 < kL0 && li > i) ? o0[(i >= 0 && i < kL0) ? i : 0] : 0.f);
        }
        gmx_f4 x;
        x.x = half ? c[0] : xl.x;
        x.y = half ? c[1] : xl.y;
        x.z = half ? c[2] : xl.z;
        x.w = half ? c[3] : xl.w;
        if (q < kQS) {
          // layer-1 / final rows: layer-0 outputs, own-layer outputs before this mixer (all of
          // them for the final mixer), the skip input, padding
          float sm[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int idx = 4 * q + e;
            float v = 0.f;
            if (idx < kL0) {
              v = o0[idx];
            } else if (idx < kL0 + kL1) {
              const int i = idx - kL0;
              v = (is_fin || k1 > i) ? o1[i] : (k1 == i ? skip : 0.f);
            } else if (idx == kL0 + kL1) {
              v = is_fin ? skip : 0.f;
            }
            sm[e] = v;
          }
          x.x = is_l0 ? x.x : sm[0];
          x.y = is_l0 ? x.y : sm[1];
          x.z = is_l0 ? x.z : sm[2];
          x.w = is_l0 ? x.w : sm[3];
        } else {
          x.x = is_l0 ? x.x : 0.f;
          x.y = is_l0 ? x.y : 0.f;
          x.z = is_l0 ? x.z : 0.f;
          x.w = is_l0 ? x.w : 0.f;
        }
        w[q].x = (w[q].x - upd * x.x) * scl;
        w[q].y = (w[q].y - upd * x.y) * scl;
        w[q].z = (w[q].z - upd * x.z) * scl;
        w[q].w = (w[q].w - upd * x.w) * scl;
      }
    }
    landed();  // requested a whole bit's work ago: no stall
  }
  evict(act && dirty);
  if (owner && do_learn) {
    scal[0] = steps;
    scal[1] = max_steps;
    scal[2] = seen_cnt;
  }
}

// Eligible: 256 inputs (shape 0) or the reference's 90 inputs (shape 1), 24 layer-0 + 8 layer-1 +
// final, one skip input, batched Predict(+Learn) (the host checks that before calling).
extern "C" hipError_t gmx_launch_wide_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args, int n_streams,
                                             int has_mask, int n_inputs, hipStream_t stream) {
  (void)hipGetLastError();
  const dim3 grid(n_streams), block(64);
  if (n_inputs == 256) {
    if (has_mask)
      hipLaunchKernelGGL((gmx_wide_kernel<256, 144, true>), grid, block, 0, stream, tp_dev, *args);
    else
      hipLaunchKernelGGL((gmx_wide_kernel<256, 144, false>), grid, block, 0, stream, tp_dev, *args);
  } else if (n_inputs == 90) {
    if (has_mask)
      hipLaunchKernelGGL((gmx_wide_kernel<90, 64, true>), grid, block, 0, stream, tp_dev, *args);
    else
      hipLaunchKernelGGL((gmx_wide_kernel<90, 64, false>), grid, block, 0, stream, tp_dev, *args);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
