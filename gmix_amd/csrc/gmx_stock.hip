// gmx_stock.hip -- the bank kernel for the reference's own shape: 90 model predictions,
// 24 / 8 / 1 mixers, one skip input (Predictor::AddMixers, predictor.cpp:251-358).
//
// Same arithmetic, same order as gmx_kernels.hip (the general kernel) -- and the same floats,
// which tests/test_gpu_stock.py asserts -- but a different data flow, possible because every
// row of this shape fits in registers (weight_size <= 113 floats):
//   * lane m of the wave IS mixer m (33 lanes used) for everything: it loads its own row from
//     HBM (29 x global_load_dwordx4, one 16-byte chunk per instruction), keeps it in 29 float4
//     registers, runs its chain on them, updates them and stores them back.  No LDS staging of
//     weights, no row/lane transposition, no per-row passes: the update of all 33 rows is one
//     sweep over the register file (all lanes in parallel), the prefetch of all 33 rows is 29
//     load instructions instead of 33 LDS-DMA issues.
//   * the row of bit t+1 is prefetched into a second register set while bit t is computed and
//     selected per lane at the end of the bit; a mixer whose gate context did not change keeps
//     its registers (no load, no select) -- the common case on real data.
//   * values that cross mixers (cascade outputs, layer outputs, skip input) are wave-uniform
//     scalars obtained with v_readlane; the 90 inputs are broadcast-read from LDS.
//   * all vector-memory instructions are issued from inline asm in a fixed order, with one
//     counted s_waitcnt vmcnt(#stores) per bit (see gmx_kernels.hip for why); batches of
//     loads / stores run under a lane mask set inside the asm (layer-1/final rows are 64
//     floats: only chunks 0..15 exist for them).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmx_internal.h"
#include "gmx_math.h"

typedef float gmx_f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kN = 90, kNPad = 92, kL0 = 24, kL1 = 8, kM = 33, kNQ = 29;
constexpr int kQA = 16;  // chunks every mixer's stored row has (64 floats); the rest: layer 0 only

__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void dma16s(uint64_t sbase, uint32_t voff, uint32_t lds_byte_off) {
  uint32_t saved;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(saved)
      : "s"(sbase), "v"(voff), "s"(lds_byte_off)
      : "memory");
}
__device__ __forceinline__ void vld8(uint64_t& d, const uint64_t* p) {
  asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void vld4(uint32_t& d, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=a"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void vld1(uint32_t& d, const uint8_t* p) {
  asm volatile("global_load_ubyte %0, %1, off" : "=a"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void vst8(void* p, uint64_t v) {
  asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void vst4(void* p, float v) {
  asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// Chunks 0..15 of every row whose lane is in `mask` (non-empty); chunks 16..28 likewise.
// The destination is the ACCUMULATOR half of the register file ("a" constraint; gfx90a+ VMEM
// can target AGPRs): a wave has at most 256 architected VGPRs, and with two rows resident the
// compiler would otherwise park freshly "loaded" VGPRs in AGPRs right behind the load asm --
// i.e. copy them before the data has arrived.  Landing the prefetch in AGPRs leaves nothing
// to park; the row moves to VGPRs after the counted wait.
__device__ __forceinline__ void load_rows_a(gmx_f4* r, const void* p, uint64_t mask) {
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %16, exec\n\ts_mov_b64 exec, %18\n\t"
      "global_load_dwordx4 %0, %17, off\n\t"
      "global_load_dwordx4 %1, %17, off offset:16\n\t"
      "global_load_dwordx4 %2, %17, off offset:32\n\t"
      "global_load_dwordx4 %3, %17, off offset:48\n\t"
      "global_load_dwordx4 %4, %17, off offset:64\n\t"
      "global_load_dwordx4 %5, %17, off offset:80\n\t"
      "global_load_dwordx4 %6, %17, off offset:96\n\t"
      "global_load_dwordx4 %7, %17, off offset:112\n\t"
      "global_load_dwordx4 %8, %17, off offset:128\n\t"
      "global_load_dwordx4 %9, %17, off offset:144\n\t"
      "global_load_dwordx4 %10, %17, off offset:160\n\t"
      "global_load_dwordx4 %11, %17, off offset:176\n\t"
      "global_load_dwordx4 %12, %17, off offset:192\n\t"
      "global_load_dwordx4 %13, %17, off offset:208\n\t"
      "global_load_dwordx4 %14, %17, off offset:224\n\t"
      "global_load_dwordx4 %15, %17, off offset:240\n\t"
      "s_mov_b64 exec, %16"
      : "=&a"(r[0]), "=&a"(r[1]), "=&a"(r[2]), "=&a"(r[3]), "=&a"(r[4]), "=&a"(r[5]), "=&a"(r[6]),
        "=&a"(r[7]), "=&a"(r[8]), "=&a"(r[9]), "=&a"(r[10]), "=&a"(r[11]), "=&a"(r[12]),
        "=&a"(r[13]), "=&a"(r[14]), "=&a"(r[15]), "=&s"(saved)
      : "v"(p), "s"(mask)
      : "memory");
}
__device__ __forceinline__ void load_rows_b(gmx_f4* r, const void* p, uint64_t mask) {
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %13, exec\n\ts_mov_b64 exec, %15\n\t"
      "global_load_dwordx4 %0, %14, off offset:256\n\t"
      "global_load_dwordx4 %1, %14, off offset:272\n\t"
      "global_load_dwordx4 %2, %14, off offset:288\n\t"
      "global_load_dwordx4 %3, %14, off offset:304\n\t"
      "global_load_dwordx4 %4, %14, off offset:320\n\t"
      "global_load_dwordx4 %5, %14, off offset:336\n\t"
      "global_load_dwordx4 %6, %14, off offset:352\n\t"
      "global_load_dwordx4 %7, %14, off offset:368\n\t"
      "global_load_dwordx4 %8, %14, off offset:384\n\t"
      "global_load_dwordx4 %9, %14, off offset:400\n\t"
      "global_load_dwordx4 %10, %14, off offset:416\n\t"
      "global_load_dwordx4 %11, %14, off offset:432\n\t"
      "global_load_dwordx4 %12, %14, off offset:448\n\t"
      "s_mov_b64 exec, %13"
      : "=&a"(r[16]), "=&a"(r[17]), "=&a"(r[18]), "=&a"(r[19]), "=&a"(r[20]), "=&a"(r[21]),
        "=&a"(r[22]), "=&a"(r[23]), "=&a"(r[24]), "=&a"(r[25]), "=&a"(r[26]), "=&a"(r[27]),
        "=&a"(r[28]), "=&s"(saved)
      : "v"(p), "s"(mask)
      : "memory");
}
__device__ __forceinline__ void store_rows_a(const gmx_f4* r, void* p, uint64_t mask) {
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\ts_mov_b64 exec, %18\n\t"
      "global_store_dwordx4 %17, %1, off\n\t"
      "global_store_dwordx4 %17, %2, off offset:16\n\t"
      "global_store_dwordx4 %17, %3, off offset:32\n\t"
      "global_store_dwordx4 %17, %4, off offset:48\n\t"
      "global_store_dwordx4 %17, %5, off offset:64\n\t"
      "global_store_dwordx4 %17, %6, off offset:80\n\t"
      "global_store_dwordx4 %17, %7, off offset:96\n\t"
      "global_store_dwordx4 %17, %8, off offset:112\n\t"
      "global_store_dwordx4 %17, %9, off offset:128\n\t"
      "global_store_dwordx4 %17, %10, off offset:144\n\t"
      "global_store_dwordx4 %17, %11, off offset:160\n\t"
      "global_store_dwordx4 %17, %12, off offset:176\n\t"
      "global_store_dwordx4 %17, %13, off offset:192\n\t"
      "global_store_dwordx4 %17, %14, off offset:208\n\t"
      "global_store_dwordx4 %17, %15, off offset:224\n\t"
      "global_store_dwordx4 %17, %16, off offset:240\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved)
      : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]),
        "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11]), "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15]),
        "v"(p), "s"(mask)
      : "memory");
}
__device__ __forceinline__ void store_rows_b(const gmx_f4* r, void* p, uint64_t mask) {
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\ts_mov_b64 exec, %15\n\t"
      "global_store_dwordx4 %14, %1, off offset:256\n\t"
      "global_store_dwordx4 %14, %2, off offset:272\n\t"
      "global_store_dwordx4 %14, %3, off offset:288\n\t"
      "global_store_dwordx4 %14, %4, off offset:304\n\t"
      "global_store_dwordx4 %14, %5, off offset:320\n\t"
      "global_store_dwordx4 %14, %6, off offset:336\n\t"
      "global_store_dwordx4 %14, %7, off offset:352\n\t"
      "global_store_dwordx4 %14, %8, off offset:368\n\t"
      "global_store_dwordx4 %14, %9, off offset:384\n\t"
      "global_store_dwordx4 %14, %10, off offset:400\n\t"
      "global_store_dwordx4 %14, %11, off offset:416\n\t"
      "global_store_dwordx4 %14, %12, off offset:432\n\t"
      "global_store_dwordx4 %14, %13, off offset:448\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved)
      : "v"(r[16]), "v"(r[17]), "v"(r[18]), "v"(r[19]), "v"(r[20]), "v"(r[21]), "v"(r[22]),
        "v"(r[23]), "v"(r[24]), "v"(r[25]), "v"(r[26]), "v"(r[27]), "v"(r[28]), "v"(p), "s"(mask)
      : "memory");
}
// Pins the prefetched row behind the preceding s_waitcnt (asm volatile statements keep order).
__device__ __forceinline__ void pin_rows(gmx_f4* r) {
  asm volatile("" : "+a"(r[0]), "+a"(r[1]), "+a"(r[2]), "+a"(r[3]), "+a"(r[4]), "+a"(r[5]), "+a"(r[6]),
               "+a"(r[7]), "+a"(r[8]), "+a"(r[9]), "+a"(r[10]), "+a"(r[11]), "+a"(r[12]), "+a"(r[13]),
               "+a"(r[14]));
  asm volatile("" : "+a"(r[15]), "+a"(r[16]), "+a"(r[17]), "+a"(r[18]), "+a"(r[19]), "+a"(r[20]),
               "+a"(r[21]), "+a"(r[22]), "+a"(r[23]), "+a"(r[24]), "+a"(r[25]), "+a"(r[26]),
               "+a"(r[27]), "+a"(r[28]));
}

#define GMX_W(r, j) ((r)[(j) >> 2][(j) & 3])

// ---- the arithmetic of one bit, shared by the batched and the session kernel ---------------

// Only active_models are visited (mixer.cpp:57-59): silent slots contribute nothing.  `mword`
// is this lane's word of the active mask (lanes 0..2 hold the three words).
__device__ __forceinline__ void stock_mask_inputs(float* xin, uint32_t mword, int lane) {
  const uint32_t c = (uint32_t)lane * 4u;
  const uint32_t wd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c >> 5) << 2, (int)mword);
  if (c < (uint32_t)kNPad) {
    float4 v = *(float4*)(xin + c);
    const uint32_t b = wd >> (c & 31u);
    v.x = (b & 1u) ? v.x : 0.f;
    v.y = (b & 2u) ? v.y : 0.f;
    v.z = (b & 4u) ? v.z : 0.f;
    v.w = (b & 8u) ? v.w : 0.f;
    *(float4*)(xin + c) = v;
  }
}

// 33 x Mixer::Predict (mixer.cpp:51-106): returns this lane's mixer output; out0v = the
// layer-0 outputs (lanes 0..23), which the other lanes read with v_readlane.
__device__ __forceinline__ float stock_forward(const gmx_f4 (&w)[kNQ], const float* xin, float skip0,
                                               bool seen, int lane, float& out0v) {
  const bool is_l0 = lane < kL0, is_up = lane >= kL0 && lane < kM;
  // ---- layer 0, inputs 0..89 (mixer.cpp:56-59): every lane runs the chain on its own
  //      registers; only the layer-0 lanes keep the result --------------------------------
  float a0 = 0.f;
#pragma unroll
  for (int q = 0; q < (kN + 3) / 4; ++q) {
    const float4 xq = *(const float4*)(xin + 4 * q);  // same address in every lane: broadcast
    a0 = a0 + xq.x * w[q].x;
    a0 = a0 + xq.y * w[q].y;
    if (4 * q + 2 < kN) a0 = a0 + xq.z * w[q].z;
    if (4 * q + 3 < kN) a0 = a0 + xq.w * w[q].w;
    // keep the broadcast reads near their use: hoisting all 23 of them costs 92 VGPRs next to
    // the two resident rows and tips the kernel into scratch spills
    if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  float acc = (is_l0 && seen) ? a0 : 0.f;
  // ---- layer-0 cascade (mixer.cpp:60-64) -------------------------------------------------
#pragma unroll
  for (int i = 0; i + 1 < kL0; ++i) {
    const float o = rl_f(acc, i);
    const float tv = acc + o * GMX_W(w, kN + i);
    acc = (is_l0 && lane > i && seen) ? tv : acc;
  }
  out0v = acc;
  // ---- layer 1 and final: the 24 layer-0 outputs (mixer.cpp:66-68, 82-84) -----------------
  float a1 = 0.f;
#pragma unroll
  for (int i = 0; i < kL0; ++i) a1 = a1 + rl_f(out0v, i) * GMX_W(w, i);
  acc = is_up ? (seen ? a1 : 0.f) : acc;
  // ---- layer-1 cascade, each mixer's skip input first (mixer.cpp:69-80) -------------------
#pragma unroll
  for (int i = 0; i < kL1; ++i) {
    const float ts = acc + skip0 * GMX_W(w, kL0 + i);  // lane 24+i: skip weight at 24+i
    acc = (lane == kL0 + i && seen) ? ts : acc;
    const float o = rl_f(acc, kL0 + i);
    const float tv = acc + o * GMX_W(w, kL0 + i);  // later lanes: cascade weight at 24+i
    acc = (lane > kL0 + i && lane < kM && seen) ? tv : acc;
  }
  // The loop above also ran the final mixer's 8 layer-1 terms (lane 32 > 24+i); its skip
  // input closes the chain (mixer.cpp:85-97).
  const float ts = acc + skip0 * GMX_W(w, kL0 + kL1);
  acc = (lane == kM - 1 && seen) ? ts : acc;
  return acc;
}

// The weight update of 33 x Mixer::Learn (mixer.cpp:129-175) as one sweep over the register
// file.  Weight j of lane m multiplies
//   layer 0 (m < 24): input j (j < 90), then output j-90 of the earlier mixers (j-90 < m)
//   layer 1 / final : layer-0 output j (j < 24), layer-1 output j-24 (j < m), skip (j == m)
// and nothing beyond its weight_size (the stored padding stays zero).  `scl` is the
// every-1024th-visit shrink or exactly 1.0f.
__device__ __forceinline__ void stock_update(gmx_f4 (&w)[kNQ], const float* xin, float skip0, float upd,
                                             float scl, float acc, float out0v, int lane) {
  const bool is_l0 = lane < kL0;
  const float upd0 = is_l0 ? upd : 0.f;  // elements only layer 0 has: others see "- 0 * x"
#pragma unroll
  for (int q = 0; q < kNQ; ++q) {
    if ((q & 3) == 0) __builtin_amdgcn_sched_barrier(0);  // see stock_forward
    if (4 * q >= kL0 + kL1 + 1 + 3 && 4 * q + 3 < kN) {
      // inputs 36..87: plain x for layer 0, nothing for layers 1/2 -- whole float4s, so the
      // compiler can use packed multiplies / adds
      const float4 xr = *(const float4*)(xin + 4 * q);
      const gmx_f4 xq = {xr.x, xr.y, xr.z, xr.w};
      w[q] = (w[q] - upd0 * xq) * scl;
      continue;
    }
    float4 xq = (float4){0.f, 0.f, 0.f, 0.f};
    if (4 * q < kNPad) xq = *(const float4*)(xin + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * q + e;
      if (j >= kN + kL0 - 1) continue;  // padding: zero, stays zero
      const float xj = e == 0 ? xq.x : (e == 1 ? xq.y : (e == 2 ? xq.z : xq.w));
      float wj = e == 0 ? w[q].x : (e == 1 ? w[q].y : (e == 2 ? w[q].z : w[q].w));
      if (j >= kL0 + kL1 + 1 && j < kN) {
        wj = (wj - upd0 * xj) * scl;
      } else {
        float v;
        if (j >= kN) {
          v = (is_l0 && lane > j - kN) ? rl_f(out0v, j - kN) : 0.f;
        } else {
          float v1;  // layer-1 / final lanes
          if (j < kL0) v1 = rl_f(out0v, j);
          else if (j < kL0 + kL1) v1 = (lane > j) ? rl_f(acc, j) : (lane == j ? skip0 : 0.f);
          else v1 = (lane == j) ? skip0 : 0.f;
          v = is_l0 ? xj : v1;
        }
        wj = (wj - upd * v) * scl;
      }
      if (e == 0) w[q].x = wj; else if (e == 1) w[q].y = wj; else if (e == 2) w[q].z = wj; else w[q].w = wj;
    }
  }
}

// Per-lane state of one mixer between bits: Mixer's counters (mixer.h:33-38), the resident
// row's tag / MixerData::steps, and whether registers are ahead of HBM (write-back).
struct StockLane {
  uint64_t steps, max_steps, seen_cnt;
  uint64_t rs_w;
  uint32_t tag;
  bool dirty;
};

// Scalar part of Mixer::Learn (mixer.cpp:108-128): returns the update factor, advances the
// counters.  dec = float(0.9 / pow(1e-7*steps_+0.8, 0.8)) comes from the host (mixer.cpp:111).
__device__ __forceinline__ float stock_learn_scalars(StockLane& st, float dec, float lr, float pl,
                                                     uint32_t bit, bool is_mx, float& scl) {
  const double dd = (double)dec * (1.5 - ((double)st.rs_w) / (double)st.max_steps);  // mixer.cpp:112
  const float decay = (float)dd;
  const float upd = decay * lr * (pl - (float)bit);  // mixer.cpp:123
  const uint64_t rs_new = st.rs_w + 1;
  scl = ((rs_new & 1023u) == 0) ? (1.0f - 3.0e-6f) : 1.0f;  // mixer.cpp:173-175
  if (is_mx) {
    ++st.steps;
    if (rs_new > st.max_steps) st.max_steps = rs_new;
    if (st.rs_w == 0) ++st.seen_cnt;  // FindOrCreateMixerData (mixer.cpp:44-46)
    st.rs_w = rs_new;
    st.dirty = true;  // row and counter go back to HBM when the row is replaced (write-back)
  }
  return upd;
}

// Write-back of the lanes in `ev` (their resident row and its step counter).
__device__ __forceinline__ void stock_evict(const gmx_f4 (&w)[kNQ], const StockLane& st, bool ev,
                                            uint8_t* w_tab, uint64_t* rs_tab, uint32_t row_bytes) {
  const uint64_t mask_l0 = (1ull << kL0) - 1;
  const uint64_t em = __ballot(ev);
  if (em) {
    uint8_t* dst = w_tab + (uint64_t)st.tag * row_bytes;
    store_rows_a(w, dst, em);
    if (em & mask_l0) store_rows_b(w, dst, em & mask_l0);
    if (ev) vst8(rs_tab + st.tag, st.rs_w);
  }
}

// Take over the prefetched row (per lane).
__device__ __forceinline__ void stock_adopt(gmx_f4 (&w)[kNQ], const gmx_f4 (&wn)[kNQ], bool need) {
  // (one code path on purpose: a second, select-free copy for "every mixer changes rows" pushed
  // the kernel over the register budget into scratch spills)
#pragma unroll
  for (int q = 0; q < kNQ; ++q) {
    w[q].x = need ? wn[q].x : w[q].x;
    w[q].y = need ? wn[q].y : w[q].y;
    w[q].z = need ? wn[q].z : w[q].z;
    w[q].w = need ? wn[q].w : w[q].w;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Batched kernel: T bits from record arrays, one wave per stream (see file header).
// ---------------------------------------------------------------------------------------
template <bool HAS_MASK>
__global__ void __launch_bounds__(64)
gmx_stock_kernel(const GmxTopoDev* __restrict__ tp, const GmxRunArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int rec = a.rec_base + (int)blockIdx.x;
  const int s = a.stream_base + (int)blockIdx.x;
  const uint64_t T = a.T;
  if (T == 0) return;
  const int MW = tp->mask_words;  // 3
  uint8_t* const bank = a.banks + (uint64_t)s * tp->bank_bytes;
  const bool do_predict = (a.mode & GMX_MODE_PREDICT) != 0;
  const bool do_learn = (a.mode & GMX_MODE_LEARN) != 0;
  const bool do_latch = (a.mode & GMX_MODE_LATCH) != 0;

  float* const in0 = lds + tp->lds_in0;  // [2][in0_sz]: the inputs of the bit, double-buffered
  const uint32_t in0_sz = tp->in0_sz;
  uint64_t* const s_tab = (uint64_t*)(lds + tp->lds_misc);  // expf's table (lgkmcnt, not vmcnt)
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];

  const bool is_mx = lane < kM;
  const GmxMixerDev d = tp->mx[is_mx ? lane : 0];
  const int skip_idx = tp->skip_idx[0];
  const uint64_t mask_l0 = (1ull << kL0) - 1;

  const uint64_t RS = a.rec_stride;
  const float* const pred_s = a.pred + (uint64_t)rec * RS * kNPad;
  const uint32_t* const mask_s = HAS_MASK ? a.mask + (uint64_t)rec * RS * MW : nullptr;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS * kM;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  const float* const dec_s = a.decay + (uint64_t)a.decay_idx[blockIdx.x] * T;
  float* const p_s = a.p_out + (uint64_t)rec * RS;
  float* const oa_s = a.out_all ? a.out_all + (uint64_t)rec * RS * kM : nullptr;
  float* const latch_s = a.latch_out + (uint64_t)s * kM;

  uint64_t* const scal = (uint64_t*)(bank + tp->scal_off) + 3 * lane;
  StockLane st = {0, 1, 0, 0, 0xffffffffu, false};
  if (is_mx) {
    st.steps = scal[0];
    st.max_steps = scal[1];
    st.seen_cnt = scal[2];
  }
  uint64_t* const rs_tab = (uint64_t*)(bank + d.rs_off);
  uint8_t* const w_tab = bank + d.w_off;
  const uint32_t row_bytes = d.stride * 4u;  // 512 (layer 0) or 256

  // stores issued per bit after the prefetch loads (rows are written back at replacement
  // time, inside the commit, i.e. before the NEXT bit's prefetch)
  const int n_st = 1 + (oa_s ? 1 : 0) + (do_latch ? 1 : 0);

  gmx_f4 w[kNQ], wn[kNQ];  // this lane's row in use / the prefetched one
#pragma unroll
  for (int q = 0; q < kNQ; ++q) {
    w[q] = (gmx_f4){0.f, 0.f, 0.f, 0.f};
    wn[q] = (gmx_f4){0.f, 0.f, 0.f, 0.f};
  }
  uint64_t nm_cur = 0;  // lanes whose mixer changes rows at this bit
  uint32_t xb = 0;

  uint32_t ctx_nn = is_mx ? ctx_s[lane] : 0;
  uint32_t row_n = 0, mask_n = ~0u, mask_c = ~0u, bit_n = 0, bit_c = 0;
  uint32_t dec_n = 0, dec_c = 0;
  uint64_t rs_ld = 0;
  bool need = false;
  const uint32_t lds_base = lds_addr(lds);
  const uint32_t lane16 = (uint32_t)lane * 16u;

  for (uint64_t t = 0; t <= T; ++t) {
    // ================= prefetch bit t ==================================================
    if (t < T) {
      row_n = ctx_nn % d.table_size;  // FindMixerData (mixer.cpp:32)
      need = is_mx && row_n != st.tag;
      const uint8_t* src = w_tab + (uint64_t)row_n * row_bytes;
      const uint64_t nm = __ballot(need);
      nm_cur = nm;
      if (nm) load_rows_a(wn, src, nm);
      if (nm & mask_l0) load_rows_b(wn, src, nm & mask_l0);
      if (need) vld8(rs_ld, rs_tab + row_n);
      {
        const uint64_t g = (uint64_t)(pred_s + t * (uint64_t)kNPad);
        const uint32_t dstx = lds_base + (tp->lds_in0 + (xb ^ 1u) * in0_sz) * 4u;
        if (lane16 < kNPad * 4u) dma16s(g, lane16, dstx);
      }
      if (HAS_MASK) {
        mask_n = 0u;
        if (lane < MW) vld4(mask_n, mask_s + t * (uint64_t)MW + lane);
      }
      vld1(bit_n, bits_s + t);
      vld4(dec_n, dec_s + t);
      ctx_nn = 0u;
      if (is_mx && t + 1 < T) vld4(ctx_nn, ctx_s + (t + 1) * (uint64_t)kM + lane);
    }
    if (t > 0) {
      // ================= compute bit t-1 ================================================
      const uint64_t tc = t - 1;
      float* const xin = in0 + xb * in0_sz;
      const bool seen = is_mx && st.rs_w != 0;  // unseen row = no row: p = 0 (mixer.cpp:52-55)
      const float skip0 = xin[skip_idx];         // raw, possibly stale (mixer.cpp:76-79)
      if (HAS_MASK) stock_mask_inputs(xin, mask_c, lane);
      float acc, out0v;
      if (do_predict) {
        acc = stock_forward(w, xin, skip0, seen, lane, out0v);
      } else {
        // Learn-only call of the per-bit surface: outputs were latched by the forward call.
        acc = is_mx ? latch_s[lane] : 0.f;
        out0v = acc;
      }
      const float pl = gmx_logistic_tab(acc, s_tab);  // Sigmoid::Logistic of every mixer's output
      if (lane == kM - 1) vst4(p_s + tc, gmx_clamp_prob(pl));  // predictor.cpp:369-375
      if (oa_s && is_mx) vst4(oa_s + tc * (uint64_t)kM + lane, acc);
      if (do_latch && is_mx) vst4(latch_s + lane, acc);
      if (do_learn) {
        float scl;
        const float upd = stock_learn_scalars(st, __uint_as_float(dec_c), d.lr, pl, bit_c, is_mx, scl);
        stock_update(w, xin, skip0, upd, scl, acc, out0v, lane);
      }
    }
    // ================= commit the prefetch issued above ==================================
    if (t < T) {
      if (t == 0) vmcnt<0>(); else {
        // everything older than this bit's stores has landed
        switch (n_st) {
          case 1: vmcnt<1>(); break;
          case 2: vmcnt<2>(); break;
          case 3: vmcnt<3>(); break;
          default: vmcnt<0>(); break;
        }
      }
      pin_rows(wn);
      asm volatile("" : "+a"(rs_ld), "+a"(mask_n), "+a"(bit_n), "+a"(dec_n), "+a"(ctx_nn));
      if (nm_cur) {  // wave-uniform: some mixer changes rows
        // write-back: a row that was updated goes to HBM (with its step counter) only now that
        // it is being replaced
        stock_evict(w, st, need && st.dirty, w_tab, rs_tab, row_bytes);
        stock_adopt(w, wn, need);
        if (need) {
          st.tag = row_n;
          st.rs_w = rs_ld;
          st.dirty = false;
        }
      }
      xb ^= 1u;
      mask_c = mask_n;
      bit_c = bit_n;
      dec_c = dec_n;
    }
  }
  vmcnt<0>();
  stock_evict(w, st, is_mx && st.dirty, w_tab, rs_tab, row_bytes);  // flush what is only in registers
  if (is_mx && do_learn) {
    scal[0] = st.steps;
    scal[1] = st.max_steps;
    scal[2] = st.seen_cnt;
  }
}

extern "C" hipError_t gmx_launch_stock_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args,
                                              int n_streams, unsigned lds_bytes, int has_mask,
                                              hipStream_t stream) {
  (void)hipGetLastError();
  if (has_mask)
    hipLaunchKernelGGL(gmx_stock_kernel<true>, dim3(n_streams), dim3(64), lds_bytes, stream, tp_dev, *args);
  else
    hipLaunchKernelGGL(gmx_stock_kernel<false>, dim3(n_streams), dim3(64), lds_bytes, stream, tp_dev, *args);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Session kernel: the per-bit surface (Predict / Perceive / Learn one bit at a time, the only
// way a decoder can drive the mixers -- coder/decoder.cpp:19-39 learns the bit from Predict's
// result) without a kernel launch per call.  One persistent wave per stream takes commands
// from a mailbox in host-coherent pinned memory; between the forward and the learn command
// of a bit everything stays in registers / LDS.  The wave leaves by itself -- writing its rows
// back -- on a stop command or after `idle_ticks` of s_memrealtime (100 MHz) without work,
// so a vanished host cannot leave it spinning; the host relaunches it on demand.  A wave
// started between the forward and the learn of a bit gets `replay` and recomputes the
// forward from the payload still in the mailbox before it takes commands.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
gmx_stock_session_kernel(const GmxTopoDev* __restrict__ tp, uint8_t* banks, int stream,
                         GmxMailbox* mb, unsigned long long idle_ticks, int replay_forward) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  uint8_t* const bank = banks + (uint64_t)stream * tp->bank_bytes;
  float* const xin = lds + tp->lds_in0;
  uint64_t* const s_tab = (uint64_t*)(lds + tp->lds_misc);
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];
  const bool is_mx = lane < kM;
  const GmxMixerDev d = tp->mx[is_mx ? lane : 0];
  const int skip_idx = tp->skip_idx[0];
  const uint64_t mask_l0 = (1ull << kL0) - 1;
  uint64_t* const scal = (uint64_t*)(bank + tp->scal_off) + 3 * lane;
  StockLane st = {0, 1, 0, 0, 0xffffffffu, false};
  if (is_mx) {
    st.steps = scal[0];
    st.max_steps = scal[1];
    st.seen_cnt = scal[2];
  }
  uint64_t* const rs_tab = (uint64_t*)(bank + d.rs_off);
  uint8_t* const w_tab = bank + d.w_off;
  const uint32_t row_bytes = d.stride * 4u;

  gmx_f4 w[kNQ], wn[kNQ];
#pragma unroll
  for (int q = 0; q < kNQ; ++q) {
    w[q] = (gmx_f4){0.f, 0.f, 0.f, 0.f};
    wn[q] = (gmx_f4){0.f, 0.f, 0.f, 0.f};
  }
  volatile GmxMailbox* const vm = mb;
  // A command published before this wave started (or while its predecessor was leaving) is
  // still pending: resume from the last COMPLETED command word.
  uint32_t seen = __hip_atomic_load(&mb->done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  float acc = 0.f, out0v = 0.f, pl = 0.5f, skip0 = 0.f, dec = 0.f;
  bool learned = true;  // nothing to learn from yet
  bool replay = replay_forward != 0;
  uint32_t exit_state = GMX_MB_EXIT_IDLE;
  for (;;) {
    uint32_t word = seen;
    uint32_t cmd = GMX_MB_FORWARD;
    if (!replay) {
      // ---- wait for the next command (bounded) -------------------------------------------
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      bool idle = false;
      for (;;) {
        word = __hip_atomic_load(&mb->cmd_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (word != seen) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) { idle = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (idle) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      cmd = word & GMX_MB_CMD_MASK;
    }
    if (cmd == GMX_MB_STOP) { exit_state = GMX_MB_EXIT_STOP; seen = word; break; }
    if (cmd == GMX_MB_FORWARD) {
      // the blackboard of this bit, straight from the host's pinned memory
      const uint32_t ctx = is_mx ? vm->ctx[lane] : 0u;
      const uint32_t mword = lane < 3 ? vm->mask[lane] : 0u;
      dec = __uint_as_float(vm->dec_bits);
      if (lane * 4 < kNPad) {
        float4 v;
        v.x = vm->pred[4 * lane + 0];
        v.y = vm->pred[4 * lane + 1];
        v.z = vm->pred[4 * lane + 2];
        v.w = vm->pred[4 * lane + 3];
        *(float4*)(xin + 4 * lane) = v;
      }
      const uint32_t row_n = ctx % d.table_size;  // FindMixerData (mixer.cpp:32)
      const bool need = is_mx && row_n != st.tag;
      const uint64_t nm = __ballot(need);
      if (nm) {
        stock_evict(w, st, need && st.dirty, w_tab, rs_tab, row_bytes);
        const uint8_t* src = w_tab + (uint64_t)row_n * row_bytes;
        uint64_t rs_ld = 0;
        load_rows_a(wn, src, nm);
        if (nm & mask_l0) load_rows_b(wn, src, nm & mask_l0);
        if (need) vld8(rs_ld, rs_tab + row_n);
        vmcnt<0>();
        pin_rows(wn);
        asm volatile("" : "+a"(rs_ld));
        stock_adopt(w, wn, need);
        if (need) {
          st.tag = row_n;
          st.rs_w = rs_ld;
          st.dirty = false;
        }
      }
      skip0 = xin[skip_idx];  // raw, possibly stale (mixer.cpp:76-79)
      stock_mask_inputs(xin, mword, lane);
      const bool seen_row = is_mx && st.rs_w != 0;
      acc = stock_forward(w, xin, skip0, seen_row, lane, out0v);
      pl = gmx_logistic_tab(acc, s_tab);
      if (is_mx) vm->outs[lane] = acc;
      if (lane == kM - 1) vm->p = gmx_clamp_prob(pl);  // predictor.cpp:369-375
      learned = false;
    } else if ((cmd == GMX_MB_LEARN0 || cmd == GMX_MB_LEARN1) && !learned) {
      float scl;
      const float upd = stock_learn_scalars(st, dec, d.lr, pl, cmd == GMX_MB_LEARN1 ? 1u : 0u, is_mx, scl);
      stock_update(w, xin, skip0, upd, scl, acc, out0v, lane);
      learned = true;  // a second Learn for the same Predict is a protocol error: ignored
    }
    if (replay) {
      replay = false;
      continue;
    }
    seen = word;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(&mb->done_seq, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // leave: everything that is only in registers goes back to HBM
  stock_evict(w, st, is_mx && st.dirty, w_tab, rs_tab, row_bytes);
  if (is_mx) {
    scal[0] = st.steps;
    scal[1] = st.max_steps;
    scal[2] = st.seen_cnt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  if (lane == 0) {
    if (exit_state == GMX_MB_EXIT_STOP)
      __hip_atomic_store(&mb->done_seq, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->state, exit_state, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

extern "C" hipError_t gmx_launch_stock_session(const GmxTopoDev* tp_dev, uint8_t* banks, int stream_idx,
                                               GmxMailbox* mb, unsigned long long idle_ticks,
                                               int replay_forward, unsigned lds_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_stock_session_kernel, dim3(1), dim3(64), lds_bytes, stream, tp_dev, banks,
                     stream_idx, mb, idle_ticks, replay_forward);
  return hipGetLastError();
}
