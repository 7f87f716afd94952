// gmx_single.hip -- the throughput kernel for banks that are ONE layer-0 mixer
// (BASELINE.json configs[1]: 256 inputs, one 2^16-row gate table per stream).
//
// Per coded bit and stream the reference does (mixer.cpp:51-176): pick row ctx % 65536, one
// strict left-to-right fp32 dot product of 256 terms, one logistic, one scaled update of the
// same 256 weights.  Bytes: 1 KiB row in, 1 KiB row out, 1 KiB of inputs -- 0.5 flop/byte, so
// the kernel is HBM-bound by construction and everything below is about keeping enough
// independent 1-KiB row requests in flight while honouring the sequential sum.
//
// Mapping (CDNA4, wave64), parameterised by LPS = lanes per stream (16, 32 or 64):
//   * a wave owns 64/LPS streams; each stream lives in LPS consecutive lanes.  Lane l of a
//     stream holds elements {4*LPS*k + 4l .. +3 : k < K} of the weight row and of the inputs
//     (K float4 each), so every global_load_dwordx4 / store moves 16*LPS contiguous bytes per
//     stream.
//   * the dot product is a chain that walks the LPS lanes of the stream K times:
//         acc = rotate_by_one_lane(acc) + p.x;  acc += p.y;  acc += p.z;  acc += p.w
//     executed LPS*K times by all lanes; lane i's accumulator is the true prefix sum at step
//     i (mod LPS), the DPP rotate (row_ror:1 / wave_ror:1) hands it to lane i+1, and after the
//     last step the last lane holds the sum in exactly the reference's order
//     (mixer.cpp:57-59).  One VALU instruction per element, no LDS, no cross-stream traffic.
//   * rows, inputs and row-step counters of the next NSLOT-1 bits are in flight in a static
//     ring of register slots (no LDS staging: the data is consumed by the lanes that loaded
//     it).  hipcc's automatic s_waitcnt placement drains such a ring at the loop header
//     (its loop-carried analysis is conservative), so every vector-memory instruction of the
//     loop is issued from inline asm in a fixed order and ONE counted s_waitcnt vmcnt(N) per
//     bit, with N = the number of younger memory instructions, releases exactly the slot that
//     is about to be used.  Slot registers are named as operands of that wait, so no use can
//     move above it; a premature read would return the previous record's bytes and is what
//     the parity tests on 2^16-row tables (every row a fresh HBM miss) would catch.
//   * the chain of bit t+1 is issued in the same basic block as the logistic / update /
//     stores of bit t, so the dependent-add latency of the one is filled by the other.  It is
//     speculative: if bit t+1 gates the row bit t has just updated (or any later slot holds
//     it), the updated row is handed over in registers and that chain is redone -- rare for
//     32-bit contexts, so it sits behind one wave-uniform branch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmx_internal.h"
#include "gmx_math.h"

typedef float gmx_f4 __attribute__((ext_vector_type(4)));

// ---- vector-memory instructions issued by hand (see file header) ---------------------------
// Rows and inputs carry the non-temporal hint: a row is read once and written once per visit and
// the tables are far larger than any cache, so keeping its lines in L2 only delays the write-back
// and evicts nothing useful later.  Measured: 1.57e9 -> 1.79e9 bits/s (0.60 -> 0.69 of 8 TB/s).
template <int OFF>
__device__ __forceinline__ void gmx_ld16(gmx_f4& d, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gmx_ld8(uint64_t& d, const uint64_t* p) {
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gmx_ld4(uint32_t& d, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gmx_ld1(uint32_t& d, const uint8_t* p) {
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gmx_st16(float* p, const gmx_f4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off offset:%2 nt" : : "v"(p), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gmx_st8(uint64_t* p, uint64_t v) {
  asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void gmx_st4(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

template <int K>
struct GmxSlot {
  gmx_f4 w[K];
  gmx_f4 x[K];
  uint32_t mk[K];  // HAS_MASK: the active-mask word that covers this lane's quad of chunk k
  uint64_t rs;   // MixerData::steps of the row (long-term-memory.h:29)
  uint32_t row;
  uint32_t bit;
  uint32_t dec;  // float bits
};

// s_waitcnt vmcnt(N) that also pins the registers it releases: naming them as read-write
// operands keeps every use behind the wait and every earlier def in front of it.
template <int N>
__device__ __forceinline__ void gmx_wait_slot(GmxSlot<1>& a) {
  asm volatile("s_waitcnt vmcnt(%c5)"
               : "+v"(a.w[0]), "+v"(a.x[0]), "+v"(a.rs), "+v"(a.bit), "+v"(a.dec)
               : "n"(N));
}
template <int N>
__device__ __forceinline__ void gmx_wait_slot(GmxSlot<2>& a) {
  asm volatile("s_waitcnt vmcnt(%c7)"
               : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.x[0]), "+v"(a.x[1]), "+v"(a.rs), "+v"(a.bit),
                 "+v"(a.dec)
               : "n"(N));
}
template <int N>
__device__ __forceinline__ void gmx_wait_slot(GmxSlot<4>& a) {
  asm volatile("s_waitcnt vmcnt(%c11)"
               : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]), "+v"(a.w[3]), "+v"(a.x[0]), "+v"(a.x[1]),
                 "+v"(a.x[2]), "+v"(a.x[3]), "+v"(a.rs), "+v"(a.bit), "+v"(a.dec)
               : "n"(N));
}
// The mask words of a slot ride on the same wait: pinned right behind it.
template <int K>
__device__ __forceinline__ void gmx_pin_mask(GmxSlot<K>& a) {
  if (K == 1) asm volatile("" : "+v"(a.mk[0]));
  if (K == 2) asm volatile("" : "+v"(a.mk[0]), "+v"(a.mk[1 % K]));
  if (K == 4) asm volatile("" : "+v"(a.mk[0]), "+v"(a.mk[1 % K]), "+v"(a.mk[2 % K]), "+v"(a.mk[3 % K]));
}
// Pins older, already released values behind the preceding wait (asm volatile statements keep
// their order).
__device__ __forceinline__ void gmx_pin(uint64_t& rs, uint32_t& bit, uint32_t& dec, uint32_t& cq) {
  asm volatile("" : "+v"(rs), "+v"(bit), "+v"(dec), "+v"(cq));
}

// Hand the accumulator of lane i-1 to lane i inside a stream of LPS lanes, at step `i` of a
// pass over the stream (`first_pass`: nothing precedes lane 0, its input is the initial 0).
template <int LPS>
__device__ __forceinline__ float gmx_handoff(float acc, int i, bool first_pass, int lane) {
  const int a = __float_as_int(acc);
  if (LPS == 64)  // wave_ror:1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, a, 0x13C, 0xf, 0xf, false));
  if (LPS == 16)  // row_ror:1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, a, 0x121, 0xf, 0xf, false));
  // LPS == 32: two DPP rows per stream
  if (i == 16)    // lane 15 -> the next row (rows 1 and 3 written, the others keep acc)
    return __int_as_float(__builtin_amdgcn_update_dpp(a, a, 0x142, 0xa, 0xf, false));
  if (i == 0 && !first_pass)  // lane 31 -> lane 0 of the same stream
    return __shfl(acc, (lane & 32) | 31);
  return __int_as_float(__builtin_amdgcn_update_dpp(0, a, 0x121, 0xf, 0xf, false));
}

// One piece (chunk q: 4*LPS elements) of the strict left-to-right dot product
// (mixer.cpp:56-59); after the last piece the last lane of every stream holds the sum.
template <int K, int LPS>
__device__ __forceinline__ float gmx_chain_piece(const GmxSlot<K>& c, int q, float acc, int lane) {
  const float px = c.x[q].x * c.w[q].x, py = c.x[q].y * c.w[q].y;
  const float pz = c.x[q].z * c.w[q].z, pw = c.x[q].w * c.w[q].w;
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    acc = gmx_handoff<LPS>(acc, i, q == 0, lane) + px;
    acc = acc + py;
    acc = acc + pz;
    acc = acc + pw;
  }
  return acc;
}
template <int K, int LPS>
__device__ __forceinline__ float gmx_chain(const GmxSlot<K>& c, int lane) {
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < K; ++q) acc = gmx_chain_piece<K, LPS>(c, q, acc, lane);
  return acc;
}

// LPS   : lanes per stream (a wave carries 64/LPS streams)
// K     : float4 chunks per lane (n_inputs <= 4*LPS*K)
// NSLOT : register slots in the prefetch ring (bits in flight, the current one included)
// FULL  : n_inputs == n_pad == stride == 4*LPS*K -- no ragged edges (the benchmark shape)
template <int LPS, int K, int NSLOT, bool FULL, bool WANT_OUT, bool LEARN, bool HAS_MASK>
__global__ void __launch_bounds__(64)
gmx_single_kernel(const GmxTopoDev* __restrict__ tp, const GmxRunArgs a) {
  constexpr int G = 64 / LPS;
  constexpr int CH = 4 * LPS;  // elements per chunk
  // vector-memory instructions per bit, in issue order: p (+out) stores, K row stores + the
  // row-step store (LEARN), 2K+3 refill loads (+K mask words), 1 context load.
  constexpr int VM_PER_BIT = 1 + (WANT_OUT ? 1 : 0) + (LEARN ? K + 1 : 0) + 2 * K + 3 + (HAS_MASK ? K : 0) + 1;
  // At the top of bit t the youngest data needed are slot t+1's, loaded during bit
  // t+1-NSLOT and followed by that bit's context load and all of bits t+2-NSLOT .. t-1.
  constexpr int VM_WAIT = 1 + (NSLOT - 2) * VM_PER_BIT;
  static_assert(VM_WAIT <= 63, "vmcnt is a 6-bit counter");
  static_assert(CH * 4 * (K - 1) < 4096, "immediate offset of global_load is 12 bits");
  const int lane = threadIdx.x;
  const int g = lane / LPS, l = lane % LPS;
  const int li = (int)blockIdx.x * G + g;  // launch-local stream index of these LPS lanes
  // expf's 2^(i/32) table in LDS: its lookups use lgkmcnt, not the vmcnt queue the prefetched
  // rows are in.  (One wave per block: no barrier needed, LDS ops of a wave are in order.)
  __shared__ uint64_t s_tab[32];
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];
  if (li >= a.n_streams) return;  // whole streams leave together
  const int rec = a.rec_base + li;
  const int s = a.stream_base + li;
  const uint64_t T = a.T;
  if (T == 0) return;

  const int N = tp->n, NPAD = tp->n_pad;
  const uint32_t stride = FULL ? (uint32_t)(CH * K) : tp->mx[0].stride;
  const uint32_t table = tp->mx[0].table_size;
  const float lr = tp->mx[0].lr;
  uint8_t* const bank = a.banks + (uint64_t)s * tp->bank_bytes;
  float* const wtab = (float*)(bank + tp->mx[0].w_off);
  uint64_t* const rstab = (uint64_t*)(bank + tp->mx[0].rs_off);
  uint64_t* const scal = (uint64_t*)(bank + tp->scal_off);

  const uint64_t RS = a.rec_stride;
  const uint32_t xstride = FULL ? (uint32_t)(CH * K) : (uint32_t)NPAD;
  const float* const pred_s = a.pred + (uint64_t)rec * RS * xstride;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  const float* const dec_s = a.decay + (uint64_t)a.decay_idx[li] * T;
  const int MW = tp->mask_words;
  const uint32_t* const mask_s = HAS_MASK ? a.mask + (uint64_t)rec * RS * MW : nullptr;
  float* const p_s = a.p_out + (uint64_t)rec * RS;
  float* const oa_s = WANT_OUT ? a.out_all + (uint64_t)rec * RS : nullptr;

  uint64_t steps = scal[0], max_steps = scal[1], seen_cnt = scal[2];

  GmxSlot<K> sl[NSLOT];
  uint32_t ctxq[NSLOT];  // contexts of the bits that will refill each slot next

  // 2K+3 loads of bit tb into slot d (past the end of the batch: a harmless re-read of the
  // last record, results never used).  Ragged shapes clamp out-of-row lanes to the row start.
  auto issue = [&](GmxSlot<K>& d, uint32_t ctx, uint64_t tb) {
    const uint64_t tt = tb < T ? tb : T - 1;
    d.row = ctx % table;  // FindMixerData (mixer.cpp:32)
    const float* wr = wtab + (uint64_t)d.row * stride + 4 * l;
    const float* xr = pred_s + tt * (uint64_t)xstride + 4 * l;
    if (FULL) {
      gmx_ld16<0>(d.w[0], wr);
      if (K > 1) gmx_ld16<CH * 4>(d.w[1 % K], wr);
      if (K > 2) gmx_ld16<CH * 8>(d.w[2 % K], wr);
      if (K > 3) gmx_ld16<CH * 12>(d.w[3 % K], wr);
      gmx_ld16<0>(d.x[0], xr);
      if (K > 1) gmx_ld16<CH * 4>(d.x[1 % K], xr);
      if (K > 2) gmx_ld16<CH * 8>(d.x[2 % K], xr);
      if (K > 3) gmx_ld16<CH * 12>(d.x[3 % K], xr);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e = CH * k + 4 * l;
        gmx_ld16<0>(d.w[k], (uint32_t)e < stride ? wr + CH * k : wr - 4 * l);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e = CH * k + 4 * l;
        gmx_ld16<0>(d.x[k], e < NPAD ? xr + CH * k : xr - 4 * l);
      }
    }
    if (HAS_MASK) {
      // the active_models word of each of this lane's quads (a quad never straddles a word)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e = CH * k + 4 * l;
        gmx_ld4(d.mk[k], mask_s + tt * (uint64_t)MW + (e < NPAD ? (e >> 5) : 0));
      }
    }
    gmx_ld8(d.rs, rstab + d.row);
    gmx_ld1(d.bit, bits_s + tt);
    gmx_ld4(d.dec, dec_s + tt);
  };
  // Only active_models are visited (mixer.cpp:57-59): a silent slot contributes nothing to the sum
  // and its weight does not move -- both exactly what a zero input gives.
  auto mask_inputs = [&](GmxSlot<K>& d) {
    if (!HAS_MASK) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t b = d.mk[k] >> ((uint32_t)(CH * k + 4 * l) & 31u);
      d.x[k].x = (b & 1u) ? d.x[k].x : 0.f;
      d.x[k].y = (b & 2u) ? d.x[k].y : 0.f;
      d.x[k].z = (b & 4u) ? d.x[k].z : 0.f;
      d.x[k].w = (b & 8u) ? d.x[k].w : 0.f;
    }
  };
  // Ragged shapes: zero what lies outside the row / the inputs (after the data has landed).
  auto trim = [&](GmxSlot<K>& d) {
    if (FULL) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e = CH * k + 4 * l;
      const bool wok = (uint32_t)e < stride;
      d.w[k].x = wok ? d.w[k].x : 0.f;
      d.w[k].y = wok ? d.w[k].y : 0.f;
      d.w[k].z = wok ? d.w[k].z : 0.f;
      d.w[k].w = wok ? d.w[k].w : 0.f;
      d.x[k].x = (e + 0 < N) ? d.x[k].x : 0.f;  // inputs past n_inputs must not contribute
      d.x[k].y = (e + 1 < N) ? d.x[k].y : 0.f;
      d.x[k].z = (e + 2 < N) ? d.x[k].z : 0.f;
      d.x[k].w = (e + 3 < N) ? d.x[k].w : 0.f;
    }
  };

  // prologue: contexts of the first 2*NSLOT bits, the first NSLOT slots, then drain.
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const uint64_t tb = (uint64_t)k < T ? (uint64_t)k : T - 1;
    ctxq[k] = ctx_s[tb];
  }
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    issue(sl[k], ctxq[k], (uint64_t)k);
    const uint64_t tb = (uint64_t)(k + NSLOT) < T ? (uint64_t)(k + NSLOT) : T - 1;
    gmx_ld4(ctxq[k], ctx_s + tb);
  }
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    gmx_wait_slot<0>(sl[k]);
    if (HAS_MASK) gmx_pin_mask(sl[k]);
    gmx_pin(sl[k].rs, sl[k].bit, sl[k].dec, ctxq[k]);
  }
  trim(sl[0]);
  mask_inputs(sl[0]);
  float acc_cur = gmx_chain<K, LPS>(sl[0], lane);

  for (uint64_t t0 = 0; t0 < T; t0 += NSLOT) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const uint64_t t = t0 + k;
      if (t >= T) break;  // wave-uniform: only the last, partial group of NSLOT bits
      GmxSlot<K>& c = sl[k];
      GmxSlot<K>& nx = sl[(k + 1) % NSLOT];
      // ---- release slot t+1 (rows, inputs) and with it everything older ------------------
      gmx_wait_slot<VM_WAIT>(nx);
      if (HAS_MASK) gmx_pin_mask(nx);
      gmx_pin(c.rs, c.bit, c.dec, ctxq[k]);
      trim(nx);
      mask_inputs(nx);
      // ---- speculative Mixer::Predict chain of bit t+1, spread over the stages of bit t ----
      float acc_nxt = gmx_chain_piece<K, LPS>(nx, 0, 0.f, lane);
      // ---- finish Mixer::Predict of bit t ---------------------------------------------
      float out = __shfl(acc_cur, lane | (LPS - 1));  // the stream's last lane holds the sum
      const bool seen = c.rs != 0;             // unseen row = no row: p = 0 (mixer.cpp:52-55)
      out = seen ? out : 0.f;
      const float pl = gmx_logistic_tab(out, s_tab);  // Sigmoid::Logistic, used by Predict and Learn
      const float prob = gmx_clamp_prob(pl);   // clamp of Predictor::Predict (predictor.cpp:370-374)
      gmx_st4(p_s + t, prob);                  // the lanes of a stream write the same word
      if (WANT_OUT) gmx_st4(oa_s + t, out);
      if (K > 1) acc_nxt = gmx_chain_piece<K, LPS>(nx, 1 % K, acc_nxt, lane);
      if (!LEARN) {
        if (K > 2) acc_nxt = gmx_chain_piece<K, LPS>(nx, 2 % K, acc_nxt, lane);
        if (K > 3) acc_nxt = gmx_chain_piece<K, LPS>(nx, 3 % K, acc_nxt, lane);
      } else {
        // ---- Mixer::Learn (mixer.cpp:108-176) -----------------------------------------
        const double dd = (double)__uint_as_float(c.dec) * (1.5 - ((double)c.rs) / (double)max_steps);
        const float decay = (float)dd;
        const float upd = decay * lr * (pl - (float)c.bit);
        const uint64_t rs_new = c.rs + 1;
        ++steps;
        max_steps = rs_new > max_steps ? rs_new : max_steps;
        seen_cnt += (c.rs == 0) ? 1u : 0u;
        // weight regularisation on every 1024th visit (mixer.cpp:173-175); multiplying by
        // 1.0f otherwise is exact and keeps this block free of branches
        const float sc = ((rs_new & 1023u) == 0) ? (1.0f - 3.0e-6f) : 1.0f;
        if (K > 2) acc_nxt = gmx_chain_piece<K, LPS>(nx, 2 % K, acc_nxt, lane);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          c.w[q].x = (c.w[q].x - upd * c.x[q].x) * sc;
          c.w[q].y = (c.w[q].y - upd * c.x[q].y) * sc;
          c.w[q].z = (c.w[q].z - upd * c.x[q].z) * sc;
          c.w[q].w = (c.w[q].w - upd * c.x[q].w) * sc;
        }
        float* wr = wtab + (uint64_t)c.row * stride + 4 * l;
        if (FULL) {
          gmx_st16<0>(wr, c.w[0]);
          if (K > 1) gmx_st16<CH * 4>(wr, c.w[1 % K]);
          if (K > 2) gmx_st16<CH * 8>(wr, c.w[2 % K]);
          if (K > 3) gmx_st16<CH * 12>(wr, c.w[3 % K]);
        } else {
          // every chunk has lanes inside the row, so each store below is one (partially
          // masked) instruction on every path: the vmcnt bookkeeping does not change
#pragma unroll
          for (int q = 0; q < K; ++q)
            if ((uint32_t)(CH * q + 4 * l) < stride) gmx_st16<0>(wr + CH * q, c.w[q]);
        }
        gmx_st8(rstab + c.row, rs_new);
        if (K > 3) acc_nxt = gmx_chain_piece<K, LPS>(nx, 3 % K, acc_nxt, lane);
        // A newer copy of this row may already sit in a later slot (same gate context again
        // within NSLOT bits): hand the updated row over in registers, redo the chain of bit
        // t+1 if it was built on the stale copy.  Slots t+2.. may still be in flight, so
        // they are drained first (rare path).
        bool hit[NSLOT];
        bool any_hit = false;
#pragma unroll
        for (int j = 1; j < NSLOT; ++j) {
          hit[j] = sl[(k + j) % NSLOT].row == c.row && (t + j) < T;
          any_hit |= hit[j];
        }
        if (__any(any_hit)) {
#pragma unroll
          for (int j = 2; j < NSLOT; ++j) gmx_wait_slot<0>(sl[(k + j) % NSLOT]);
#pragma unroll
          for (int j = 1; j < NSLOT; ++j) {
            GmxSlot<K>& o = sl[(k + j) % NSLOT];
            if (hit[j]) {
#pragma unroll
              for (int q = 0; q < K; ++q) o.w[q] = c.w[q];
              o.rs = rs_new;
            }
          }
          if (__any(hit[1])) acc_nxt = gmx_chain<K, LPS>(nx, lane);
        }
      }
      // ---- refill this slot with bit t + NSLOT, and fetch the context after that ----------
      issue(c, ctxq[k], t + NSLOT);
      {
        const uint64_t tn = t + 2 * NSLOT;
        gmx_ld4(ctxq[k], ctx_s + (tn < T ? tn : T - 1));
      }
      acc_cur = acc_nxt;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last refills are still in flight
  if (LEARN) {
    scal[0] = steps;
    scal[1] = max_steps;
    scal[2] = seen_cnt;
  }
}

template <int LPS, int K, int NSLOT, bool FULL, bool HAS_MASK>
static hipError_t launch_kfm(const GmxTopoDev* tp_dev, const GmxRunArgs& a, hipStream_t stream) {
  constexpr int G = 64 / LPS;
  const dim3 grid((a.n_streams + G - 1) / G), block(64);
  const bool learn = (a.mode & GMX_MODE_LEARN) != 0;
  if (a.out_all) {
    if (learn)
      hipLaunchKernelGGL((gmx_single_kernel<LPS, K, NSLOT, FULL, true, true, HAS_MASK>), grid, block, 0, stream, tp_dev, a);
    else
      hipLaunchKernelGGL((gmx_single_kernel<LPS, K, NSLOT, FULL, true, false, HAS_MASK>), grid, block, 0, stream, tp_dev, a);
  } else {
    if (learn)
      hipLaunchKernelGGL((gmx_single_kernel<LPS, K, NSLOT, FULL, false, true, HAS_MASK>), grid, block, 0, stream, tp_dev, a);
    else
      hipLaunchKernelGGL((gmx_single_kernel<LPS, K, NSLOT, FULL, false, false, HAS_MASK>), grid, block, 0, stream, tp_dev, a);
  }
  return hipGetLastError();
}
template <int LPS, int K, int NSLOT, bool FULL>
static hipError_t launch_kf(const GmxTopoDev* tp_dev, const GmxRunArgs& a, hipStream_t stream) {
  // batches with an active mask: one slot less in the ring where the mask words would push the
  // counted wait past the 6-bit vmcnt
  constexpr int VM_M = 1 + 1 + (K + 1) + 2 * K + 3 + K + 1;
  constexpr int NS_M = (1 + (NSLOT - 2) * VM_M <= 63) ? NSLOT : NSLOT - 1;
  return a.mask ? launch_kfm<LPS, K, NS_M, FULL, true>(tp_dev, a, stream)
                : launch_kfm<LPS, K, NSLOT, FULL, false>(tp_dev, a, stream);
}

template <int LPS, int K, int NSLOT>
static hipError_t launch_k(const GmxTopoDev* tp_dev, const GmxRunArgs& a, bool full,
                           hipStream_t stream) {
  return full ? launch_kf<LPS, K, NSLOT, true>(tp_dev, a, stream)
              : launch_kf<LPS, K, NSLOT, false>(tp_dev, a, stream);
}

// Eligible banks: exactly one layer-0 mixer, n_inputs <= 256, batched Predict(+Learn) mode, with or
// without an active mask.  The host side checks that before calling.
// `variant` picks the lane mapping: 0 = default for the shape, 16/32/64 = lanes per stream.
extern "C" hipError_t gmx_launch_single_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args,
                                               int n_inputs, int variant, hipStream_t stream) {
  (void)hipGetLastError();
  const GmxRunArgs& a = *args;
  if (n_inputs <= 64) return launch_k<16, 1, 4>(tp_dev, a, n_inputs == 64, stream);
  if (n_inputs <= 128) {
    if (variant == 16) return launch_k<16, 2, 4>(tp_dev, a, n_inputs == 128, stream);
    return launch_k<32, 1, 6>(tp_dev, a, n_inputs == 128, stream);
  }
  // Measured on MI355X (2048-3584 streams of 256 inputs): 16 and 32 lanes per stream run
  // within 2% of each other, 64 lanes (twice the waves, same issue-bound SIMDs) 10-15% behind.
  const bool full = n_inputs == 256;
  if (variant == 165) return launch_k<16, 4, 5>(tp_dev, a, full, stream);  // tuning: one more slot in flight
  if (variant == 325) return launch_k<32, 2, 5>(tp_dev, a, full, stream);
  if (variant == 64) return launch_k<64, 1, 6>(tp_dev, a, full, stream);
  if (variant == 32) return launch_k<32, 2, 4>(tp_dev, a, full, stream);
  return launch_k<16, 4, 4>(tp_dev, a, full, stream);
}
