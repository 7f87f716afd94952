// gmx_indirect.hip -- the reference's Indirect models (models/indirect.cpp:28-69) for T bits of
// many streams: one wave per stream, lane k = model k.
//
// Per bit a model reads ONE byte pair -- its two states for the table index
// ((context << 8) + bit_context) % size -- maps each state to a logit through a 256-entry table,
// nudges those two logits towards the coded bit and steps the two state machines.  That is
// byte/integer work bound by the latency of one 2-byte access per model and bit:
//   * both one-byte tables of a model are interleaved into one u16 table (one access, one sector);
//     within a byte the 8 indices of a model fall into one 512-byte block, so after the first
//     touch the accesses are L2 hits;
//   * the index of bit t+1 is known from the records, so its entry is fetched while bit t is
//     processed (and patched from registers in the rare case that both are the same entry);
//   * the 2 x 256 logits of every model (82 KiB for the stock 41 models) and the two
//     next-state tables live in LDS for the whole launch;
//   * outputs go straight into the prediction / active-mask records of a mixer batch when one
//     is attached: no host round trip between the models and the mixers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gmx_internal.h"
#include "gmx_math.h"

// Loads of the block pipeline are issued from inline asm: left to itself hipcc sinks them next to
// their uses, which puts one memory latency back into every bit.
__device__ __forceinline__ void ind_ld32(uint32_t& d, const uint32_t* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ind_ld32o(uint32_t& d, const void* p) {
  asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ind_ld8o(uint32_t& d, const void* p) {
  asm volatile("global_load_ubyte %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
template <int J, int D>
__device__ __forceinline__ void ind_block_records(uint32_t* ctx_n, uint32_t* bc_n, uint32_t* bit_n,
                                                  const uint32_t* ctx_p, uint32_t k_bytes, const uint32_t* bc_p,
                                                  const uint8_t* bit_p) {
  if constexpr (J < D) {
    ind_ld32(ctx_n[J], (const uint32_t*)((const uint8_t*)ctx_p + (uint64_t)J * k_bytes));
    ind_ld32o<4 * J>(bc_n[J], bc_p);
    ind_ld8o<J>(bit_n[J], bit_p);
    ind_block_records<J + 1, D>(ctx_n, bc_n, bit_n, ctx_p, k_bytes, bc_p, bit_p);
  }
}
__device__ __forceinline__ void ind_ld16(uint32_t& d, const uint16_t* p) {
  asm volatile("global_load_ushort %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void ind_ld8(uint32_t& d, const uint8_t* p) {
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}

// OUT: write {predictions, active} records; MX: write into a mixer batch; LEARN: Indirect::Learn.
template <bool OUT, bool MX, bool LEARN>
__global__ void __launch_bounds__(64)
gmx_indirect_kernel(const GmxIndDev* __restrict__ dv, const GmxIndRunArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int K = dv->k;
  const int s = a.stream_base + (int)blockIdx.x;
  const int rec = a.rec_base + (int)blockIdx.x;
  const uint64_t T = a.T;
  if (T == 0) return;
  uint8_t* const bank = a.banks + (uint64_t)s * dv->bank_bytes;

  // LDS: [k][512] logits | exp2 table (32 x u64) | ns_next[512] | rm_next[512] | mask words
  float* const ptab = lds;
  uint64_t* const s_tab = (uint64_t*)(lds + (size_t)K * 512);
  uint8_t* const nsn = (uint8_t*)(s_tab + 32);
  uint8_t* const rmn = nsn + 512;
  uint32_t* const mw = (uint32_t*)(rmn + 512);
  float* const gp = (float*)(bank + dv->pred_off);
  for (int i = lane; i < K * 512; i += 64) ptab[i] = gp[i];
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];
  for (int i = lane; i < 512; i += 64) {
    nsn[i] = dv->ns_next[i];
    rmn[i] = dv->rm_next[i];
  }
  __syncthreads();

  const bool on = lane < K;
  // idle lanes mirror model 0: same values to the same addresses, so the loop needs no exec
  // masking at all (a divergent region around the stores also makes hipcc drain vmcnt per bit)
  const int ml = on ? lane : 0;
  const GmxIndModelDev d = dv->m[ml];
  uint16_t* const tab = (uint16_t*)(bank + d.tab_off);
  float* const slots = (float*)(bank + dv->slots_off);
  float va = slots[2 * ml], vb = slots[2 * ml + 1];
  float* const nsp = ptab + (size_t)ml * 512;
  float* const rmp = nsp + 256;

  const uint64_t RS = a.rec_stride;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS * K;
  const uint32_t* const bc_s = a.bc + (uint64_t)rec * RS;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  float* const po = OUT ? a.pred_out + (uint64_t)rec * RS * 2 * K : nullptr;
  uint8_t* const ao = OUT ? a.act_out + (uint64_t)rec * RS * 2 * K : nullptr;
  float* const mxp = MX ? a.mx_pred + (uint64_t)rec * a.mx_rec_stride * a.mx_n_pad : nullptr;
  uint32_t* const mxm = MX ? a.mx_mask + (uint64_t)rec * a.mx_rec_stride * a.mx_mask_words : nullptr;
  uint8_t* const mxb = MX ? a.mx_bits + (uint64_t)rec * a.mx_rec_stride : nullptr;
  const int MW = a.mx_mask_words;
  // bits of the attached mask that belong to the Indirect models (cleared and rewritten per bit)
  uint32_t own = 0;
  if (MX) {
    if (lane < MW) mw[lane] = 0;
    __syncthreads();
    if (on) {
      atomicOr(&mw[d.slot_a >> 5], 1u << (d.slot_a & 31));
      atomicOr(&mw[d.slot_b >> 5], 1u << (d.slot_b & 31));
    }
    __syncthreads();
    if (lane < MW) own = mw[lane];
    __syncthreads();
  }

  // Bits are processed in blocks of D.  The table indices of a block depend on the records only,
  // so its D entries are fetched together -- one memory latency per block instead of one per
  // bit -- behind the records of the next block; an entry that the block itself rewrites
  // before using it again is patched from registers.  The body is branch-free per lane
  // (selects): with one wave per stream every exec-mask detour costs.  (Fetching block b+1's
  // entries during block b was tried: what it hides is less than what patching across blocks
  // costs -- the loop is bound by its dependent LDS/logistic chain, not by the fetch.)
  constexpr int D = 8;
  uint32_t prev_ctx = 0, base = 0;
  bool have_base = false;
  // ((context << 8) + bit_context) % size (indirect.cpp:31-32, 32-bit wrap).  A model's context
  // changes once per byte, so the residue of (context << 8) is kept.
  auto index_of = [&](uint32_t ctx, uint32_t bcu) -> uint32_t {
    const bool moved = !have_base || ctx != prev_ctx;
    if (__ballot(moved)) {  // wave-uniform
      const uint32_t nb = (ctx << 8) % d.size;
      base = moved ? nb : base;
      prev_ctx = ctx;
      have_base = true;
    }
    if (bcu < 256u) {  // always true for the reference's bit_context
      const uint32_t ix = base + bcu;
      return ix >= d.size ? ix - d.size : ix;
    }
    return ((ctx << 8) + bcu) % d.size;
  };
  auto load_records = [&](uint32_t* c, uint32_t* bcv, uint32_t* bv, uint64_t t_first) {
    if (t_first + D <= T) {
      ind_block_records<0, D>(c, bcv, bv, ctx_s + t_first * K + ml, (uint32_t)K * 4u, bc_s + t_first,
                              bits_s + t_first);
    } else {
#pragma unroll
      for (int j = 0; j < D; ++j) {  // the last blocks: clamped (values past T are not used)
        uint64_t t = t_first + j;
        t = t < T ? t : T - 1;
        ind_ld32(c[j], ctx_s + t * K + ml);
        ind_ld32(bcv[j], bc_s + t);
        ind_ld8o<0>(bv[j], bits_s + t);
      }
    }
  };
  uint32_t ctx_r[D], bc_r[D], bit_r[D];
  load_records(ctx_r, bc_r, bit_r, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < D; ++j) asm volatile("" : "+v"(ctx_r[j]), "+v"(bc_r[j]), "+v"(bit_r[j]));

  for (uint64_t t0 = 0; t0 < T; t0 += D) {
    // records of the next block first: vector memory returns in order, so once this block's
    // entries (issued behind them) are there, nothing is left to wait for while the bits run
    uint32_t ctx_n[D], bc_n[D], bit_n[D], idx[D], e[D];
    load_records(ctx_n, bc_n, bit_n, t0 + D);
    // Inside one byte (one context, bit_context growing) the D indices are distinct: the usual
    // case, blocks being bytes.  Otherwise an entry may come twice and is patched below.
    bool plain = true;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const uint32_t bcu = __builtin_amdgcn_readfirstlane(bc_r[j]);
      idx[j] = index_of(ctx_r[j], bcu);
      ind_ld16(e[j], tab + idx[j]);
      plain = plain && bcu < 256u;
      if (j > 0) plain = plain && ctx_r[j] == ctx_r[0] && bcu > __builtin_amdgcn_readfirstlane(bc_r[j - 1]);
    }
    const bool patch = __ballot(!plain) != 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < D; ++j)  // pins the loaded registers behind the wait
      asm volatile("" : "+v"(e[j]), "+v"(ctx_n[j]), "+v"(bc_n[j]), "+v"(bit_n[j]));
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const uint64_t t = t0 + j;
      if (t >= T) break;
      const int bit = (int)__builtin_amdgcn_readfirstlane(bit_r[j]);
      // ---- Indirect::Predict (indirect.cpp:28-46) ------------------------------------------
      const uint32_t ns = e[j] & 255u, rm = e[j] >> 8;
      const float qa = nsp[ns], qb = rmp[rm];
      const bool seen_a = ns != 255u, seen_b = rm != 0u;  // never-seen states leave the slot alone
      va = seen_a ? qa : va;
      vb = seen_b ? qb : vb;
      const bool act_a = seen_a && qa != 0.f;  // SetLogitPrediction: a zero logit is stored, not active
      const bool act_b = seen_b && qb != 0.f;
      if (OUT) {
        *(float2*)(po + t * 2 * K + 2 * ml) = make_float2(va, vb);
        *(uchar2*)(ao + t * 2 * K + 2 * ml) = make_uchar2(act_a, act_b);
      }
      if (MX) {
        mxp[t * a.mx_n_pad + d.slot_a] = va;
        mxp[t * a.mx_n_pad + d.slot_b] = vb;
        if (lane < MW) mw[lane] = 0;
        __syncthreads();
        if (act_a) atomicOr(&mw[d.slot_a >> 5], 1u << (d.slot_a & 31));
        if (act_b) atomicOr(&mw[d.slot_b >> 5], 1u << (d.slot_b & 31));
        __syncthreads();
        if (lane < MW) {
          uint32_t* wp = mxm + t * MW + lane;
          *wp = (*wp & ~own) | mw[lane];
        }
        if (lane == 0) mxb[t] = (uint8_t)bit;
        __syncthreads();
      }
      // ---- Indirect::Learn (indirect.cpp:48-69) --------------------------------------------
      if (LEARN) {
        const uint32_t sn = seen_a ? ns : 0u;  // the uninitialised state learns as state 0
        const float pa = nsp[sn];
        const float na = pa + ((float)bit - gmx_logistic_tab(pa, s_tab)) * d.lr;
        const float nb = qb + ((float)bit - gmx_logistic_tab(qb, s_tab)) * d.lr;
        const uint32_t e_new = (uint32_t)nsn[2 * sn + bit] | ((uint32_t)rmn[2 * rm + bit] << 8);
        nsp[sn] = na;
        rmp[rm] = nb;
        tab[idx[j]] = (uint16_t)e_new;
        if (patch) {
#pragma unroll
          for (int i = j + 1; i < D; ++i)  // the same entry again later in this block
            e[i] = idx[i] == idx[j] ? e_new : e[i];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      ctx_r[j] = ctx_n[j];
      bc_r[j] = bc_n[j];
      bit_r[j] = bit_n[j];
    }
  }
  slots[2 * ml] = va;
  slots[2 * ml + 1] = vb;
  __syncthreads();
  if (LEARN)
    for (int i = lane; i < K * 512; i += 64) gp[i] = ptab[i];
}

template <bool OUT, bool MX, bool LEARN>
static hipError_t ind_launch_as(const GmxIndDev* dv, const GmxIndRunArgs* args, int n_streams, unsigned lds_bytes,
                                hipStream_t stream) {
  static unsigned allowed = 48u * 1024u;  // per instantiation: dynamic LDS the runtime has been told about
  if (lds_bytes > allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)gmx_indirect_kernel<OUT, MX, LEARN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    allowed = lds_bytes;
  }
  hipLaunchKernelGGL((gmx_indirect_kernel<OUT, MX, LEARN>), dim3(n_streams), dim3(64), lds_bytes, stream, dv, *args);
  return hipGetLastError();
}

extern "C" hipError_t gmx_launch_indirect_kernel(const GmxIndDev* dv, const GmxIndRunArgs* args, int n_streams,
                                                 unsigned lds_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  const bool out = args->pred_out != nullptr, mx = args->mx_pred != nullptr, learn = args->learn != 0;
  if (out) {
    if (mx) return learn ? ind_launch_as<true, true, true>(dv, args, n_streams, lds_bytes, stream)
                         : ind_launch_as<true, true, false>(dv, args, n_streams, lds_bytes, stream);
    return learn ? ind_launch_as<true, false, true>(dv, args, n_streams, lds_bytes, stream)
                 : ind_launch_as<true, false, false>(dv, args, n_streams, lds_bytes, stream);
  }
  if (mx) return learn ? ind_launch_as<false, true, true>(dv, args, n_streams, lds_bytes, stream)
                       : ind_launch_as<false, true, false>(dv, args, n_streams, lds_bytes, stream);
  return learn ? ind_launch_as<false, false, true>(dv, args, n_streams, lds_bytes, stream)
               : ind_launch_as<false, false, false>(dv, args, n_streams, lds_bytes, stream);
}

// ---------------------------------------------------------------------------------------
// Session kernel: Indirect::Predict / Indirect::Learn one bit at a time without a kernel launch per
// call (a decoder learns the bit from Predict's own result, coder/decoder.cpp:19-39).  One persistent
// wave per stream, the protocol of gmx_stock_session_kernel: commands from a mailbox, the logit
// tables resident in LDS, the table entry of the bit in a register between forward and learn; the
// wave leaves -- writing the logit tables and the slot values back -- on GMX_MB_STOP or after
// `idle_ticks` of s_memrealtime without a command.  `replay_forward`: started between the forward and
// the learn of a bit, it recomputes the forward from the contexts still in the mailbox first.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
gmx_indirect_session_kernel(const GmxIndDev* __restrict__ dv, uint8_t* banks, int stream, GmxIndMbCmd* mc,
                            GmxIndMbReply* mr, unsigned long long idle_ticks, int replay_forward) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int K = dv->k;
  uint8_t* const bank = banks + (uint64_t)stream * dv->bank_bytes;
  float* const ptab = lds;
  uint64_t* const s_tab = (uint64_t*)(lds + (size_t)K * 512);
  uint8_t* const nsn = (uint8_t*)(s_tab + 32);
  uint8_t* const rmn = nsn + 512;
  float* const gp = (float*)(bank + dv->pred_off);
  for (int i = lane; i < K * 512; i += 64) ptab[i] = gp[i];
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];
  for (int i = lane; i < 512; i += 64) {
    nsn[i] = dv->ns_next[i];
    rmn[i] = dv->rm_next[i];
  }
  __syncthreads();
  const bool on = lane < K;
  const int ml = on ? lane : 0;  // idle lanes mirror model 0 (same values to the same addresses)
  const GmxIndModelDev d = dv->m[ml];
  uint16_t* const tab = (uint16_t*)(bank + d.tab_off);
  float* const slots = (float*)(bank + dv->slots_off);
  float va = slots[2 * ml], vb = slots[2 * ml + 1];
  float* const nsp = ptab + (size_t)ml * 512;
  float* const rmp = nsp + 256;
  volatile GmxIndMbReply* const vr = mr;
  const volatile GmxIndMbCmd* const vc = mc;

  uint32_t seen = __hip_atomic_load(&mr->done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint32_t idx = 0, e = 0;
  bool have_fwd = false, dirty = false;
  bool replay = replay_forward != 0;
  uint32_t exit_state = GMX_MB_EXIT_IDLE;
  for (;;) {
    uint32_t word = seen, cmd = GMX_MB_FORWARD;
    uint32_t slot = (uint32_t)(replay_forward - 1) & 1u;
    if (!replay) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      bool idle = false;
      for (;;) {
        word = __hip_atomic_load(&mc->cmd_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (word != seen) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) { idle = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (idle) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      cmd = word & GMX_MB_CMD_MASK;
      slot = (word >> GMX_MB_SLOT_SHIFT) & 1u;
    }
    if (cmd == GMX_MB_STOP) { exit_state = GMX_MB_EXIT_STOP; seen = word; break; }
    const bool with_learn = cmd == GMX_MB_LEARN0 || cmd == GMX_MB_LEARN1 || cmd == GMX_MB_LEARN0_FWD ||
                            cmd == GMX_MB_LEARN1_FWD;
    const bool with_forward = cmd == GMX_MB_FORWARD || cmd == GMX_MB_LEARN0_FWD || cmd == GMX_MB_LEARN1_FWD;
    if (with_learn && have_fwd) {
      // ---- Indirect::Learn (indirect.cpp:48-69) on the entry the forward latched ------------
      const int bit = (cmd == GMX_MB_LEARN1 || cmd == GMX_MB_LEARN1_FWD) ? 1 : 0;
      const uint32_t ns = e & 255u, rm = e >> 8;
      const uint32_t sn = ns != 255u ? ns : 0u;  // the uninitialised state learns as state 0
      const float pa = nsp[sn], qb = rmp[rm];
      const float na = pa + ((float)bit - gmx_logistic_tab(pa, s_tab)) * d.lr;
      const float nb = qb + ((float)bit - gmx_logistic_tab(qb, s_tab)) * d.lr;
      const uint32_t e_new = (uint32_t)nsn[2 * sn + bit] | ((uint32_t)rmn[2 * rm + bit] << 8);
      nsp[sn] = na;
      rmp[rm] = nb;
      tab[idx] = (uint16_t)e_new;
      have_fwd = false;
      dirty = true;
    }
    if (with_forward) {
      // ---- Indirect::Predict (indirect.cpp:28-46) -------------------------------------------
      const uint32_t ctx = vc->ctx[slot][ml];
      const uint32_t bcu = vc->bit_context[slot];
      idx = ((ctx << 8) + bcu) % d.size;  // indirect.cpp:31-32, 32-bit wrap
      uint32_t ev;
      asm volatile("s_waitcnt vmcnt(0)\n\tglobal_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(ev) : "v"(tab + idx) : "memory");  // behind the entry store of the learn above
      e = ev;
      const uint32_t ns = e & 255u, rm = e >> 8;
      const float qa = nsp[ns], qb = rmp[rm];
      const bool seen_a = ns != 255u, seen_b = rm != 0u;  // never-seen states leave the slot alone
      va = seen_a ? qa : va;
      vb = seen_b ? qb : vb;
      const uint64_t act_a = __ballot(on && seen_a && qa != 0.f);  // a zero logit is stored, not active
      const uint64_t act_b = __ballot(on && seen_b && qb != 0.f);
      have_fwd = true;
      if (!replay) {
        if (on) {
          vr->pred[2 * lane] = va;
          vr->pred[2 * lane + 1] = vb;
        }
        if (lane == 0) {
          vr->active_a = act_a;
          vr->active_b = act_b;
        }
      }
    }
    if (replay) {
      replay = false;
      continue;
    }
    seen = word;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(&mr->done_seq, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // leave: what is only in LDS / registers goes back to the bank
  if (on) {
    slots[2 * lane] = va;
    slots[2 * lane + 1] = vb;
  }
  __syncthreads();
  if (dirty)
    for (int i = lane; i < K * 512; i += 64) gp[i] = ptab[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  if (lane == 0) {
    if (exit_state == GMX_MB_EXIT_STOP)
      __hip_atomic_store(&mr->done_seq, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mr->state, exit_state, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

extern "C" hipError_t gmx_launch_indirect_session(const GmxIndDev* dv, uint8_t* banks, int stream_idx, GmxIndMbCmd* mc,
                                                  GmxIndMbReply* mr, unsigned long long idle_ticks,
                                                  int replay_forward, unsigned lds_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  static unsigned allowed = 48u * 1024u;
  if (lds_bytes > allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)gmx_indirect_session_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    allowed = lds_bytes;
  }
  hipLaunchKernelGGL(gmx_indirect_session_kernel, dim3(1), dim3(64), lds_bytes, stream, dv, banks, stream_idx, mc,
                     mr, idle_ticks, replay_forward);
  return hipGetLastError();
}

// Fill every model's table with "never seen" (nonstationary 255, run map 0) and zero the rest.
__global__ void gmx_indirect_init_kernel(uint8_t* banks, uint64_t bank_bytes, uint64_t tab_bytes) {
  const uint64_t n16 = tab_bytes / 16;  // tables first, 16-byte granules
  uint8_t* bank = banks + (uint64_t)blockIdx.y * bank_bytes;
  const uint4 v = make_uint4(0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu);
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
    ((uint4*)bank)[i] = v;
  const uint64_t rest = (bank_bytes - tab_bytes) / 16;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < rest; i += (uint64_t)gridDim.x * blockDim.x)
    ((uint4*)(bank + tab_bytes))[i] = z;
}

extern "C" hipError_t gmx_launch_indirect_init(uint8_t* banks, uint64_t bank_bytes, uint64_t tab_bytes,
                                               int n_streams, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_indirect_init_kernel, dim3(256, n_streams), dim3(256), 0, stream, banks, bank_bytes,
                     tab_bytes);
  return hipGetLastError();
}
