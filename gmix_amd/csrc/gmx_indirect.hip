// gmx_indirect.hip -- the reference's Indirect models (models/indirect.cpp:28-69) for T bits of
// many streams: two waves per stream (one per state of a model), lane k = model k.
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

#include <type_traits>

#include "gmx_internal.h"
#include "gmx_math.h"
#include "gmx_step_dev.h"

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

// Phase profile (-DGMX_IND_PROF, `make -C gmix_amd/csrc prof`): s_memtime ticks wave 0 of block 0 spends in
// the stages of a full block, summed; read with gmx_indirect_prof_read.
#ifdef GMX_IND_PROF
__device__ unsigned long long gmx_ind_prof[8];
#define IND_STAMP(i)                                             \
  do {                                                           \
    if (prof_on) {                                               \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      prof_acc[i] += now_ - tprev_;                              \
      tprev_ = now_;                                             \
    }                                                            \
  } while (0)
#else
#define IND_STAMP(i)
#endif

// OUT: write {predictions, active} records; MX: write into a mixer batch; LEARN: Indirect::Learn.
//
// Two waves per stream.  A model is two independent predictors that share nothing but the table index:
// the nonstationary state (low byte of the entry, logits nsp, next-state table ns_next, blackboard slot
// "-indirect") and the run-map state (high byte, rmp, rm_next, slot "-run_map").  With one wave per SIMD
// the loop is bound by the instructions the wave issues per bit, most of them the two Sigmoid::Logistic;
// wave 0 of the block takes the first half, wave 1 the second, each reads and writes only its own byte of
// an entry, its own logit table and its own output slots -- they never wait for each other.
template <bool OUT, bool MX, bool LEARN>
__global__ void __launch_bounds__(128)
gmx_indirect_kernel(const GmxIndDev* __restrict__ dv, const GmxIndRunArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0: nonstationary, 1: run map
  const int K = dv->k;
  const int s = a.stream_base + (int)blockIdx.x;
  const int rec = a.rec_base + (int)blockIdx.x;
  const uint64_t T = a.T_list ? a.T_list[blockIdx.x] : a.T;  // (wave-uniform: one stream per block)
  if (T == 0) return;
  uint8_t* const bank = a.banks + (uint64_t)s * dv->bank_bytes;

  // LDS: [k][512] logits | exp2 table (32 x u64) | ns_next[512] | rm_next[512] | mask words [2][8]
  float* const ptab = lds;
  uint64_t* const s_tab = (uint64_t*)(lds + (size_t)K * 512);
  uint8_t* const nsn = (uint8_t*)(s_tab + 32);
  uint8_t* const rmn = nsn + 512;
  uint32_t* const mw = (uint32_t*)(rmn + 512) + 8 * half;
  float* const gp = (float*)(bank + dv->pred_off);
  uint32_t big = 0;  // any logit of the stream at or beyond 32 in magnitude (or not a number)?
  for (int i = threadIdx.x; i < K * 512; i += 128) {
    const float v = gp[i];
    ptab[i] = v;
    big |= (gmx_f2u(v) & 0x7fffffffu) >= 0x42000000u ? 1u : 0u;
  }
  if (threadIdx.x < 32) s_tab[threadIdx.x] = gmx_exp2f_tab[threadIdx.x];
  for (int i = threadIdx.x; i < 512; i += 128) {
    nsn[i] = dv->ns_next[i];
    rmn[i] = dv->rm_next[i];
  }
  // Logits that start below 32 in magnitude stay there while learning rates are at most 1: a step adds
  // (bit - logistic(p)) * lr, at most lr below 17 and nothing at all beyond it (1 - logistic(17) is under half an ulp of
  // 17 already).  Then every logistic of the launch may take the short way of gmx_math.h (no special cases of expf,
  // no scaling steps in the division) without asking again bit by bit: `small` is block-uniform.
  const int any_big = __syncthreads_or((int)big);

  const bool on = lane < K;
  const int ml = on ? lane : 0;
  const GmxIndModelDev d = dv->m[ml];
  const bool small = !any_big && __ballot(!(__builtin_fabsf(d.lr) <= 1.0f)) == 0;
  uint8_t* const tab = bank + d.tab_off + half;  // this half's byte of entry i: tab[2 * i]
  float* const slots = (float*)(bank + dv->slots_off);
  float val = slots[2 * ml + half];
  float* const lp = ptab + (size_t)ml * 512 + 256 * half;  // this half's 256 logits
  const uint8_t* const nextp = half ? rmn : nsn;
  const int slot = half ? d.slot_b : d.slot_a;
  const uint32_t unseen = half ? 0u : 255u;  // run-map state 0 / nonstationary state 255: never seen

  const uint64_t RS = a.rec_stride;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS * K;
  const uint32_t* const bc_s = a.bc + (uint64_t)rec * RS;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  float* const po = OUT ? a.pred_out + (uint64_t)rec * RS * 2 * K + 2 * ml + half : nullptr;
  uint8_t* const ao = OUT ? a.act_out + (uint64_t)rec * RS * 2 * K + 2 * ml + half : nullptr;
  float* const mxp = MX ? a.mx_pred + (uint64_t)rec * a.mx_rec_stride * a.mx_n_pad + slot : nullptr;
  uint32_t* const mxm = MX ? a.mx_mask + (uint64_t)rec * a.mx_rec_stride * a.mx_mask_words : nullptr;
  uint8_t* const mxb = MX ? a.mx_bits + (uint64_t)rec * a.mx_rec_stride : nullptr;
  const int MW = a.mx_mask_words;  // <= 8 (the launcher checks)
  // bits of the attached mask that belong to this half's slots (cleared and rewritten per bit).  The mask
  // words are composed in LDS by the wave alone: LDS keeps one wave's accesses in order, no barrier.
  uint32_t own = 0;
  if (MX) {
    if (lane < MW) mw[lane] = 0;
    if (on) atomicOr(&mw[slot >> 5], 1u << (slot & 31));
    if (lane < MW) own = mw[lane];
  }

  // The lanes without a model sit the loop out: a vector-memory instruction whose lanes go to different
  // lines costs the wave time per active lane.
  if (on) {
#ifdef GMX_IND_PROF
    const bool prof_on = blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    // Bits are processed in blocks of D = 8.  The table indices of a block depend on the records only, so
    // its D entries are fetched while the block before it is processed, behind the records of the block
    // after it: no memory latency is left in the loop.  What the block in between writes into an entry the
    // fetched block uses is patched in from registers.
    //
    // The reference's own records are "byte-shaped": a block is one byte -- one context per model, the
    // bit_context + 1 of bit j j+1 bits long.  Then the D entries of a block are distinct, entry j of a block can
    // only coincide with entry j of the block before it (same context, same bit_context) unless the index
    // ranges of two different contexts overlap (`near`), the index is an add and a wrap, and nothing in a
    // bit's code depends on lane-uniform branches.  Any other shape of records takes the general code: the
    // residue per bit, entries that repeat inside a block patched pairwise, the next block's entries fetched
    // again behind this block's stores.  With one wave per SIMD the loop is bound by the instructions it
    // issues: the shaped path is written to need few.
    constexpr int D = 8;
    struct Recs {
      uint32_t ctx[D], bc[D], bit[D];
    };
    struct Block {
      uint32_t idx[D], e[D];
      bool shaped;  // wave-uniform: byte-shaped for every model
      bool clean;   // wave-uniform: shaped, and no model's index range overlaps that of the block before
    };
    uint32_t prev_ctx = 0, base = 0;
    bool have_base = false;
    // x % size without a division: q = mulhi(x, floor(2^32 / size)) is the quotient or one less
    const uint32_t magic = d.size == 1u ? 0xffffffffu : (uint32_t)(0x100000000ull / d.size);
    auto residue = [&](uint32_t x) -> uint32_t {
      uint32_t r = x - __umulhi(x, magic) * d.size;
      r = r < r - d.size ? r : r - d.size;  // min(r, r - size): r - size wraps when r < size
      return r < r - d.size ? r : r - d.size;
    };
    // ((context << 8) + bit_context) % size (indirect.cpp:31-32, 32-bit wrap), any records.  A model's
    // context changes once per byte, so the residue of (context << 8) is kept.
    auto index_of = [&](uint32_t ctx, uint32_t bcu) -> uint32_t {
      const bool moved = !have_base || ctx != prev_ctx;
      if (__ballot(moved)) {  // wave-uniform
        const uint32_t nb = residue(ctx << 8);
        base = moved ? nb : base;
        prev_ctx = ctx;
        have_base = true;
      }
      if (bcu < 256u) {
        const uint32_t ix = base + bcu;
        return ix >= d.size ? ix - d.size : ix;
      }
      return residue((ctx << 8) + bcu);
    };
    auto load_records = [&](Recs& r, uint64_t t_first) {
      if (t_first + D <= T) {
        ind_block_records<0, D>(r.ctx, r.bc, r.bit, ctx_s + t_first * K + ml, (uint32_t)K * 4u, bc_s + t_first,
                                bits_s + t_first);
      } else {
#pragma unroll
        for (int j = 0; j < D; ++j) {  // the last blocks: clamped (values past T are not used)
          uint64_t t = t_first + j;
          t = t < T ? t : T - 1;
          ind_ld32(r.ctx[j], ctx_s + t * K + ml);
          ind_ld32(r.bc[j], bc_s + t);
          ind_ld8o<0>(r.bit[j], bits_s + t);
        }
      }
    };
    auto pin_records = [&](Recs& r) {
#pragma unroll
      for (int j = 0; j < D; ++j) asm volatile("" : "+v"(r.ctx[j]), "+v"(r.bc[j]), "+v"(r.bit[j]));
    };
    auto pin_entries = [&](Block& blk) {
#pragma unroll
      for (int j = 0; j < D; ++j) asm volatile("" : "+v"(blk.e[j]));
    };
    // indices of a block and the loads of its entries (this wave's byte of each)
    auto block_indices = [&](Block& blk, const Recs& r) {
      uint32_t odd = 0;  // branch-free: any bit set = not byte-shaped
#pragma unroll
      for (int j = 0; j < D; ++j) {
        odd |= ((r.bc[j] + 1u) >> j) ^ 1u;  // bit_context = recent_bits - 1 (basic-contexts.cpp:33)
        if (j > 0) odd |= r.ctx[j] ^ r.ctx[0];
      }
      blk.shaped = __ballot(odd != 0u) == 0;
      const uint32_t base_before = base;
      const bool had = have_base;
      if (blk.shaped) {
        const bool moved = !have_base || r.ctx[0] != prev_ctx;
        if (__ballot(moved)) {
          const uint32_t nb = residue(r.ctx[0] << 8);
          base = moved ? nb : base;
        }
        prev_ctx = r.ctx[0];
        have_base = true;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const uint32_t ix = base + r.bc[j];
          blk.idx[j] = ix >= d.size ? ix - d.size : ix;
        }
      } else {
#pragma unroll
        for (int j = 0; j < D; ++j) blk.idx[j] = index_of(r.ctx[j], __builtin_amdgcn_readfirstlane(r.bc[j]));
      }
      IND_STAMP(5);
#pragma unroll
      for (int j = 0; j < D; ++j) ind_ld8(blk.e[j], tab + 2ull * blk.idx[j]);
      IND_STAMP(6);
      const uint32_t dd = base >= base_before ? base - base_before : base_before - base;
      const bool near = had && dd != 0u && (dd < 256u || d.size - dd < 256u);
      blk.clean = blk.shaped && __ballot(near) == 0;
    };
    // stores a full block issues behind the next block's entry loads (vector memory returns in order)
    constexpr int kStoresPerBlock = D * ((OUT ? 2 : 0) + (MX ? 1 : 0) + (LEARN ? 1 : 0));

    // running output positions (bit t): one add per bit instead of a 64-bit multiply
    float* po_t = po;
    uint8_t* ao_t = ao;
    float* mxp_t = mxp;
    uint32_t* mxm_t = MX ? mxm + lane : nullptr;
    uint8_t* mxb_t = mxb;
    const uint32_t K2 = 2u * (uint32_t)K;
    const bool bit_writer = half == 0 && lane == 0;

    // one bit of block `cur`: Indirect::Predict, the outputs, Indirect::Learn; e_new = the entry afterwards
    // In a byte-shaped block the states of all D bits are known at its start, so the logits of bit j+1 are read
    // while bit j is computed -- BEFORE bit j's logit is written -- and what bit j writes is forwarded in a
    // register when bit j+1 reads the same state (every other lane-bit on the bench's records): the LDS round
    // trip leaves the chain of dependent operations, which is what this loop waits for.
    float q_fwd = 0.f, p0_fwd = 0.f;  // the logits of the next bit of a shaped block: lp[e[j+1]] and lp[0]
    auto do_bit = [&](Block& cur, int j, uint32_t bit, uint32_t& e_new, auto shaped_tag, auto small_tag) {
      constexpr bool kShaped = decltype(shaped_tag)::value;
      constexpr bool kSmall = decltype(small_tag)::value;
      // ---- Indirect::Predict (indirect.cpp:28-46) ------------------------------------------
      const uint32_t st = cur.e[j];
      const bool first = !kShaped || !LEARN || j == 0;
      const float q = first ? lp[st] : q_fwd;
      const float p0 = first ? lp[0] : p0_fwd;  // what an uninitialised nonstationary state learns at
      float q_raw = 0.f, p0_raw = 0.f;
      if (kShaped && LEARN && j + 1 < D) {  // the next bit's, as they stand before this bit learns
        q_raw = lp[cur.e[j + 1]];
        p0_raw = lp[0];
      }
      const bool seen = st != unseen;  // a never-seen state leaves the slot alone
      val = seen ? q : val;
      const bool act = seen && q != 0.f;  // SetLogitPrediction: a zero logit is stored, not active
      if (OUT) {
        *po_t = val;
        *ao_t = (uint8_t)act;
        po_t += K2;
        ao_t += K2;
      }
      if (MX) {
        *mxp_t = val;
        mxp_t += a.mx_n_pad;
        if (lane < MW) mw[lane] = 0;
        if (act) atomicOr(&mw[slot >> 5], 1u << (slot & 31));
        if (lane < MW) {
          // the other wave rewrites its own bits of the same words: two atomics whose order does not matter
          atomicAnd(mxm_t, ~own);
          atomicOr(mxm_t, mw[lane]);
        }
        mxm_t += MW;
        if (bit_writer) *mxb_t = (uint8_t)bit;
        ++mxb_t;
      }
      // ---- Indirect::Learn (indirect.cpp:48-69) --------------------------------------------
      e_new = st;
      if (LEARN) {
        uint32_t sn = st;
        float p = q;
        if (half == 0) {  // the uninitialised nonstationary state learns as state 0 (the run map at its state)
          sn = seen ? st : 0u;
          p = seen ? q : p0;
        }
        const float n = p + ((float)bit - (kSmall ? gmx_logistic_short(p, s_tab) : gmx_logistic_tab(p, s_tab))) * d.lr;
        e_new = nextp[2 * sn + bit];
        lp[sn] = n;
        tab[2ull * cur.idx[j]] = (uint8_t)e_new;
        if (kShaped && j + 1 < D) {
          q_fwd = cur.e[j + 1] == sn ? n : q_raw;
          p0_fwd = sn == 0u ? n : p0_raw;
        }
        if (!kShaped) {
#pragma unroll
          for (int i = j + 1; i < D; ++i)  // the same entry again later in this block
            cur.e[i] = cur.idx[i] == cur.idx[j] ? e_new : cur.e[i];
        }
      }
    };

    Recs rn, rf;      // records of the next block (landed) and of the one after it (in flight)
    uint32_t bit_c[D];  // the bits of the current block
    Block cur, nxt;
    {
      Recs r0;
      load_records(r0, 0);
      load_records(rn, D);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pin_records(r0);
      pin_records(rn);
      block_indices(cur, r0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pin_entries(cur);
#pragma unroll
      for (int j = 0; j < D; ++j) bit_c[j] = r0.bit[j];
    }
    uint64_t t0 = 0;
    for (; t0 + D <= T; t0 += D) {  // full blocks
      IND_STAMP(4);
      load_records(rf, t0 + 2 * D);
      IND_STAMP(0);
      block_indices(nxt, rn);
      IND_STAMP(1);
      uint32_t e_new[D];
      if (cur.shaped && small) {  // one branch per block, not one per bit
#pragma unroll
        for (int j = 0; j < D; ++j) do_bit(cur, j, bit_c[j], e_new[j], std::true_type{}, std::true_type{});
      } else if (cur.shaped) {
#pragma unroll
        for (int j = 0; j < D; ++j) do_bit(cur, j, bit_c[j], e_new[j], std::true_type{}, std::false_type{});
      } else {
#pragma unroll
        for (int j = 0; j < D; ++j) do_bit(cur, j, bit_c[j], e_new[j], std::false_type{}, std::false_type{});
      }
      IND_STAMP(2);
      if (t0 + D < T) {
        if (!LEARN || (cur.shaped && nxt.clean)) {
          // the entries were asked for before this block's stores: they are there once at most those stores
          // are outstanding
          if (kStoresPerBlock > 0) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kStoresPerBlock) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          pin_entries(nxt);
          if (LEARN) {
#pragma unroll
            for (int j = 0; j < D; ++j) nxt.e[j] = nxt.idx[j] == cur.idx[j] ? e_new[j] : nxt.e[j];
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < D; ++j) ind_ld8(nxt.e[j], tab + 2ull * nxt.idx[j]);  // behind the stores: current
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          pin_entries(nxt);
        }
        pin_records(rf);
      }
      IND_STAMP(3);
      cur = nxt;
#pragma unroll
      for (int j = 0; j < D; ++j) bit_c[j] = rn.bit[j];
      rn = rf;
    }
    if (t0 < T) {  // the last, partial block
      uint32_t e_new;
      cur.shaped = false;  // its records past T are clamped copies: entries repeat
#pragma unroll
      for (int j = 0; j < D; ++j)
        if (t0 + j < T) do_bit(cur, j, bit_c[j], e_new, std::false_type{}, std::false_type{});
    }
#ifdef GMX_IND_PROF
    if (prof_on)
      for (int i = 0; i < 8; ++i) atomicAdd(&gmx_ind_prof[i], prof_acc[i]);
#endif
  }  // if (on)
  if (on) slots[2 * ml + half] = val;
  __syncthreads();
  if (LEARN)
    for (int i = threadIdx.x; i < K * 512; i += 128) gp[i] = ptab[i];
}

#ifdef GMX_IND_PROF
extern "C" int gmx_indirect_prof_read(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_ind_prof), sizeof(unsigned long long) * 8);
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(gmx_ind_prof), z, sizeof z);
  }
  return (int)e;
}
#endif

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
  hipLaunchKernelGGL((gmx_indirect_kernel<OUT, MX, LEARN>), dim3(n_streams), dim3(128), lds_bytes, stream, dv, *args);
  return hipGetLastError();
}

extern "C" hipError_t gmx_launch_indirect_kernel(const GmxIndDev* dv, const GmxIndRunArgs* args, int n_streams,
                                                 unsigned lds_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (args->mx_pred && (args->mx_mask_words < 1 || args->mx_mask_words > 8)) return hipErrorInvalidValue;
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
        const uint32_t chain_word = vc->chain_word;
        if (chain_word) {
          // ---- hand the predictions to the mixers' session and ring it (see GmxIndMbCmd) -----------
          GmxMbCmd* const mmc = (GmxMbCmd*)vc->chain_mc;
          volatile GmxMbPayload* const mp = &mmc->slot[vc->chain_slot & 1u];
          uint32_t* const mwl = (uint32_t*)(rmn + 512);  // four mask words, composed in LDS (one wave: in order)
          if (lane < 4) mwl[lane] = 0;
          if (on) {
            mp->pred[d.slot_a] = va;
            mp->pred[d.slot_b] = vb;
            if ((act_a >> lane) & 1ull) atomicOr(&mwl[d.slot_a >> 5], 1u << (d.slot_a & 31));
            if ((act_b >> lane) & 1ull) atomicOr(&mwl[d.slot_b >> 5], 1u << (d.slot_b & 31));
          }
          if (lane < 4) mp->mask[lane] = mp->mask[lane] | mwl[lane];  // the host left these bits clear
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          if (lane == 0) __hip_atomic_store(&mmc->cmd_seq, chain_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

// ---------------------------------------------------------------------------------------
// Lock step (gmx_chainstep.inc): ONE bit of every stream per launch, the coded bit of a forward known only a
// launch later -- S decoders side by side (coder/decoder.cpp:19-39).  One wave per stream, lane = model.  What a
// stream asks of a step is in what[s]: bit 0 = Indirect::Learn (indirect.cpp:48-69) with bits[s] on the entry the
// stream's last forward latched, bit 1 = Indirect::Predict (indirect.cpp:28-46) on this step's contexts; 0 = the
// stream sits the step out.  Between launches a model keeps {table index, entry, "a forward waits for its learn"} in
// `latch`; logits and slot values are read and written in the bank itself (two logits per model and bit: no LDS copy
// of the 82 KiB of tables for one bit's work).  Predictions go straight into the mixers' records of the same step.
// WITH_LSTM: the LSTM's bit prediction of the same step (gmx_step_dev.h) first, in the same launch -- a launch of its
// own cost the step 4.5 us for 0.5 us of work; lstm_prediction_context reaches the model that reads it through LDS.
template <bool WITH_LSTM>
__global__ void __launch_bounds__(64)
gmx_indirect_step_kernel(const GmxIndDev* __restrict__ dv, const GmxIndStepArgs a, const GmxLstmDev* __restrict__ ldv,
                         const GmxLstmBitArgs la) {
  __shared__ uint64_t s_tab[32];
  __shared__ uint8_t nsn[512], rmn[512];
  __shared__ uint32_t mwl[8];
  __shared__ __attribute__((aligned(16))) float pr[WITH_LSTM ? GMX_L_NO : 4];
  const int lane = threadIdx.x;
  const int s = blockIdx.x;
  // (the step's first launch: the stream's control words and records come in from pinned host memory with it -- a copy
  // kernel in front cost the step 6 us; a.what and everything else below point into the device copy it fills)
  if (a.up.n > 0) gmx_step_upload(a.up, s, lane);
  // (what the stream asks is tested only where it is first needed: everything up to there is requested together with it)
  const uint32_t what = a.what[s];
  const int K = dv->k;
  if (lane < 32) s_tab[lane] = gmx_exp2f_tab[lane];
  for (int i = lane; i < 512; i += 64) {
    nsn[i] = dv->ns_next[i];
    rmn[i] = dv->rm_next[i];
  }
  if (lane < 8) mwl[lane] = 0;
  // The LSTM's part begins: its byte distribution into LDS, and lstm_prediction_context -- a constant of the byte, so
  // the model that reads it has it now and ALL of the Indirect models' loads below can be under way while lane 0 walks
  // the distribution (the walk used to come first and everything else after it: two round trips more in the chain).
  uint32_t lstm_ctx = 0;
  if (WITH_LSTM) lstm_ctx = gmx_lstm_bitstep_begin(ldv, la, s, pr, lane);
  const bool on = lane < K;
  uint8_t* const bank = a.banks + (uint64_t)s * dv->bank_bytes;
  uint32_t* const L = a.latch + ((uint64_t)s * 64 + lane) * 4;  // {index, entry, have, -}
  GmxIndModelDev d = dv->m[on ? lane : 0];
  // Three dependent trips to memory instead of the seven of learn-then-predict written out in order: (1) everything
  // that depends on nothing -- the latch, the step's context, the slots; (2) the learn's two logits AND the predict's
  // table entry; (3) the predict's two logits.  What the learn writes and the predict would read again -- the entry,
  // when the index repeats; a logit, when the state does -- is handed over in registers; the stores go out last.
  uint16_t* const tab = (uint16_t*)(bank + d.tab_off);
  float* const nsp = (float*)(bank + dv->pred_off) + (size_t)(on ? lane : 0) * 512;
  float* const rmp = nsp + 256;
  float* const slots = (float*)(bank + dv->slots_off);
  uint32_t idx_old = 0, e_old = 0, have = 0, ctx = 0;
  float va = 0.f, vb = 0.f;
  const int bit = a.bits[s] ? 1 : 0;
  const uint32_t bcu = a.bc[s];
  if (on) {
    idx_old = L[0];
    e_old = L[1];
    have = L[2];
    ctx = a.ctx[(uint64_t)s * K + lane];
    va = slots[2 * lane];
    vb = slots[2 * lane + 1];
  }
  if (!(what & 3u)) return;  // the stream sits the step out
  const bool do_pred = (what & 2u) != 0;
  const bool with_lstm = WITH_LSTM && do_pred;
  if (with_lstm && lane == la.ind_ctx_col) ctx = lstm_ctx;
  const bool do_learn = on && (what & 1u) && have;
  // ---- trip 2
  const uint32_t ns_o = e_old & 255u, rm_o = e_old >> 8;
  const uint32_t sn_o = ns_o != 255u ? ns_o : 0u;  // the uninitialised state learns as state 0
  float pa = 0.f, pb = 0.f;
  if (do_learn) {
    pa = nsp[sn_o];
    pb = rmp[rm_o];
  }
  uint32_t idx = idx_old, e = e_old;
  if (on && do_pred) {
    idx = ((ctx << 8) + bcu) % d.size;  // indirect.cpp:31-32, 32-bit wrap
    e = tab[idx];
  }
  __syncthreads();  // (the tables and the distribution are in LDS)
  if (with_lstm) {  // ... while those are on their way: lane 0's walk down the distribution
    uint32_t act;
    gmx_lstm_bitstep_walk(ldv, la, s, what, lstm_ctx, pr, lane, /*mask_to_global=*/false, act);
    if (lane == 0 && act) atomicOr(&mwl[la.slot >> 5], 1u << (la.slot & 31));  // (the host left the slot's bit clear)
  }
  if (on) {
    float na = 0.f, nb = 0.f;
    uint32_t e_upd = 0;
    if (do_learn) {  // Indirect::Learn (indirect.cpp:48-69)
      na = pa + ((float)bit - gmx_logistic_tab(pa, s_tab)) * d.lr;
      nb = pb + ((float)bit - gmx_logistic_tab(pb, s_tab)) * d.lr;
      e_upd = (uint32_t)nsn[2 * sn_o + bit] | ((uint32_t)rmn[2 * rm_o + bit] << 8);
      if (do_pred && idx == idx_old) e = e_upd;
      have = 0;
    }
    if (do_pred) {  // Indirect::Predict (indirect.cpp:28-46)
      const uint32_t ns = e & 255u, rm = e >> 8;
      const bool seen_a = ns != 255u, seen_b = rm != 0u;  // never-seen states leave the slot alone
      // ---- trip 3
      float qa = 0.f, qb = 0.f;
      if (seen_a) qa = (do_learn && ns == sn_o) ? na : nsp[ns];
      if (seen_b) qb = (do_learn && rm == rm_o) ? nb : rmp[rm];
      if (seen_a) slots[2 * lane] = va = qa;
      if (seen_b) slots[2 * lane + 1] = vb = qb;
      const bool act_a = seen_a && qa != 0.f, act_b = seen_b && qb != 0.f;  // a zero logit is stored, not active
      if (act_a) atomicOr(&mwl[d.slot_a >> 5], 1u << (d.slot_a & 31));
      if (act_b) atomicOr(&mwl[d.slot_b >> 5], 1u << (d.slot_b & 31));
      have = 1;
      if (a.mx_pred) {
        float* const mp = a.mx_pred + (uint64_t)s * a.mx_n_pad;
        mp[d.slot_a] = va;
        mp[d.slot_b] = vb;
      }
      if (a.pred_out) {
        a.pred_out[((uint64_t)s * K + lane) * 2] = va;
        a.pred_out[((uint64_t)s * K + lane) * 2 + 1] = vb;
        a.act_out[((uint64_t)s * K + lane) * 2] = (uint8_t)act_a;
        a.act_out[((uint64_t)s * K + lane) * 2 + 1] = (uint8_t)act_b;
      }
    }
    if (do_learn) {
      nsp[sn_o] = na;
      rmp[rm_o] = nb;
      tab[idx_old] = (uint16_t)e_upd;
    }
    L[0] = idx;
    L[1] = e;
    L[2] = have;
  }
  __syncthreads();
  // the mask words of the mixers' record: the host left the models' bits clear
  if ((what & 2u) && a.mx_mask && lane < a.mx_mask_words && lane < 8)
    a.mx_mask[(uint64_t)s * a.mx_mask_words + lane] |= mwl[lane];
}

extern "C" hipError_t gmx_launch_indirect_step(const GmxIndDev* dv, const GmxIndStepArgs* args, int n_streams,
                                               hipStream_t stream) {
  (void)hipGetLastError();
  GmxLstmBitArgs none;
  memset(&none, 0, sizeof none);
  hipLaunchKernelGGL(gmx_indirect_step_kernel<false>, dim3(n_streams), dim3(64), 0, stream, dv, *args,
                     (const GmxLstmDev*)nullptr, none);
  return hipGetLastError();
}
// ... with the LSTM's bit prediction of the same step in front
extern "C" hipError_t gmx_launch_models_step(const GmxIndDev* dv, const GmxIndStepArgs* args, const GmxLstmDev* ldv,
                                             const GmxLstmBitArgs* largs, int n_streams, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_indirect_step_kernel<true>, dim3(n_streams), dim3(64), 0, stream, dv, *args, ldv, *largs);
  return hipGetLastError();
}
