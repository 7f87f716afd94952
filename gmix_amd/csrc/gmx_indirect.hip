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
  const GmxIndModelDev d = dv->m[on ? lane : 0];
  uint16_t* const tab = (uint16_t*)(bank + d.tab_off);
  float* const slots = (float*)(bank + dv->slots_off);
  float va = on ? slots[2 * lane] : 0.f, vb = on ? slots[2 * lane + 1] : 0.f;
  float* const nsp = ptab + (size_t)lane * 512;
  float* const rmp = nsp + 256;

  const uint64_t RS = a.rec_stride;
  const uint32_t* const ctx_s = a.ctx + (uint64_t)rec * RS * K;
  const uint32_t* const bc_s = a.bc + (uint64_t)rec * RS;
  const uint8_t* const bits_s = a.bits + (uint64_t)rec * RS;
  float* const po = a.pred_out ? a.pred_out + (uint64_t)rec * RS * 2 * K : nullptr;
  uint8_t* const ao = a.act_out ? a.act_out + (uint64_t)rec * RS * 2 * K : nullptr;
  float* const mxp = a.mx_pred ? a.mx_pred + (uint64_t)rec * a.mx_rec_stride * a.mx_n_pad : nullptr;
  uint32_t* const mxm = a.mx_pred ? a.mx_mask + (uint64_t)rec * a.mx_rec_stride * a.mx_mask_words : nullptr;
  uint8_t* const mxb = a.mx_pred ? a.mx_bits + (uint64_t)rec * a.mx_rec_stride : nullptr;
  const int MW = a.mx_mask_words;
  // bits of the attached mask that belong to the Indirect models (cleared and rewritten per bit)
  uint32_t own = 0;
  if (mxp) {
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

  // entry of bit 0
  uint32_t idx = on ? (uint32_t)(((ctx_s[lane] << 8) + bc_s[0]) % d.size) : 0u;  // indirect.cpp:31-32
  uint32_t e = on ? tab[idx] : 0x00ffu;
  for (uint64_t t = 0; t < T; ++t) {
    // fetch the entry of bit t+1 (its index does not depend on this bit's outcome)
    uint32_t idx_n = 0, e_n = 0x00ffu;
    if (on && t + 1 < T) {
      idx_n = (uint32_t)(((ctx_s[(t + 1) * K + lane] << 8) + bc_s[t + 1]) % d.size);
      e_n = tab[idx_n];
    }
    const int bit = bits_s[t];
    // ---- Indirect::Predict (indirect.cpp:28-46) --------------------------------------------
    const uint32_t ns = e & 255u, rm = e >> 8;
    bool act_a = false, act_b = false;
    if (on) {
      if (ns != 255u) {
        va = nsp[ns];
        act_a = va != 0.f;  // SetLogitPrediction: a zero logit is stored, not active
      }
      if (rm != 0u) {
        vb = rmp[rm];
        act_b = vb != 0.f;
      }
      if (po) {
        *(float2*)(po + t * 2 * K + 2 * lane) = make_float2(va, vb);
        *(uchar2*)(ao + t * 2 * K + 2 * lane) = make_uchar2(act_a, act_b);
      }
      if (mxp) {
        mxp[t * a.mx_n_pad + d.slot_a] = va;
        mxp[t * a.mx_n_pad + d.slot_b] = vb;
      }
    }
    if (mxp) {
      if (lane < MW) mw[lane] = 0;
      __syncthreads();
      if (act_a) atomicOr(&mw[d.slot_a >> 5], 1u << (d.slot_a & 31));
      if (act_b) atomicOr(&mw[d.slot_b >> 5], 1u << (d.slot_b & 31));
      __syncthreads();
      if (lane < MW) {
        uint32_t* w = mxm + t * MW + lane;
        *w = (*w & ~own) | mw[lane];
      }
      if (lane == 0) mxb[t] = (uint8_t)bit;
      __syncthreads();
    }
    // ---- Indirect::Learn (indirect.cpp:48-69) ----------------------------------------------
    if (a.learn && on) {
      const uint32_t sn = ns == 255u ? 0u : ns;  // the uninitialised state learns as state 0
      const float pa = nsp[sn];
      nsp[sn] = pa + ((float)bit - gmx_logistic_tab(pa, s_tab)) * d.lr;
      const float pb = rmp[rm];
      rmp[rm] = pb + ((float)bit - gmx_logistic_tab(pb, s_tab)) * d.lr;
      const uint32_t e_new = (uint32_t)nsn[2 * sn + bit] | ((uint32_t)rmn[2 * rm + bit] << 8);
      tab[idx] = (uint16_t)e_new;
      if (idx_n == idx && t + 1 < T) e_n = e_new;  // same entry twice in a row: take it from here
    }
    idx = idx_n;
    e = e_n;
  }
  if (on) {
    slots[2 * lane] = va;
    slots[2 * lane + 1] = vb;
  }
  __syncthreads();
  if (a.learn)
    for (int i = lane; i < K * 512; i += 64) gp[i] = ptab[i];
}

extern "C" hipError_t gmx_indirect_kernel_set_lds(unsigned lds_bytes) {
  return hipFuncSetAttribute((const void*)gmx_indirect_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds_bytes);
}

extern "C" hipError_t gmx_launch_indirect_kernel(const GmxIndDev* dv, const GmxIndRunArgs* args, int n_streams,
                                                 unsigned lds_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_indirect_kernel, dim3(n_streams), dim3(64), lds_bytes, stream, dv, *args);
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
