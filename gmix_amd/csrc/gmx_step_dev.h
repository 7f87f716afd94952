// gmx_step_dev.h -- device code shared by the lock-step kernels of gmx_lstm.hip and gmx_indirect.hip
// (gmx_chainstep.inc): the LSTM's prediction of one bit, as a function a block of 64 threads calls.
#ifndef GMX_STEP_DEV_H_
#define GMX_STEP_DEV_H_

#include "gmx_internal.h"
#include "gmx_math.h"

// The head of a step (GmxStepUpload): block s brings stream s's slices of the step's pinned host block into the device
// copy -- every load in flight before the first store, so the block pays the link's latency once -- and makes them
// visible to its own later loads.  Called by all 64 threads of a one-wave block.
__device__ __forceinline__ void gmx_step_upload(const GmxStepUpload& u, int s, int lane) {
  uint32_t total = 0;  // in units: 4 bytes, or 1 byte for the byte arrays
  for (int e = 0; e < u.n; ++e) total += (u.bps[e] & 3u) ? u.bps[e] : u.bps[e] >> 2;
  for (uint32_t base = 0; base < total; base += 256) {
    uint32_t v[4], at[4];
    bool word[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t i = base + (uint32_t)lane + 64u * (uint32_t)r;
      at[r] = 0xffffffffu;
      word[r] = true;
      v[r] = 0;
      if (i < total) {
        for (int e = 0; e < u.n; ++e) {
          const bool w = !(u.bps[e] & 3u);
          const uint32_t n = w ? u.bps[e] >> 2 : u.bps[e];
          if (i < n) {
            at[r] = u.off[e] + (uint32_t)s * u.bps[e] + (w ? 4u * i : i);
            word[r] = w;
            break;
          }
          i -= n;
        }
        if (word[r])
          v[r] = *(const uint32_t*)(u.src + at[r]);
        else
          v[r] = u.src[at[r]];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (at[r] == 0xffffffffu) continue;
      if (word[r])
        *(uint32_t*)(u.dst + at[r]) = v[r];
      else
        u.dst[at[r]] = (uint8_t)v[r];
    }
  }
  __threadfence();  // the stores are in L2 and this CU's L1 holds nothing older before anybody reads them back
  __syncthreads();
}

// LstmModel::Predict's walk down the byte distribution for ONE bit of stream s (lstm-model.cpp:34-48; see
// gmx_lstm_bitstep_kernel), in two parts so that a caller can put its own loads in between.  Every thread of the (one-wave)
// block calls both (`pr`: 256 floats of LDS).  gmx_lstm_bitstep_begin copies the byte distribution into LDS and returns
// lstm_prediction_context (a constant of the byte: Lstm::Predict stored it when the byte opened); gmx_lstm_bitstep_walk
// is lane 0's: the range, the two ordered sums, the prediction into the mixers' record.  mask_bit_out = whether
// SetPrediction marked the slot active.  mask_to_global: set / clear the slot's bit in the mixers' mask record here (else
// the caller does).
__device__ __forceinline__ uint32_t gmx_lstm_bitstep_begin(const GmxLstmDev* __restrict__ dvp, const GmxLstmBitArgs& a, int s,
                                                           float* pr, int lane) {
  float* const B = a.banks + (uint64_t)s * dvp->bank_floats;
  const uint32_t* const scal = (const uint32_t*)(B + dvp->scal);
  const float4 q = ((const float4*)(B + dvp->probs))[lane];
  const uint32_t c = scal[4];
  ((float4*)pr)[lane] = q;
  return c;
}
__device__ __forceinline__ void gmx_lstm_bitstep_walk(const GmxLstmDev* __restrict__ dvp, const GmxLstmBitArgs& a, int s,
                                                      uint32_t what, uint32_t c, const float* pr, int lane,
                                                      bool mask_to_global, uint32_t& mask_bit_out) {
  float* const B = a.banks + (uint64_t)s * dvp->bank_floats;
  uint32_t* const scal = (uint32_t*)(B + dvp->scal);
  mask_bit_out = 0;
  if (lane != 0) return;
  int top, bot;
  if (what & 4u) {
    top = 255;
    bot = 0;
  } else {
    top = (int)scal[8];
    bot = (int)scal[9];
    const int mid_before = bot + ((top - bot) / 2);
    if (a.bits[s])
      bot = mid_before + 1;
    else
      top = mid_before;
  }
  const int mid = bot + ((top - bot) / 2);
  // std::accumulate from the first element up, over the upper half, then on over the lower half.  The halves are
  // powers of two long and aligned to their length: sixteen values are fetched at a time, added one after the other
  // (the adds wait for their batch, not each for its own read).
  auto ordered = [&](float acc, int lo, int n) {
    int i = lo;
    for (; n >= 16; n -= 16, i += 16) {
      const float4 q0 = *(const float4*)(pr + i), q1 = *(const float4*)(pr + i + 4), q2 = *(const float4*)(pr + i + 8),
                   q3 = *(const float4*)(pr + i + 12);
      acc += q0.x; acc += q0.y; acc += q0.z; acc += q0.w;
      acc += q1.x; acc += q1.y; acc += q1.z; acc += q1.w;
      acc += q2.x; acc += q2.y; acc += q2.z; acc += q2.w;
      acc += q3.x; acc += q3.y; acc += q3.z; acc += q3.w;
    }
    for (; n >= 4; n -= 4, i += 4) {
      const float4 q4 = *(const float4*)(pr + i);
      acc += q4.x; acc += q4.y; acc += q4.z; acc += q4.w;
    }
    for (; n > 0; --n, ++i) acc += pr[i];
    return acc;
  };
  const float num = ordered(0.0f, mid + 1, top - mid);
  const float denom = ordered(num, bot, mid - bot + 1);
  float prediction = __uint_as_float(scal[5]);
  bool active = false;
  if (denom != 0.0f) {  // (a silent bit leaves the slot as it was)
    const float p = num / denom;
    prediction = gmx_logit(p);
    active = p != 0.5f;
    scal[5] = __float_as_uint(prediction);
  }
  scal[8] = (uint32_t)top;
  scal[9] = (uint32_t)bot;
  a.mx_pred[(uint64_t)s * a.mx_n_pad + a.slot] = prediction;
  if (mask_to_global) {
    uint32_t* const w = a.mx_mask + (uint64_t)s * a.mx_mask_words + (a.slot >> 5);
    const uint32_t m = 1u << (a.slot & 31);
    *w = active ? (*w | m) : (*w & ~m);
  }
  if (a.mixer_ctx_col >= 0) a.mx_ctx[(uint64_t)s * a.mx_m + a.mixer_ctx_col] = c;
  if (a.ind_ctx && a.ind_ctx_col >= 0) a.ind_ctx[(uint64_t)s * a.ind_k + a.ind_ctx_col] = c;
  mask_bit_out = active ? 1u : 0u;
}
// Both, one after the other (gmx_lstm_bitstep_kernel).
__device__ __forceinline__ void gmx_lstm_bitstep_body(const GmxLstmDev* __restrict__ dvp, const GmxLstmBitArgs& a, int s,
                                                      uint32_t what, float* pr, int lane, bool mask_to_global,
                                                      uint32_t& ctx_out, uint32_t& mask_bit_out) {
  ctx_out = gmx_lstm_bitstep_begin(dvp, a, s, pr, lane);
  __syncthreads();
  gmx_lstm_bitstep_walk(dvp, a, s, what, ctx_out, pr, lane, mask_to_global, mask_bit_out);
}

#endif  // GMX_STEP_DEV_H_
