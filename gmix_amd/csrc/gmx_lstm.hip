// gmx_lstm.hip -- the reference's LSTM byte model (models/lstm-model.cpp, lstm.cpp,
// lstm-layer.cpp) for many streams: one 256-thread workgroup per stream, bytes in sequence.
//
// Every float operation of the reference is performed once, in the reference's order: the dot
// products are serial chains (mul, then add), expression sums run from the last element down
// and plain valarray sums from the first up (libstdc++), layer norm / activations / Adam are the
// same elementwise formulas; tanhf, expf, logf are gmx_math.h's restatements of the glibc
// routines the reference calls.  Parallelism comes from what the reference leaves independent:
//   * the 3 gates x 50 cells of a forward pass: one wave per gate, one lane per cell, the gate
//     weights stored input-major ([input][cell]) so that the 307-step chains of a wave read
//     coalesced rows;
//   * the 256 outputs of the output layer (51-step chains, one thread each) and of its SGD step;
//   * in the backward pass the 50 hidden-error chains (256 steps each), the three gates, and the
//     563 x 50 independent accumulators / Adam updates of each gate;
//   * the 8 bit predictions of a byte (their ranges follow from the byte), one lane each.
// The chains are latency-bound if their operands are fetched as they go (hipcc waits for every
// small batch of loads before the adds that use it: 58 us per forward pass), so every chain first
// issues the loads of a long stretch (up to 104 weights) and then runs its adds on registers.
// At thousands of streams the kernel is HBM-bound, and half of its traffic used to be the gradient
// accumulators (NeuronLayer::update_, read and written once per epoch of a backward pass).  They
// are now never stored: the epochs only record each gate's final error vector, and after the last
// epoch every accumulator is formed in a register -- update[cell][256+r] = sum over the epochs, in
// the reference's order, of error[epoch][cell] * input[epoch][r], the layer inputs being kept
// input-major for this -- and consumed by Adam on the spot.  The one-hot symbol columns, of which
// an epoch touches one row, stay in HBM.  oracle/gmx_oracle_lstm.c is the line-by-line
// specification.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gmx_internal.h"
#include "gmx_math.h"
#include "gmx_step_dev.h"

// Loads a chain issues before its adds (GMX_LSTM_STRETCH weights, GMX_LSTM_Q float4s) against
// workgroups per CU (GMX_LSTM_BLOCKS): shorter stretches cost latency per byte, fewer registers
// let more streams run at once.  Measured at 4096 streams: 2 workgroups per CU (stretches of 104)
// 7.9e6 bytes/s, 3 per CU (77) 8.5e6, 4 per CU (52) 8.4e6; the longer stretches win at few streams.
// After the later changes to the backward pass the balance moved: stretches of 154 (two
// load-then-add rounds per gate chain; the whole chain at once, 307, spills) with two workgroups
// per CU give 1.14e7 bytes/s at 4096 streams and 7.9e6 at 256, stretches of 77 with three per CU
// 1.06e7 and 6.65e6.  Both builds exist; GMX_LSTM_BUILD=3 in the environment picks the second.
// (Tried and dropped: the non-temporal hint on the weight and ring-slot accesses -- unlike the mixer
// rows they ARE reused from L2 / the memory-side cache: 5-12% slower; a dense [input][50] pitch for the gate matrices instead of [input][64] -- 22%
// fewer bytes, but rows that no longer start on a cache line: 15-25% slower at every stream count.)
#ifndef GMX_LSTM_STRETCH
#define GMX_LSTM_STRETCH 77
#define GMX_LSTM_Q 16
#define GMX_LSTM_BLOCKS 3
#define GMX_LSTM_ADAM 12
#endif

// Phase profile (build with -DGMX_LSTM_PROF, read with gmx_lstm_prof_read): s_memtime ticks that
// thread 0 of block 0 spends between consecutive stamps, summed per stamp id.
#ifdef GMX_LSTM_PROF
__device__ unsigned long long gmx_lstm_prof[16];
#define STAMP(i)                                                 \
  do {                                                           \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                   \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      gmx_lstm_prof[i] += now_ - tprev_;                         \
      tprev_ = now_;                                             \
    }                                                            \
  } while (0)
#else
#define STAMP(i)
#endif

namespace {

constexpr int NI = GMX_L_NI, NO = GMX_L_NO, NC = GMX_L_NC, H = GMX_L_H, LIN = GMX_L_LIN, LINP = GMX_L_LINP,
              HID = GMX_L_HID, CP = GMX_L_CP;
constexpr float kLearningRate = 0.03f, kClip = 10.0f;
constexpr uint32_t kUpdateLimit = 3000;

constexpr int kTileRows = 32;  // layer-input rows of the deferred accumulation staged at a time

struct Lds {
  alignas(16) float probs[NO];   // LstmModel::probs_ / softmax scratch
  alignas(16) float xt[kTileRows][GMX_L_HP];  // input-major layer inputs of a tile of rows
  float xin[LINP];        // layer input of the epoch at hand
  float hid[CP];          // Lstm::hidden_ (hid[50] = 1)
  float herr[CP];         // Lstm::hidden_error_
  alignas(16) float err[NO];  // output-layer error of an epoch
  float nrm[3][CP];       // per gate: pre-norm sums / products for the ordered reductions
  float act[3][CP];       // per gate: activated state of the forward pass / final errors backward
  float fsum[3][CP];
  float red[8];
  uint32_t ired[4];
  uint64_t exptab[32];
};

// acc + p[0] + p[1] + ... + p[n-1], strictly in that order, n a multiple of 4 up to 4*Q4, p 16-byte
// aligned in LDS: the reads are issued as independent 16-byte loads, the adds follow in order.
template <int Q4>
__device__ __forceinline__ float ordered_sum4(float acc, const float* p, int n) {
  float4 v[Q4];
#pragma unroll
  for (int q = 0; q < Q4; ++q) v[q] = *(const float4*)(p + 4 * (4 * q < n ? q : 0));
#pragma unroll
  for (int q = 0; q < Q4; ++q)
    if (4 * q < n) {
      acc += v[q].x;
      acc += v[q].y;
      acc += v[q].z;
      acc += v[q].w;
    }
  return acc;
}

__device__ __forceinline__ float clipf(float a) { return a < -kClip ? -kClip : (a > kClip ? kClip : a); }

}  // namespace

// SESSION: one persistent block that takes its work from a mailbox instead of record arrays (a decoder
// knows a byte only after its eight predictions, coder/decoder.cpp:19-39): every command is one pass
// through the loop below with the phases it names, the model's state stays in LDS and registers between
// commands and goes back to the bank when the block leaves (GMX_MB_STOP, or idle_ticks without a command).
template <int kStretch, int kBlocks, bool SESSION>
__global__ void __launch_bounds__(256, kBlocks)
gmx_lstm_kernel(const GmxLstmDev* __restrict__ dvp, const GmxLstmRunArgs a) {
  __shared__ Lds L;
  const GmxLstmDev& dv = *dvp;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s = blockIdx.x;  // record stream; its bank:
  // bytes of this stream in the launch (streams of different lengths share one: files end at different bytes)
  const uint64_t NB = (!SESSION && a.n_list) ? a.n_list[s] : a.n_bytes;
  if (!SESSION && NB == 0) return;
  float* const B = a.banks + (uint64_t)(a.stream_base + s) * dv.bank_floats;
  uint32_t* const scal = (uint32_t*)(B + dv.scal);
  uint32_t* const hist = (uint32_t*)(B + dv.input_history);
  float* const out_layer = B + dv.out_layer;
  const float* const ppm_s = a.ppm + (uint64_t)s * a.rec_stride * NI;
  const uint8_t* const bytes_s = a.bytes + (uint64_t)s * a.rec_stride;
  const float* const adam_s = a.adam + (uint64_t)s * a.max_bptt * 4;
  float* const pred_s = a.pred_out + (uint64_t)s * a.rec_stride * 8;
  uint8_t* const act_s = a.act_out + (uint64_t)s * a.rec_stride * 8;
  uint32_t* const ctx_s = a.ctx_out + (uint64_t)s * a.rec_stride;

  if (tid < 32) L.exptab[tid] = gmx_exp2f_tab[tid];
  if (tid < CP) {
    L.hid[tid] = (B + dv.hidden)[tid];
    L.herr[tid] = (B + dv.hidden_error)[tid];
  }
  L.probs[tid] = (B + dv.probs)[tid];
  uint32_t epoch = scal[0], l_epoch = scal[1], update_steps = scal[2], last_byte = scal[3], context = scal[4];
  uint32_t coded = scal[6];
  float prediction = __uint_as_float(scal[5]);
  if (a.last_byte >= 0) last_byte = (uint32_t)a.last_byte;
  uint32_t bptt_done = 0;
  // what a pass through the loop does: fixed for a launch over records, named by each command of a session
  uint32_t phases = a.phases;
  bool learn = a.learn != 0;
  const float* adam_row = adam_s;
  volatile GmxLstmMbCmd* const vc = a.mc;
  volatile GmxLstmMbReply* const vr = a.mb;
  uint32_t seen = 0, word = 0, exit_state = GMX_MB_EXIT_IDLE;
  bool fwd_after = false;  // the Predict half of a GMX_MB_LEARN0_FWD is still to run
  if (SESSION) seen = __hip_atomic_load(&a.mb->done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();

#ifdef GMX_LSTM_PROF
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
  // per-lane copies of what every byte reads and only a backward pass changes (layer-norm gain and
  // bias of this wave's gate), of the cell state, and of the NEXT byte's PPM input: requested by an
  // inline-asm load while the current byte is computed, collected by the s_waitcnt at the end of the
  // same loop iteration (on the other side of the back edge the compiler could copy the register
  // before the data is in)
  float gam = 0.0f, bet = 0.0f, cst = 0.0f, ppm_n = 0.0f;
  if (wave < 3 && lane < NC) {
    gam = (B + dv.gate[wave].gamma)[lane];
    bet = (B + dv.gate[wave].beta)[lane];
  }
  if (tid < NC) cst = (B + dv.state)[tid];
  auto ppm_request = [&](uint64_t nn) {
    const float* p = ppm_s + (nn < NB ? nn : NB - 1) * NI + tid;
    // into an AGPR: the value is in flight from here to ppm_landed(), and an accumulation register is nothing
    // the compiler would copy or park in scratch in between (this kernel spills a few VGPRs)
    asm volatile("global_load_dword %0, %1, off" : "=a"(ppm_n) : "v"(p) : "memory");
  };
  auto ppm_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" : "+a"(ppm_n)); };
  if (!SESSION && (phases & 1u)) {
    ppm_request(0);
    ppm_landed();
  }
  // a session's answer to the command just run (not yet after the Perceive half of a GMX_MB_LEARN0_FWD)
  auto command_done = [&]() {
    if (!SESSION || fwd_after) return;
    if (phases & 1u) {
      vr->probs[tid] = L.probs[tid];
      if (tid == 0) vr->context = context;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    seen = word;
    if (tid == 0) __hip_atomic_store(&a.mb->done_seq, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // The 8 bit predictions of a byte (lstm-model.cpp:34-48) by the first 8 lanes of one wave: lane k sums the two
  // halves of bit k's range of L.probs in order, lane 0 then walks the bits (a silent bit keeps the previous
  // prediction) and stores the records.  Nothing later in the byte needs them, so a launch over whole records lets
  // the FOURTH wave do byte n's while the other three run the gate chains of byte n+1 (L.probs is not rewritten
  // before that byte's softmax): 2.5 of a byte's 26 us leave the chain of dependent phases.
  auto bit_predictions = [&](uint32_t byte_b, uint64_t n_b, uint32_t context_b) {
    float logit = 0.0f, state = 0.0f;
    if (lane < 8) {
      const int k = lane;
      const int size = 256 >> k, half = size >> 1;
      const int bot = k == 0 ? 0 : (int)((byte_b >> (8 - k)) << (8 - k));
      const int mid = bot + half - 1, top = bot + size - 1;
      // std::accumulate from 0.0f over the upper half, then on over the lower half
      float num = 0.0f, denom;
      if (half >= 4) {
#pragma unroll 1
        for (int o = 0; o < half; o += 64) num = ordered_sum4<16>(num, L.probs + mid + 1 + o, half - o < 64 ? half - o : 64);
        denom = num;
#pragma unroll 1
        for (int o = 0; o < half; o += 64) denom = ordered_sum4<16>(denom, L.probs + bot + o, half - o < 64 ? half - o : 64);
      } else {
        for (int i = mid + 1; i <= top; ++i) num += L.probs[i];
        denom = num;
        for (int i = bot; i <= mid; ++i) denom += L.probs[i];
      }
      // SetPrediction (short-term-memory.cpp:187-191); a silent bit (denom == 0) keeps the slot
      const float p = num / denom;
      logit = gmx_logit(p);
      state = denom != 0.0f ? (p == 0.5f ? 1.0f : 2.0f) : 0.0f;  // 0 silent, 1 inactive, 2 active
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // in bit order: a silent bit leaves the previous prediction
      const float st = __shfl(state, k), lg = __shfl(logit, k);
      if (lane == 0) {
        if (st != 0.0f) prediction = lg;
        pred_s[n_b * 8 + k] = prediction;
        act_s[n_b * 8 + k] = st == 2.0f ? 1 : 0;
      }
    }
    if (lane == 0) ctx_s[n_b] = context_b;
  };
  const bool defer_bits = !SESSION && a.phases == 7u;  // whole records: wave 3 takes the bits a byte late
  bool bits_owed = false;
  uint32_t owed_byte = 0, owed_context = 0;
  uint64_t owed_n = 0;
  for (uint64_t n = 0; SESSION || n < NB; ++n) {
    uint32_t byte = 0;
    if (SESSION) {
      if (fwd_after) {  // second half of GMX_MB_LEARN0_FWD
        fwd_after = false;
        phases = 1u;
        learn = false;
      } else {
        if (tid == 0) {
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          uint32_t w = seen;
          for (;;) {
            w = __hip_atomic_load(&a.mc->cmd_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (w != seen) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.idle_ticks) break;
            __builtin_amdgcn_s_sleep(1);
          }
          L.ired[2] = w;
        }
        __syncthreads();
        word = L.ired[2];
        __syncthreads();
        if (word == seen) break;  // idle: leave (the state goes back to the bank below)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const uint32_t cmd = word & GMX_MB_CMD_MASK;
        if (cmd == GMX_MB_STOP) {
          exit_state = GMX_MB_EXIT_STOP;
          seen = word;
          break;
        }
        fwd_after = cmd == GMX_MB_LEARN0_FWD;
        phases = cmd == GMX_MB_FORWARD ? 1u : 4u;
        learn = cmd != GMX_MB_FORWARD;
      }
      byte = vc->byte;
      adam_row = (const float*)a.mc->adam;
      bptt_done = 0;
      if (phases & 1u) {
        last_byte = vc->last_byte;
        ppm_n = vc->ppm[tid];
      }
    } else {
      byte = bytes_s[n];
    }
    STAMP(0);
    bool sgd_done = false;
    // ======================= Lstm::Predict (lstm.cpp:95-123) ================================
    if (phases & 1u) {
      const uint32_t e = epoch, le = l_epoch;
      float* const lin = B + dv.layer_input + (uint64_t)e * LINP;
      // SetInput + the recurrent part of the layer input (lstm.cpp:45-50, :98-100)
      L.xin[tid] = ppm_n;
      if (!SESSION) ppm_request(n + 1);
      if (tid < NC) L.xin[NI + tid] = L.hid[tid];
      if (tid == 0) L.xin[LIN - 1] = 1.0f;
      if (tid < NC) (B + dv.last_state + (uint64_t)le * CP)[tid] = cst;  // lstm-layer.cpp:200
      __syncthreads();
      for (int j = tid; j < LIN; j += 256) {
        lin[j] = L.xin[j];
        (B + dv.lin_t)[(uint64_t)j * GMX_L_HP + e] = L.xin[j];
      }
      STAMP(1);  // inputs
      // LstmLayer::ForwardPass(NeuronLayer&) (lstm-layer.cpp:221-241): wave = gate, lane = cell
      if (wave < 3 && lane < NC) {
        const GmxLstmGateOff& g = dv.gate[wave];
        const float* w = B + g.weights;
        float f = w[(uint64_t)last_byte * CP + lane];
        // blocks of four layer inputs, 16 bytes per lane and load; a stretch = kB blocks in flight
        constexpr int kB = (kStretch + 3) / 4;
        const float* wl = w + gmx_l_mat(NO, lane);
#pragma unroll 1
        for (int q0 = 0; q0 < GMX_L_LINB; q0 += kB) {
          float4 wq[kB];
#pragma unroll
          for (int u = 0; u < kB; ++u)
            wq[u] = *(const float4*)(wl + (uint64_t)(q0 + u < GMX_L_LINB ? q0 + u : GMX_L_LINB - 1) * (CP * 4));
#pragma unroll
          for (int u = 0; u < kB; ++u) {
            const int j = 4 * (q0 + u);
            if (j + 0 < LIN) f += L.xin[j + 0] * wq[u].x;
            if (j + 1 < LIN) f += L.xin[j + 1] * wq[u].y;
            if (j + 2 < LIN) f += L.xin[j + 2] * wq[u].z;
            if (j + 3 < LIN) f += L.xin[j + 3] * wq[u].w;
          }
        }
        L.nrm[wave][lane] = f;
      } else if (wave == 3 && bits_owed) {
        bit_predictions(owed_byte, owed_n, owed_context);
      }
      bits_owed = false;
      __syncthreads();
      STAMP(2);  // gate chains
      if (wave < 3 && lane == 0) {
        float sq = L.nrm[wave][NC - 1] * L.nrm[wave][NC - 1];  // expression .sum(): last element first
#pragma unroll 1
        for (int i0 = NC - 1 - 7; i0 >= 0; i0 -= 7) {  // the other 49, seven LDS reads at a time
          float nv[7];
#pragma unroll
          for (int u = 0; u < 7; ++u) nv[u] = L.nrm[wave][i0 + u];
#pragma unroll
          for (int u = 6; u >= 0; --u) sq += nv[u] * nv[u];
        }
        const float iv = 1.0f / sqrtf((sq / (float)NC) + 1e-5f);
        L.red[wave] = iv;
        (B + dv.gate[wave].ivar)[le] = iv;
      }
      __syncthreads();
      if (wave < 3 && lane < NC) {
        const GmxLstmGateOff& g = dv.gate[wave];
        const float nv = L.nrm[wave][lane] * L.red[wave];
        (B + g.norm + (uint64_t)le * CP)[lane] = nv;
        float st = nv * gam + bet;
        st = wave == 1 ? gmx_tanhf(st) : gmx_logistic_tab(st, L.exptab);  // lstm-layer.cpp:205-211
        (B + g.state + (uint64_t)le * CP)[lane] = st;
        L.act[wave][lane] = st;
      }
      __syncthreads();
      if (tid < NC) {  // lstm-layer.cpp:212-217
        const float f = L.act[0][tid];
        const float igs = 1.0f - f;
        float st = cst * f;
        st = st + L.act[1][tid] * igs;
        const float ts = gmx_tanhf(st);
        cst = st;
        (B + dv.input_gate_state + (uint64_t)le * CP)[tid] = igs;
        (B + dv.state)[tid] = st;
        (B + dv.tanh_state + (uint64_t)le * CP)[tid] = ts;
        L.hid[tid] = L.act[2][tid] * ts;
      }
      __syncthreads();
      STAMP(3);  // norm, activations, cell
      // output layer + softmax (lstm.cpp:106-118): thread = output symbol
      const float* ol = out_layer + (uint64_t)e * HID * NO;
      float sum = 0.0f;
      float ov[HID];
#pragma unroll
      for (int j = 0; j < HID; ++j) ov[j] = ol[(uint64_t)j * NO + tid];
#pragma unroll
      for (int j = 0; j < HID; ++j) sum += L.hid[j] * ov[j];
      // max_out = max(0, every sum): order does not matter for a maximum
      float mx = sum > 0.0f ? sum : 0.0f;
      for (int o = 32; o > 0; o >>= 1) {
        const float other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
      }
      if (lane == 0) L.red[4 + wave] = mx;
      __syncthreads();
      mx = L.red[4];
      for (int k = 1; k < 4; ++k) mx = L.red[4 + k] > mx ? L.red[4 + k] : mx;
      L.probs[tid] = gmx_expf_tab(sum - mx, L.exptab);
      __syncthreads();
      STAMP(4);  // output layer, max, expf
      if (tid == 0) {  // valarray::sum(): first element first (0 + p[0] is p[0])
        float t = 0.0f;
#pragma unroll 1
        for (int o = 0; o < NO; o += 64) t = ordered_sum4<16>(t, L.probs + o, 64);
        L.red[0] = t;
      }
      __syncthreads();
      const float p = L.probs[tid] / L.red[0];
      __syncthreads();
      L.probs[tid] = p;
      (B + dv.output + (uint64_t)e * NO)[tid] = p;
      epoch = epoch + 1 == H ? 0 : epoch + 1;
      l_epoch = l_epoch + 1 == H ? 0 : l_epoch + 1;
      // The output layer's own step (lstm.cpp:86-92) belongs to Lstm::Perceive of this byte, but
      // when no backward pass comes first (epoch != 0) nothing it reads changes until then: it
      // starts from the matrix column this thread still holds, so do it now and read the ring
      // slot once instead of twice.  (Only when this launch also perceives, and knows the byte.)
      sgd_done = false;
      if (learn && (phases & 4u) && epoch != 0) {
        const float error = ((uint32_t)tid == byte) ? (p - 1.0f) : p;
        const float lr_e = kLearningRate * error;
        float* dst = out_layer + (uint64_t)epoch * HID * NO;
#pragma unroll
        for (int j = 0; j < HID; ++j) dst[(uint64_t)j * NO + tid] = ov[j] - lr_e * L.hid[j];
        sgd_done = true;
      }
      // lstm_prediction_context: the first symbol with the largest probability (lstm-model.cpp:25-33)
      float pm = p;
      for (int o = 32; o > 0; o >>= 1) {
        const float other = __shfl_xor(pm, o);
        pm = other > pm ? other : pm;
      }
      if (lane == 0) L.red[4 + wave] = pm;
      if (tid == 0) L.ired[0] = 0xffffffffu;
      __syncthreads();
      pm = L.red[4];
      for (int k = 1; k < 4; ++k) pm = L.red[4 + k] > pm ? L.red[4 + k] : pm;
      if (p == pm) atomicMin(&L.ired[0], (uint32_t)tid);
      __syncthreads();
      context = pm > 0.0f ? L.ired[0] : 0u;
    }
    STAMP(5);  // softmax sum, divide, context, early SGD
    // ======================= the 8 bit predictions (lstm-model.cpp:34-48) ====================
    if (phases & 2u) {
      if (defer_bits) {  // by wave 3, beside the next byte's gate chains (or behind the loop)
        bits_owed = true;
        owed_byte = byte;
        owed_n = n;
        owed_context = context;
      } else if (wave == 0) {
        bit_predictions(byte, n, context);
      }
    }
    __syncthreads();
    if (phases & 6u) {  // the byte is known from here on
      last_byte = byte;
      coded = 1;
    }
    if (!learn || !(phases & 4u)) {
      if (!SESSION && (phases & 1u)) ppm_landed();
      __syncthreads();
      command_done();
      continue;
    }
    STAMP(6);  // bits
    // ======================= Lstm::Perceive (lstm.cpp:52-93) ================================
    const uint32_t last_epoch = epoch == 0 ? H - 1 : epoch - 1;
    const uint32_t old_input = hist[last_epoch];
    __syncthreads();
    if (tid == 0) hist[last_epoch] = byte;
    __syncthreads();
    if (epoch == 0) {
      const float alpha = adam_row[bptt_done * 4 + 0], d1 = adam_row[bptt_done * 4 + 1], d2 = adam_row[bptt_done * 4 + 2];
      for (int ep = H - 1; ep >= 0; --ep) {
        STAMP(10);  // (previous epoch: neuron backward, clips)
        // output-layer error of this epoch and its pull on the hidden state (lstm.cpp:61-69)
        {
          const float o = (B + dv.output + (uint64_t)ep * NO)[tid];
          L.err[tid] = ((uint32_t)tid == hist[ep]) ? (o - 1.0f) : o;
          for (int j = tid; j < LIN; j += 256) L.xin[j] = (B + dv.layer_input + (uint64_t)ep * LINP)[j];
        }
        __syncthreads();
        {
          // hidden_error_[j] += sum_i lstm_output_layer[ep][i][j] * error[i], i = 0..255 in order
          // (lstm.cpp:61-69).  Four lanes share a row j of the [hidden][symbol] slab: lane s holds
          // columns 16m + 4s .. + 3, so a load instruction moves whole 64-byte lines and all 16 of
          // a lane are in flight at once (one memory latency per epoch instead of four).  The sum
          // walks the four lanes in column order: acc = quad_rotate(acc) + p executed by all of
          // them, lane s holding the true prefix at step s.
          const int j = tid >> 2, sq = tid & 3;
          const int jr = j < NC ? j : 0;  // threads 200.. shadow row 0 and do not store
          const float* ol = out_layer + ((uint64_t)ep * HID + jr) * NO + 4 * sq;
          float4 ov[GMX_LSTM_Q];
#pragma unroll
          for (int m = 0; m < GMX_LSTM_Q; ++m) ov[m] = *(const float4*)(ol + 16 * m);
          float he = L.herr[jr];  // picked up by lane 0 through the first rotate
#pragma unroll
          for (int m = 0; m < GMX_LSTM_Q; ++m) {
            const float4 ev = *(const float4*)(L.err + 16 * m + 4 * sq);
            const float px = ov[m].x * ev.x, py = ov[m].y * ev.y, pz = ov[m].z * ev.z, pw = ov[m].w * ev.w;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              he = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(he), 0x93, 0xf, 0xf, false)) + px;
              he += py;
              he += pz;
              he += pw;
            }
          }
          __syncthreads();  // every lane has read its herr before the last lanes overwrite it
          if (sq == 3 && j < NC) L.herr[j] = he;
        }
        __syncthreads();
        const uint32_t prev_epoch = ep == 0 ? H - 1 : ep - 1;
        const uint32_t symbol = ep == 0 ? old_input : hist[prev_epoch];
        STAMP(7);  // error + hidden-error chain
        // LstmLayer::BackwardPass (lstm-layer.cpp:252-292)
        if (tid < NC) {
          const int i = tid;
          float stored = (B + dv.stored_error)[i], serr = (B + dv.state_error)[i];
          if (ep == H - 1) {
            stored = L.herr[i];
            serr = 0.0f;
          } else {
            stored += L.herr[i];
          }
          const float ts = (B + dv.tanh_state + (uint64_t)ep * CP)[i];
          const float og = (B + dv.gate[2].state + (uint64_t)ep * CP)[i];
          const float in = (B + dv.gate[1].state + (uint64_t)ep * CP)[i];
          const float fg = (B + dv.gate[0].state + (uint64_t)ep * CP)[i];
          const float igs = (B + dv.input_gate_state + (uint64_t)ep * CP)[i];
          const float ls = (B + dv.last_state + (uint64_t)ep * CP)[i];
          L.act[2][i] = ts * stored * og * (1.0f - og);
          serr += stored * og * (1.0f - (ts * ts));
          L.act[1][i] = serr * igs * (1.0f - (in * in));
          L.act[0][i] = (ls - in) * serr * fg * igs;
          L.herr[i] = 0.0f;
          if (ep > 0) {
            serr *= fg;
            stored = 0.0f;
          }
          (B + dv.stored_error)[i] = stored;
          (B + dv.state_error)[i] = serr;
        }
        if (ep == 0 && update_steps < kUpdateLimit) ++update_steps;
        __syncthreads();
        STAMP(8);  // layer backward
        // LstmLayer::BackwardPass(NeuronLayer&) (lstm-layer.cpp:294-355): wave = gate, lane = cell
        float err = 0.0f, nv = 0.0f;
        if (wave < 3 && lane < NC) {
          const GmxLstmGateOff& g = dv.gate[wave];
          if (ep == H - 1) {
            (B + g.gamma_u)[lane] = 0.0f;
            (B + g.beta_u)[lane] = 0.0f;
            float* up = B + g.update;
            for (int r = 0; r < NO; ++r) up[(uint64_t)r * CP + lane] = 0.0f;  // the symbol columns
            // transpose_ (lstm-layer.cpp:300-304): only a checkpoint ever reads this copy, the
            // chains below take the recurrent weights from the matrix itself (unchanged until Adam)
            for (int j = 0; j < HID; ++j)
              (B + g.transpose)[(uint64_t)j * CP + lane] = (B + g.weights)[gmx_l_mat(NO + NI + j, lane)];
          }
          err = L.act[wave][lane];
          nv = (B + g.norm + (uint64_t)ep * CP)[lane];
          (B + g.beta_u)[lane] += err;
          (B + g.gamma_u)[lane] += err * nv;
          err *= gam * (B + g.ivar)[ep];
          L.nrm[wave][lane] = err * nv;
        }
        __syncthreads();
        if (wave < 3 && lane == 0) {
          float t = L.nrm[wave][NC - 1];  // expression .sum(): last element first
#pragma unroll 1
          for (int i0 = NC - 1 - 7; i0 >= 0; i0 -= 7) {
            float nv[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) nv[u] = L.nrm[wave][i0 + u];
#pragma unroll
            for (int u = 6; u >= 0; --u) t += nv[u];
          }
          L.red[wave] = t / (float)NC;
        }
        __syncthreads();
        if (wave < 3 && lane < NC) {
          err -= L.red[wave] * nv;
          L.act[wave][lane] = err;
        }
        __syncthreads();
        if (wave < 3 && lane < NC) {
          const GmxLstmGateOff& g = dv.gate[wave];
          if (ep > 0) {  // through the recurrent weights (transpose_ is a snapshot of them)
            const float* wr = B + g.weights + gmx_l_mat(NO + NI + lane, 0);  // this row's cells are 4 floats apart
            float f = 0.0f;
            float rv[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) rv[j] = wr[4 * j];
#pragma unroll
            for (int j = 0; j < NC; ++j) f += L.act[wave][j] * rv[j];
            L.fsum[wave][lane] = f;
          }
          (B + g.update)[(uint64_t)symbol * CP + lane] += err;       // lstm-layer.cpp:342
          (B + dv.errs + ((uint64_t)wave * H + ep) * CP)[lane] = err;  // for the deferred accumulation
        }
        __syncthreads();
        if (tid < NC) {  // the three gates add to stored_error_ in their order, then the clips
          float stored = (B + dv.stored_error)[tid];
          if (ep > 0) {
            stored += L.fsum[0][tid];
            stored += L.fsum[1][tid];
            stored += L.fsum[2][tid];
          }
          (B + dv.stored_error)[tid] = clipf(stored);
          (B + dv.state_error)[tid] = clipf((B + dv.state_error)[tid]);
          L.herr[tid] = clipf(L.herr[tid]);
        }
        __syncthreads();
      }
      __syncthreads();
      STAMP(10);
      // update_[cell][256 + r] (lstm-layer.cpp:341) formed in a register -- the epochs' products
      // added in the reference's order, 99 down to 0 -- and handed to Adam (lstm-layer.cpp:12-35)
      // right away; then the symbol columns and the layer-norm parameters.
      {
        const float beta1 = 0.025f, beta2 = 0.9999f, eps = 1e-6f;
        auto adam1 = [&](float gr, float* mp, float* vp, float* wp) {
          float m = *mp * beta1;
          m += (1.0f - beta1) * gr;
          float v = *vp * beta2;
          v += (1.0f - beta2) * gr * gr;
          *mp = m;
          *vp = v;
          *wp -= alpha * ((m / d1) / (sqrtf(v / d2 + eps)));
        };
        const int cl = lane < NC ? lane : 0;  // idle lanes shadow cell 0 and do not store
#pragma unroll 1
        for (int g = 0; g < 3; ++g) {
          const GmxLstmGateOff& go = dv.gate[g];
          float ereg[H];
#pragma unroll
          for (int ep = 0; ep < H; ++ep) ereg[ep] = (B + dv.errs + ((uint64_t)g * H + ep) * CP)[cl];
          STAMP(11);  // (the gate's error vectors into registers)
#pragma unroll 1
          for (int r0 = 0; r0 < LIN; r0 += kTileRows) {
            // stage the tile's rows (one row = one input over the 100 epochs) in LDS, coalesced
            __syncthreads();
            for (int i = tid; i < kTileRows * GMX_L_HP; i += 256) {
              const int rr = r0 + i / GMX_L_HP;
              (&L.xt[0][0])[i] = rr < LIN ? (B + dv.lin_t)[(uint64_t)r0 * GMX_L_HP + i] : 0.0f;
            }
            __syncthreads();
            STAMP(12);  // (tile staged)
            // this wave's rows of the tile: wave, wave + 4, ...; the Adam operands of the next row
            // are requested before the 100-term sum of the current one runs
            constexpr int kMine = kTileRows / 4;
            auto row_ix = [&](int u) {
              const int r = r0 + wave + 4 * u;
              return gmx_l_mat(NO + (r < LIN ? r : LIN - 1), cl);
            };
            float am = (B + go.m)[row_ix(0)], av = (B + go.v)[row_ix(0)], aw = (B + go.weights)[row_ix(0)];
#pragma unroll
            for (int u = 0; u < kMine; ++u) {
              float nm = 0.0f, nvv = 0.0f, nw = 0.0f;
              if (u + 1 < kMine) {
                nm = (B + go.m)[row_ix(u + 1)];
                nvv = (B + go.v)[row_ix(u + 1)];
                nw = (B + go.weights)[row_ix(u + 1)];
              }
              const float4* xr = (const float4*)L.xt[wave + 4 * u];
              float acc = 0.0f;
#pragma unroll
              for (int q = H / 4 - 1; q >= 0; --q) {  // epochs 99 down to 0
                const float4 x = xr[q];
                acc += ereg[4 * q + 3] * x.w;
                acc += ereg[4 * q + 2] * x.z;
                acc += ereg[4 * q + 1] * x.y;
                acc += ereg[4 * q + 0] * x.x;
              }
              const int r = r0 + wave + 4 * u;
              if (r < LIN && lane < NC) {
                const uint64_t ix = gmx_l_mat(NO + r, lane);
                float m = am * beta1;
                m += (1.0f - beta1) * acc;
                float v = av * beta2;
                v += (1.0f - beta2) * acc * acc;
                (B + go.update)[ix] = acc;  // NeuronLayer::update_ is part of the model's checkpoint
                (B + go.m)[ix] = m;
                (B + go.v)[ix] = v;
                (B + go.weights)[ix] = aw - alpha * ((m / d1) / (sqrtf(v / d2 + eps)));
              }
              am = nm;
              av = nvv;
              aw = nw;
            }
            STAMP(13);  // (sums + Adam of the tile's rows)
          }
          if (lane < NC) {  // the symbol columns: 16 rows' operands requested at a time
#pragma unroll 1
            for (int r0 = wave; r0 < NO; r0 += 4 * 16) {
              float ug[16], um[16], uv[16], uw[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) {
                const uint64_t ix = (uint64_t)(r0 + 4 * u) * CP + lane;
                ug[u] = (B + go.update)[ix];
                um[u] = (B + go.m)[ix];
                uv[u] = (B + go.v)[ix];
                uw[u] = (B + go.weights)[ix];
              }
#pragma unroll
              for (int u = 0; u < 16; ++u) {
                const uint64_t ix = (uint64_t)(r0 + 4 * u) * CP + lane;
                float m = um[u] * beta1;
                m += (1.0f - beta1) * ug[u];
                float v = uv[u] * beta2;
                v += (1.0f - beta2) * ug[u] * ug[u];
                (B + go.m)[ix] = m;
                (B + go.v)[ix] = v;
                (B + go.weights)[ix] = uw[u] - alpha * ((m / d1) / (sqrtf(v / d2 + eps)));
              }
            }
          }
          STAMP(14);  // (Adam of the symbol columns)
        }
        if (wave < 3 && lane < NC) {
          const GmxLstmGateOff& go = dv.gate[wave];
          adam1((B + go.gamma_u)[lane], B + go.gamma_m + lane, B + go.gamma_v + lane, B + go.gamma + lane);
          adam1((B + go.beta_u)[lane], B + go.beta_m + lane, B + go.beta_v + lane, B + go.beta + lane);
        }
      }
      __syncthreads();
      if (wave < 3 && lane < NC) {  // Adam has moved the layer-norm parameters
        gam = (B + dv.gate[wave].gamma)[lane];
        bet = (B + dv.gate[wave].beta)[lane];
      }
      ++bptt_done;
    }
    STAMP(9);  // deferred accumulation + Adam (backward bytes)
    // the output layer's own step (lstm.cpp:86-92) when the forward pass could not take it along
    if (!sgd_done) {
      const float o = (B + dv.output + (uint64_t)last_epoch * NO)[tid];
      const float error = ((uint32_t)tid == byte) ? (o - 1.0f) : o;
      const float le = kLearningRate * error;
      const float* src = out_layer + (uint64_t)last_epoch * HID * NO;
      float* dst = out_layer + (uint64_t)epoch * HID * NO;
      float sv[HID];
#pragma unroll
      for (int j = 0; j < HID; ++j) sv[j] = src[(uint64_t)j * NO + tid];
#pragma unroll
      for (int j = 0; j < HID; ++j) dst[(uint64_t)j * NO + tid] = sv[j] - le * L.hid[j];
    }
    if (!SESSION && (phases & 1u)) ppm_landed();
    __syncthreads();
    command_done();
  }
  if (bits_owed && wave == 3) bit_predictions(owed_byte, owed_n, owed_context);  // the last byte's
  // state back to the bank
  if (tid < CP) {
    (B + dv.hidden)[tid] = L.hid[tid];
    (B + dv.hidden_error)[tid] = L.herr[tid];
  }
  (B + dv.probs)[tid] = L.probs[tid];
  if (tid == 0) {
    scal[0] = epoch;
    scal[1] = l_epoch;
    scal[2] = update_steps;
    scal[3] = last_byte;
    scal[4] = context;
    if (!defer_bits) scal[5] = __float_as_uint(prediction);
    scal[6] = coded;
  }
  if (defer_bits && tid == 192) scal[5] = __float_as_uint(prediction);  // (wave 3's lane 0 walked the bits)
  if (SESSION) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (tid == 0) {
      if (exit_state == GMX_MB_EXIT_STOP)
        __hip_atomic_store(&a.mb->done_seq, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.mb->state, exit_state, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

#ifdef GMX_LSTM_PROF
extern "C" int gmx_lstm_prof_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gmx_lstm_prof), sizeof(unsigned long long) * 16);
}
#endif

// cus: compute units the launch may use.  With at most one workgroup per unit the build that allows itself 512
// registers goes: all 77 weight quads of a gate chain in flight at once instead of two stretches of 39 (ONE stream:
// 21.9 us per byte instead of 24.7; 256 streams: 32.0 instead of 34.8, scripts/lstm_phase_profile.py).
extern "C" hipError_t gmx_launch_lstm_kernel(const GmxLstmDev* dv, const GmxLstmRunArgs* args, int n_streams, int cus,
                                             hipStream_t stream) {
  (void)hipGetLastError();
  int build = (cus > 0 && n_streams <= cus) ? 1 : 2;
  if (const char* e = getenv("GMX_LSTM_BUILD")) {  // tuning: force the 1-, 2- or 3-workgroups-per-CU build
    if (e[0] == '1') build = 1;
    if (e[0] == '2') build = 2;
    if (e[0] == '3') build = 3;
  }
  if (build == 1)
    hipLaunchKernelGGL((gmx_lstm_kernel<308, 1, false>), dim3(n_streams), dim3(256), 0, stream, dv, *args);
  else if (build == 2)
    hipLaunchKernelGGL((gmx_lstm_kernel<154, 2, false>), dim3(n_streams), dim3(256), 0, stream, dv, *args);
  else
    hipLaunchKernelGGL((gmx_lstm_kernel<GMX_LSTM_STRETCH, GMX_LSTM_BLOCKS, false>), dim3(n_streams), dim3(256), 0, stream,
                       dv, *args);
  return hipGetLastError();
}

// The per-byte session of ONE stream (args->stream_base; args->mc / mb / idle_ticks set).
extern "C" hipError_t gmx_launch_lstm_session(const GmxLstmDev* dv, const GmxLstmRunArgs* args, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL((gmx_lstm_kernel<154, 2, true>), dim3(1), dim3(256), 0, stream, dv, *args);
  return hipGetLastError();
}

// Scatter of a batch's results into the record arrays of the models downstream (device to
// device): the prediction of bit k of byte n goes to slot `slot` of record 8n+k of a mixer batch
// (and its active flag into that record's mask), lstm_prediction_context of byte n into column
// mixer_ctx_col of the 8 mixer records and column ind_ctx_col of the 8 Indirect records.
struct GmxLstmScatterArgs {
  const float* pred;       // [S][rec_stride][8]
  const uint8_t* act;
  const uint32_t* ctx;     // [S][rec_stride]
  uint64_t rec_stride, n_bytes;
  float* mx_pred;          // [S][mx_stride][mx_n_pad] or null
  uint32_t* mx_mask;       // [S][mx_stride][mx_mask_words]
  uint32_t* mx_ctx;        // [S][mx_stride][mx_m]
  uint64_t mx_stride;
  int32_t mx_n_pad, mx_mask_words, mx_m, slot, mixer_ctx_col;
  uint32_t* ind_ctx;       // [S][ind_stride][ind_k] or null
  uint64_t ind_stride;
  int32_t ind_k, ind_ctx_col;
};

__global__ void __launch_bounds__(256) gmx_lstm_scatter_kernel(const GmxLstmScatterArgs a) {
  const int s = blockIdx.y;
  const uint64_t bit = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (bit >= a.n_bytes * 8) return;
  const uint64_t n = bit >> 3;
  const uint32_t c = a.ctx[(uint64_t)s * a.rec_stride + n];
  if (a.mx_pred) {
    const uint64_t r = (uint64_t)s * a.mx_stride + bit;
    a.mx_pred[r * a.mx_n_pad + a.slot] = a.pred[((uint64_t)s * a.rec_stride + n) * 8 + (bit & 7)];
    uint32_t* w = a.mx_mask + r * a.mx_mask_words + (a.slot >> 5);
    const uint32_t m = 1u << (a.slot & 31);
    *w = a.act[((uint64_t)s * a.rec_stride + n) * 8 + (bit & 7)] ? (*w | m) : (*w & ~m);
    if (a.mixer_ctx_col >= 0) a.mx_ctx[r * a.mx_m + a.mixer_ctx_col] = c;
  }
  if (a.ind_ctx) a.ind_ctx[((uint64_t)s * a.ind_stride + bit) * a.ind_k + a.ind_ctx_col] = c;
}

extern "C" hipError_t gmx_launch_lstm_scatter(const GmxLstmScatterArgs* args, int n_streams, hipStream_t stream) {
  (void)hipGetLastError();
  const unsigned blocks = (unsigned)((args->n_bytes * 8 + 255) / 256);
  hipLaunchKernelGGL(gmx_lstm_scatter_kernel, dim3(blocks, n_streams), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Lock step (gmx_chainstep.inc): the LSTM's prediction of ONE bit of every stream per launch -- LstmModel::Predict's
// walk down the byte distribution (lstm-model.cpp:34-48) with the byte's bits arriving a launch apart, as S decoders
// side by side produce them (coder/decoder.cpp:19-39).  The distribution is the bank's (Lstm::Predict ran in the
// launch before a byte's first bit: what[s] bit 2), the range [bot, top] lives in the bank's scalars between
// launches (scal[8], scal[9]; mid_ follows from them), the prediction goes where ShortTermMemory::SetPrediction
// (short-term-memory.cpp:187-191) puts it -- slot `slot` of the mixers' record of the same step, with its active bit --
// and lstm_prediction_context (lstm-model.cpp:25-33) into the gate-context / Indirect-context columns that read it.
// One wave per stream; the two ordered sums (std::accumulate from the first element up) are one lane's.
__global__ void __launch_bounds__(64)
gmx_lstm_bitstep_kernel(const GmxLstmDev* __restrict__ dvp, const GmxLstmBitArgs a) {
  __shared__ __attribute__((aligned(16))) float pr[GMX_L_NO];
  const int s = blockIdx.x;
  const uint32_t what = a.what[s];
  if (!(what & 2u)) return;
  uint32_t ctx_unused, mask_bit_unused;
  gmx_lstm_bitstep_body(dvp, a, s, what, pr, (int)threadIdx.x, /*mask_to_global=*/true, ctx_unused, mask_bit_unused);
}

extern "C" hipError_t gmx_launch_lstm_bitstep(const GmxLstmDev* dv, const GmxLstmBitArgs* args, int n_streams,
                                              hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(gmx_lstm_bitstep_kernel, dim3(n_streams), dim3(64), 0, stream, dv, *args);
  return hipGetLastError();
}
