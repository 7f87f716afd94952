/* gmx_lstm_synth.h -- deterministic synthetic input for the LSTM byte model (TEST INFRASTRUCTURE,
 * shared by oracle/ref_build/ref_lstm_harness.cpp and the C restatement).
 *
 * Bytes: a noisy first-order process the network can learn:
 *   r = rnd(); next = (r % 4 == 0) ? (r >> 8) & 255 : (prev * 5 + 17) & 255 ... & the alphabet mask
 * ppm_predictions (what ModPPMD leaves in ShortTermMemory, mod_ppmd.cpp:1655-1661): 256 values
 *   >= 1, the true next byte boosted half of the time, then divided by their (left-to-right) sum.
 */
#ifndef GMX_LSTM_SYNTH_H_
#define GMX_LSTM_SYNTH_H_

#include <stdint.h>

typedef struct gmx_lstm_synth {
  uint64_t s;
  uint32_t prev;
  uint32_t mask;   /* alphabet mask (255 = all bytes, 15 = a 16-symbol stream) */
} gmx_lstm_synth;

static inline uint32_t gmx_lstm_rnd(gmx_lstm_synth* g) {
  uint64_t s = g->s;
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  g->s = s;
  return (uint32_t)(s >> 11);
}

static inline void gmx_lstm_synth_init(gmx_lstm_synth* g, uint64_t seed, uint32_t mask) {
  g->s = seed ? seed : 0x9E3779B97F4A7C15ull;
  g->prev = 0;
  g->mask = mask;
}

/* Next byte of the stream and the PPM distribution a predictor would see BEFORE coding it. */
static inline uint32_t gmx_lstm_synth_byte(gmx_lstm_synth* g, float* ppm) {
  uint32_t r = gmx_lstm_rnd(g);
  uint32_t next = ((r & 3u) == 0 ? (r >> 8) : (g->prev * 5u + 17u)) & g->mask;
  uint32_t boost = gmx_lstm_rnd(g) & 1u;
  for (int i = 0; i < 256; ++i) ppm[i] = (float)(1u + (gmx_lstm_rnd(g) % 7u));
  if (boost) ppm[next] = 900.0f;
  float sum = ppm[0];
  for (int i = 1; i < 256; ++i) sum += ppm[i];
  for (int i = 0; i < 256; ++i) ppm[i] /= sum;
  g->prev = next;
  return next;
}

#endif /* GMX_LSTM_SYNTH_H_ */
