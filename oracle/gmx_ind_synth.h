/* gmx_ind_synth.h -- deterministic synthetic input streams for the Indirect models
 * (TEST INFRASTRUCTURE, shared by oracle/ref_build/ref_indirect_harness.cpp, the C restatement
 * and the on-device twin, like gmx_synth.h for the mixers).
 *
 * The stream has the byte structure the reference's contexts have (basic-contexts.cpp:21-39):
 *   recent_bits starts at 1; after each coded bit recent_bits = 2*recent_bits + bit; at 256 a
 *   byte is complete: recent_bits = 1 and every model's context is redrawn;
 *   bit_context = recent_bits - 1.
 * Context of model k at a byte boundary: c = rnd(); if (ctx_mod[k % 4]) c %= ctx_mod[k % 4]
 *   (small moduli make contexts come back, so states leave "never seen" and predictions
 *   train; modulus 0 = full 32-bit hashes: (c << 8) wraps and collides like a real hash).
 * Coded bit: r = rnd(); bit = ((ctx[0] ^ (bit_context * 7)) & 1) xor ((r % 5) == 0)
 *   -- learnable from the first context, noisy.
 */
#ifndef GMX_IND_SYNTH_H_
#define GMX_IND_SYNTH_H_

#include <stdint.h>

typedef struct gmx_ind_synth {
  uint64_t s;
  int32_t k;              /* models */
  uint32_t ctx_mod[4];
  uint32_t recent_bits;
  uint64_t t;
} gmx_ind_synth;

static inline uint32_t gmx_ind_rnd(gmx_ind_synth* g) {
  uint64_t s = g->s;
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  g->s = s;
  return (uint32_t)(s >> 11);
}

static inline void gmx_ind_synth_init(gmx_ind_synth* g, uint64_t seed, int k, const uint32_t ctx_mod[4]) {
  g->s = seed ? seed : 0x9E3779B97F4A7C15ull;
  g->k = k;
  for (int i = 0; i < 4; ++i) g->ctx_mod[i] = ctx_mod[i];
  g->recent_bits = 1;
  g->t = 0;
}

/* Contexts and bit_context for the NEXT bit (call before Predict). */
static inline uint32_t gmx_ind_synth_contexts(gmx_ind_synth* g, uint32_t* ctx) {
  if (g->recent_bits == 1) {
    for (int j = 0; j < g->k; ++j) {
      uint32_t c = gmx_ind_rnd(g);
      uint32_t m = g->ctx_mod[j & 3];
      if (m) c %= m;
      ctx[j] = c;
    }
  }
  return g->recent_bits - 1;
}

/* The coded bit (call after Predict); advances the byte state. */
static inline int gmx_ind_synth_bit(gmx_ind_synth* g, const uint32_t* ctx) {
  uint32_t r = gmx_ind_rnd(g);
  uint32_t bc = g->recent_bits - 1;
  int bit = (int)(((ctx[0] ^ (bc * 7u)) & 1u) ^ ((r % 5u) == 0));
  g->recent_bits = g->recent_bits * 2 + (uint32_t)bit;
  if (g->recent_bits >= 256) g->recent_bits = 1;
  g->t++;
  return bit;
}

#endif /* GMX_IND_SYNTH_H_ */
