/* gmx_synth.h -- deterministic synthetic mixer-input streams (TEST INFRASTRUCTURE).
 *
 * One generator, shared verbatim by
 *   - oracle/ref_build/ref_mixer_harness.cpp  (drives the reference's own Mixer class),
 *   - oracle/gmx_oracle.c                     (CPU restatement, exported for tests/bench),
 * so that golden vectors produced HERE from the reference can be re-derived on the GPU
 * box from a seed alone (the reference itself never travels).
 *
 * Stream definition (BASELINE.json configs[1]):
 *   rnd()  = xorshift64 (s^=s<<13; s^=s>>7; s^=s<<17; return (uint32_t)(s>>11))
 *   per bit: for i<N   x_i = ((int)(rnd()%2001) - 1000) / 250.0f      (grid on [-4,4])
 *            for j<M   ctx_j = rnd()                                  (32-bit gate context)
 *            ... Predict ...
 *            bit = rnd() & 1
 * Extensions (off by default, used by the parity tests to reach the reference's edge
 * cases: silent models / stale skip inputs, repeated rows, rows that persist over bits):
 *   zero_mod k>0 : before x_i, d=rnd(); model i is SILENT this bit iff d%k==0 (then no x draw;
 *                  its slot in predictions[] keeps the stale value, like a reference model
 *                  that does not call SetLogitPrediction -- short-term-memory.cpp:193-197)
 *   ctx_mode 1   : ctx_j = rnd() % ctx_mod
 *   ctx_mode 2/3 : as 0/1 but contexts are only redrawn every 8th bit (byte-boundary contexts)
 *   ctx_mode 4/5 : as 2/3, and the contexts GMX_SYNTH_BITLEVEL names (j = 2, 11, 26, 29: the
 *                  positions of the reference predictor's four bit-level gate contexts,
 *                  predictor.cpp:103-186) are redrawn on every other bit too, in index order --
 *                  the row-change pattern of a real gmix run: two layer-0 and two layer-1 rows
 *                  move on every bit, all 33 at a byte boundary
 *   bit_mode 1   : r=rnd(); bit = (x_0 > 0) xor ((r&7)==0) -- a learnable stream, so weights
 *                  grow and outputs leave the neighbourhood of 0 (exercises the squash/clamp)
 */
#ifndef GMX_SYNTH_H_
#define GMX_SYNTH_H_

#include <stdint.h>

#define GMX_SYNTH_SEED 0x9E3779B97F4A7C15ull
/* bit j set = gate context j is bit-level in ctx_mode 4/5 */
#define GMX_SYNTH_BITLEVEL ((1ull << 2) | (1ull << 11) | (1ull << 26) | (1ull << 29))

typedef struct gmx_synth {
  uint64_t s;        /* xorshift64 state */
  int32_t n;         /* model predictions per bit */
  int32_t m;         /* gate contexts per bit (= mixers) */
  int32_t ctx_mode;  /* 0..5, see above */
  uint32_t ctx_mod;  /* modulus for ctx_mode 1/3 */
  uint32_t zero_mod; /* 0 = every model speaks */
  int32_t bit_mode;  /* 0 = random bits, 1 = learnable bits */
  uint64_t t;        /* bits generated so far */
} gmx_synth;

static inline uint32_t gmx_synth_rnd(gmx_synth* g) {
  uint64_t s = g->s;
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  g->s = s;
  return (uint32_t)(s >> 11);
}

static inline void gmx_synth_init(gmx_synth* g, uint64_t seed, int n, int m, int ctx_mode,
                                  uint32_t ctx_mod, uint32_t zero_mod, int bit_mode) {
  g->s = seed ? seed : GMX_SYNTH_SEED;
  g->n = n;
  g->m = m;
  g->ctx_mode = ctx_mode;
  g->ctx_mod = ctx_mod ? ctx_mod : 1;
  g->zero_mod = zero_mod;
  g->bit_mode = bit_mode;
  g->t = 0;
}

/* Advance one bit.
 *   pred[n]   : persistent raw prediction slots (stale when silent)
 *   active[n] : 1 iff model i spoke this bit with a non-zero logit
 *               (SetLogitPrediction marks a zero logit inactive, short-term-memory.cpp:195)
 *   ctx[m]    : persistent gate contexts
 * returns the coded bit. */
static inline int gmx_synth_step(gmx_synth* g, float* pred, uint8_t* active, uint32_t* ctx) {
  for (int i = 0; i < g->n; ++i) {
    if (g->zero_mod) {
      uint32_t d = gmx_synth_rnd(g);
      if (d % g->zero_mod == 0) {
        active[i] = 0;
        continue;
      }
    }
    float x = (float)((int)(gmx_synth_rnd(g) % 2001u) - 1000) / 250.0f;
    pred[i] = x;
    active[i] = (x != 0.0f);
  }
  int redraw = (g->ctx_mode < 2) || ((g->t & 7) == 0);
  if (redraw) {
    for (int j = 0; j < g->m; ++j) {
      uint32_t c = gmx_synth_rnd(g);
      if (g->ctx_mode & 1) c %= g->ctx_mod;
      ctx[j] = c;
    }
  } else if (g->ctx_mode >= 4) {
    for (int j = 0; j < g->m && j < 64; ++j) {
      if (!((GMX_SYNTH_BITLEVEL >> j) & 1)) continue;
      uint32_t c = gmx_synth_rnd(g);
      if (g->ctx_mode & 1) c %= g->ctx_mod;
      ctx[j] = c;
    }
  }
  g->t++;
  uint32_t r = gmx_synth_rnd(g);
  if (g->bit_mode == 1) return (int)((pred[0] > 0.0f) ^ ((r & 7u) == 0));
  return (int)(r & 1u);
}

#endif /* GMX_SYNTH_H_ */
