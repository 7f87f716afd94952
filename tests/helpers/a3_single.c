/* a3_single.c -- SURVEY.md Appendix A.3, row "256 | 1/0/0 | 2 000 000 | -357.038172 | 957fee36",
 * through the oracle.  The appendix's recipe reads "every context variable = rnd()"; the survey's
 * probe kept a context variable for the (absent) final mixer as well, i.e. it drew ONE MORE 32-bit
 * number per bit than there are mixers -- with that draw the published answer comes out exactly
 * (without it: 679de36f, which is what round 1 recorded as "not reproduced").  The two 24/8/1 rows
 * have a final mixer, so their draw count was never in question.
 * Returns the FNV-style hash of the appendix; *acc receives the double sum of the outputs. */
#include <stdint.h>
#include <string.h>

typedef struct gmxo_bank gmxo_bank;
gmxo_bank* gmxo_create(int, int, const int*, int, const int*, const uint32_t*, const float*);
void gmxo_destroy(gmxo_bank*);
float gmxo_predict(gmxo_bank*, const float*, const int*, int, const uint32_t*, float*);
void gmxo_learn(gmxo_bank*, int);

static uint64_t s;
static uint32_t rnd(void) {
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  return (uint32_t)(s >> 11);
}

uint32_t a3_single(uint64_t T, int unused_context_draws, double* acc_out) {
  s = 0x9E3779B97F4A7C15ull;
  int N = 256, layer[1] = {0}, skip[1] = {1};
  uint32_t table[1] = {1u << 16};
  float lr[1] = {0.005f};
  gmxo_bank* b = gmxo_create(N, 1, skip, 1, layer, table, lr);
  float x[256], out[1];
  int act[256];
  uint32_t h = 0;
  double acc = 0;
  for (uint64_t t = 0; t < T; ++t) {
    int na = 0;
    for (int i = 0; i < N; ++i) {
      x[i] = ((int)(rnd() % 2001) - 1000) / 250.0f;
      if (x[i] != 0) act[na++] = i; /* SetLogitPrediction (short-term-memory.cpp:193-197) */
    }
    uint32_t ctx = rnd();
    for (int k = 0; k < unused_context_draws; ++k) rnd();
    gmxo_predict(b, x, act, na, &ctx, out);
    uint32_t bits;
    memcpy(&bits, out, 4);
    h = h * 16777619u ^ bits;
    acc += out[0];
    gmxo_learn(b, rnd() & 1);
  }
  gmxo_destroy(b);
  *acc_out = acc;
  return h;
}
