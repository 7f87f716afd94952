/* gmx_oracle_ind.c -- CPU restatement of the reference's Indirect models.
 *
 * TEST INFRASTRUCTURE (same rules as gmx_oracle.c): only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it; the product (libgmxmix.so) never links it.
 * Pinned by tests/golden/ind_*.npz, which tests/golden/make_golden.py produces by running the
 * REAL reference class through oracle/ref_build/ref_indirect_harness.cpp.
 *
 * Restates (file:line in /root/reference/src):
 *   IndirectMemory                       memory/long-term-memory.h:11-25
 *   Indirect::Indirect                   models/indirect.cpp:5-26   (table = table_size*256+1)
 *   Indirect::Predict                    models/indirect.cpp:28-46
 *   Indirect::Learn                      models/indirect.cpp:48-69
 *   Indirect::GetMemoryUsage             models/indirect.cpp:71-78
 *   ShortTermMemory::SetLogitPrediction  memory/short-term-memory.cpp:193-197
 *   LongTermMemory::WriteToDisk, indirect section   memory/long-term-memory.cpp:8-32
 * The two state machines (contexts/nonstationary.cpp, contexts/run-map.cpp) are DATA owned by
 * ShortTermMemory in the reference; here, as in the product ABI, the caller passes them in as
 * two 256x2 next-state tables (the fixtures carry them as dumped by the reference itself).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gmx_ind_synth.h"

float gmxo_logistic(float p); /* gmx_oracle.c: Sigmoid::Logistic with the system libm expf */

typedef struct {
  uint32_t size;       /* table_size * 256 + 1 */
  float lr;
  uint8_t* ns;         /* nonstationary_table, 255 = never seen */
  uint8_t* rm;         /* run_map_table, 0 = never seen */
  float nsp[256], rmp[256];
} ind_model;

typedef struct gmxo_ind {
  int k;
  ind_model* m;
  uint8_t ns_next[512], rm_next[512];
  float* pred;         /* [2k] blackboard slots: stale when a model stays silent */
  uint8_t* active;     /* [2k] */
  uint32_t* idx;       /* [k] table index of the last Predict (contexts do not move before Learn) */
} gmxo_ind;

void gmxo_ind_destroy(gmxo_ind* b) {
  if (!b) return;
  for (int i = 0; i < b->k; ++i) {
    free(b->m[i].ns);
    free(b->m[i].rm);
  }
  free(b->m);
  free(b->pred);
  free(b->active);
  free(b->idx);
  free(b);
}

gmxo_ind* gmxo_ind_create(int k, const uint32_t* table_size, const float* lr, const uint8_t* ns_next,
                          const uint8_t* rm_next) {
  gmxo_ind* b = (gmxo_ind*)calloc(1, sizeof *b);
  b->k = k;
  b->m = (ind_model*)calloc(k, sizeof(ind_model));
  for (int i = 0; i < k; ++i) {
    ind_model* m = &b->m[i];
    m->size = table_size[i] * 256u + 1u; /* indirect.cpp:15-19 */
    m->lr = lr[i];
    m->ns = (uint8_t*)malloc(m->size);
    m->rm = (uint8_t*)calloc(m->size, 1);
    memset(m->ns, 255, m->size);         /* long-term-memory.h:13 */
  }
  memcpy(b->ns_next, ns_next, 512);
  memcpy(b->rm_next, rm_next, 512);
  b->pred = (float*)calloc(2 * k, sizeof(float));
  b->active = (uint8_t*)calloc(2 * k, 1);
  b->idx = (uint32_t*)calloc(k, sizeof(uint32_t));
  return b;
}

/* K x Indirect::Predict.  pred_out/active_out: [2k], slot 2i = "-indirect", 2i+1 = "-run_map". */
void gmxo_ind_predict(gmxo_ind* b, const uint32_t* ctx, uint32_t bit_context, float* pred_out,
                      uint8_t* active_out) {
  for (int i = 0; i < b->k; ++i) {
    ind_model* m = &b->m[i];
    uint32_t c = ((ctx[i] << 8) + bit_context) % m->size; /* indirect.cpp:31-32, unsigned wrap */
    b->idx[i] = c;
    b->active[2 * i] = b->active[2 * i + 1] = 0;
    int s = m->ns[c];
    if (s != 255) { /* indirect.cpp:35-38 */
      float p = m->nsp[s];
      b->pred[2 * i] = p;
      b->active[2 * i] = (p != 0.0f); /* short-term-memory.cpp:195 */
    }
    int r = m->rm[c];
    if (r != 0) { /* indirect.cpp:41-44 */
      float p = m->rmp[r];
      b->pred[2 * i + 1] = p;
      b->active[2 * i + 1] = (p != 0.0f);
    }
  }
  if (pred_out) memcpy(pred_out, b->pred, 2 * b->k * sizeof(float));
  if (active_out) memcpy(active_out, b->active, 2 * b->k);
}

/* K x Indirect::Learn. */
void gmxo_ind_learn(gmxo_ind* b, int bit) {
  for (int i = 0; i < b->k; ++i) {
    ind_model* m = &b->m[i];
    uint32_t c = b->idx[i];
    int s = m->ns[c];
    if (s == 255) s = 0; /* indirect.cpp:53-56 */
    m->nsp[s] += ((float)bit - gmxo_logistic(m->nsp[s])) * m->lr; /* indirect.cpp:57-60 */
    m->ns[c] = b->ns_next[2 * s + bit];
    int r = m->rm[c];
    m->rmp[r] += ((float)bit - gmxo_logistic(m->rmp[r])) * m->lr; /* indirect.cpp:63-66 */
    m->rm[c] = b->rm_next[2 * r + bit];
  }
}

void gmxo_ind_run(gmxo_ind* b, uint64_t T, const uint32_t* ctx, const uint32_t* bit_context,
                  const uint8_t* bits, uint64_t nolearn_from, float* pred_out, uint8_t* active_out) {
  for (uint64_t t = 0; t < T; ++t) {
    gmxo_ind_predict(b, ctx + t * b->k, bit_context[t], pred_out ? pred_out + t * 2 * b->k : 0,
                     active_out ? active_out + t * 2 * b->k : 0);
    if (t < nolearn_from) gmxo_ind_learn(b, bits[t]);
  }
}

uint64_t gmxo_ind_memory_usage(const gmxo_ind* b, int i) {
  return 12ull + 256 * 4 * 2 + 2ull * b->m[i].size; /* indirect.cpp:71-78 */
}

/* Indirect section of LongTermMemory::WriteToDisk (long-term-memory.cpp:8-32). */
size_t gmxo_ind_export(const gmxo_ind* b, uint8_t* out, size_t cap) {
  size_t n = 0;
#define PUT(ptr, len)                                    \
  do {                                                   \
    if (out && n + (len) <= cap) memcpy(out + n, (ptr), (len)); \
    n += (len);                                          \
  } while (0)
  for (int i = 0; i < b->k; ++i) {
    const ind_model* m = &b->m[i];
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < m->size; ++j) cnt += m->ns[j] != 255;
    PUT(&cnt, 4);
    if (cnt < m->size / 3) {
      for (uint32_t j = 0; j < m->size; ++j)
        if (m->ns[j] != 255) {
          PUT(&j, 4);
          PUT(&m->ns[j], 1);
          PUT(&m->rm[j], 1);
        }
    } else {
      PUT(m->ns, m->size);
      PUT(m->rm, m->size);
    }
    PUT(m->nsp, 1024);
    PUT(m->rmp, 1024);
  }
#undef PUT
  return n;
}

/* The synthetic stream of gmx_ind_synth.h as arrays: ctx[T][k], bit_context[T], bits[T]. */
void gmxo_ind_synth_fill(uint64_t seed, int k, const uint32_t* ctx_mod, uint64_t T, uint32_t* ctx,
                         uint32_t* bit_context, uint8_t* bits) {
  gmx_ind_synth g;
  gmx_ind_synth_init(&g, seed, k, ctx_mod);
  uint32_t* cur = (uint32_t*)calloc(k, sizeof(uint32_t));
  for (uint64_t t = 0; t < T; ++t) {
    bit_context[t] = gmx_ind_synth_contexts(&g, cur);
    memcpy(ctx + t * k, cur, k * sizeof(uint32_t));
    bits[t] = (uint8_t)gmx_ind_synth_bit(&g, cur);
  }
  free(cur);
}

/* The harness' running checksum over (prediction bits, active flag) of every slot and bit. */
uint64_t gmxo_ind_fnv64(const float* pred, const uint8_t* active, uint64_t n, uint64_t h) {
  for (uint64_t i = 0; i < n; ++i) {
    uint32_t u;
    memcpy(&u, &pred[i], 4);
    h = (h ^ u) * 0x100000001b3ull;
    h = (h ^ active[i]) * 0x100000001b3ull;
  }
  return h;
}
