/* gmx_oracle.c -- CPU restatement of gmix's mixer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped path (gmix_amd/csrc, HIP) shares no
 * code with it and fails loudly when its extension is missing.
 *
 * Parity status: PINNED.  The reference's own tests hold no numeric vectors for this path
 * (SURVEY.md section 4), so the pin is the reference itself run in the build container:
 * oracle/ref_build/ compiles the reference's Mixer sources where they lie and
 * tests/golden/make_golden.py records its outputs; tests/test_oracle.py checks this
 * restatement against those fixtures bit for bit (outputs, probabilities, counters and the
 * serialised mixer section).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Arithmetic notes: plain C on x86-64 SSE2 evaluates float expressions in float
 * (FLT_EVAL_METHOD 0); build with -O2 -ffp-contract=off and no -march so that no FMA is
 * formed -- this is the "strict" build SURVEY.md section 8c designates as the oracle.
 * expf / pow come from the system libm, exactly as in the reference (sigmoid.cpp:5,
 * mixer.cpp:111).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gmx_synth.h"

typedef struct {
  int layer;             /* 0, 1 or 2 (mixer.h:13) */
  uint32_t table_size;   /* rows (mixer.cpp:15) */
  float lr;              /* learning_rate_ (mixer.h:38) */
  int output_index;      /* ShortTermMemory::AddMixer (short-term-memory.cpp:199-213) */
  int weight_size;       /* mixer.cpp:17-26 */
  uint64_t steps;        /* steps_      : Learn calls so far (mixer.cpp:9,124) */
  uint64_t max_steps;    /* max_steps_  : starts at 1 (mixer.cpp:8,126-128) */
  uint64_t contexts_seen;/* mixer.cpp:45 */
  float* w;              /* dense [table_size][weight_size], zero = unseen or fresh row */
  uint64_t* row_steps;   /* MixerData::steps; 0 <=> row never allocated (long-term-memory.h:27-31) */
  uint32_t row;          /* row latched by the last predict */
} gmxo_mixer;

typedef struct gmxo_bank {
  int n;       /* num_predictions */
  int n_skip;  /* models_with_skip_connection.size() */
  int* skip;   /* indices into predictions */
  int m, l0, l1, has_final;
  gmxo_mixer* mx;
  /* blackboard (short-term-memory.h:43-58,134-142) */
  float* predictions;
  int* active;
  int n_active;
  float* out0;
  float* out1;
  float final_out;
} gmxo_bank;

/* Sigmoid::Logistic (mixer/sigmoid.cpp:5): float arithmetic, libm expf. */
float gmxo_logistic(float p) { return 1 / (1 + expf(-p)); }

/* Final squash of Predictor::Predict (predictor.cpp:369-375). */
float gmxo_squash_clamp(float out) {
  float prob = gmxo_logistic(out);
  float eps = 0.0001;
  if (prob < eps)
    prob = eps;
  else if (prob > 1 - eps)
    prob = 1 - eps;
  return prob;
}

/* First factor of the learning-rate decay (mixer.cpp:111), narrowed to float there. */
float gmxo_decay_base(uint64_t steps) { return 0.9 / pow(0.0000001 * steps + 0.8, 0.8); }

/* Mixer::Mixer + Predictor::AddMixers bookkeeping (mixer.cpp:3-27, predictor.cpp:251-358). */
gmxo_bank* gmxo_create(int n_inputs, int n_skip, const int* skip_index, int n_mixers,
                       const int* layer, const uint32_t* table_size, const float* lr) {
  gmxo_bank* b = (gmxo_bank*)calloc(1, sizeof(*b));
  b->n = n_inputs;
  b->n_skip = n_skip;
  b->skip = (int*)calloc(n_skip ? n_skip : 1, sizeof(int));
  for (int i = 0; i < n_skip; ++i) b->skip[i] = skip_index[i];
  b->m = n_mixers;
  b->mx = (gmxo_mixer*)calloc(n_mixers, sizeof(gmxo_mixer));
  for (int j = 0; j < n_mixers; ++j) {
    gmxo_mixer* x = &b->mx[j];
    x->layer = layer[j];
    x->table_size = table_size[j];
    x->lr = lr[j];
    x->max_steps = 1;
    if (layer[j] == 0) {
      x->output_index = b->l0++;
      x->weight_size = n_inputs + x->output_index;
    } else if (layer[j] == 1) {
      x->output_index = b->l1++;
      x->weight_size = b->l0 + x->output_index + n_skip;
    } else {
      x->output_index = b->l0 + b->l1 + 1;
      x->weight_size = b->l0 + b->l1 + n_skip;
      b->has_final = 1;
    }
    x->w = (float*)calloc((size_t)x->table_size * x->weight_size, sizeof(float));
    x->row_steps = (uint64_t*)calloc(x->table_size, sizeof(uint64_t));
  }
  b->predictions = (float*)calloc(n_inputs ? n_inputs : 1, sizeof(float));
  b->active = (int*)calloc(n_inputs ? n_inputs : 1, sizeof(int));
  b->out0 = (float*)calloc(b->l0 ? b->l0 : 1, sizeof(float));
  b->out1 = (float*)calloc(b->l1 ? b->l1 : 1, sizeof(float));
  return b;
}

void gmxo_destroy(gmxo_bank* b) {
  if (!b) return;
  for (int j = 0; j < b->m; ++j) {
    free(b->mx[j].w);
    free(b->mx[j].row_steps);
  }
  free(b->mx);
  free(b->skip);
  free(b->predictions);
  free(b->active);
  free(b->out0);
  free(b->out1);
  free(b);
}

/* Mixer::Predict (mixer.cpp:51-106).  An unseen row is "no row": p = 0 (mixer.cpp:52-55). */
static void mixer_predict(gmxo_bank* b, gmxo_mixer* x, uint32_t context) {
  x->row = context % x->table_size; /* FindMixerData, mixer.cpp:29-37 */
  float p = 0;
  if (x->row_steps[x->row] != 0) {
    const float* w = x->w + (size_t)x->row * x->weight_size;
    if (x->layer == 0) {
      for (int a = 0; a < b->n_active; ++a) {
        int i = b->active[a];
        p += b->predictions[i] * w[i];
      }
      for (int i = 0; i < x->output_index; ++i) p += b->out0[i] * w[b->n + i];
    } else if (x->layer == 1) {
      for (int i = 0; i < b->l0; ++i) p += b->out0[i] * w[i];
      for (int i = 0; i < x->output_index; ++i) p += b->out1[i] * w[b->l0 + i];
      int offset = b->l0 + x->output_index;
      for (int i = 0; i < b->n_skip; ++i) p += b->predictions[b->skip[i]] * w[offset + i];
    } else {
      for (int i = 0; i < b->l0; ++i) p += b->out0[i] * w[i];
      for (int i = 0; i < b->l1; ++i) p += b->out1[i] * w[b->l0 + i];
      int offset = b->l0 + b->l1;
      for (int i = 0; i < b->n_skip; ++i) p += b->predictions[b->skip[i]] * w[offset + i];
    }
  }
  if (x->layer == 2)
    b->final_out = p;
  else if (x->layer == 1)
    b->out1[x->output_index] = p;
  else
    b->out0[x->output_index] = p;
}

/* Mixer::Learn (mixer.cpp:108-176). */
static void mixer_learn(gmxo_bank* b, gmxo_mixer* x, int new_bit) {
  /* FindOrCreateMixerData (mixer.cpp:39-49): the dense row is already zero. */
  if (x->row_steps[x->row] == 0) ++x->contexts_seen;
  float* w = x->w + (size_t)x->row * x->weight_size;
  uint64_t* rs = &x->row_steps[x->row];
  float decay = 0.9 / pow(0.0000001 * x->steps + 0.8, 0.8);
  decay *= 1.5 - ((1.0 * *rs) / x->max_steps);
  float p;
  if (x->layer == 2)
    p = gmxo_logistic(b->final_out);
  else if (x->layer == 1)
    p = gmxo_logistic(b->out1[x->output_index]);
  else
    p = gmxo_logistic(b->out0[x->output_index]);
  float update = decay * x->lr * (p - new_bit);
  ++x->steps;
  ++*rs;
  if (*rs > x->max_steps) x->max_steps = *rs;
  if (x->layer == 0) {
    for (int a = 0; a < b->n_active; ++a) {
      int i = b->active[a];
      w[i] -= update * b->predictions[i];
    }
    for (int i = 0; i < x->output_index; ++i) w[i + b->n] -= update * b->out0[i];
  } else if (x->layer == 1) {
    for (int i = 0; i < b->l0; ++i) w[i] -= update * b->out0[i];
    for (int i = 0; i < x->output_index; ++i) w[i + b->l0] -= update * b->out1[i];
    int offset = b->l0 + x->output_index;
    for (int i = 0; i < b->n_skip; ++i) w[i + offset] -= update * b->predictions[b->skip[i]];
  } else {
    for (int i = 0; i < b->l0; ++i) w[i] -= update * b->out0[i];
    for (int i = 0; i < b->l1; ++i) w[b->l0 + i] -= update * b->out1[i];
    int offset = b->l0 + b->l1;
    for (int i = 0; i < b->n_skip; ++i) w[i + offset] -= update * b->predictions[b->skip[i]];
  }
  if ((*rs & 1023) == 0) {
    const float c = 1.0f - 3.0e-6f;
    for (int i = 0; i < x->weight_size; ++i) w[i] *= c;
  }
}

/* The mixer slice of Predictor::Predict (predictor.cpp:360-376).
 *   predictions[n] : raw blackboard (stale slots allowed), active[n_active] ascending
 *                    model indices (short-term-memory.cpp:187-197), ctx[m] gate contexts in
 *                    mixer construction order.  out_all[m] (nullable) receives the logit
 *                    outputs: layer 0, then layer 1, then final.  Returns the clamped
 *                    probability computed from the LAST mixer's output. */
float gmxo_predict(gmxo_bank* b, const float* predictions, const int* active, int n_active,
                   const uint32_t* ctx, float* out_all) {
  memcpy(b->predictions, predictions, sizeof(float) * b->n);
  memcpy(b->active, active, sizeof(int) * n_active);
  b->n_active = n_active;
  for (int j = 0; j < b->m; ++j) mixer_predict(b, &b->mx[j], ctx[j]);
  float last = 0;
  for (int j = 0; j < b->m; ++j) {
    gmxo_mixer* x = &b->mx[j];
    float o = x->layer == 2 ? b->final_out
                            : (x->layer == 1 ? b->out1[x->output_index] : b->out0[x->output_index]);
    if (out_all) out_all[j] = o;
    last = o;
  }
  return gmxo_squash_clamp(last);
}

/* Predictor::Perceive + the mixer slice of Predictor::Learn (predictor.cpp:378-387). */
void gmxo_learn(gmxo_bank* b, int bit) {
  for (int j = 0; j < b->m; ++j) mixer_learn(b, &b->mx[j], bit);
}

/* Batched driver: T bits of {raw predictions[n], active flags[n], ctx[m], bit}; Learn is
 * called for bits t < nolearn_from only (generation mode, runner-utils.cpp:199-209). */
void gmxo_run(gmxo_bank* b, uint64_t T, const float* pred, const uint8_t* active_flags,
              const uint32_t* ctx, const uint8_t* bits, uint64_t nolearn_from, float* p_final,
              float* out_all) {
  int* act = (int*)malloc(sizeof(int) * (b->n ? b->n : 1));
  for (uint64_t t = 0; t < T; ++t) {
    int na = 0;
    for (int i = 0; i < b->n; ++i)
      if (active_flags[t * b->n + i]) act[na++] = i;
    float p = gmxo_predict(b, pred + t * b->n, act, na, ctx + t * b->m,
                           out_all ? out_all + t * b->m : 0);
    if (p_final) p_final[t] = p;
    if (t < nolearn_from) gmxo_learn(b, bits[t]);
  }
  free(act);
}

/* Mixer::WriteToDisk for every mixer in order (mixer.cpp:178-182): 3 x u64 each. */
size_t gmxo_export_short(const gmxo_bank* b, void* buf, size_t cap) {
  size_t need = (size_t)b->m * 24;
  if (!buf || cap < need) return need;
  uint64_t* o = (uint64_t*)buf;
  for (int j = 0; j < b->m; ++j) {
    o[3 * j + 0] = b->mx[j].steps;
    o[3 * j + 1] = b->mx[j].max_steps;
    o[3 * j + 2] = b->mx[j].contexts_seen;
  }
  return need;
}

/* Mixer section of LongTermMemory::WriteToDisk (long-term-memory.cpp:35-55). */
size_t gmxo_export_long(const gmxo_bank* b, void* buf, size_t cap) {
  size_t need = 0;
  for (int j = 0; j < b->m; ++j) {
    const gmxo_mixer* x = &b->mx[j];
    need += 8;
    for (uint32_t r = 0; r < x->table_size; ++r)
      if (x->row_steps[r]) need += 12 + 4 * (size_t)x->weight_size;
  }
  if (!buf || cap < need) return need;
  uint8_t* o = (uint8_t*)buf;
  for (int j = 0; j < b->m; ++j) {
    const gmxo_mixer* x = &b->mx[j];
    uint32_t cnt = 0;
    for (uint32_t r = 0; r < x->table_size; ++r)
      if (x->row_steps[r]) ++cnt;
    uint32_t input_size = cnt ? (uint32_t)x->weight_size : 0;
    memcpy(o, &cnt, 4);
    memcpy(o + 4, &input_size, 4);
    o += 8;
    for (uint32_t r = 0; r < x->table_size; ++r) {
      if (!x->row_steps[r]) continue;
      memcpy(o, &r, 4);
      memcpy(o + 4, &x->row_steps[r], 8);
      memcpy(o + 12, x->w + (size_t)r * x->weight_size, 4 * (size_t)x->weight_size);
      o += 12 + 4 * (size_t)x->weight_size;
    }
  }
  return need;
}

/* Mixer::GetMemoryUsage (mixer.cpp:197-205). */
uint64_t gmxo_memory_usage(const gmxo_bank* b, int j) {
  const gmxo_mixer* x = &b->mx[j];
  uint64_t usage = 29;
  int mixer_data_size = x->weight_size * 4 + 12;
  usage += x->contexts_seen * mixer_data_size;
  usage += 8 * (uint64_t)x->table_size;
  return usage;
}

/* Encoder::Discretize (coder/encoder.cpp:8): the 16-bit probability the coder consumes. */
uint32_t gmxo_discretize(float p) { return 1 + 65534 * p; }

/* Binary arithmetic coder, encoder side (coder/encoder.cpp:10-34), for compressed-bytes
 * parity checks: encodes bits[T] with p[T]; returns bytes written (out may be NULL to size). */
size_t gmxo_encode(uint64_t T, const uint8_t* bits, const float* p, uint8_t* out, size_t cap) {
  uint32_t x1 = 0, x2 = 0xffffffffu;
  size_t n = 0;
  for (uint64_t t = 0; t < T; ++t) {
    const uint32_t pr = gmxo_discretize(p[t]);
    const uint32_t xmid = x1 + ((x2 - x1) >> 16) * pr + (((x2 - x1) & 0xffff) * pr >> 16);
    if (bits[t])
      x2 = xmid;
    else
      x1 = xmid + 1;
    while (((x1 ^ x2) & 0xff000000u) == 0) {
      if (out && n < cap) out[n] = (uint8_t)(x2 >> 24);
      ++n;
      x1 <<= 8;
      x2 = (x2 << 8) + 255;
    }
  }
  /* Flush (encoder.cpp:27-34) */
  while (((x1 ^ x2) & 0xff000000u) == 0) {
    if (out && n < cap) out[n] = (uint8_t)(x2 >> 24);
    ++n;
    x1 <<= 8;
    x2 = (x2 << 8) + 255;
  }
  if (out && n < cap) out[n] = (uint8_t)(x2 >> 24);
  ++n;
  return n;
}

/* Synthetic stream (gmx_synth.h) materialised for tests and bench:
 *   pred[T][n] raw slots, active[T][n], ctx[T][m], bits[T]. */
void gmxo_synth_fill(uint64_t seed, int n, int m, int ctx_mode, uint32_t ctx_mod,
                     uint32_t zero_mod, int bit_mode, uint64_t T, float* pred, uint8_t* active,
                     uint32_t* ctx, uint8_t* bits) {
  gmx_synth g;
  gmx_synth_init(&g, seed, n, m, ctx_mode, ctx_mod, zero_mod, bit_mode);
  float* p = (float*)calloc(n ? n : 1, sizeof(float));
  uint8_t* a = (uint8_t*)calloc(n ? n : 1, 1);
  uint32_t* c = (uint32_t*)calloc(m ? m : 1, sizeof(uint32_t));
  for (uint64_t t = 0; t < T; ++t) {
    bits[t] = (uint8_t)gmx_synth_step(&g, p, a, c);
    memcpy(pred + t * n, p, sizeof(float) * n);
    memcpy(active + t * n, a, n);
    memcpy(ctx + t * m, c, sizeof(uint32_t) * m);
  }
  free(p);
  free(a);
  free(c);
}

/* Stateful variant for long streams generated chunk by chunk. */
typedef struct gmxo_stream {
  gmx_synth g;
  float* p;
  uint8_t* a;
  uint32_t* c;
} gmxo_stream;

gmxo_stream* gmxo_stream_new(uint64_t seed, int n, int m, int ctx_mode, uint32_t ctx_mod,
                             uint32_t zero_mod, int bit_mode) {
  gmxo_stream* s = (gmxo_stream*)calloc(1, sizeof(*s));
  gmx_synth_init(&s->g, seed, n, m, ctx_mode, ctx_mod, zero_mod, bit_mode);
  s->p = (float*)calloc(n ? n : 1, sizeof(float));
  s->a = (uint8_t*)calloc(n ? n : 1, 1);
  s->c = (uint32_t*)calloc(m ? m : 1, sizeof(uint32_t));
  return s;
}

void gmxo_stream_next(gmxo_stream* s, uint64_t T, float* pred, uint8_t* active, uint32_t* ctx,
                      uint8_t* bits) {
  const int n = s->g.n, m = s->g.m;
  for (uint64_t t = 0; t < T; ++t) {
    bits[t] = (uint8_t)gmx_synth_step(&s->g, s->p, s->a, s->c);
    memcpy(pred + t * n, s->p, sizeof(float) * n);
    memcpy(active + t * n, s->a, n);
    memcpy(ctx + t * m, s->c, sizeof(uint32_t) * m);
  }
}

void gmxo_stream_free(gmxo_stream* s) {
  if (!s) return;
  free(s->p);
  free(s->a);
  free(s->c);
  free(s);
}

/* libm probes used by tests/test_math.py to pin the product's own expf against the libm the
 * reference would link on this machine. */
float gmxo_libm_expf(float x) { return expf(x); }
void gmxo_libm_expf_array(const float* x, float* y, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) y[i] = expf(x[i]);
}

/* Checksum the harness prints: per bit, FNV-1a-style over the bit patterns of out_all[m]
 * then of p (ref_build/ref_mixer_harness.cpp).  h0 lets callers chain batches. */
uint64_t gmxo_fnv64(uint64_t h0, uint64_t T, int m, const float* out_all, const float* p) {
  uint64_t h = h0 ? h0 : 1469598103934665603ull;
  for (uint64_t t = 0; t < T; ++t) {
    for (int k = 0; k < m; ++k) {
      uint32_t b;
      memcpy(&b, &out_all[t * m + k], 4);
      h = (h ^ b) * 1099511628211ull;
    }
    uint32_t pb;
    memcpy(&pb, &p[t], 4);
    h = (h ^ pb) * 1099511628211ull;
  }
  return h;
}
