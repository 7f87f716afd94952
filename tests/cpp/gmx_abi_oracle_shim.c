/* gmx_abi_oracle_shim.c -- TEST INFRASTRUCTURE, CPU only.  NOT a fallback and never shipped:
 * the product library has no CPU path (GMX_ERR_NO_DEVICE).  This file answers the C-ABI
 * entry points gmx_model_adapter.h and gmx_batched.h call with the oracle's restatement, so that the adapter's
 * HOST logic (registration, staging through LongTermMemory::mixers, checkpoint order, Copy) can
 * be run against the real reference in the build container, where there is no GPU
 * (tests/test_dropin_cpu.py).  The GPU proof is tests/test_gpu_dropin.py, against libgmxmix.so. */
#include "../../oracle/gmx_oracle.c"

#include "../../include/gmxmix.h"

struct gmx_group {
  gmxo_bank** bs;  /* one oracle bank per stream */
  gmxo_bank* b;    /* == bs[0] */
  int S, n, m;
};

const char* gmx_strerror(int s) { return s == GMX_OK ? "ok" : s == GMX_ERR_FORMAT ? "malformed checkpoint" : "error"; }
const char* gmx_last_error(void) { return ""; }
int gmx_device_pci_bus_id(int device, char* buf, size_t len) {
  (void)device;
  (void)buf;
  (void)len;
  return GMX_ERR_NO_DEVICE;
}

int gmx_group_create(gmx_group** out, const gmx_topology* t, int n_streams, int device) {
  (void)device;
  if (!out || !t || n_streams < 1) return GMX_ERR_INVALID;
  int layer[64];
  uint32_t table[64];
  float lr[64];
  int skip[8];
  for (int j = 0; j < t->n_mixers; ++j) {
    layer[j] = t->mixers[j].layer;
    table[j] = t->mixers[j].table_size;
    lr[j] = t->mixers[j].learning_rate;
  }
  for (int i = 0; i < t->n_skip; ++i) skip[i] = t->skip_index[i];
  gmx_group* g = (gmx_group*)calloc(1, sizeof(*g));
  g->n = t->n_inputs;
  g->m = t->n_mixers;
  g->S = n_streams;
  g->bs = (gmxo_bank**)calloc((size_t)n_streams, sizeof(gmxo_bank*));
  for (int s = 0; s < n_streams; ++s)
    g->bs[s] = gmxo_create(t->n_inputs, t->n_skip, skip, t->n_mixers, layer, table, lr);
  g->b = g->bs[0];
  *out = g;
  return GMX_OK;
}

void gmx_group_destroy(gmx_group* g) {
  if (!g) return;
  for (int s = 0; s < g->S; ++s) gmxo_destroy(g->bs[s]);
  free(g->bs);
  free(g);
}
int gmx_group_n_mixers(const gmx_group* g) { return g ? g->m : GMX_ERR_INVALID; }
int gmx_group_n_inputs(const gmx_group* g) { return g ? g->n : GMX_ERR_INVALID; }
int gmx_group_n_streams(const gmx_group* g) { return g ? g->S : GMX_ERR_INVALID; }

int gmx_bank_forward(gmx_group* g, int stream, const float* predictions, const int32_t* active, int n_active,
                     const uint32_t* contexts, float* p_final, float* out_all) {
  if (!g || stream < 0 || stream >= g->S || n_active < 0) return GMX_ERR_INVALID;
  *p_final = gmxo_predict(g->bs[stream], predictions, active, n_active, contexts, out_all);
  return GMX_OK;
}

/* ---- the batched surface the run-ahead compressor uses (gmix_amd/host/gmx_batched.h): plain host arrays, the
 * "device work" done by the oracle when gmx_group_run_ragged is called ---- */
struct gmx_batch {
  gmx_group* g;
  uint64_t T;
  int n_pad, mw;
  float *pred, *p, *out;
  uint32_t *mask, *ctx;
  uint8_t* bits;
  float* last;
  unsigned flags;
};
int gmx_batch_create(gmx_batch** out, gmx_group* g, uint64_t max_bits, unsigned flags) {
  if (!out || !g || !max_bits ||
      (flags != (GMX_BATCH_OUTPUTS | GMX_BATCH_MASK) && flags != (GMX_BATCH_LAST_OUTPUTS | GMX_BATCH_MASK)))
    return GMX_ERR_INVALID;
  gmx_batch* b = (gmx_batch*)calloc(1, sizeof(*b));
  const size_t R = (size_t)g->S * max_bits;
  b->g = g;
  b->T = max_bits;
  b->n_pad = (g->n + 3) / 4 * 4;
  b->mw = (g->n + 31) / 32;
  b->pred = (float*)calloc(R * b->n_pad, 4);
  b->mask = (uint32_t*)calloc(R * b->mw, 4);
  b->ctx = (uint32_t*)calloc(R * g->m, 4);
  b->bits = (uint8_t*)calloc(R, 1);
  b->p = (float*)calloc(R, 4);
  b->out = (float*)calloc(R * g->m, 4);
  b->last = (float*)calloc((size_t)g->S * g->m, 4);
  b->flags = flags;
  *out = b;
  return GMX_OK;
}
void gmx_batch_destroy(gmx_batch* b) {
  if (!b) return;
  free(b->pred);
  free(b->mask);
  free(b->ctx);
  free(b->bits);
  free(b->p);
  free(b->out);
  free(b->last);
  free(b);
}
int gmx_batch_n_pad(const gmx_batch* b) { return b->n_pad; }
uint64_t gmx_batch_max_bits(const gmx_batch* b) { return b->T; }
int gmx_shim_batch_m(const gmx_batch* b) { return b->g->m; } /* for gmx_lstm_feed in gmx_abi_oracle_shim2.c */
int gmx_batch_mask_words(const gmx_batch* b) { return b->mw; }
float* gmx_batch_predictions(gmx_batch* b) { return b->pred; }
uint32_t* gmx_batch_active_mask(gmx_batch* b) { return b->mask; }
uint32_t* gmx_batch_contexts(gmx_batch* b) { return b->ctx; }
uint8_t* gmx_batch_bits(gmx_batch* b) { return b->bits; }
const float* gmx_batch_p(gmx_batch* b) { return b->p; }
const float* gmx_batch_outputs(gmx_batch* b) { return (b->flags & GMX_BATCH_OUTPUTS) ? b->out : 0; }
const float* gmx_batch_last_outputs(gmx_batch* b) { return (b->flags & GMX_BATCH_LAST_OUTPUTS) ? b->last : 0; }
int gmx_batch_upload(gmx_batch* b, uint64_t n) { return (b && n <= b->T) ? GMX_OK : GMX_ERR_INVALID; }
int gmx_batch_download(gmx_batch* b, uint64_t n) { return (b && n <= b->T) ? GMX_OK : GMX_ERR_INVALID; }
int gmx_batch_wait(gmx_batch* b) { return b ? GMX_OK : GMX_ERR_INVALID; }
int gmx_group_run_ragged(gmx_group* g, gmx_batch* b, const uint64_t* n_bits, int learn) {
  if (!g || !b || b->g != g || !n_bits) return GMX_ERR_INVALID;
  int32_t act[2048];
  for (int s = 0; s < g->S; ++s) {
    if (n_bits[s] > b->T) return GMX_ERR_INVALID;
    for (uint64_t t = 0; t < n_bits[s]; ++t) {
      const size_t r = (size_t)s * b->T + t;
      int na = 0;
      for (int i = 0; i < g->n; ++i)
        if (b->mask[r * b->mw + (i >> 5)] >> (i & 31) & 1u) act[na++] = i;
      b->p[r] = gmxo_predict(g->bs[s], b->pred + r * b->n_pad, act, na, b->ctx + r * g->m, b->out + r * g->m);
      if (learn) gmxo_learn(g->bs[s], b->bits[r]);
      if (t + 1 == n_bits[s]) memcpy(b->last + (size_t)s * g->m, b->out + r * g->m, (size_t)g->m * 4);
    }
  }
  return GMX_OK;
}

/* The two calls in a row, the Indirect models' results put into the mixers' inputs (the product hands them over
 * on the device).  The Indirect side lives in gmx_abi_oracle_shim2.c, linked only into the chain builds. */
extern int gmx_indirect_forward(gmx_indirect* ib, int stream, const uint32_t* contexts, uint32_t bit_context,
                                float* predictions, uint8_t* active) __attribute__((weak));
extern int gmx_shim_indirect_slots(gmx_indirect* ib, int* n, const int (**slots)[2]) __attribute__((weak));
int gmx_chain_forward(gmx_indirect* ib, gmx_group* g, int stream, const uint32_t* ind_contexts, uint32_t bit_context,
                      const float* predictions, const int32_t* active_models, int n_active, const uint32_t* contexts,
                      float* p_final, float* out_all, float* ind_predictions, uint8_t* ind_active) {
  if (!gmx_indirect_forward || !gmx_shim_indirect_slots || !ib || !g || stream < 0 || stream >= g->S || n_active < 0) return GMX_ERR_INVALID;
  int k = 0;
  const int (*slots)[2] = 0;
  gmx_shim_indirect_slots(ib, &k, &slots);
  float ip[128], pr[512];
  uint8_t ia[128], own[512], on[512];
  int32_t act[512];
  int rc = gmx_indirect_forward(ib, stream, ind_contexts, bit_context, ip, ia);
  if (rc) return rc;
  memset(own, 0, sizeof own);
  memset(on, 0, sizeof on);
  memcpy(pr, predictions, (size_t)g->n * sizeof(float));
  for (int i = 0; i < k; ++i)
    for (int h = 0; h < 2; ++h) {
      own[slots[i][h]] = 1;
      pr[slots[i][h]] = ip[2 * i + h];
      on[slots[i][h]] = ia[2 * i + h];
    }
  for (int i = 0; i < n_active; ++i)
    if (!own[active_models[i]]) on[active_models[i]] = 1;
  int na = 0;
  for (int i = 0; i < g->n; ++i)
    if (on[i]) act[na++] = i;
  rc = gmx_bank_forward(g, stream, pr, act, na, contexts, p_final, out_all);
  if (rc) return rc;
  if (ind_predictions) memcpy(ind_predictions, ip, (size_t)2 * k * sizeof(float));
  if (ind_active) memcpy(ind_active, ia, (size_t)2 * k);
  return GMX_OK;
}

int gmx_bank_learn(gmx_group* g, int stream, int bit) {
  if (!g || stream < 0 || stream >= g->S) return GMX_ERR_INVALID;
  gmxo_learn(g->bs[stream], bit);
  return GMX_OK;
}

int gmx_bank_export(gmx_group* g, int stream, void* long_buf, size_t* long_bytes, void* short_buf,
                    size_t* short_bytes) {
  if (!g || stream < 0 || stream >= g->S) return GMX_ERR_INVALID;
  gmxo_bank* bk = g->bs[stream];
  size_t nl = gmxo_export_long(bk, 0, 0), ns = gmxo_export_short(bk, 0, 0);
  if (long_buf) gmxo_export_long(bk, long_buf, nl);
  if (short_buf) gmxo_export_short(bk, short_buf, ns);
  *long_bytes = nl;
  *short_bytes = ns;
  return GMX_OK;
}

/* Mixer::ReadFromDisk x M + the mixer section of LongTermMemory::ReadFromDisk (mixer.cpp:184-188,
 * long-term-memory.cpp:134-149) */
int gmx_bank_import(gmx_group* g, int stream, const void* long_buf, size_t long_bytes, const void* short_buf,
                    size_t short_bytes) {
  if (!g || stream < 0 || stream >= g->S || short_bytes != (size_t)g->b->m * 24) return GMX_ERR_FORMAT;
  gmxo_bank* bk = g->bs[stream];
  const uint8_t* p = (const uint8_t*)long_buf;
  const uint8_t* end = p + long_bytes;
  const uint64_t* sh = (const uint64_t*)short_buf;
  for (int j = 0; j < bk->m; ++j) {
    gmxo_mixer* x = &bk->mx[j];
    memset(x->w, 0, sizeof(float) * (size_t)x->table_size * x->weight_size);
    memset(x->row_steps, 0, sizeof(uint64_t) * x->table_size);
    x->steps = sh[3 * j];
    x->max_steps = sh[3 * j + 1];
    x->contexts_seen = sh[3 * j + 2];
    if (end - p < 8) return GMX_ERR_FORMAT;
    uint32_t cnt, in;
    memcpy(&cnt, p, 4);
    memcpy(&in, p + 4, 4);
    p += 8;
    if (cnt && in != (uint32_t)x->weight_size) return GMX_ERR_FORMAT;
    for (uint32_t i = 0; i < cnt; ++i) {
      if ((size_t)(end - p) < 12 + 4 * (size_t)in) return GMX_ERR_FORMAT;
      uint32_t r;
      memcpy(&r, p, 4);
      if (r >= x->table_size) return GMX_ERR_FORMAT;
      memcpy(&x->row_steps[r], p + 4, 8);
      memcpy(x->w + (size_t)r * x->weight_size, p + 12, 4 * (size_t)in);
      p += 12 + 4 * (size_t)in;
    }
  }
  return p == end ? GMX_OK : GMX_ERR_FORMAT;
}

int gmx_bank_copy(gmx_group* dst, int ds, gmx_group* src, int ss) {
  if (!dst || !src || ds < 0 || ds >= dst->S || ss < 0 || ss >= src->S || dst->b->m != src->b->m) return GMX_ERR_INVALID;
  for (int j = 0; j < dst->b->m; ++j) {
    gmxo_mixer *a = &dst->bs[ds]->mx[j], *b = &src->bs[ss]->mx[j];
    if (a->table_size != b->table_size || a->weight_size != b->weight_size) return GMX_ERR_INVALID;
    memcpy(a->w, b->w, sizeof(float) * (size_t)a->table_size * a->weight_size);
    memcpy(a->row_steps, b->row_steps, sizeof(uint64_t) * a->table_size);
    a->steps = b->steps;
    a->max_steps = b->max_steps;
    a->contexts_seen = b->contexts_seen;
  }
  return GMX_OK;
}

int gmx_bank_memory_usage(gmx_group* g, int stream, int mixer, uint64_t* bytes) {
  if (!g || stream < 0 || stream >= g->S || mixer < 0 || mixer >= g->b->m) return GMX_ERR_INVALID;
  *bytes = gmxo_memory_usage(g->bs[stream], mixer);
  return GMX_OK;
}
