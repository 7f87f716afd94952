/* gmx_abi_oracle_shim2.c -- TEST INFRASTRUCTURE, CPU only (see gmx_abi_oracle_shim.c): the Indirect and
 * LSTM entry points gmx_model_adapter.h calls, answered by the oracle's restatements, so that the host
 * logic of gmx::GpuIndirect / gmx::GpuLstmModel can be run inside the real reference without a GPU
 * (oracle/_ref/ref_tester_chain_shim, tests/test_dropin_cpu.py).  Never shipped, never a fallback. */
#include "../../oracle/gmx_oracle_ind.c"
#include "../../oracle/gmx_oracle_lstm.c"

#include "../../include/gmxmix.h"

/* ---- Indirect ---------------------------------------------------------------------------------- */
struct gmx_indirect {
  gmxo_ind** bs; /* one oracle bank per stream */
  gmxo_ind* b;   /* == bs[0] */
  int S, n;
  int slot[64][2];
};
#define IND_BANK(ib, stream) ((ib) && (stream) >= 0 && (stream) < (ib)->S ? (ib)->bs[stream] : 0)
/* for gmx_chain_forward in gmx_abi_oracle_shim.c: where the models' predictions go on the blackboard */
int gmx_shim_indirect_slots(gmx_indirect* ib, int* n, const int (**slots)[2]) {
  *n = ib->n;
  *slots = ib->slot;
  return 0;
}

int gmx_indirect_create(gmx_indirect** out, const gmx_indirect_desc* models, int n_models, const uint8_t* ns_next,
                        const uint8_t* rm_next, int n_streams, int device) {
  (void)device;
  if (!out || n_streams < 1 || n_models > 64) return GMX_ERR_INVALID;
  uint32_t ts[64];
  float lr[64];
  for (int i = 0; i < n_models; ++i) {
    ts[i] = models[i].table_size;
    lr[i] = models[i].learning_rate;
  }
  gmx_indirect* ib = (gmx_indirect*)calloc(1, sizeof *ib);
  ib->n = n_models;
  for (int i = 0; i < n_models; ++i) {
    ib->slot[i][0] = models[i].slot_indirect;
    ib->slot[i][1] = models[i].slot_run_map;
  }
  ib->S = n_streams;
  ib->bs = (gmxo_ind**)calloc((size_t)n_streams, sizeof(gmxo_ind*));
  for (int s = 0; s < n_streams; ++s) ib->bs[s] = gmxo_ind_create(n_models, ts, lr, ns_next, rm_next);
  ib->b = ib->bs[0];
  *out = ib;
  return GMX_OK;
}
void gmx_indirect_destroy(gmx_indirect* ib) {
  if (!ib) return;
  for (int s = 0; s < ib->S; ++s) gmxo_ind_destroy(ib->bs[s]);
  free(ib->bs);
  free(ib);
}
int gmx_indirect_forward(gmx_indirect* ib, int stream, const uint32_t* contexts, uint32_t bit_context, float* predictions,
                         uint8_t* active) {
  if (!IND_BANK(ib, stream)) return GMX_ERR_INVALID;
  gmxo_ind_predict(ib->bs[stream], contexts, bit_context, predictions, active);
  return GMX_OK;
}
int gmx_indirect_learn(gmx_indirect* ib, int stream, int bit) {
  if (!IND_BANK(ib, stream)) return GMX_ERR_INVALID;
  gmxo_ind_learn(ib->bs[stream], bit);
  return GMX_OK;
}
int gmx_indirect_export(gmx_indirect* ib, int stream, void* buf, size_t* bytes) {
  if (!IND_BANK(ib, stream) || !bytes) return GMX_ERR_INVALID;
  const size_t n = gmxo_ind_export(ib->bs[stream], 0, 0);
  if (buf) gmxo_ind_export(ib->bs[stream], (uint8_t*)buf, n);
  *bytes = n;
  return GMX_OK;
}
/* the indirect section of LongTermMemory::ReadFromDisk (long-term-memory.cpp:111-132) into a bank */
int gmx_indirect_import(gmx_indirect* ib, int stream, const void* buf, size_t bytes) {
  if (!IND_BANK(ib, stream)) return GMX_ERR_INVALID;
  const uint8_t* p = (const uint8_t*)buf;
  const uint8_t* end = p + bytes;
  for (int i = 0; i < ib->b->k; ++i) {
    ind_model* m = &ib->bs[stream]->m[i];
    if (end - p < 4) return GMX_ERR_FORMAT;
    uint32_t cnt;
    memcpy(&cnt, p, 4);
    p += 4;
    memset(m->ns, 255, m->size);
    memset(m->rm, 0, m->size);
    if (cnt < m->size / 3) {
      if ((size_t)(end - p) < (size_t)cnt * 6) return GMX_ERR_FORMAT;
      for (uint32_t j = 0; j < cnt; ++j) {
        uint32_t key;
        memcpy(&key, p, 4);
        if (key >= m->size) return GMX_ERR_FORMAT;
        m->ns[key] = p[4];
        m->rm[key] = p[5];
        p += 6;
      }
    } else {
      if ((size_t)(end - p) < 2 * (size_t)m->size) return GMX_ERR_FORMAT;
      memcpy(m->ns, p, m->size);
      memcpy(m->rm, p + m->size, m->size);
      p += 2 * (size_t)m->size;
    }
    if (end - p < 2048) return GMX_ERR_FORMAT;
    memcpy(m->nsp, p, 1024);
    memcpy(m->rmp, p + 1024, 1024);
    p += 2048;
  }
  return p == end ? GMX_OK : GMX_ERR_FORMAT;
}
int gmx_indirect_copy(gmx_indirect* dst, int ds, gmx_indirect* src, int ss) {
  if (!IND_BANK(dst, ds) || !IND_BANK(src, ss) || dst->b->k != src->b->k) return GMX_ERR_INVALID;
  for (int i = 0; i < dst->b->k; ++i) {
    ind_model *a = &dst->bs[ds]->m[i], *b = &src->bs[ss]->m[i];
    if (a->size != b->size) return GMX_ERR_INVALID;
    memcpy(a->ns, b->ns, a->size);
    memcpy(a->rm, b->rm, a->size);
    memcpy(a->nsp, b->nsp, sizeof a->nsp);
    memcpy(a->rmp, b->rmp, sizeof a->rmp);
  }
  memcpy(dst->bs[ds]->pred, src->bs[ss]->pred, 2 * dst->b->k * sizeof(float));
  return GMX_OK;
}
int gmx_indirect_slots_get(gmx_indirect* ib, int stream, float* values) {
  if (!IND_BANK(ib, stream) || !values) return GMX_ERR_INVALID;
  memcpy(values, ib->bs[stream]->pred, 2 * ib->b->k * sizeof(float));
  return GMX_OK;
}
int gmx_indirect_slots_set(gmx_indirect* ib, int stream, const float* values) {
  if (!IND_BANK(ib, stream) || !values) return GMX_ERR_INVALID;
  memcpy(ib->bs[stream]->pred, values, 2 * ib->b->k * sizeof(float));
  return GMX_OK;
}
int gmx_indirect_memory_usage(gmx_indirect* ib, int model, uint64_t* bytes) {
  if (!ib || model < 0 || model >= ib->b->k) return GMX_ERR_INVALID;
  *bytes = gmxo_ind_memory_usage(ib->b, model);
  return GMX_OK;
}

/* ---- LSTM -------------------------------------------------------------------------------------- */
/* the byte the last Lstm::Perceive stored (lstm.cpp:53-55): what a bank remembers as last_byte after an import */
static uint32_t gmxo_lstm_last_input(const gmxo_lstm* l) {
  int last_epoch = (int)l->epoch - 1;
  if (last_epoch == -1) last_epoch = H - 1;
  return l->input_history[last_epoch];
}
typedef struct lstm_stream {
  gmxo_lstm* l;
  int forward_pending; /* the product refuses checkpoints between forward and perceive: so does this */
  uint32_t last_byte, context; /* what the bank remembers between batched runs */
  float prediction;
} lstm_stream;
struct gmx_lstm {
  lstm_stream* st;
  int S;
};
#define LSTM_ST(l, stream) ((l) && (stream) >= 0 && (stream) < (l)->S ? &(l)->st[stream] : 0)

int gmx_lstm_create(gmx_lstm** out, int n_streams, int device) {
  (void)device;
  if (!out || n_streams < 1) return GMX_ERR_INVALID;
  gmx_lstm* l = (gmx_lstm*)calloc(1, sizeof *l);
  l->S = n_streams;
  l->st = (lstm_stream*)calloc((size_t)n_streams, sizeof(lstm_stream));
  /* gmxo_lstm_create draws the reference's initial weights from rand() (it restates LstmLayer's constructor); the
   * product's gmx_lstm_create touches no generator, and the reference's generation test samples from rand() afterwards
   * (runner-utils.cpp:19, tester.cpp:296): the draws are made on a state of their own (glibc: rand() is random()) */
  char scratch[256];
  char* caller_state = initstate(1u, scratch, sizeof scratch);
  for (int s = 0; s < n_streams; ++s) l->st[s].l = gmxo_lstm_create();
  setstate(caller_state);
  *out = l;
  return GMX_OK;
}
void gmx_lstm_destroy(gmx_lstm* l) {
  if (!l) return;
  for (int s = 0; s < l->S; ++s) gmxo_lstm_destroy(l->st[s].l);
  free(l->st);
  free(l);
}
int gmx_lstm_set_weights(gmx_lstm* l, int stream, const float* weights) {
  if (!LSTM_ST(l, stream)) return GMX_ERR_INVALID;
  gmxo_lstm_set_weights(l->st[stream].l, weights);
  return GMX_OK;
}
int gmx_lstm_forward(gmx_lstm* l, int stream, int last_byte, const float* ppm, float* probs, uint32_t* context) {
  if (!LSTM_ST(l, stream)) return GMX_ERR_INVALID;
  float pr[256];
  uint32_t ctx = 0;
  gmxo_lstm_predict_byte(l->st[stream].l, ppm, (uint32_t)last_byte, pr, &ctx);
  if (probs) memcpy(probs, pr, sizeof pr);
  if (context) *context = ctx;
  l->st[stream].context = ctx;
  l->st[stream].forward_pending = 1;
  return GMX_OK;
}
int gmx_lstm_perceive(gmx_lstm* l, int stream, int byte) {
  if (!LSTM_ST(l, stream)) return GMX_ERR_INVALID;
  gmxo_lstm_perceive_byte(l->st[stream].l, (uint32_t)byte);
  l->st[stream].last_byte = (uint32_t)byte;
  l->st[stream].forward_pending = 0;
  return GMX_OK;
}
int gmx_lstm_export(gmx_lstm* l, int stream, void* long_buf, size_t* long_bytes, void* short_buf, size_t* short_bytes) {
  if (!LSTM_ST(l, stream)) return GMX_ERR_INVALID;
  lstm_stream* st = &l->st[stream];
  *long_bytes = gmxo_lstm_export_long(st->l, 0);
  *short_bytes = gmxo_lstm_export_short(st->l, 0);
  if (long_buf) gmxo_lstm_export_long(st->l, (uint8_t*)long_buf);
  if (short_buf) gmxo_lstm_export_short(st->l, (uint8_t*)short_buf);
  return GMX_OK;
}
int gmx_lstm_import(gmx_lstm* l, int stream, const void* long_buf, size_t long_bytes, const void* short_buf,
                    size_t short_bytes) {
  if (!LSTM_ST(l, stream)) return GMX_ERR_INVALID;
  lstm_stream* st = &l->st[stream];
  if (gmxo_lstm_import_long(st->l, (const uint8_t*)long_buf, long_bytes)) return GMX_ERR_FORMAT;
  if (gmxo_lstm_import_short(st->l, (const uint8_t*)short_buf, short_bytes)) return GMX_ERR_FORMAT;
  st->forward_pending = 1; /* (the file does not say: include/gmxmix.h) */
  st->last_byte = gmxo_lstm_last_input(st->l); /* the newest entry of input_history_ (include/gmxmix.h) */
  return GMX_OK;
}
int gmx_lstm_copy(gmx_lstm* dst, int ds, gmx_lstm* src, int ss) {
  if (!LSTM_ST(dst, ds) || !LSTM_ST(src, ss)) return GMX_ERR_INVALID;
  lstm_stream *d = &dst->st[ds], *sr = &src->st[ss];
  const size_t nl = gmxo_lstm_export_long(sr->l, 0), ns = gmxo_lstm_export_short(sr->l, 0);
  uint8_t* a = (uint8_t*)malloc(nl);
  uint8_t* b = (uint8_t*)malloc(ns);
  gmxo_lstm_export_long(sr->l, a);
  gmxo_lstm_export_short(sr->l, b);
  gmxo_lstm_import_long(d->l, a, nl);
  gmxo_lstm_import_short(d->l, b, ns);
  d->forward_pending = sr->forward_pending;
  d->last_byte = sr->last_byte;
  d->context = sr->context;
  d->prediction = sr->prediction;
  free(a);
  free(b);
  return GMX_OK;
}
int gmx_lstm_memory_usage(gmx_lstm* l, uint64_t* bytes) {
  if (!l) return GMX_ERR_INVALID;
  *bytes = 7017924ull; /* LstmModel::GetMemoryUsage (lstm-model.cpp:87-101): a constant of the architecture, the value
                          the reference's own model reports in tests/golden/lstm_*.npz */
  return GMX_OK;
}

/* ---- the batched surfaces the run-ahead compressor uses with the whole chain on the device
 * (gmix_amd/host/gmx_model_adapter.h, MixerPool::Lead): plain host arrays; the "device work" is done by the oracle when
 * the run / feed calls are made, in the order the product queues it ---- */
int gmx_indirect_n_models(const gmx_indirect* ib) { return ib ? ib->n : GMX_ERR_INVALID; }
struct gmx_ind_batch {
  gmx_indirect* ib;
  uint64_t T;
  uint32_t *ctx, *bc;
  uint8_t *bits, *act;
  float* pred;
};
int gmx_ind_batch_create(gmx_ind_batch** out, gmx_indirect* ib, uint64_t max_bits) {
  if (!out || !ib || !max_bits) return GMX_ERR_INVALID;
  gmx_ind_batch* b = (gmx_ind_batch*)calloc(1, sizeof *b);
  const size_t R = (size_t)ib->S * max_bits;
  b->ib = ib;
  b->T = max_bits;
  b->ctx = (uint32_t*)calloc(R * ib->n, 4);
  b->bc = (uint32_t*)calloc(R, 4);
  b->bits = (uint8_t*)calloc(R, 1);
  b->pred = (float*)calloc(R * 2 * ib->n, 4);
  b->act = (uint8_t*)calloc(R * 2 * ib->n, 1);
  *out = b;
  return GMX_OK;
}
void gmx_ind_batch_destroy(gmx_ind_batch* b) {
  if (!b) return;
  free(b->ctx);
  free(b->bc);
  free(b->bits);
  free(b->pred);
  free(b->act);
  free(b);
}
uint32_t* gmx_ind_batch_contexts(gmx_ind_batch* b) { return b->ctx; }
uint32_t* gmx_ind_batch_bit_contexts(gmx_ind_batch* b) { return b->bc; }
uint8_t* gmx_ind_batch_bits(gmx_ind_batch* b) { return b->bits; }
const float* gmx_ind_batch_predictions(gmx_ind_batch* b) { return b->pred; }
const uint8_t* gmx_ind_batch_active(gmx_ind_batch* b) { return b->act; }
int gmx_ind_batch_upload(gmx_ind_batch* b, uint64_t n) { return (b && n <= b->T) ? GMX_OK : GMX_ERR_INVALID; }
int gmx_ind_batch_download(gmx_ind_batch* b, uint64_t n) { return (b && n <= b->T) ? GMX_OK : GMX_ERR_INVALID; }
int gmx_ind_batch_wait(gmx_ind_batch* b) { return b ? GMX_OK : GMX_ERR_INVALID; }
/* the mixer batch's arrays (gmx_abi_oracle_shim.c) */
extern float* gmx_batch_predictions(gmx_batch* b);
extern uint32_t* gmx_batch_active_mask(gmx_batch* b);
extern uint32_t* gmx_batch_contexts(gmx_batch* b);
extern uint8_t* gmx_batch_bits(gmx_batch* b);
extern int gmx_batch_n_pad(const gmx_batch* b);
extern int gmx_batch_mask_words(const gmx_batch* b);
extern uint64_t gmx_batch_max_bits(const gmx_batch* b);
extern int gmx_shim_batch_m(const gmx_batch* b);
int gmx_indirect_run_ragged(gmx_indirect* ib, gmx_ind_batch* b, const uint64_t* n_bits, int learn, gmx_batch* into) {
  if (!ib || !b || b->ib != ib || !n_bits) return GMX_ERR_INVALID;
  const int k = ib->n;
  for (int s = 0; s < ib->S; ++s) {
    if (n_bits[s] > b->T) return GMX_ERR_INVALID;
    const size_t r0 = (size_t)s * b->T;
    gmxo_ind_run(ib->bs[s], n_bits[s], b->ctx + r0 * k, b->bc + r0, b->bits + r0, learn ? n_bits[s] : 0,
                 b->pred + r0 * 2 * k, b->act + r0 * 2 * k);
    if (!into) continue;
    /* gmx_indirect_run's `into`: predictions at the models' slots, their active bits, the coded bits */
    const int n_pad = gmx_batch_n_pad(into), mw = gmx_batch_mask_words(into);
    const uint64_t MT = gmx_batch_max_bits(into);
    for (uint64_t t = 0; t < n_bits[s]; ++t) {
      float* mp = gmx_batch_predictions(into) + ((size_t)s * MT + t) * n_pad;
      uint32_t* mm = gmx_batch_active_mask(into) + ((size_t)s * MT + t) * mw;
      for (int i = 0; i < k; ++i)
        for (int h = 0; h < 2; ++h) {
          const int slot = ib->slot[i][h];
          mp[slot] = b->pred[(r0 + t) * 2 * k + 2 * i + h];
          if (b->act[(r0 + t) * 2 * k + 2 * i + h])
            mm[slot >> 5] |= 1u << (slot & 31);
          else
            mm[slot >> 5] &= ~(1u << (slot & 31));
        }
      gmx_batch_bits(into)[(size_t)s * MT + t] = b->bits[r0 + t];
    }
  }
  return GMX_OK;
}

struct gmx_lstm_batch {
  gmx_lstm* l;
  uint64_t NB;
  float *ppm, *pred;
  uint8_t *bytes, *act;
  uint32_t* ctx;
};
int gmx_lstm_batch_create(gmx_lstm_batch** out, gmx_lstm* l, uint64_t max_bytes) {
  if (!out || !l || !max_bytes) return GMX_ERR_INVALID;
  gmx_lstm_batch* b = (gmx_lstm_batch*)calloc(1, sizeof *b);
  const size_t R = (size_t)l->S * max_bytes;
  b->l = l;
  b->NB = max_bytes;
  b->ppm = (float*)calloc(R * 256, 4);
  b->bytes = (uint8_t*)calloc(R, 1);
  b->pred = (float*)calloc(R * 8, 4);
  b->act = (uint8_t*)calloc(R * 8, 1);
  b->ctx = (uint32_t*)calloc(R, 4);
  *out = b;
  return GMX_OK;
}
void gmx_lstm_batch_destroy(gmx_lstm_batch* b) {
  if (!b) return;
  free(b->ppm);
  free(b->bytes);
  free(b->pred);
  free(b->act);
  free(b->ctx);
  free(b);
}
float* gmx_lstm_batch_ppm(gmx_lstm_batch* b) { return b->ppm; }
uint8_t* gmx_lstm_batch_bytes(gmx_lstm_batch* b) { return b->bytes; }
const float* gmx_lstm_batch_predictions(gmx_lstm_batch* b) { return b->pred; }
const uint8_t* gmx_lstm_batch_active(gmx_lstm_batch* b) { return b->act; }
const uint32_t* gmx_lstm_batch_contexts(gmx_lstm_batch* b) { return b->ctx; }
int gmx_lstm_batch_upload(gmx_lstm_batch* b, uint64_t n) { return (b && n <= b->NB) ? GMX_OK : GMX_ERR_INVALID; }
int gmx_lstm_batch_download(gmx_lstm_batch* b, uint64_t n) { return (b && n <= b->NB) ? GMX_OK : GMX_ERR_INVALID; }
int gmx_lstm_batch_wait(gmx_lstm_batch* b) { return b ? GMX_OK : GMX_ERR_INVALID; }
int gmx_lstm_run_ragged(gmx_lstm* l, gmx_lstm_batch* b, const uint64_t* n_bytes, int learn) {
  if (!l || !b || b->l != l || !n_bytes) return GMX_ERR_INVALID;
  for (int s = 0; s < l->S; ++s) {
    if (n_bytes[s] > b->NB) return GMX_ERR_INVALID;
    if (!n_bytes[s]) continue;
    lstm_stream* st = &l->st[s];
    const size_t r0 = (size_t)s * b->NB;
    gmxo_lstm_run(st->l, n_bytes[s], b->ppm + r0 * 256, b->bytes + r0, learn, &st->last_byte, &st->prediction,
                  &st->context, b->pred + r0 * 8, b->act + r0 * 8, b->ctx + r0);
  }
  return GMX_OK;
}
int gmx_lstm_feed(gmx_lstm* l, gmx_lstm_batch* b, uint64_t n_bytes, gmx_batch* mb, int slot, int mixer_ctx_col,
                  gmx_ind_batch* ib, int ind_ctx_col) {
  if (!l || !b || b->l != l || n_bytes > b->NB) return GMX_ERR_INVALID;
  for (int s = 0; s < l->S; ++s)
    for (uint64_t n = 0; n < n_bytes; ++n)
      for (int k = 0; k < 8; ++k) {
        const size_t r = (size_t)s * b->NB + n;
        const uint64_t t = 8 * n + k;
        if (mb) {
          const uint64_t MT = gmx_batch_max_bits(mb);
          const int n_pad = gmx_batch_n_pad(mb), mw = gmx_batch_mask_words(mb), m = gmx_shim_batch_m(mb);
          gmx_batch_predictions(mb)[((size_t)s * MT + t) * n_pad + slot] = b->pred[r * 8 + k];
          uint32_t* w = gmx_batch_active_mask(mb) + ((size_t)s * MT + t) * mw + (slot >> 5);
          *w = b->act[r * 8 + k] ? (*w | 1u << (slot & 31)) : (*w & ~(1u << (slot & 31)));
          if (mixer_ctx_col >= 0) gmx_batch_contexts(mb)[((size_t)s * MT + t) * m + mixer_ctx_col] = b->ctx[r];
        }
        if (ib) ib->ctx[((size_t)s * ib->T + t) * ib->ib->n + ind_ctx_col] = b->ctx[r];
      }
  return GMX_OK;
}


/* ---- gmx_chainstep: S decoders in lock step through LSTM -> Indirect models -> mixers, one step per coded bit
 * (gmx_amd/csrc/gmx_chainstep.inc), the "device work" done by the oracle stream by stream ---- */
extern int gmx_group_n_mixers(const gmx_group* g);
extern int gmx_group_n_inputs(const gmx_group* g);
extern int gmx_group_n_streams(const gmx_group* g);
extern int gmx_bank_forward(gmx_group* g, int stream, const float* predictions, const int32_t* active, int n_active,
                            const uint32_t* contexts, float* p_final, float* out_all);
extern int gmx_bank_learn(gmx_group* g, int stream, int bit);
typedef struct cs_stream {
  int pending, recent_bits, new_bit;
  uint32_t last_byte, context;
  float prediction;
  float cur_ppm[256];
} cs_stream;
struct gmx_chainstep {
  gmx_group* g;
  gmx_indirect* ib;
  gmx_lstm* l;
  int S, n, n_pad, mw, m, k, lstm_slot, mcol, icol;
  float *pred, *ppm, *p, *outs;
  uint32_t *mask, *ctx, *ictx, *ibc;
  uint8_t *bits, *what;
  cs_stream* st;
};
int gmx_chainstep_create(gmx_chainstep** out, gmx_group* g, gmx_indirect* ib, gmx_lstm* l, int lstm_slot, int mixer_ctx_col,
                         int ind_ctx_col) {
  if (!out || !g) return GMX_ERR_INVALID;
  gmx_chainstep* cs = (gmx_chainstep*)calloc(1, sizeof *cs);
  cs->g = g;
  cs->ib = ib;
  cs->l = l;
  cs->S = gmx_group_n_streams(g);
  cs->n = gmx_group_n_inputs(g);
  cs->m = gmx_group_n_mixers(g);
  cs->n_pad = (cs->n + 3) / 4 * 4;
  cs->mw = (cs->n + 31) / 32;
  cs->k = ib ? ib->n : 0;
  cs->lstm_slot = lstm_slot;
  cs->mcol = l ? mixer_ctx_col : -1;
  cs->icol = (l && ib) ? ind_ctx_col : -1;
  const size_t S = (size_t)cs->S;
  cs->pred = (float*)calloc(S * cs->n_pad, 4);
  cs->mask = (uint32_t*)calloc(S * cs->mw, 4);
  cs->ctx = (uint32_t*)calloc(S * cs->m, 4);
  cs->ictx = (uint32_t*)calloc(S * (cs->k ? cs->k : 1), 4);
  cs->ibc = (uint32_t*)calloc(S, 4);
  cs->ppm = (float*)calloc(S * 256, 4);
  cs->p = (float*)calloc(S, 4);
  cs->outs = (float*)calloc(S * cs->m, 4);
  cs->bits = (uint8_t*)calloc(S, 1);
  cs->what = (uint8_t*)calloc(S, 1);
  cs->st = (cs_stream*)calloc(S, sizeof(cs_stream));
  for (int s = 0; s < cs->S; ++s) cs->st[s].recent_bits = 1;
  *out = cs;
  return GMX_OK;
}
void gmx_chainstep_destroy(gmx_chainstep* cs) {
  if (!cs) return;
  void* v[] = {cs->pred, cs->mask, cs->ctx, cs->ictx, cs->ibc, cs->ppm, cs->p, cs->outs, cs->bits, cs->what, cs->st};
  for (size_t i = 0; i < sizeof v / sizeof v[0]; ++i) free(v[i]);
  free(cs);
}
int gmx_chainstep_n_streams(const gmx_chainstep* cs) { return cs ? cs->S : GMX_ERR_INVALID; }
float* gmx_chainstep_predictions(gmx_chainstep* cs) { return cs->pred; }
uint32_t* gmx_chainstep_active_mask(gmx_chainstep* cs) { return cs->mask; }
uint32_t* gmx_chainstep_contexts(gmx_chainstep* cs) { return cs->ctx; }
uint32_t* gmx_chainstep_ind_contexts(gmx_chainstep* cs) { return cs->ib ? cs->ictx : 0; }
uint32_t* gmx_chainstep_bit_contexts(gmx_chainstep* cs) { return cs->ib ? cs->ibc : 0; }
float* gmx_chainstep_ppm(gmx_chainstep* cs) { return cs->l ? cs->ppm : 0; }
uint8_t* gmx_chainstep_bits(gmx_chainstep* cs) { return cs->bits; }
uint8_t* gmx_chainstep_what(gmx_chainstep* cs) { return cs->what; }
const float* gmx_chainstep_p(gmx_chainstep* cs) { return cs->p; }
const float* gmx_chainstep_outputs(gmx_chainstep* cs) { return cs->outs; }
int gmx_chainstep_commit(gmx_chainstep* cs, int s) { return (cs && s >= 0 && s < cs->S) ? GMX_OK : GMX_ERR_INVALID; }
int gmx_chainstep_step(gmx_chainstep* cs);
int gmx_chainstep_launch(gmx_chainstep* cs) { return gmx_chainstep_step(cs); }  /* (the CPU stand-in has nothing to overlap) */
int gmx_chainstep_wait(gmx_chainstep* cs) { return cs ? GMX_OK : GMX_ERR_INVALID; }
int gmx_chainstep_step(gmx_chainstep* cs) {
  if (!cs) return GMX_ERR_INVALID;
  for (int s = 0; s < cs->S; ++s) {
    const uint8_t w = cs->what[s];
    cs_stream* st = &cs->st[s];
    if ((w & GMX_STEP_LEARN) && !st->pending) return GMX_ERR_STATE;
    if ((w & GMX_STEP_PREDICT) && st->pending && !(w & GMX_STEP_LEARN)) return GMX_ERR_STATE;
  }
  {
    int any = 0;
    for (int s = 0; s < cs->S; ++s) any |= cs->what[s];
    if (!any) return GMX_OK;
    for (int s = 0; s < cs->S; ++s)  /* a Learn belongs to the step right behind its Predict (gmxmix.h) */
      if (cs->st[s].pending && !(cs->what[s] & GMX_STEP_LEARN)) return GMX_ERR_STATE;
  }
  for (int s = 0; s < cs->S; ++s) {
    const uint8_t w = cs->what[s];
    cs_stream* st = &cs->st[s];
    float* pr = cs->pred + (size_t)s * cs->n_pad;
    uint32_t* mk = cs->mask + (size_t)s * cs->mw;
    uint32_t* cx = cs->ctx + (size_t)s * cs->m;
    uint32_t* icx = cs->ictx + (size_t)s * (cs->k ? cs->k : 1);
    if (w & GMX_STEP_LEARN) {
      const int bit = cs->bits[s] ? 1 : 0;
      gmx_bank_learn(cs->g, s, bit);
      if (cs->ib) gmxo_ind_learn(cs->ib->bs[s], bit);
      if (cs->l) gmxo_lstm_model_learn(cs->l->st[s].l, st->recent_bits, bit);
      st->new_bit = bit;
      st->recent_bits += st->recent_bits + bit;
      if (st->recent_bits >= 256) {
        st->last_byte = (uint32_t)(st->recent_bits - 256);
        st->recent_bits = 1;
      }
      st->pending = 0;
    }
    if (w & GMX_STEP_PREDICT) {
      if (cs->l) {
        if (st->recent_bits == 1) memcpy(st->cur_ppm, cs->ppm + (size_t)s * 256, sizeof st->cur_ppm);
        const int act = gmxo_lstm_model_predict(cs->l->st[s].l, st->recent_bits, st->last_byte, st->new_bit, st->cur_ppm,
                                                &st->prediction, &st->context, 0);
        pr[cs->lstm_slot] = st->prediction;
        if (act)
          mk[cs->lstm_slot >> 5] |= 1u << (cs->lstm_slot & 31);
        else
          mk[cs->lstm_slot >> 5] &= ~(1u << (cs->lstm_slot & 31));
        if (cs->mcol >= 0) cx[cs->mcol] = st->context;
        if (cs->icol >= 0) icx[cs->icol] = st->context;
      }
      if (cs->ib) {
        float ip[128];
        uint8_t ia[128];
        gmxo_ind_predict(cs->ib->bs[s], icx, cs->ibc[s], ip, ia);
        for (int i = 0; i < cs->k; ++i)
          for (int h = 0; h < 2; ++h) {
            const int slot = cs->ib->slot[i][h];
            pr[slot] = ip[2 * i + h];
            if (ia[2 * i + h]) mk[slot >> 5] |= 1u << (slot & 31);
          }
      }
      int32_t act[2048];
      int na = 0;
      for (int i = 0; i < cs->n; ++i)
        if (mk[i >> 5] >> (i & 31) & 1u) act[na++] = i;
      gmx_bank_forward(cs->g, s, pr, act, na, cx, &cs->p[s], cs->outs + (size_t)s * cs->m);
      st->pending = 1;
    }
  }
  return GMX_OK;
}
