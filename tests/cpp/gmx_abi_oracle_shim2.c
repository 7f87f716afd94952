/* gmx_abi_oracle_shim2.c -- TEST INFRASTRUCTURE, CPU only (see gmx_abi_oracle_shim.c): the Indirect and
 * LSTM entry points gmx_model_adapter.h calls, answered by the oracle's restatements, so that the host
 * logic of gmx::GpuIndirect / gmx::GpuLstmModel can be run inside the real reference without a GPU
 * (oracle/_ref/ref_tester_chain_shim, tests/test_dropin_cpu.py).  Never shipped, never a fallback. */
#include "../../oracle/gmx_oracle_ind.c"
#include "../../oracle/gmx_oracle_lstm.c"

#include "../../include/gmxmix.h"

/* ---- Indirect ---------------------------------------------------------------------------------- */
struct gmx_indirect {
  gmxo_ind* b;
  int n;
  int slot[64][2];
};
/* for gmx_chain_forward in gmx_abi_oracle_shim.c: where the models' predictions go on the blackboard */
int gmx_shim_indirect_slots(gmx_indirect* ib, int* n, const int (**slots)[2]) {
  *n = ib->n;
  *slots = ib->slot;
  return 0;
}

int gmx_indirect_create(gmx_indirect** out, const gmx_indirect_desc* models, int n_models, const uint8_t* ns_next,
                        const uint8_t* rm_next, int n_streams, int device) {
  (void)device;
  if (!out || n_streams != 1 || n_models > 64) return GMX_ERR_INVALID;
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
  ib->b = gmxo_ind_create(n_models, ts, lr, ns_next, rm_next);
  *out = ib;
  return GMX_OK;
}
void gmx_indirect_destroy(gmx_indirect* ib) {
  if (!ib) return;
  gmxo_ind_destroy(ib->b);
  free(ib);
}
int gmx_indirect_forward(gmx_indirect* ib, int stream, const uint32_t* contexts, uint32_t bit_context, float* predictions,
                         uint8_t* active) {
  if (!ib || stream) return GMX_ERR_INVALID;
  gmxo_ind_predict(ib->b, contexts, bit_context, predictions, active);
  return GMX_OK;
}
int gmx_indirect_learn(gmx_indirect* ib, int stream, int bit) {
  if (!ib || stream) return GMX_ERR_INVALID;
  gmxo_ind_learn(ib->b, bit);
  return GMX_OK;
}
int gmx_indirect_export(gmx_indirect* ib, int stream, void* buf, size_t* bytes) {
  if (!ib || stream || !bytes) return GMX_ERR_INVALID;
  const size_t n = gmxo_ind_export(ib->b, 0, 0);
  if (buf) gmxo_ind_export(ib->b, (uint8_t*)buf, n);
  *bytes = n;
  return GMX_OK;
}
/* the indirect section of LongTermMemory::ReadFromDisk (long-term-memory.cpp:111-132) into a bank */
int gmx_indirect_import(gmx_indirect* ib, int stream, const void* buf, size_t bytes) {
  if (!ib || stream) return GMX_ERR_INVALID;
  const uint8_t* p = (const uint8_t*)buf;
  const uint8_t* end = p + bytes;
  for (int i = 0; i < ib->b->k; ++i) {
    ind_model* m = &ib->b->m[i];
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
  if (!dst || !src || ds || ss || dst->b->k != src->b->k) return GMX_ERR_INVALID;
  for (int i = 0; i < dst->b->k; ++i) {
    ind_model *a = &dst->b->m[i], *b = &src->b->m[i];
    if (a->size != b->size) return GMX_ERR_INVALID;
    memcpy(a->ns, b->ns, a->size);
    memcpy(a->rm, b->rm, a->size);
    memcpy(a->nsp, b->nsp, sizeof a->nsp);
    memcpy(a->rmp, b->rmp, sizeof a->rmp);
  }
  memcpy(dst->b->pred, src->b->pred, 2 * dst->b->k * sizeof(float));
  return GMX_OK;
}
int gmx_indirect_memory_usage(gmx_indirect* ib, int model, uint64_t* bytes) {
  if (!ib || model < 0 || model >= ib->b->k) return GMX_ERR_INVALID;
  *bytes = gmxo_ind_memory_usage(ib->b, model);
  return GMX_OK;
}

/* ---- LSTM -------------------------------------------------------------------------------------- */
struct gmx_lstm {
  gmxo_lstm* l;
  int forward_pending; /* the product refuses checkpoints between forward and perceive: so does this */
};

int gmx_lstm_create(gmx_lstm** out, int n_streams, int device) {
  (void)device;
  if (!out || n_streams != 1) return GMX_ERR_INVALID;
  gmx_lstm* l = (gmx_lstm*)calloc(1, sizeof *l);
  l->l = gmxo_lstm_create();
  *out = l;
  return GMX_OK;
}
void gmx_lstm_destroy(gmx_lstm* l) {
  if (!l) return;
  gmxo_lstm_destroy(l->l);
  free(l);
}
int gmx_lstm_set_weights(gmx_lstm* l, int stream, const float* weights) {
  if (!l || stream) return GMX_ERR_INVALID;
  gmxo_lstm_set_weights(l->l, weights);
  return GMX_OK;
}
int gmx_lstm_forward(gmx_lstm* l, int stream, int last_byte, const float* ppm, float* probs, uint32_t* context) {
  if (!l || stream) return GMX_ERR_INVALID;
  float pr[256];
  uint32_t ctx = 0;
  gmxo_lstm_predict_byte(l->l, ppm, (uint32_t)last_byte, pr, &ctx);
  if (probs) memcpy(probs, pr, sizeof pr);
  if (context) *context = ctx;
  l->forward_pending = 1;
  return GMX_OK;
}
int gmx_lstm_perceive(gmx_lstm* l, int stream, int byte) {
  if (!l || stream) return GMX_ERR_INVALID;
  gmxo_lstm_perceive_byte(l->l, (uint32_t)byte);
  l->forward_pending = 0;
  return GMX_OK;
}
int gmx_lstm_export(gmx_lstm* l, int stream, void* long_buf, size_t* long_bytes, void* short_buf, size_t* short_bytes) {
  if (!l || stream) return GMX_ERR_INVALID;
  if ((long_buf || short_buf) && l->forward_pending) return GMX_ERR_STATE;
  *long_bytes = gmxo_lstm_export_long(l->l, 0);
  *short_bytes = gmxo_lstm_export_short(l->l, 0);
  if (long_buf) gmxo_lstm_export_long(l->l, (uint8_t*)long_buf);
  if (short_buf) gmxo_lstm_export_short(l->l, (uint8_t*)short_buf);
  return GMX_OK;
}
int gmx_lstm_import(gmx_lstm* l, int stream, const void* long_buf, size_t long_bytes, const void* short_buf,
                    size_t short_bytes) {
  if (!l || stream) return GMX_ERR_INVALID;
  if (gmxo_lstm_import_long(l->l, (const uint8_t*)long_buf, long_bytes)) return GMX_ERR_FORMAT;
  if (gmxo_lstm_import_short(l->l, (const uint8_t*)short_buf, short_bytes)) return GMX_ERR_FORMAT;
  l->forward_pending = 0;
  return GMX_OK;
}
int gmx_lstm_copy(gmx_lstm* dst, int ds, gmx_lstm* src, int ss) {
  if (!dst || !src || ds || ss) return GMX_ERR_INVALID;
  const size_t nl = gmxo_lstm_export_long(src->l, 0), ns = gmxo_lstm_export_short(src->l, 0);
  uint8_t* a = (uint8_t*)malloc(nl);
  uint8_t* b = (uint8_t*)malloc(ns);
  gmxo_lstm_export_long(src->l, a);
  gmxo_lstm_export_short(src->l, b);
  gmxo_lstm_import_long(dst->l, a, nl);
  gmxo_lstm_import_short(dst->l, b, ns);
  dst->forward_pending = src->forward_pending;
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
