/* gmx_oracle_lstm.c -- CPU restatement of the reference's LSTM byte model (SURVEY.md section 8f
 * rank 3).  TEST INFRASTRUCTURE: same rules as gmx_oracle.c.  Pinned by tests/golden/lstm_*.npz,
 * made by running the REAL LstmModel through oracle/ref_build/ref_lstm_harness.cpp.
 *
 * Restates (file:line in /root/reference/src):
 *   LstmModel::LstmModel / Predict / Learn   models/lstm-model.cpp:5-15, :17-49, :51-60
 *   Lstm::Lstm / SetInput / Perceive / Predict models/lstm.cpp:9-43, :45-50, :52-93, :95-123
 *   LstmLayer::LstmLayer (weight init)        models/lstm-layer.cpp:155-195
 *   LstmLayer::ForwardPass (both)             models/lstm-layer.cpp:197-241
 *   LstmLayer::BackwardPass (both), ClipGradients, Adam   :243-355, :10-35
 *   ShortTermMemory::SetPrediction, Sigmoid::Logit        short-term-memory.cpp:187-191, sigmoid.cpp:7-13
 * One layer (the reference builds Lstm(256, 256, 50, 1, 100, 0.03, 10)); the `layer > 0` branch of
 * the backward pass never runs and is not restated.  libm (tanhf, expf, logf, powf, sqrtf) and
 * rand() are the system's, exactly as in the reference; valarray expressions are written out
 * element by element in the order libstdc++ evaluates them (sums left to right from element 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float gmxo_logistic(float p); /* gmx_oracle.c */

enum { NI = 256, NO = 256, NC = 50, H = 100, LIN = NI + NC + 1 /* 307 */, W = LIN + NO /* 563 */, HID = NC + 1 };

typedef struct {
  float error[NC], ivar[H], gamma[NC], gamma_u[NC], gamma_m[NC], gamma_v[NC];
  float beta[NC], beta_u[NC], beta_m[NC], beta_v[NC];
  float state[H][NC], update[NC][W], m[NC][W], v[NC][W], transpose[W - NO - NI][NC], norm[H][NC];
  float weights[NC][W]; /* LongTermMemory::neuron_layer_weights[layer_index_] */
} neuron_layer;

typedef struct gmxo_lstm {
  /* Lstm */
  uint32_t input_history[H];
  float hidden[HID], hidden_error[NC];
  float layer_input[H][LIN];
  float output[H][NO];
  float (*out_layer)[NO][HID]; /* LongTermMemory::lstm_output_layer[H] */
  uint32_t epoch;
  /* LstmLayer */
  float state[NC], state_error[NC], stored_error[NC];
  float tanh_state[H][NC], input_gate_state[H][NC], last_state[H][NC];
  uint32_t l_epoch;
  uint64_t update_steps;
  neuron_layer* gate[3]; /* forget, input node, output */
  /* LstmModel */
  int top, mid, bot;
  float probs[NO];
} gmxo_lstm;

static const float kLearningRate = 0.03f, kClip = 10.0f;
static const uint64_t kUpdateLimit = 3000;

void gmxo_lstm_destroy(gmxo_lstm* l) {
  if (!l) return;
  for (int g = 0; g < 3; ++g) free(l->gate[g]);
  free(l->out_layer);
  free(l);
}

/* The constructor chain, consuming rand() like LstmLayer::LstmLayer (lstm-layer.cpp:179-194):
 * the caller has seeded rand() the way the reference's Predictor does (srand(0xDEADBEEF),
 * predictor.cpp:18) -- or passes its own weights afterwards with gmxo_lstm_set_weights. */
gmxo_lstm* gmxo_lstm_create(void) {
  gmxo_lstm* l = (gmxo_lstm*)calloc(1, sizeof *l);
  l->out_layer = calloc(H, sizeof *l->out_layer);
  for (int g = 0; g < 3; ++g) {
    l->gate[g] = (neuron_layer*)calloc(1, sizeof(neuron_layer));
    for (int i = 0; i < NC; ++i) l->gate[g]->gamma[i] = 1.0f;
  }
  l->hidden[HID - 1] = 1;
  for (int e = 0; e < H; ++e) {
    l->layer_input[e][LIN - 1] = 1;
    for (int i = 0; i < NO; ++i) l->output[e][i] = (float)(1.0 / NO);
  }
  float val = sqrtf(6.0f / (float)(NI + NO));
  float low = -val, range = 2 * val;
  for (int i = 0; i < NC; ++i) {
    for (int j = 0; j < W; ++j)
      for (int g = 0; g < 3; ++g)
        l->gate[g]->weights[i][j] = low + ((float)rand() / (float)RAND_MAX) * range;
    l->gate[0]->weights[i][W - 1] = 1;
  }
  l->top = 255;
  l->mid = 127;
  l->bot = 0;
  for (int i = 0; i < NO; ++i) l->probs[i] = (float)(1.0 / 256);
  return l;
}

void gmxo_lstm_get_weights(const gmxo_lstm* l, float* w /* [3][NC][W] */) {
  for (int g = 0; g < 3; ++g) memcpy(w + (size_t)g * NC * W, l->gate[g]->weights, sizeof l->gate[g]->weights);
}
void gmxo_lstm_set_weights(gmxo_lstm* l, const float* w) {
  for (int g = 0; g < 3; ++g) memcpy(l->gate[g]->weights, w + (size_t)g * NC * W, sizeof l->gate[g]->weights);
}

/* LstmLayer::ForwardPass(NeuronLayer&, ...) (lstm-layer.cpp:221-241) */
static void neuron_forward(gmxo_lstm* l, neuron_layer* n, const float* input, int symbol) {
  const uint32_t e = l->l_epoch;
  for (int i = 0; i < NC; ++i) {
    float f = n->weights[i][symbol];
    for (int j = 0; j < LIN; ++j) f += input[j] * n->weights[i][NO + j];
    n->norm[e][i] = f;
  }
  /* .sum() of a valarray EXPRESSION runs from the last element down (libstdc++
   * bits/valarray_after.h, _Expr::sum); .sum() of a plain valarray from the first up */
  float s = n->norm[e][NC - 1] * n->norm[e][NC - 1];
  for (int i = NC - 2; i >= 0; --i) s += n->norm[e][i] * n->norm[e][i];
  n->ivar[e] = 1.0f / sqrtf((s / NC) + 1e-5f);
  for (int i = 0; i < NC; ++i) n->norm[e][i] *= n->ivar[e];
  for (int i = 0; i < NC; ++i) n->state[e][i] = n->norm[e][i] * n->gamma[i] + n->beta[i];
}

/* LstmLayer::ForwardPass (lstm-layer.cpp:197-219) */
static void layer_forward(gmxo_lstm* l, const float* input, int symbol) {
  const uint32_t e = l->l_epoch;
  memcpy(l->last_state[e], l->state, sizeof l->state);
  for (int g = 0; g < 3; ++g) neuron_forward(l, l->gate[g], input, symbol);
  for (int i = 0; i < NC; ++i) {
    l->gate[0]->state[e][i] = gmxo_logistic(l->gate[0]->state[e][i]);
    l->gate[1]->state[e][i] = tanhf(l->gate[1]->state[e][i]);
    l->gate[2]->state[e][i] = gmxo_logistic(l->gate[2]->state[e][i]);
  }
  for (int i = 0; i < NC; ++i) l->input_gate_state[e][i] = 1.0f - l->gate[0]->state[e][i];
  for (int i = 0; i < NC; ++i) l->state[i] *= l->gate[0]->state[e][i];
  for (int i = 0; i < NC; ++i) l->state[i] += l->gate[1]->state[e][i] * l->input_gate_state[e][i];
  for (int i = 0; i < NC; ++i) l->tanh_state[e][i] = tanhf(l->state[i]);
  for (int i = 0; i < NC; ++i) l->hidden[i] = l->gate[2]->state[e][i] * l->tanh_state[e][i];
  if (++l->l_epoch == H) l->l_epoch = 0;
}

/* Lstm::Predict (lstm.cpp:95-123); Lstm::SetInput happened before (lstm-model.cpp:21) */
static const float* lstm_predict(gmxo_lstm* l, const float* ppm, uint32_t last_byte) {
  const uint32_t e = l->epoch;
  memcpy(l->layer_input[e], ppm, NI * sizeof(float));              /* SetInput */
  memcpy(l->layer_input[e] + NI, l->hidden, NC * sizeof(float));   /* lstm.cpp:98-100 */
  layer_forward(l, l->layer_input[e], (int)last_byte);
  float max_out = 0;
  for (int i = 0; i < NO; ++i) {
    float sum = 0;
    for (int j = 0; j < HID; ++j) sum += l->hidden[j] * l->out_layer[e][i][j];
    l->output[e][i] = sum;
    max_out = sum > max_out ? sum : max_out; /* std::max(sum, max_out) */
  }
  for (int i = 0; i < NO; ++i) l->output[e][i] = expf(l->output[e][i] - max_out);
  float s = l->output[e][0];
  for (int i = 1; i < NO; ++i) s += l->output[e][i];
  for (int i = 0; i < NO; ++i) l->output[e][i] /= s;
  if (++l->epoch == H) l->epoch = 0;
  return l->output[e];
}

static void clip(float* a) {
  for (int i = 0; i < NC; ++i) {
    if (a[i] < -kClip) a[i] = -kClip;
    else if (a[i] > kClip) a[i] = kClip;
  }
}

/* Adam (lstm-layer.cpp:12-35) on n elements */
static void adam(float* g, float* m, float* v, float* w, int n, float t, uint64_t limit) {
  const float beta1 = 0.025, beta2 = 0.9999, eps = 1e-6f;
  float alpha;
  if (t < limit) alpha = kLearningRate * 0.1f / sqrtf(5e-5f * t + 1.0f);
  else alpha = kLearningRate * 0.1f / sqrtf(5e-5f * limit + 1.0f);
  for (int j = 0; j < n; ++j) m[j] *= beta1;
  for (int j = 0; j < n; ++j) m[j] += (1.0f - beta1) * g[j];
  for (int j = 0; j < n; ++j) v[j] *= beta2;
  for (int j = 0; j < n; ++j) v[j] += (1.0f - beta2) * g[j] * g[j];
  float d1, d2;
  if (t < limit) {
    d1 = (float)(1.0f - powf(beta1, t));
    d2 = (float)(1.0f - powf(beta2, t));
  } else {
    d1 = (float)(1.0f - powf(beta1, limit));
    d2 = (float)(1.0f - powf(beta2, limit));
  }
  for (int j = 0; j < n; ++j) w[j] -= alpha * ((m[j] / d1) / (sqrtf(v[j] / d2 + eps)));
}

/* LstmLayer::BackwardPass(NeuronLayer&, ...) (lstm-layer.cpp:294-355), layer == 0 */
static void neuron_backward(gmxo_lstm* l, neuron_layer* n, const float* input, int epoch, int symbol) {
  if (epoch == H - 1) {
    memset(n->gamma_u, 0, sizeof n->gamma_u);
    memset(n->beta_u, 0, sizeof n->beta_u);
    for (int i = 0; i < NC; ++i) {
      memset(n->update[i], 0, sizeof n->update[i]);
      for (int j = 0; j < W - NO - NI; ++j) n->transpose[j][i] = n->weights[i][j + NO + NI];
    }
  }
  for (int i = 0; i < NC; ++i) n->beta_u[i] += n->error[i];
  for (int i = 0; i < NC; ++i) n->gamma_u[i] += n->error[i] * n->norm[epoch][i];
  for (int i = 0; i < NC; ++i) n->error[i] *= n->gamma[i] * n->ivar[epoch];
  float s = n->error[NC - 1] * n->norm[epoch][NC - 1]; /* expression sum: last element first */
  for (int i = NC - 2; i >= 0; --i) s += n->error[i] * n->norm[epoch][i];
  s = s / NC;
  for (int i = 0; i < NC; ++i) n->error[i] -= s * n->norm[epoch][i];
  if (epoch > 0) {
    for (int i = 0; i < NC; ++i) {
      float f = 0;
      for (int j = 0; j < NC; ++j) f += n->error[j] * n->transpose[i][j];
      l->stored_error[i] += f;
    }
  }
  for (int i = 0; i < NC; ++i) {
    for (int j = 0; j < LIN; ++j) n->update[i][NO + j] += n->error[i] * input[j];
    n->update[i][symbol] += n->error[i];
  }
  if (epoch == 0) {
    for (int i = 0; i < NC; ++i) adam(n->update[i], n->m[i], n->v[i], n->weights[i], W, (float)l->update_steps, kUpdateLimit);
    adam(n->gamma_u, n->gamma_m, n->gamma_v, n->gamma, NC, (float)l->update_steps, kUpdateLimit);
    adam(n->beta_u, n->beta_m, n->beta_v, n->beta, NC, (float)l->update_steps, kUpdateLimit);
  }
}

/* LstmLayer::BackwardPass (lstm-layer.cpp:252-292) */
static void layer_backward(gmxo_lstm* l, const float* input, int epoch, int symbol) {
  neuron_layer *fg = l->gate[0], *in = l->gate[1], *og = l->gate[2];
  if (epoch == H - 1) {
    memcpy(l->stored_error, l->hidden_error, sizeof l->stored_error);
    memset(l->state_error, 0, sizeof l->state_error);
  } else {
    for (int i = 0; i < NC; ++i) l->stored_error[i] += l->hidden_error[i];
  }
  for (int i = 0; i < NC; ++i)
    og->error[i] = l->tanh_state[epoch][i] * l->stored_error[i] * og->state[epoch][i] * (1.0f - og->state[epoch][i]);
  for (int i = 0; i < NC; ++i)
    l->state_error[i] += l->stored_error[i] * og->state[epoch][i] * (1.0f - (l->tanh_state[epoch][i] * l->tanh_state[epoch][i]));
  for (int i = 0; i < NC; ++i)
    in->error[i] = l->state_error[i] * l->input_gate_state[epoch][i] * (1.0f - (in->state[epoch][i] * in->state[epoch][i]));
  for (int i = 0; i < NC; ++i)
    fg->error[i] = (l->last_state[epoch][i] - in->state[epoch][i]) * l->state_error[i] * fg->state[epoch][i] * l->input_gate_state[epoch][i];
  memset(l->hidden_error, 0, sizeof l->hidden_error);
  if (epoch > 0) {
    for (int i = 0; i < NC; ++i) l->state_error[i] *= fg->state[epoch][i];
    memset(l->stored_error, 0, sizeof l->stored_error);
  } else if (l->update_steps < kUpdateLimit) {
    ++l->update_steps;
  }
  neuron_backward(l, fg, input, epoch, symbol);
  neuron_backward(l, in, input, epoch, symbol);
  neuron_backward(l, og, input, epoch, symbol);
  clip(l->state_error);
  clip(l->stored_error);
  clip(l->hidden_error);
}

/* Lstm::Perceive (lstm.cpp:52-93) */
static void lstm_perceive(gmxo_lstm* l, uint32_t input) {
  int last_epoch = (int)l->epoch - 1;
  if (last_epoch == -1) last_epoch = H - 1;
  uint32_t old_input = l->input_history[last_epoch];
  l->input_history[last_epoch] = input;
  if (l->epoch == 0) {
    for (int epoch = H - 1; epoch >= 0; --epoch) {
      for (uint32_t i = 0; i < NO; ++i) {
        float error = (i == l->input_history[epoch]) ? (l->output[epoch][i] - 1) : l->output[epoch][i];
        for (int j = 0; j < NC; ++j) l->hidden_error[j] += l->out_layer[epoch][i][j] * error;
      }
      int prev_epoch = epoch - 1;
      if (prev_epoch == -1) prev_epoch = H - 1;
      uint32_t symbol = l->input_history[prev_epoch];
      if (epoch == 0) symbol = old_input;
      layer_backward(l, l->layer_input[epoch], epoch, (int)symbol);
    }
  }
  for (uint32_t i = 0; i < NO; ++i) {
    float error = (i == input) ? (l->output[last_epoch][i] - 1) : l->output[last_epoch][i];
    float le = kLearningRate * error;
    for (int j = 0; j < HID; ++j) l->out_layer[l->epoch][i][j] = l->out_layer[last_epoch][i][j];
    for (int j = 0; j < HID; ++j) l->out_layer[l->epoch][i][j] -= le * l->hidden[j];
  }
}

/* Sigmoid::Logit (sigmoid.cpp:7-13) */
static float logit(float p) {
  if (p < 0.0001) p = 0.0001;
  else if (p > 0.9999) p = 0.9999;
  return logf(p / (1 - p));
}

/* The bit-level half of LstmModel::Predict alone (lstm-model.cpp:34-48), for callers that hold the byte distribution
 * themselves (the per-byte device surface hands out probs_; the range is the host's): advances top/mid/bot by new_bit
 * (first != 0: a new byte, the full range) and returns what SetPrediction would do -- 1 active, 0 stored but silent
 * or nothing stored (*prediction untouched when denom == 0). */
int gmxo_lstm_bit_from_probs(const float* probs, int first, int new_bit, int* top, int* mid, int* bot, float* prediction) {
  if (first) {
    *top = 255;
    *bot = 0;
  } else if (new_bit) {
    *bot = *mid + 1;
  } else {
    *top = *mid;
  }
  *mid = *bot + ((*top - *bot) / 2);
  float num = 0.0f;
  for (int i = *mid + 1; i <= *top; ++i) num += probs[i];
  float denom = num;
  for (int i = *bot; i <= *mid; ++i) denom += probs[i];
  if (denom != 0) {
    float p = num / denom;
    *prediction = logit(p);
    return p == 0.5 ? 0 : 1;
  }
  return 0;
}

/* LstmModel::Predict (lstm-model.cpp:17-49).  recent_bits / last_byte / new_bit as the context
 * models left them in ShortTermMemory.  *prediction keeps its old value when the model stays
 * silent (denom == 0); returns 1 if SetPrediction marked it active, else 0;
 * *context = ShortTermMemory::lstm_prediction_context (only rewritten at byte boundaries). */
int gmxo_lstm_model_predict(gmxo_lstm* l, int recent_bits, uint32_t last_byte, int new_bit, const float* ppm,
                            float* prediction, uint32_t* context, float* probs_out) {
  if (recent_bits == 1) {
    const float* p = lstm_predict(l, ppm, last_byte);
    memcpy(l->probs, p, sizeof l->probs);
    l->top = 255;
    l->bot = 0;
    float max_pred = 0;
    *context = 0;
    for (int i = 0; i < 256; ++i)
      if (l->probs[i] > max_pred) {
        max_pred = l->probs[i];
        *context = (uint32_t)i;
      }
  } else {
    if (new_bit) l->bot = l->mid + 1;
    else l->top = l->mid;
  }
  l->mid = l->bot + ((l->top - l->bot) / 2);
  float num = 0.0f;
  for (int i = l->mid + 1; i <= l->top; ++i) num += l->probs[i];
  float denom = num;
  for (int i = l->bot; i <= l->mid; ++i) denom += l->probs[i];
  if (probs_out) memcpy(probs_out, l->probs, sizeof l->probs);
  if (denom != 0) {
    float p = num / denom;
    *prediction = logit(p);          /* SetPrediction (short-term-memory.cpp:187-191) */
    return p == 0.5 ? 0 : 1;
  }
  return 0;
}

/* LstmModel::Learn (lstm-model.cpp:51-60) */
void gmxo_lstm_model_learn(gmxo_lstm* l, int recent_bits, int new_bit) {
  int current_byte = recent_bits * 2 + new_bit;
  if (current_byte >= 256) lstm_perceive(l, (uint32_t)(current_byte - 256));
}

/* FNV-1a over the learned state: output layer ring, gate weights, Adam moments -- what
 * LongTermMemory::WriteToDisk's LSTM section (long-term-memory.cpp:57-68) and the layers'
 * WriteToDisk hold, hashed instead of stored (5 MiB per instance). */
uint64_t gmxo_lstm_state_hash(const gmxo_lstm* l) {
  uint64_t h = 0xcbf29ce484222325ull;
  const uint8_t* p;
#define MIX(ptr, len)                               \
  for (p = (const uint8_t*)(ptr); p < (const uint8_t*)(ptr) + (len); ++p) h = (h ^ *p) * 0x100000001b3ull
  MIX(l->out_layer, (size_t)H * NO * HID * 4);
  for (int g = 0; g < 3; ++g) {
    MIX(l->gate[g]->weights, sizeof l->gate[g]->weights);
    MIX(l->gate[g]->m, sizeof l->gate[g]->m);
    MIX(l->gate[g]->v, sizeof l->gate[g]->v);
    MIX(l->gate[g]->gamma, sizeof l->gate[g]->gamma);
    MIX(l->gate[g]->beta, sizeof l->gate[g]->beta);
  }
  MIX(l->hidden, sizeof l->hidden);
  MIX(l->state, sizeof l->state);
#undef MIX
  return h;
}

/* ---- persistence ------------------------------------------------------------------------
 * LstmModel::WriteToDisk / ReadFromDisk (lstm-model.cpp:62-76) with everything they reach:
 * Lstm::WriteToDisk (lstm.cpp:124-140), LstmLayer::WriteToDisk (lstm-layer.cpp:356-374) and three
 * NeuronLayer::WriteToDisk (lstm-layer.cpp:62-91) -- the model's stretch of the `.short` file --
 * and the LSTM section of LongTermMemory::WriteToDisk (long-term-memory.cpp:57-67) of the `.long`
 * file.  Raw arrays in declaration order, no headers.  dir: 0 = count, 1 = write buf, 2 = read buf. */
#define IO(ptr, len)                                        \
  do {                                                      \
    if (dir == 1) memcpy(buf + off, (ptr), (len));          \
    else if (dir == 2) memcpy((ptr), buf + off, (len));     \
    off += (len);                                           \
  } while (0)

static size_t short_walk(gmxo_lstm* l, uint8_t* buf, int dir) {
  size_t off = 0;
  IO(&l->top, 4);
  IO(&l->mid, 4);
  IO(&l->bot, 4);
  IO(l->probs, sizeof l->probs);
  IO(l->input_history, sizeof l->input_history);
  IO(l->hidden, sizeof l->hidden);
  IO(l->hidden_error, sizeof l->hidden_error);
  IO(l->layer_input, sizeof l->layer_input);
  IO(l->output, sizeof l->output);
  IO(&l->epoch, 4);
  IO(l->state, sizeof l->state);
  IO(l->state_error, sizeof l->state_error);
  IO(l->stored_error, sizeof l->stored_error);
  IO(l->tanh_state, sizeof l->tanh_state);
  IO(l->input_gate_state, sizeof l->input_gate_state);
  IO(l->last_state, sizeof l->last_state);
  IO(&l->l_epoch, 4);
  IO(&l->update_steps, 8);
  for (int g = 0; g < 3; ++g) {
    neuron_layer* n = l->gate[g];
    IO(n->error, sizeof n->error);
    IO(n->ivar, sizeof n->ivar);
    IO(n->gamma, sizeof n->gamma);
    IO(n->gamma_u, sizeof n->gamma_u);
    IO(n->gamma_m, sizeof n->gamma_m);
    IO(n->gamma_v, sizeof n->gamma_v);
    IO(n->beta, sizeof n->beta);
    IO(n->beta_u, sizeof n->beta_u);
    IO(n->beta_m, sizeof n->beta_m);
    IO(n->beta_v, sizeof n->beta_v);
    IO(n->state, sizeof n->state);
    IO(n->update, sizeof n->update);
    IO(n->m, sizeof n->m);
    IO(n->v, sizeof n->v);
    IO(n->transpose, sizeof n->transpose);
    IO(n->norm, sizeof n->norm);
  }
  return off;
}

static size_t long_walk(gmxo_lstm* l, uint8_t* buf, int dir) {
  size_t off = 0;
  IO(l->out_layer, (size_t)H * NO * HID * 4);
  for (int g = 0; g < 3; ++g) IO(l->gate[g]->weights, sizeof l->gate[g]->weights);
  return off;
}
#undef IO

/* buf == NULL: the size only */
size_t gmxo_lstm_export_short(gmxo_lstm* l, uint8_t* buf) { return short_walk(l, buf, buf ? 1 : 0); }
size_t gmxo_lstm_export_long(gmxo_lstm* l, uint8_t* buf) { return long_walk(l, buf, buf ? 1 : 0); }
int gmxo_lstm_import_short(gmxo_lstm* l, const uint8_t* buf, size_t n) {
  if (n != short_walk(l, 0, 0)) return -1;
  short_walk(l, (uint8_t*)buf, 2);
  return 0;
}
int gmxo_lstm_import_long(gmxo_lstm* l, const uint8_t* buf, size_t n) {
  if (n != long_walk(l, 0, 0)) return -1;
  long_walk(l, (uint8_t*)buf, 2);
  return 0;
}

void gmxo_srand(unsigned seed) { srand(seed); }

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}

/* FNV-1a of the gate weights in LongTermMemory order (forget, input node, output; cell-major), with
 * with_output_layer != 0 preceded by lstm_output_layer -- the two hashes the harness dumps. */
uint64_t gmxo_lstm_weights_hash(const gmxo_lstm* l, int with_output_layer) {
  uint64_t h = 0xcbf29ce484222325ull;
  if (with_output_layer) h = fnv(h, l->out_layer, (size_t)H * NO * HID * 4);
  for (int g = 0; g < 3; ++g) h = fnv(h, l->gate[g]->weights, sizeof l->gate[g]->weights);
  return h;
}

#include "gmx_lstm_synth.h"

/* The loop of oracle/ref_build/ref_lstm_harness.cpp around the restated model. */
uint64_t gmxo_lstm_run_synth2(gmxo_lstm* l, uint64_t n_bytes, uint64_t seed, uint32_t mask, uint64_t dump,
                              uint64_t nolearn_from, float* pred_out, uint8_t* act_out, uint32_t* ctx_out);
uint64_t gmxo_lstm_run_synth(gmxo_lstm* l, uint64_t n_bytes, uint64_t seed, uint32_t mask, uint64_t dump,
                             float* pred_out, uint8_t* act_out, uint32_t* ctx_out) {
  return gmxo_lstm_run_synth2(l, n_bytes, seed, mask, dump, ~0ull, pred_out, act_out, ctx_out);
}
/* ... from byte nolearn_from on without LstmModel::Learn (the harness's --nolearn-from: generation) */
uint64_t gmxo_lstm_run_synth2(gmxo_lstm* l, uint64_t n_bytes, uint64_t seed, uint32_t mask, uint64_t dump,
                              uint64_t nolearn_from, float* pred_out, uint8_t* act_out, uint32_t* ctx_out) {
  gmx_lstm_synth g;
  gmx_lstm_synth_init(&g, seed, mask);
  float ppm[256], cur_ppm[256];
  uint64_t h = 0xcbf29ce484222325ull;
  int recent_bits = 1, new_bit = 0;
  uint32_t last_byte = 0, context = 0;
  float prediction = 0;
  uint32_t byte = gmx_lstm_synth_byte(&g, ppm);
  memset(cur_ppm, 0, sizeof cur_ppm);
  for (uint64_t n = 0; n < n_bytes; ++n) {
    for (int k = 0; k < 8; ++k) {
      if (recent_bits == 1) memcpy(cur_ppm, ppm, sizeof ppm);
      uint8_t act = (uint8_t)gmxo_lstm_model_predict(l, recent_bits, last_byte, new_bit, cur_ppm, &prediction, &context, 0);
      h = fnv(h, &prediction, 4);
      h = fnv(h, &act, 1);
      if (k == 0) h = fnv(h, &context, 4);
      if (n < dump) {
        pred_out[n * 8 + k] = prediction;
        act_out[n * 8 + k] = act;
        if (k == 0) ctx_out[n] = context;
      }
      new_bit = (int)((byte >> (7 - k)) & 1u);
      if (n < nolearn_from) gmxo_lstm_model_learn(l, recent_bits, new_bit);
      recent_bits += recent_bits + new_bit;
      if (recent_bits >= 256) {
        last_byte = (uint32_t)(recent_bits - 256);
        recent_bits = 1;
      }
    }
    byte = gmx_lstm_synth_byte(&g, ppm);
  }
  return h;
}

/* The synthetic stream as arrays: ppm[n][256], bytes[n]. */
void gmxo_lstm_synth_fill(uint64_t seed, uint32_t mask, uint64_t n, float* ppm, uint8_t* bytes) {
  gmx_lstm_synth g;
  gmx_lstm_synth_init(&g, seed, mask);
  for (uint64_t i = 0; i < n; ++i) bytes[i] = (uint8_t)gmx_lstm_synth_byte(&g, ppm + i * 256);
}

/* Same bookkeeping as gmxo_lstm_run_synth on caller-supplied records (what the device kernel
 * is compared with): pred/act [n][8], ctx [n]; learn == 0 skips LstmModel::Learn. */
void gmxo_lstm_run(gmxo_lstm* l, uint64_t n_bytes, const float* ppm, const uint8_t* bytes, int learn,
                   uint32_t* last_byte_io, float* prediction_io, uint32_t* context_io, float* pred_out,
                   uint8_t* act_out, uint32_t* ctx_out) {
  int recent_bits = 1, new_bit = 0;
  uint32_t last_byte = *last_byte_io, context = *context_io;
  float prediction = *prediction_io;
  for (uint64_t n = 0; n < n_bytes; ++n) {
    for (int k = 0; k < 8; ++k) {
      uint8_t act = (uint8_t)gmxo_lstm_model_predict(l, recent_bits, last_byte, new_bit, ppm + n * 256, &prediction,
                                                     &context, 0);
      pred_out[n * 8 + k] = prediction;
      act_out[n * 8 + k] = act;
      if (k == 0) ctx_out[n] = context;
      new_bit = (int)((bytes[n] >> (7 - k)) & 1u);
      if (learn) gmxo_lstm_model_learn(l, recent_bits, new_bit);
      recent_bits += recent_bits + new_bit;
      if (recent_bits >= 256) {
        last_byte = (uint32_t)(recent_bits - 256);
        recent_bits = 1;
      }
    }
  }
  *last_byte_io = last_byte;
  *prediction_io = prediction;
  *context_io = context;
}

void gmxo_lstm_get_output_layer(const gmxo_lstm* l, float* out /* [H][NO][HID] */) {
  memcpy(out, l->out_layer, (size_t)H * NO * HID * 4);
}
uint64_t gmxo_lstm_update_steps(const gmxo_lstm* l) { return l->update_steps; }

/* Byte-level entry points (what the per-byte device surface is compared with): Lstm::Predict at a
 * byte boundary with the bookkeeping of LstmModel::Predict, and Lstm::Perceive. */
void gmxo_lstm_predict_byte(gmxo_lstm* l, const float* ppm, uint32_t last_byte, float* probs_out, uint32_t* ctx_out) {
  float prediction = 0;
  gmxo_lstm_model_predict(l, 1, last_byte, 0, ppm, &prediction, ctx_out, probs_out);
}
void gmxo_lstm_perceive_byte(gmxo_lstm* l, uint32_t byte) { lstm_perceive(l, byte); }

/* FNV-1a 64 of a byte array (the harnesses' checksum), for the tests. */
uint64_t gmxo_fnv64_bytes(const uint8_t* p, uint64_t n) { return fnv(0xcbf29ce484222325ull, p, n); }
