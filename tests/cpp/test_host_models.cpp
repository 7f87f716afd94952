// test_host_models.cpp -- the C++ host mirrors of the reference's Indirect and LstmModel
// (gmix_amd/host/gmx_models.h) driven bit by bit the way Predictor drives its models_, checked
// against the CPU oracle bit for bit, with checkpoint / copy round trips through files in the
// reference's formats.  Needs an MI355X.  Built and run by tests/test_gpu_host_cpp.py, which
// passes a directory holding ns_next.bin / rm_next.bin (the two [256][2] next-state tables).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../../gmix_amd/host/gmx_models.h"
extern "C" {
#include "../../oracle/gmx_ind_synth.h"
#include "../../oracle/gmx_lstm_synth.h"
struct gmxo_ind;
gmxo_ind* gmxo_ind_create(int, const uint32_t*, const float*, const uint8_t*, const uint8_t*);
void gmxo_ind_destroy(gmxo_ind*);
void gmxo_ind_predict(gmxo_ind*, const uint32_t*, uint32_t, float*, uint8_t*);
void gmxo_ind_learn(gmxo_ind*, int);
uint64_t gmxo_ind_memory_usage(const gmxo_ind*, int);
size_t gmxo_ind_export(const gmxo_ind*, uint8_t*, size_t);
struct gmxo_lstm;
void gmxo_srand(unsigned);
gmxo_lstm* gmxo_lstm_create(void);
void gmxo_lstm_destroy(gmxo_lstm*);
int gmxo_lstm_model_predict(gmxo_lstm*, int, uint32_t, int, const float*, float*, uint32_t*, float*);
void gmxo_lstm_model_learn(gmxo_lstm*, int, int);
size_t gmxo_lstm_export_short(gmxo_lstm*, uint8_t*);
size_t gmxo_lstm_export_long(gmxo_lstm*, uint8_t*);
}

static void Fail(const char* what, long t = -1) {
  fprintf(stderr, "Test failed: %s (step %ld)\n", what, t);
  fflush(stderr);
  abort();  // the reference's convention (tester.cpp:318-321)
}
static std::vector<char> Slurp(const std::string& path) {
  std::ifstream s(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
}
static bool SameBits(float a, float b) { return memcmp(&a, &b, 4) == 0; }

// ---- Indirect ----------------------------------------------------------------------------------
struct IndRow { unsigned table; float lr; };
static const IndRow kInd[] = {{256, 0.02f}, {3, 0.1f}, {65536, 0.005f}, {1, 0.03f}, {4096, 0.01f}, {22, 0.05f}, {1 << 18, 0.002f}};
static const int kK = sizeof(kInd) / sizeof(kInd[0]);

struct IndRig {
  std::vector<unsigned int> ctx;  // the context variables the models alias
  gmx::ShortTermMemory stm;
  gmx::IndirectBank bank;
  std::vector<std::unique_ptr<gmx::Indirect>> models;
  IndRig(const std::vector<char>& ns, const std::vector<char>& rm) : ctx(kK, 0) {
    for (int i = 0; i < kK; ++i)
      models.emplace_back(new gmx::Indirect(stm, bank, kInd[i].lr, kInd[i].table, ctx[i], "ctx" + std::to_string(i), false));
    stm.predictions.resize(stm.num_predictions);
    stm.predictions = 0;
    int rc = bank.Finalize([&](int s, int b) { return (uint8_t)ns[2 * s + b]; },
                           [&](int s, int b) { return (uint8_t)rm[2 * s + b]; });
    if (rc != GMX_OK) Fail(gmx_last_error());
  }
  void Predict() {
    stm.active_models.clear();  // predictor.cpp:361
    for (auto& m : models) m->Predict(stm, bank);
  }
  void Learn(int bit) {
    stm.new_bit = bit;
    for (auto& m : models) m->Learn(stm, bank);
  }
};

static void TestIndirect(const std::string& dir) {
  const std::vector<char> ns = Slurp(dir + "/ns_next.bin"), rm = Slurp(dir + "/rm_next.bin");
  if (ns.size() != 512 || rm.size() != 512) Fail("next-state tables missing");
  const long T = 4000;
  const uint32_t ctx_mod[4] = {40, 3, 0, 900};
  std::vector<uint32_t> table(kK);
  std::vector<float> lr(kK);
  for (int i = 0; i < kK; ++i) { table[i] = kInd[i].table; lr[i] = kInd[i].lr; }
  gmxo_ind* ob = gmxo_ind_create(kK, table.data(), lr.data(), (const uint8_t*)ns.data(), (const uint8_t*)rm.data());
  gmx_ind_synth g;
  gmx_ind_synth_init(&g, 77, kK, ctx_mod);
  std::unique_ptr<IndRig> r(new IndRig(ns, rm));
  std::vector<float> po(2 * kK);
  std::vector<uint8_t> ao(2 * kK);
  std::vector<uint32_t> cur(kK, 0);
  for (long t = 0; t < T; ++t) {
    r->stm.bit_context = gmx_ind_synth_contexts(&g, cur.data());
    for (int i = 0; i < kK; ++i) r->ctx[i] = cur[i];
    r->Predict();
    gmxo_ind_predict(ob, cur.data(), r->stm.bit_context, po.data(), ao.data());
    size_t na = 0;
    for (int k = 0; k < 2 * kK; ++k) {
      if (!SameBits(r->stm.predictions[k], po[k])) Fail("Indirect prediction differs from oracle", t);
      if (ao[k]) {
        if (na >= r->stm.active_models.size() || r->stm.active_models[na] != k) Fail("active_models order", t);
        ++na;
      }
    }
    if (na != r->stm.active_models.size()) Fail("active_models length", t);
    const int bit = gmx_ind_synth_bit(&g, cur.data());
    r->Learn(bit);
    gmxo_ind_learn(ob, bit);
    if (t == T / 2 || t == T / 2 + 700) {
      // restart from a file (t == T/2) or through Copy (later): LongTermMemory's indirect section
      std::vector<uint8_t> ref(gmxo_ind_export(ob, nullptr, 0));
      gmxo_ind_export(ob, ref.data(), ref.size());
      const std::string path = dir + "/gmx_indirect.long";
      {
        std::ofstream s(path, std::ios::binary);
        r->bank.WriteToDisk(&s);
      }
      const std::vector<char> got = Slurp(path);
      if (got.size() != ref.size() || memcmp(got.data(), ref.data(), ref.size()) != 0) Fail("indirect section differs from the reference format", t);
      std::unique_ptr<IndRig> fresh(new IndRig(ns, rm));
      if (t == T / 2) {
        std::ifstream s(path, std::ios::binary);
        fresh->bank.ReadFromDisk(&s);
      } else {
        fresh->bank.Copy(&r->bank);
      }
      // the blackboard slots of silent models are ShortTermMemory state: carried over by hand
      fresh->stm.predictions = r->stm.predictions;
      for (int i = 0; i < kK; ++i)
        if (fresh->models[i]->GetMemoryUsage(fresh->stm, fresh->bank) != gmxo_ind_memory_usage(ob, i)) Fail("GetMemoryUsage", i);
      r = std::move(fresh);
    }
  }
  if (r->bank.status() != GMX_OK) Fail("IndirectBank status");
  gmxo_ind_destroy(ob);
}

// ---- LstmModel ---------------------------------------------------------------------------------
struct LstmRig {
  gmx::ShortTermMemory stm;
  gmx::LstmBank bank;
  gmx::LstmModel model;
  LstmRig() : model(stm, bank, false) {
    if (bank.status() != GMX_OK) Fail(gmx_last_error());
    stm.predictions.resize(stm.num_predictions);
    stm.predictions = 0;
  }
};

static void TestLstm(const std::string& dir) {
  const long N = 330;
  srand(0xDEADBEEF);  // predictor.cpp:18
  std::unique_ptr<LstmRig> r(new LstmRig());
  gmxo_srand(0xDEADBEEF);
  gmxo_lstm* om = gmxo_lstm_create();
  gmx_lstm_synth g;
  gmx_lstm_synth_init(&g, 5, 63);
  float ppm[256];
  uint32_t byte = gmx_lstm_synth_byte(&g, ppm);
  float o_pred = 0;
  uint32_t o_ctx = 0;
  int o_recent = 1, o_bit = 0;
  uint32_t o_last = 0;
  std::vector<float> cur_ppm(256, 0.f);
  for (long n = 0; n < N; ++n) {
    for (int k = 0; k < 8; ++k) {
      gmx::ShortTermMemory& stm = r->stm;
      if (stm.recent_bits == 1) {  // ModPPMD refreshes ppm_predictions at the byte boundary
        for (int i = 0; i < 256; ++i) stm.ppm_predictions[i] = ppm[i];
        memcpy(cur_ppm.data(), ppm, sizeof ppm);
      }
      stm.active_models.clear();
      r->model.Predict(stm, r->bank);
      const int o_act = gmxo_lstm_model_predict(om, o_recent, o_last, o_bit, cur_ppm.data(), &o_pred, &o_ctx, nullptr);
      if (!SameBits(stm.predictions[0], o_pred)) Fail("LSTM bit prediction differs from oracle", n * 8 + k);
      if ((int)stm.active_models.size() != o_act) Fail("LSTM active flag", n * 8 + k);
      if (stm.lstm_prediction_context != o_ctx) Fail("lstm_prediction_context", n);
      stm.new_bit = (byte >> (7 - k)) & 1;
      r->model.Learn(stm, r->bank);
      o_bit = stm.new_bit;
      gmxo_lstm_model_learn(om, o_recent, o_bit);
      // BasicContexts::Predict's bookkeeping at the start of the next bit (basic-contexts.cpp:27-33)
      stm.recent_bits += stm.recent_bits + stm.new_bit;
      if (stm.recent_bits >= 256) {
        stm.last_byte = stm.recent_bits - 256;
        stm.recent_bits = 1;
      }
      o_recent = stm.recent_bits;
      o_last = stm.last_byte;
    }
    byte = gmx_lstm_synth_byte(&g, ppm);
    if (n == 129 || n == 250) {
      // checkpoint where a byte ends: LstmModel's stretch of .short, the LSTM section of .long
      std::vector<uint8_t> so(gmxo_lstm_export_short(om, nullptr)), lo(gmxo_lstm_export_long(om, nullptr));
      gmxo_lstm_export_short(om, so.data());
      gmxo_lstm_export_long(om, lo.data());
      {
        std::ofstream s(dir + "/gmx_lstm.short", std::ios::binary), l(dir + "/gmx_lstm.long", std::ios::binary);
        r->model.WriteToDisk(&s);
        r->bank.WriteToDisk(&l);
      }
      const std::vector<char> gs = Slurp(dir + "/gmx_lstm.short"), gl = Slurp(dir + "/gmx_lstm.long");
      if (gs.size() != so.size() || memcmp(gs.data(), so.data(), so.size()) != 0) Fail("LSTM .short differs from the reference format", n);
      if (gl.size() != lo.size() || memcmp(gl.data(), lo.data(), lo.size()) != 0) Fail("LSTM .long differs from the reference format", n);
      if (r->model.GetMemoryUsage(r->stm, r->bank) != 7017924ull) Fail("LstmModel::GetMemoryUsage");
      std::unique_ptr<LstmRig> fresh(new LstmRig());  // other weights (rand() has moved on) until restored
      if (n == 129) {
        std::ifstream s(dir + "/gmx_lstm.short", std::ios::binary), l(dir + "/gmx_lstm.long", std::ios::binary);
        fresh->model.ReadFromDisk(&s);  // .short before .long (predictor.cpp:412-416)
        fresh->bank.ReadFromDisk(&l);
      } else {
        fresh->model.Copy(&r->model);
        fresh->bank.Copy(&r->bank);
      }
      // ShortTermMemory state travels through its own ReadFromDisk / Copy in the reference
      fresh->stm.predictions = r->stm.predictions;
      fresh->stm.recent_bits = r->stm.recent_bits;
      fresh->stm.last_byte = r->stm.last_byte;
      fresh->stm.new_bit = r->stm.new_bit;
      fresh->stm.lstm_prediction_context = r->stm.lstm_prediction_context;
      if (fresh->bank.status() != GMX_OK) Fail("restored LstmBank status");
      r = std::move(fresh);
    }
  }
  if (r->bank.status() != GMX_OK) Fail("LstmBank status");
  gmxo_lstm_destroy(om);
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  TestIndirect(dir);
  TestLstm(dir);
  printf("Tests passed.\n");
  return 0;
}
