// ref_lstm_harness.cpp -- TEST INFRASTRUCTURE, builds only where /root/reference exists.
//
// Drives the REFERENCE's own `LstmModel` (src/models/lstm-model.{h,cpp}, lstm.cpp, lstm-layer.cpp)
// bit by bit the way `Predictor` does: srand(0xDEADBEEF) before construction (predictor.cpp:18),
// per bit the context bookkeeping of BasicContexts::Predict (basic-contexts.cpp:21-39: recent_bits,
// last_byte), ModPPMD's ppm_predictions refreshed at byte boundaries, LstmModel::Predict, new_bit,
// LstmModel::Learn.  Input: oracle/gmx_lstm_synth.h.  Ground truth for oracle/gmx_oracle_lstm.c.
//
// usage: ref_lstm_harness --bytes N [--dump D] [--seed S] [--mask M] [--nolearn-from N0] --out file
//   --nolearn-from N0: from byte N0 on LstmModel::Learn is not called (generation, runner-utils.cpp:199-209: Predict and
//   Perceive only) -- Lstm::Predict then runs on output layers and histories no Perceive has refreshed
// dump format "GMXL": u32 magic, N, D; u64 fnv of the initial gate weights;
//   D x { float probs[256]; u32 lstm_prediction_context; 8 x { float prediction; u8 active } };
//   u64 fnv over all bytes of (8 predictions, 8 active flags, context); u64 fnv of the final
//   lstm_output_layer + neuron_layer_weights bytes; u64 LstmModel::GetMemoryUsage
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "models/lstm-model.h"  // reference (via -I/root/reference/src)

extern "C" {
#include "../gmx_lstm_synth.h"
}

template <typename T>
static void Put(std::ofstream& f, const T& v) {
  f.write(reinterpret_cast<const char*>(&v), sizeof(v));
}
static uint64_t Fnv(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}

int main(int argc, char** argv) {
  uint64_t N = 300, dump = 0, seed = 0, nolearn_from = ~0ull;
  uint32_t mask = 255;
  std::string out_path;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> std::string { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "--bytes") N = strtoull(next().c_str(), 0, 0);
    else if (a == "--dump") dump = strtoull(next().c_str(), 0, 0);
    else if (a == "--seed") seed = strtoull(next().c_str(), 0, 0);
    else if (a == "--mask") mask = (uint32_t)strtoul(next().c_str(), 0, 0);
    else if (a == "--nolearn-from") nolearn_from = strtoull(next().c_str(), 0, 0);
    else if (a == "--out") out_path = next();
    else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
  }
  if (out_path.empty()) { fprintf(stderr, "--out required\n"); return 2; }
  if (dump > N) dump = N;

  srand(0xDEADBEEF);  // predictor.cpp:18
  ShortTermMemory stm;
  LongTermMemory ltm;
  LstmModel model(stm, ltm, false);
  stm.predictions.resize(stm.num_predictions);
  stm.predictions = 0;

  std::ofstream out(out_path, std::ios::binary);
  Put(out, (uint32_t)0x4c584d47u);  // "GMXL"
  Put(out, (uint32_t)N);
  Put(out, (uint32_t)dump);
  uint64_t hw = 0xcbf29ce484222325ull;
  for (auto& layer : ltm.neuron_layer_weights)
    for (auto& row : layer.weights) hw = Fnv(hw, &row[0], row.size() * 4);
  Put(out, hw);

  gmx_lstm_synth g;
  gmx_lstm_synth_init(&g, seed, mask);
  float ppm[256];
  uint64_t h = 0xcbf29ce484222325ull;
  uint32_t byte = gmx_lstm_synth_byte(&g, ppm);  // first byte to code and what PPM says about it
  for (uint64_t n = 0; n < N; ++n) {
    for (int k = 0; k < 8; ++k) {
      // BasicContexts::Predict's bookkeeping happened for this bit (recent_bits, last_byte);
      // ModPPMD::Predict refreshed ppm_predictions at the byte boundary
      if (stm.recent_bits == 1)
        for (int i = 0; i < 256; ++i) stm.ppm_predictions[i] = ppm[i];
      stm.active_models.clear();
      model.Predict(stm, ltm);
      float p = stm.predictions[0];
      uint8_t act = stm.active_models.empty() ? 0 : 1;
      h = Fnv(h, &p, 4);
      h = Fnv(h, &act, 1);
      if (k == 0) {
        uint32_t c = stm.lstm_prediction_context;
        h = Fnv(h, &c, 4);
        if (n < dump) {
          // probs_ is private: the byte distribution is what Lstm::Predict returned; recover it
          // through the bit predictions instead?  No: dump via WriteToDisk below at the end only.
        }
      }
      if (n < dump) {
        Put(out, p);
        Put(out, act);
        if (k == 0) Put(out, (uint32_t)stm.lstm_prediction_context);
      }
      stm.new_bit = (byte >> (7 - k)) & 1;
      if (n < nolearn_from) model.Learn(stm, ltm);
      // what BasicContexts::Predict does at the start of the next bit (basic-contexts.cpp:27-33)
      stm.recent_bits += stm.recent_bits + stm.new_bit;
      if (stm.recent_bits >= 256) {
        stm.last_byte = stm.recent_bits - 256;
        stm.recent_bits = 1;
      }
    }
    byte = gmx_lstm_synth_byte(&g, ppm);
  }
  Put(out, h);
  uint64_t hl = 0xcbf29ce484222325ull;
  for (auto& x : ltm.lstm_output_layer)
    for (auto& y : x) hl = Fnv(hl, &y[0], y.size() * 4);
  for (auto& layer : ltm.neuron_layer_weights)
    for (auto& row : layer.weights) hl = Fnv(hl, &row[0], row.size() * 4);
  Put(out, hl);
  Put(out, (uint64_t)model.GetMemoryUsage(stm, ltm));
  // LstmModel::WriteToDisk: top_, mid_, bot_, probs_, then the Lstm's own state (lstm-model.cpp:62-68)
  std::string tmp = out_path + ".short";
  {
    std::ofstream sf(tmp, std::ios::binary);
    model.WriteToDisk(&sf);
  }
  std::ifstream sf(tmp, std::ios::binary);
  std::string bytes((std::istreambuf_iterator<char>(sf)), std::istreambuf_iterator<char>());
  uint64_t hs = Fnv(0xcbf29ce484222325ull, bytes.data(), bytes.size());
  Put(out, (uint64_t)bytes.size());
  Put(out, hs);
  out.write(bytes.data(), 12 + 1024);  // top_, mid_, bot_, probs_ of the last byte
  remove(tmp.c_str());
  return 0;
}
