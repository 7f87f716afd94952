// ref_trace.cpp -- TEST INFRASTRUCTURE, builds only where /root/reference exists.
//
// Runs the REFERENCE's whole Predictor (all 88 feature models + 33 mixers, predictor.cpp)
// over the first bytes of a file exactly like runner_utils::Compress does
// (runner-utils.cpp:43-67: Predict -> Encode -> Perceive -> Learn per bit), and records what
// crosses the mixer boundary on every bit: the raw predictions blackboard, active_models, the
// 33 gate contexts the mixers alias, the coded bit, the 33 mixer outputs and the probability
// Predict() returned.  tests/golden/make_golden.py turns the dump into a fixture, so the HIP
// path can be checked against real feature-model inputs on a box where the reference is absent.
//
// usage: ref_trace <input file> <n_bytes> <out file> [analysis 0|1]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define private public  // the mixer-facing state of Predictor / Mixer is private in the reference
#include "mixer/mixer.h"
#include "predictor.h"
#undef private

template <typename T>
static void Put(std::ofstream& f, const T& v) {
  f.write(reinterpret_cast<const char*>(&v), sizeof(v));
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: ref_trace <input> <n_bytes> <out> [analysis]\n");
    return 2;
  }
  const uint64_t n_bytes = strtoull(argv[2], 0, 0);
  const int analysis = argc > 4 ? atoi(argv[4]) : 0;
  std::ifstream is(argv[1], std::ios::binary);
  if (!is.is_open()) return 1;
  std::vector<unsigned char> text(n_bytes);
  is.read(reinterpret_cast<char*>(text.data()), n_bytes);
  if ((uint64_t)is.gcount() != n_bytes) return 1;

  srand(0xDEADBEEF);  // runner.cpp:37
  Predictor p;
  if (analysis) p.EnableAnalysis(8 * n_bytes / 1000 + 1);
  ShortTermMemory& stm = p.short_term_memory_;
  std::vector<Mixer*> mixers;
  for (auto& m : p.models_)
    if (Mixer* mx = dynamic_cast<Mixer*>(m.get())) mixers.push_back(mx);
  const uint32_t n = stm.num_predictions, M = mixers.size();
  const uint32_t L0 = stm.num_layer0_mixers, L1 = stm.num_layer1_mixers;

  std::ofstream f(argv[3], std::ios::binary);
  Put(f, (uint32_t)0x54584D47u);  // "GMXT"
  Put(f, (uint32_t)1);
  Put(f, n);
  Put(f, M);
  Put(f, L0);
  Put(f, L1);
  Put(f, (uint32_t)stm.models_with_skip_connection.size());
  for (int idx : stm.models_with_skip_connection) Put(f, (uint32_t)idx);
  Put(f, (uint64_t)(8 * n_bytes));
  // the topology as constructed (mixer.h private fields)
  for (Mixer* mx : mixers) {
    Put(f, (int32_t)mx->layer_number_);
    Put(f, (uint32_t)p.long_term_memory_.mixers[mx->memory_index_].mixer_table.size());
    Put(f, (float)mx->learning_rate_);
    Put(f, (int32_t)mx->weight_size_);
  }
  std::vector<uint8_t> act(n);
  for (uint64_t pos = 0; pos < n_bytes; ++pos) {
    unsigned char c = text[pos];
    for (int j = 7; j >= 0; --j) {
      int bit = (c >> j) & 1;
      float prob = p.Predict();
      // --- the mixer boundary, as it stood when the mixers ran ---
      f.write(reinterpret_cast<const char*>(&stm.predictions[0]), 4 * n);
      memset(act.data(), 0, n);
      for (int i : stm.active_models) act[i] = 1;
      f.write(reinterpret_cast<const char*>(act.data()), n);
      for (Mixer* mx : mixers) Put(f, (uint32_t)mx->context_);
      Put(f, (uint8_t)bit);
      for (uint32_t k = 0; k < L0; ++k) Put(f, (float)stm.mixer_layer0_outputs[k]);
      for (uint32_t k = 0; k < L1; ++k) Put(f, (float)stm.mixer_layer1_outputs[k]);
      Put(f, (float)stm.final_mixer_output);
      Put(f, prob);
      p.Perceive(bit);
      p.Learn();
    }
  }
  // persistent mixer state: Mixer::WriteToDisk x33, and the mixer section of the .long file
  // written by the reference's own LongTermMemory::WriteToDisk on a memory that holds
  // nothing but the mixers.
  std::string tmp = std::string(argv[3]) + ".tmp";
  {
    std::ofstream s(tmp, std::ios::binary);
    for (Mixer* mx : mixers) mx->WriteToDisk(&s);
  }
  {
    std::ifstream s(tmp, std::ios::binary);
    std::vector<char> b((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
    Put(f, (uint64_t)b.size());
    f.write(b.data(), b.size());
  }
  {
    LongTermMemory only_mixers;
    only_mixers.mixers = std::move(p.long_term_memory_.mixers);
    std::ofstream s(tmp, std::ios::binary);
    only_mixers.WriteToDisk(&s);
  }
  {
    std::ifstream s(tmp, std::ios::binary);
    std::vector<char> b((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
    size_t len = b.size() >= 8 ? b.size() - 8 : 0;  // drop the (empty) history length
    Put(f, (uint64_t)len);
    f.write(b.data(), len);
  }
  remove(tmp.c_str());
  f.close();
  return 0;
}
