// ref_mixer_harness.cpp -- TEST INFRASTRUCTURE, builds only where /root/reference exists.
//
// Drives the REFERENCE's own `Mixer` class (src/mixer/mixer.{h,cpp}) with the synthetic
// stream of oracle/gmx_synth.h, exactly the way `Predictor` drives it
// (predictor.cpp:360-387), and dumps what it computed.  This file is our driver; the
// reference translation units are compiled where they lie (see Makefile) and linked in.
// Its output is the ground truth the C restatement (oracle/gmx_oracle.c) and the HIP path
// are pinned against (tests/golden/, made by tests/golden/make_golden.py).
//
// usage: ref_mixer_harness --n N --topo "layer:table:lr,layer:table:lr,..." [--skip i,j]
//          --bits T [--dump D] [--seed S] [--ctx-mode k --ctx-mod m] [--zero-mod z] [--bit-mode b]
//          [--nolearn-from T0] --out file
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "mixer/mixer.h"  // reference (via -I/root/reference/src)

extern "C" {
#include "../gmx_synth.h"
}

struct TopoEntry {
  int layer;
  unsigned table;
  double lr;
};

static std::vector<std::string> Split(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t e = s.find(sep, pos);
    if (e == std::string::npos) e = s.size();
    if (e > pos) out.push_back(s.substr(pos, e - pos));
    pos = e + 1;
  }
  return out;
}

template <typename T>
static void Put(std::ofstream& f, const T& v) {
  f.write(reinterpret_cast<const char*>(&v), sizeof(v));
}

int main(int argc, char** argv) {
  int n = 256;
  std::string topo_s = "0:65536:0.005";
  std::string skip_s = "1";
  uint64_t T = 1000, dump = 0, seed = 0, nolearn_from = ~0ull;
  int ctx_mode = 0, bit_mode = 0;
  unsigned ctx_mod = 1, zero_mod = 0;
  std::string out_path;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> std::string { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "--n") n = atoi(next().c_str());
    else if (a == "--topo") topo_s = next();
    else if (a == "--skip") skip_s = next();
    else if (a == "--bits") T = strtoull(next().c_str(), 0, 0);
    else if (a == "--dump") dump = strtoull(next().c_str(), 0, 0);
    else if (a == "--seed") seed = strtoull(next().c_str(), 0, 0);
    else if (a == "--ctx-mode") ctx_mode = atoi(next().c_str());
    else if (a == "--ctx-mod") ctx_mod = strtoul(next().c_str(), 0, 0);
    else if (a == "--bit-mode") bit_mode = atoi(next().c_str());
    else if (a == "--zero-mod") zero_mod = strtoul(next().c_str(), 0, 0);
    else if (a == "--nolearn-from") nolearn_from = strtoull(next().c_str(), 0, 0);
    else if (a == "--out") out_path = next();
    else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
  }
  if (out_path.empty()) { fprintf(stderr, "--out required\n"); return 2; }
  if (dump > T) dump = T;

  std::vector<TopoEntry> topo;
  for (auto& e : Split(topo_s, ',')) {
    auto f = Split(e, ':');
    if (f.size() != 3) { fprintf(stderr, "bad topo entry %s\n", e.c_str()); return 2; }
    topo.push_back({atoi(f[0].c_str()), (unsigned)strtoul(f[1].c_str(), 0, 0),
                    strtod(f[2].c_str(), 0)});
  }
  const int M = topo.size();

  // Mirror of the Predictor constructor's set-up (predictor.cpp:17-40).
  ShortTermMemory stm;
  LongTermMemory ltm;
  for (int i = 0; i < n; ++i) stm.AddPrediction("synthetic", false, nullptr);
  if (skip_s != "none")
    for (auto& s : Split(skip_s, ',')) stm.models_with_skip_connection.push_back(atoi(s.c_str()));
  std::vector<unsigned int> ctx(M, 0);  // the context variables the mixers alias (mixer.h:31)
  std::vector<std::unique_ptr<Mixer>> mixers;
  for (int j = 0; j < M; ++j) {
    // learning-rate literals are doubles narrowed to the float parameter (predictor.cpp:254+).
    mixers.emplace_back(new Mixer(stm, ltm, ctx[j], topo[j].lr, topo[j].layer, topo[j].table,
                                  "m", false));
  }
  stm.predictions.resize(stm.num_predictions);
  stm.predictions = 0;
  stm.mixer_layer0_outputs.resize(stm.num_layer0_mixers);
  stm.mixer_layer0_outputs = 0;
  stm.mixer_layer1_outputs.resize(stm.num_layer1_mixers);
  stm.mixer_layer1_outputs = 0;
  const int L0 = stm.num_layer0_mixers, L1 = stm.num_layer1_mixers;
  const int has_final = (M > L0 + L1) ? 1 : 0;

  gmx_synth g;
  gmx_synth_init(&g, seed, n, M, ctx_mode, ctx_mod, zero_mod, bit_mode);
  std::vector<float> pred(n, 0.0f);
  std::vector<uint8_t> active(n, 0);

  std::ofstream f(out_path, std::ios::binary);
  Put(f, (uint32_t)0x44584D47u);  // "GMXD"
  Put(f, (uint32_t)1);
  Put(f, (uint32_t)n);
  Put(f, (uint32_t)M);
  Put(f, (uint32_t)L0);
  Put(f, (uint32_t)L1);
  Put(f, (uint32_t)has_final);
  Put(f, (uint32_t)stm.models_with_skip_connection.size());
  Put(f, (uint64_t)T);
  Put(f, (uint64_t)dump);

  uint32_t h32 = 0;
  double acc = 0;
  uint64_t h64 = 1469598103934665603ull;
  std::vector<float> outs(M);
  for (uint64_t t = 0; t < T; ++t) {
    int bit = gmx_synth_step(&g, pred.data(), active.data(), ctx.data());
    // Predictor::Predict (predictor.cpp:360-376) with analysis off: stale slots stay.
    stm.active_models.clear();
    for (int i = 0; i < n; ++i) {
      if (active[i]) {
        stm.SetLogitPrediction(pred[i], i);
      } else {
        stm.predictions[i] = pred[i];  // what a model that wrote earlier left behind
      }
    }
    for (auto& m : mixers) m->Predict(stm, ltm);
    for (int k = 0; k < L0; ++k) outs[k] = stm.mixer_layer0_outputs[k];
    for (int k = 0; k < L1; ++k) outs[L0 + k] = stm.mixer_layer1_outputs[k];
    if (has_final) outs[L0 + L1] = stm.final_mixer_output;
    float out = outs[M - 1];
    float prob = Sigmoid::Logistic(out);
    float eps = 0.0001;
    if (prob < eps)
      prob = eps;
    else if (prob > 1 - eps)
      prob = 1 - eps;
    uint32_t ob;
    memcpy(&ob, &out, 4);
    h32 = h32 * 16777619u ^ ob;
    acc += out;
    for (int k = 0; k < M; ++k) {
      uint32_t b;
      memcpy(&b, &outs[k], 4);
      h64 = (h64 ^ b) * 1099511628211ull;
    }
    uint32_t pb;
    memcpy(&pb, &prob, 4);
    h64 = (h64 ^ pb) * 1099511628211ull;
    if (t < dump) {
      f.write(reinterpret_cast<const char*>(outs.data()), 4 * M);
      Put(f, prob);
    }
    stm.new_bit = bit;  // Predictor::Perceive (predictor.cpp:378-381)
    if (t < nolearn_from)
      for (auto& m : mixers) m->Learn(stm, ltm);  // Predictor::Learn (predictor.cpp:383-387)
  }
  Put(f, h32);
  Put(f, acc);
  Put(f, h64);

  // Persistent state: Mixer::WriteToDisk (3 x u64 each, mixer.cpp:178-182) and the mixer
  // section of LongTermMemory::WriteToDisk (long-term-memory.cpp:35-55).
  std::string tmp = out_path + ".tmp";
  {
    std::ofstream s(tmp, std::ios::binary);
    for (auto& m : mixers) m->WriteToDisk(&s);
  }
  {
    std::ifstream s(tmp, std::ios::binary);
    std::vector<char> b((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
    Put(f, (uint64_t)b.size());
    f.write(b.data(), b.size());
  }
  {
    std::ofstream s(tmp, std::ios::binary);
    FILE* keep = stdout;  // LongTermMemory::WriteToDisk printf()s section sizes
    (void)keep;
    ltm.WriteToDisk(&s);
  }
  {
    std::ifstream s(tmp, std::ios::binary);
    std::vector<char> b((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
    // With no indirect / lstm / match memories the file is the mixer section followed by
    // the u64 history length (0): strip those trailing 8 bytes.
    size_t len = b.size() >= 8 ? b.size() - 8 : 0;
    Put(f, (uint64_t)len);
    f.write(b.data(), len);
  }
  remove(tmp.c_str());
  // Per-mixer GetMemoryUsage (mixer.cpp:197-205).
  for (auto& m : mixers) Put(f, (uint64_t)m->GetMemoryUsage(stm, ltm));
  f.close();
  fprintf(stderr, "N=%d M=%d (%d/%d/%d) T=%llu acc=%.6f hash=%08x h64=%016llx\n", n, M, L0, L1,
          has_final, (unsigned long long)T, acc, h32, (unsigned long long)h64);
  return 0;
}
