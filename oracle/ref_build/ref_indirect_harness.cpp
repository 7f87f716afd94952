// ref_indirect_harness.cpp -- TEST INFRASTRUCTURE, builds only where /root/reference exists.
//
// Drives the REFERENCE's own `Indirect` class (src/models/indirect.{h,cpp}) with the synthetic
// stream of oracle/gmx_ind_synth.h the way `Predictor` drives it (predictor.cpp:360-387: clear
// active_models, every model Predict, set new_bit, every model Learn) and dumps what it
// computed, plus the two state machines the models consult (ShortTermMemory::nonstationary,
// ::run_map) as tables.  Ground truth for oracle/gmx_oracle_ind.c and the HIP path.
//
// usage: ref_indirect_harness --models "table:lr,table:lr,..." --bits T [--dump D] [--seed S]
//          [--ctx-mod a,b,c,d] [--nolearn-from T0] --out file
// dump format "GMXI": u32 magic, K, T, D; u8 nonstationary_next[256][2]; u8 run_map_next[256][2];
//   D x { float pred[2K]; u8 active[2K] }; u64 fnv over all T bits of (pred bits, active);
//   u64 usage[K]; u64 long_len; long bytes (LongTermMemory::WriteToDisk: indirect section first)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "models/indirect.h"  // reference (via -I/root/reference/src)

extern "C" {
#include "../gmx_ind_synth.h"
}

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
  std::string models_s = "256:0.02", out_path, mod_s = "0,0,0,0";
  uint64_t T = 1000, dump = 0, seed = 0, nolearn_from = ~0ull;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> std::string { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "--models") models_s = next();
    else if (a == "--bits") T = strtoull(next().c_str(), 0, 0);
    else if (a == "--dump") dump = strtoull(next().c_str(), 0, 0);
    else if (a == "--seed") seed = strtoull(next().c_str(), 0, 0);
    else if (a == "--ctx-mod") mod_s = next();
    else if (a == "--nolearn-from") nolearn_from = strtoull(next().c_str(), 0, 0);
    else if (a == "--out") out_path = next();
    else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
  }
  if (out_path.empty()) { fprintf(stderr, "--out required\n"); return 2; }
  if (dump > T) dump = T;
  uint32_t ctx_mod[4] = {0, 0, 0, 0};
  {
    auto f = Split(mod_s, ',');
    for (size_t i = 0; i < 4 && i < f.size(); ++i) ctx_mod[i] = (uint32_t)strtoul(f[i].c_str(), 0, 0);
  }

  ShortTermMemory stm;
  LongTermMemory ltm;
  auto entries = Split(models_s, ',');
  const int K = (int)entries.size();
  std::vector<unsigned int> ctx(K, 0);
  std::vector<std::unique_ptr<Indirect>> models;
  for (int k = 0; k < K; ++k) {
    auto f = Split(entries[k], ':');
    models.emplace_back(new Indirect(stm, ltm, (float)strtod(f[1].c_str(), 0),
                                     (unsigned)strtoul(f[0].c_str(), 0, 0), ctx[k], "m", false));
  }
  stm.predictions.resize(stm.num_predictions);  // predictor.cpp:29-30
  stm.predictions = 0;

  std::ofstream out(out_path, std::ios::binary);
  Put(out, (uint32_t)0x49584d47u);  // "GMXI"
  Put(out, (uint32_t)K);
  Put(out, (uint32_t)T);
  Put(out, (uint32_t)dump);
  for (int s = 0; s < 256; ++s)
    for (int b = 0; b < 2; ++b) Put(out, (uint8_t)stm.nonstationary.Next(s, b));
  for (int s = 0; s < 256; ++s)
    for (int b = 0; b < 2; ++b) Put(out, (uint8_t)stm.run_map.Next(s, b));

  gmx_ind_synth g;
  gmx_ind_synth_init(&g, seed, K, ctx_mod);
  uint64_t h = 0xcbf29ce484222325ull;
  std::vector<uint8_t> active(2 * K);
  for (uint64_t t = 0; t < T; ++t) {
    stm.bit_context = gmx_ind_synth_contexts(&g, ctx.data());
    stm.active_models.clear();
    for (auto& m : models) m->Predict(stm, ltm);
    std::fill(active.begin(), active.end(), 0);
    for (int i : stm.active_models) active[i] = 1;
    for (int i = 0; i < 2 * K; ++i) {
      float v = stm.predictions[i];
      uint32_t u;
      memcpy(&u, &v, 4);
      h = (h ^ u) * 0x100000001b3ull;
      h = (h ^ active[i]) * 0x100000001b3ull;
      if (t < dump) Put(out, v);
    }
    if (t < dump) out.write((const char*)active.data(), 2 * K);
    stm.new_bit = gmx_ind_synth_bit(&g, ctx.data());
    if (t < nolearn_from)
      for (auto& m : models) m->Learn(stm, ltm);
  }
  Put(out, h);
  for (auto& m : models) Put(out, (uint64_t)m->GetMemoryUsage(stm, ltm));
  // LongTermMemory::WriteToDisk (long-term-memory.cpp:6-34): the indirect section comes first
  std::string tmp = out_path + ".long";
  {
    std::ofstream lf(tmp, std::ios::binary);
    ltm.WriteToDisk(&lf);
  }
  std::ifstream lf(tmp, std::ios::binary);
  std::string bytes((std::istreambuf_iterator<char>(lf)), std::istreambuf_iterator<char>());
  Put(out, (uint64_t)bytes.size());
  out.write(bytes.data(), bytes.size());
  remove(tmp.c_str());
  return 0;
}
