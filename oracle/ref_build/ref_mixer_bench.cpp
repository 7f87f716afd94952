// ref_mixer_bench.cpp -- TEST/BENCH INFRASTRUCTURE (CPU baseline, kind "reference").
//
// Times the REFERENCE's own Mixer::Predict + Mixer::Learn (src/mixer/mixer.cpp:51-176),
// single thread, on the synthetic stream of oracle/gmx_synth.h.  Inputs are generated in
// chunks outside the timed bracket, so the figure is mixer-only bits/s (BASELINE.md
// section 3.1), comparable with the GPU figure whose inputs are resident in HBM.
//
// usage: ref_mixer_bench --n N --topo "layer:table:lr,..." [--skip i,j] --bits T [--seed S] [--ctx-mode 0..3 --ctx-mod K]
// prints one JSON line.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mixer/mixer.h"  // reference (via -I/root/reference/src)

extern "C" {
#include "../gmx_synth.h"
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

int main(int argc, char** argv) {
  int n = 256;
  std::string topo_s = "0:65536:0.005", skip_s = "1";
  uint64_t T = 1000000, seed = 0;
  int ctx_mode = 0;
  unsigned ctx_mod = 1;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> std::string { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "--n") n = atoi(next().c_str());
    else if (a == "--topo") topo_s = next();
    else if (a == "--skip") skip_s = next();
    else if (a == "--bits") T = strtoull(next().c_str(), 0, 0);
    else if (a == "--seed") seed = strtoull(next().c_str(), 0, 0);
    else if (a == "--ctx-mode") ctx_mode = atoi(next().c_str());
    else if (a == "--ctx-mod") ctx_mod = (unsigned)strtoul(next().c_str(), 0, 0);
    else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
  }
  ShortTermMemory stm;
  LongTermMemory ltm;
  for (int i = 0; i < n; ++i) stm.AddPrediction("synthetic", false, nullptr);
  if (skip_s != "none")
    for (auto& s : Split(skip_s, ',')) stm.models_with_skip_connection.push_back(atoi(s.c_str()));
  auto entries = Split(topo_s, ',');
  const int M = entries.size();
  std::vector<unsigned int> ctx(M, 0);
  std::vector<std::unique_ptr<Mixer>> mixers;
  for (int j = 0; j < M; ++j) {
    auto f = Split(entries[j], ':');
    mixers.emplace_back(new Mixer(stm, ltm, ctx[j], strtod(f[2].c_str(), 0), atoi(f[0].c_str()),
                                  (unsigned)strtoul(f[1].c_str(), 0, 0), "m", false));
  }
  stm.predictions.resize(stm.num_predictions);
  stm.predictions = 0;
  stm.mixer_layer0_outputs.resize(stm.num_layer0_mixers);
  stm.mixer_layer0_outputs = 0;
  stm.mixer_layer1_outputs.resize(stm.num_layer1_mixers);
  stm.mixer_layer1_outputs = 0;

  gmx_synth g;
  gmx_synth_init(&g, seed, n, M, ctx_mode, ctx_mod, 0, 0);
  const uint64_t CH = 4096;
  std::vector<float> xs(CH * n);
  std::vector<uint8_t> act(CH * n);
  std::vector<uint32_t> cs(CH * M);
  std::vector<int> bits(CH);
  std::vector<float> pred(n, 0.0f);
  std::vector<uint8_t> active(n, 0);
  double secs = 0;
  double acc = 0;
  for (uint64_t t0 = 0; t0 < T; t0 += CH) {
    uint64_t nb = (T - t0 < CH) ? (T - t0) : CH;
    for (uint64_t t = 0; t < nb; ++t) {
      bits[t] = gmx_synth_step(&g, pred.data(), active.data(), ctx.data());
      memcpy(&xs[t * n], pred.data(), 4 * n);
      memcpy(&act[t * n], active.data(), n);
      memcpy(&cs[t * M], ctx.data(), 4 * M);
    }
    auto a = std::chrono::steady_clock::now();
    for (uint64_t t = 0; t < nb; ++t) {
      // what the feature models would have left on the blackboard for this bit
      stm.active_models.clear();
      const float* x = &xs[t * n];
      const uint8_t* ac = &act[t * n];
      for (int i = 0; i < n; ++i) {
        stm.predictions[i] = x[i];
        if (ac[i]) stm.active_models.push_back(i);
      }
      for (int j = 0; j < M; ++j) ctx[j] = cs[t * M + j];
      for (auto& m : mixers) m->Predict(stm, ltm);
      stm.new_bit = bits[t];
      for (auto& m : mixers) m->Learn(stm, ltm);
      acc += (M > stm.num_layer0_mixers + stm.num_layer1_mixers)
                 ? stm.final_mixer_output
                 : stm.mixer_layer0_outputs[stm.num_layer0_mixers - 1];
    }
    auto b = std::chrono::steady_clock::now();
    secs += std::chrono::duration<double>(b - a).count();
  }
  printf("{\"bits\": %llu, \"seconds\": %.6f, \"bits_per_s\": %.1f, \"acc\": %.6f, \"n\": %d, \"mixers\": %d}\n",
         (unsigned long long)T, secs, T / secs, acc, n, M);
  return 0;
}
