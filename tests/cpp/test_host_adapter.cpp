// test_host_adapter.cpp -- the C++ host mirror (gmix_amd/host/gmx_mixer.h) driven the way the
// reference's Predictor drives its mixers, checked against the CPU oracle bit for bit, plus
// the reference tester's restart / copy / generation invariants (runner/tester.cpp:323-366)
// restricted to the mixer slice.  Needs an MI355X.  Built and run by tests/test_gpu_host_cpp.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../../gmix_amd/host/gmx_mixer.h"
extern "C" {
#include "../../oracle/gmx_synth.h"
struct gmxo_bank;
gmxo_bank* gmxo_create(int, int, const int*, int, const int*, const uint32_t*, const float*);
void gmxo_destroy(gmxo_bank*);
float gmxo_predict(gmxo_bank*, const float*, const int*, int, const uint32_t*, float*);
void gmxo_learn(gmxo_bank*, int);
size_t gmxo_export_short(const gmxo_bank*, void*, size_t);
size_t gmxo_export_long(const gmxo_bank*, void*, size_t);
}

static void Fail(const char* what, long t = -1) {
  fprintf(stderr, "Test failed: %s (bit %ld)\n", what, t);
  fflush(stderr);
  abort();  // the reference's convention (tester.cpp:318-321)
}

// predictor.cpp:254-357: (layer, table_size, learning rate literal) in construction order
struct Row { int layer; unsigned table; double lr; const char* name; };
static const Row kStock[] = {
    {0, 1 << 8, 0.005, "Mixer0(last byte)"}, {0, 1 << 8, 0.0055, "Mixer0(4th last byte)"},
    {0, 1 << 16, 0.003, "Mixer0(2nd last + recent)"}, {0, 1 << 15, 0.0045, "Mixer0(4 byte hash)"},
    {0, 1 << 8, 0.006, "Mixer0(indirect_3_24_1_8)"}, {0, 1 << 8, 0.004, "Mixer0(2nd last byte)"},
    {0, 1 << 3, 0.0005, "Mixer0(longest match)"}, {0, 1 << 16, 0.0035, "Mixer0(2 bytes)"},
    {0, 1 << 8, 0.0065, "Mixer0(3rd last byte)"}, {0, 1 << 15, 0.0025, "Mixer0(3 byte hash)"},
    {0, 1 << 8, 0.001, "Mixer0(last byte)"}, {0, 1 << 16, 0.002, "Mixer0(last byte + recent)"},
    {0, 1 << 4, 0.005, "Mixer0(interval_16_4)"}, {0, 1 << 8, 0.0045, "Mixer0(interval_16_8)"},
    {0, 1 << 12, 0.0055, "Mixer0(interval_16_12)"}, {0, 1 << 3, 0.004, "Mixer0(interval_32_3)"},
    {0, 1 << 6, 0.0035, "Mixer0(interval_32_6)"}, {0, 1 << 16, 0.006, "Mixer0(skip_0_2)"},
    {0, 1 << 12, 0.003, "Mixer0(interval_32_12)"}, {0, 1 << 4, 0.0065, "Mixer0(interval_64_4)"},
    {0, 1 << 8, 0.003, "Mixer0(interval_64_8)"}, {0, 1 << 12, 0.0025, "Mixer0(interval_64_12)"},
    {0, 1 << 8, 0.002, "Mixer0(lstm_prediction)"}, {0, 1, 0.0005, "Mixer0(no context)"},
    {1, 1 << 8, 0.0045, "Mixer1(2nd last byte)"}, {1, 1, 0.0035, "Mixer1(no context)"},
    {1, 1 << 8, 0.003, "Mixer1(recent_bits)"}, {1, 1 << 8, 0.002, "Mixer1(3rd last byte)"},
    {1, 1 << 8, 0.0025, "Mixer1(last byte)"}, {1, 1 << 8, 0.00001, "Mixer1(recent_bits)"},
    {1, 1 << 3, 0.0008, "Mixer1(longest match)"}, {1, 1, 0.0004, "Mixer1(no context)"},
    {2, 1, 0.0005, "Mixer(final layer)"}};
static const int kM = sizeof(kStock) / sizeof(kStock[0]);
static const int kN = 90;

struct Rig {
  std::vector<unsigned int> ctx;  // the context variables the mixers alias
  gmx::MixerPredictor p;
  Rig() : ctx(kM, 0) {
    for (int i = 0; i < kN; ++i) p.stm.AddPrediction("synthetic", false, nullptr);
    p.stm.models_with_skip_connection.push_back(1);  // lstm-model.cpp:12-14
    for (int j = 0; j < kM; ++j)
      p.AddMixer(ctx[j], (float)kStock[j].lr, kStock[j].layer, kStock[j].table, kStock[j].name);
    if (p.Finalize() != GMX_OK) Fail(gmx_last_error());
  }
};

struct Stream {
  gmx_synth g;
  std::vector<float> pred;
  std::vector<uint8_t> active;
  Stream(uint64_t seed) : pred(kN, 0.f), active(kN, 0) { gmx_synth_init(&g, seed, kN, kM, 3, 9, 7, 1); }
  // one bit: drives the blackboard like the feature models would; returns the coded bit
  int Step(Rig& r) {
    int bit = gmx_synth_step(&g, pred.data(), active.data(), r.ctx.data());
    r.p.BeginBit();
    for (int i = 0; i < kN; ++i) {
      if (active[i]) r.p.stm.SetLogitPrediction(pred[i], i);
      else r.p.stm.predictions[i] = pred[i];  // stale slot of a silent model
    }
    return bit;
  }
};

static std::vector<char> Slurp(const std::string& path) {
  std::ifstream s(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>());
}

static bool SameBits(float a, float b) { return memcmp(&a, &b, 4) == 0; }

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const long T = 600;
  // ---- 1. straight run against the oracle, every output of every bit ---------------------
  std::vector<float> p_straight(T);
  {
    Rig r;
    Stream st(4711);
    std::vector<int> layer(kM), skip(1, 1);
    std::vector<uint32_t> table(kM);
    std::vector<float> lr(kM);
    for (int j = 0; j < kM; ++j) { layer[j] = kStock[j].layer; table[j] = kStock[j].table; lr[j] = (float)kStock[j].lr; }
    gmxo_bank* ob = gmxo_create(kN, 1, skip.data(), kM, layer.data(), table.data(), lr.data());
    std::vector<float> oo(kM);
    for (long t = 0; t < T; ++t) {
      int bit = st.Step(r);
      float p = r.p.Predict();
      float po = gmxo_predict(ob, &r.p.stm.predictions[0], r.p.stm.active_models.data(),
                              (int)r.p.stm.active_models.size(), r.ctx.data(), oo.data());
      if (!SameBits(p, po)) Fail("probability differs from oracle", t);
      for (int k = 0; k < 24; ++k) if (!SameBits(r.p.stm.mixer_layer0_outputs[k], oo[k])) Fail("layer-0 output", t);
      for (int k = 0; k < 8; ++k) if (!SameBits(r.p.stm.mixer_layer1_outputs[k], oo[24 + k])) Fail("layer-1 output", t);
      if (!SameBits(r.p.stm.final_mixer_output, oo[32])) Fail("final output", t);
      p_straight[t] = p;
      r.p.Perceive(bit);
      r.p.Learn();
      gmxo_learn(ob, bit);
    }
    r.p.WriteCheckpoint(dir + "/gmx_straight");
    std::vector<char> so(gmxo_export_short(ob, nullptr, 0)), lo(gmxo_export_long(ob, nullptr, 0));
    gmxo_export_short(ob, so.data(), so.size());
    gmxo_export_long(ob, lo.data(), lo.size());
    if (Slurp(dir + "/gmx_straight.short") != so) Fail(".short differs from the reference format");
    if (Slurp(dir + "/gmx_straight.long") != lo) Fail(".long differs from the reference format");
    if (r.p.mixers[0]->GetMemoryUsage(r.p.stm, r.p.bank) < 29 + 8 * 256) Fail("GetMemoryUsage");
    gmxo_destroy(ob);
    if (r.p.bank.status() != GMX_OK) Fail("bank status");
  }
  // ---- 2. TestCompressionWithRestart / WithCopyRestart (tester.cpp:330-348) ---------------
  for (int use_copy = 0; use_copy < 2; ++use_copy) {
    Rig a;
    Stream st(4711);
    for (long t = 0; t < T / 2; ++t) {
      int bit = st.Step(a);
      if (!SameBits(a.p.Predict(), p_straight[t])) Fail("first half", t);
      a.p.Perceive(bit);
      a.p.Learn();
    }
    Rig b;
    // the context variables are ShortTermMemory state, which the reference restores through
    // ShortTermMemory::ReadFromDisk / Copy: outside the mixer slice, so carried over by hand
    std::copy(a.ctx.begin(), a.ctx.end(), b.ctx.begin());
    if (use_copy) {
      b.p.Copy(a.p);
    } else {
      a.p.WriteCheckpoint(dir + "/gmx_half");
      b.p.ReadCheckpoint(dir + "/gmx_half");
    }
    for (long t = T / 2; t < T; ++t) {
      int bit = st.Step(b);
      if (!SameBits(b.p.Predict(), p_straight[t])) Fail(use_copy ? "after Copy" : "after restart", t);
      b.p.Perceive(bit);
      b.p.Learn();
    }
    b.p.WriteCheckpoint(dir + "/gmx_resumed");
    if (Slurp(dir + "/gmx_resumed.short") != Slurp(dir + "/gmx_straight.short")) Fail("resumed .short");
    if (Slurp(dir + "/gmx_resumed.long") != Slurp(dir + "/gmx_straight.long")) Fail("resumed .long");
  }
  // ---- 3. TestGeneration (tester.cpp:358-366): no Learn, .long unchanged -----------------
  {
    Rig r;
    r.p.ReadCheckpoint(dir + "/gmx_straight");
    Stream st(99);
    for (long t = 0; t < 100; ++t) {
      int bit = st.Step(r);
      float p = r.p.Predict();
      if (!(p >= 0.0001f && p <= 0.9999f)) Fail("probability range", t);
      r.p.Perceive(bit);
    }
    r.p.WriteCheckpoint(dir + "/gmx_gen");
    if (Slurp(dir + "/gmx_gen.long") != Slurp(dir + "/gmx_straight.long")) Fail("generation changed .long");
    if (Slurp(dir + "/gmx_gen.short") != Slurp(dir + "/gmx_straight.short")) Fail("generation changed mixer .short");
  }
  printf("Tests passed.\n");
  return 0;
}
