// gmx_model_adapter.h -- the binding a gmix maintainer adds: `gmx::GpuMixer`, a `Model`
// (src/model.h:16-38) with EXACTLY `Mixer`'s constructor signature (src/mixer/mixer.h:17-19)
// whose work is done by libgmxmix.so on an MI355X (C ABI: include/gmxmix.h).
//
// Unlike gmx_mixer.h (a stand-alone mirror with its own blackboard type) this header is compiled
// AGAINST THE REFERENCE: it includes the reference's own model.h, so it needs -I<gmix>/src, and it
// works on the reference's own ShortTermMemory / LongTermMemory.  Switching gmix over is
//     #include "gmx_model_adapter.h"                      in src/predictor.cpp
//     new Mixer(  ->  new gmx::GpuMixer(                   33 times in Predictor::AddMixers
// and nothing else: Predictor, the runners, the coder, LongTermMemory and the tester stay as they
// are (oracle/ref_build/Makefile builds gmix and the reference's tester that way, from a patched
// temporary copy of predictor.cpp; tests/test_gpu_dropin.py runs them against the stock build).
//
// How 33 objects become one device bank
//   * Every GpuMixer registers with the bank of its Predictor (found through the address of the
//     LongTermMemory it is constructed with) and -- like Mixer::Mixer (mixer.cpp:3-27) -- with
//     ShortTermMemory::AddMixer and LongTermMemory::mixers, so num_layer0_mixers,
//     model_descriptions, mixer_index_to_model_ptr and the table sizes in LongTermMemory are what
//     the reference has.
//   * Predictor::Predict calls the models in order (predictor.cpp:366-368): the FIRST mixer's
//     Predict reads the 33 aliased context variables (mixer.h:31) and runs gmx_bank_forward for
//     all of them; it leaves mixer_layer0_outputs / mixer_layer1_outputs / final_mixer_output
//     where 33 Mixer::Predict calls would (mixer.cpp:99-105).  Learn likewise
//     (gmx_bank_learn).  Nothing between the first and the last mixer's call touches the
//     blackboard (they are consecutive entries of models_), and Learn receives the blackboard
//     const, so "read at the first mixer's call" is "read at each mixer's call".
//   * State lives on the GPU.  LongTermMemory::mixers serves as the staging area for the
//     reference's own serialisers: WriteToDisk (called for every model BEFORE
//     LongTermMemory::WriteToDisk, predictor.cpp:396-400) exports the bank into it, so the
//     reference writes the .long file itself, byte for byte; ReadFromDisk (called BEFORE
//     LongTermMemory::ReadFromDisk, predictor.cpp:412-416) notes that the tables the reference is
//     about to read must be imported, which the next call of any method does.  Copy
//     (predictor.cpp:42-48) is a device-to-device gmx_bank_copy.
//   * There is no CPU fallback: without a usable MI355X the first call prints the reason and
//     abort()s, the failure convention of the reference's own tester (tester.cpp:318-321).
#ifndef GMX_MODEL_ADAPTER_H_
#define GMX_MODEL_ADAPTER_H_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "model.h"  // the reference's (src/model.h): Model, ShortTermMemory, LongTermMemory

#include "gmxmix.h"  // include/gmxmix.h

namespace gmx {

class GpuMixer;

// All mixers of one Predictor: one gmx_group with one stream.
class GpuMixerBank {
 public:
  // The bank of the Predictor that owns `ltm` (created on first use, gone with its last mixer).
  static std::shared_ptr<GpuMixerBank> For(ShortTermMemory& stm, LongTermMemory& ltm) {
    auto& reg = Registry();
    auto it = reg.find(&ltm);
    if (it != reg.end())
      if (auto sp = it->second.lock()) return sp;
    std::shared_ptr<GpuMixerBank> sp(new GpuMixerBank(stm, ltm));
    reg[&ltm] = sp;
    return sp;
  }
  ~GpuMixerBank() {
    if (group_) gmx_group_destroy(group_);
    Registry().erase(&ltm_);
  }
  GpuMixerBank(const GpuMixerBank&) = delete;
  GpuMixerBank& operator=(const GpuMixerBank&) = delete;

  gmx_group* group() { return group_; }

 private:
  friend class GpuMixer;
  GpuMixerBank(ShortTermMemory& stm, LongTermMemory& ltm) : stm_(stm), ltm_(ltm) {}
  static std::map<const LongTermMemory*, std::weak_ptr<GpuMixerBank>>& Registry() {
    static std::map<const LongTermMemory*, std::weak_ptr<GpuMixerBank>> r;
    return r;
  }
  [[noreturn]] static void Fatal(const char* what, int rc) {
    fprintf(stderr, "\ngmx::GpuMixer: %s: %s %s\n(the mixers run on an MI355X through libgmxmix.so; there is no CPU fallback)\n",
            what, gmx_strerror(rc), rc == GMX_ERR_HIP ? gmx_last_error() : "");
    abort();
  }
  static void Check(const char* what, int rc) {
    if (rc != GMX_OK) Fatal(what, rc);
  }
  int Register(GpuMixer* m, int layer, unsigned table_size, float lr, int memory_index) {
    gmx_mixer_desc d;
    d.layer = layer;
    d.table_size = table_size;
    d.learning_rate = lr;
    descs_.push_back(d);
    mixers_.push_back(m);
    memory_index_.push_back(memory_index);
    return (int)descs_.size() - 1;
  }
  // Device bank from what the constructors registered (all of them have run by the time any
  // method of a model is called: Predictor's constructor finishes first).
  void Ensure() {
    if (group_) return;
    std::vector<int32_t> skip(stm_.models_with_skip_connection.begin(), stm_.models_with_skip_connection.end());
    gmx_topology t;
    t.n_inputs = stm_.num_predictions;
    t.n_skip = (int32_t)skip.size();
    t.skip_index = skip.data();
    t.n_mixers = (int32_t)descs_.size();
    t.mixers = descs_.data();
    const char* dev = getenv("GMX_DEVICE");
    Check("gmx_group_create", gmx_group_create(&group_, &t, 1, dev ? atoi(dev) : 0));
    outputs_.assign(descs_.size(), 0.f);
    contexts_.assign(descs_.size(), 0u);
    short_cache_.assign(descs_.size() * 24, 0);
    for (size_t j = 0; j < descs_.size(); ++j) {  // a fresh Mixer: steps_ 0, max_steps_ 1, contexts_seen_ 0
      uint64_t init[3] = {0, 1, 0};
      memcpy(&short_cache_[24 * j], init, 24);
    }
    short_in_ = short_cache_;
  }
  // Ensure + import of tables the reference's LongTermMemory::ReadFromDisk has read since
  // ReadFromDisk was called on the mixers.
  void Settle() {
    Ensure();
    if (!import_pending_) return;
    import_pending_ = false;
    std::vector<char> buf;
    auto put = [&buf](const void* p, size_t n) {
      const char* c = static_cast<const char*>(p);
      buf.insert(buf.end(), c, c + n);
    };
    for (size_t j = 0; j < descs_.size(); ++j) {  // the layout of long-term-memory.cpp:35-54
      auto& table = ltm_.mixers[memory_index_[j]].mixer_table;
      uint32_t n = 0, input_size = 0;
      for (auto& row : table)
        if (row) {
          ++n;
          input_size = (uint32_t)row->weights.size();
        }
      put(&n, 4);
      put(&input_size, 4);
      for (uint32_t c = 0; c < table.size(); ++c) {
        if (!table[c]) continue;
        uint64_t steps = table[c]->steps;
        put(&c, 4);
        put(&steps, 8);
        put(&table[c]->weights[0], 4 * table[c]->weights.size());
      }
    }
    Check("gmx_bank_import", gmx_bank_import(group_, 0, buf.data(), buf.size(), short_in_.data(), short_in_.size()));
    staged_ = true;
  }
  // Bank -> LongTermMemory::mixers (+ the 3 x u64 of every mixer), for the reference's writers.
  void Stage() {
    Settle();
    size_t nl = 0, ns = 0;
    Check("gmx_bank_export", gmx_bank_export(group_, 0, nullptr, &nl, nullptr, &ns));
    std::vector<char> l(nl ? nl : 1);
    short_cache_.assign(ns ? ns : 1, 0);
    Check("gmx_bank_export", gmx_bank_export(group_, 0, l.data(), &nl, short_cache_.data(), &ns));
    short_cache_.resize(ns);
    const char* p = l.data();
    for (size_t j = 0; j < descs_.size(); ++j) {
      auto& table = ltm_.mixers[memory_index_[j]].mixer_table;
      for (auto& row : table) row.reset();
      uint32_t n, input_size;
      memcpy(&n, p, 4);
      memcpy(&input_size, p + 4, 4);
      p += 8;
      for (uint32_t i = 0; i < n; ++i) {
        uint32_t c;
        memcpy(&c, p, 4);
        MixerData* row = new MixerData(input_size);
        memcpy(&row->steps, p + 4, 8);
        memcpy(&row->weights[0], p + 12, 4 * (size_t)input_size);
        table[c].reset(row);
        p += 12 + 4 * (size_t)input_size;
      }
    }
    staged_ = true;
  }
  // The staged rows are only for the serialisers: give the host memory back once the stream moves on.
  void Unstage() {
    if (!staged_ || import_pending_) return;
    staged_ = false;
    for (size_t j = 0; j < descs_.size(); ++j)
      for (auto& row : ltm_.mixers[memory_index_[j]].mixer_table) row.reset();
  }
  void PredictAll(ShortTermMemory& stm);
  void LearnAll(const ShortTermMemory& stm) {
    Settle();
    Check("gmx_bank_learn", gmx_bank_learn(group_, 0, stm.new_bit));
  }
  void CopyFrom(GpuMixerBank& o) {
    o.Settle();
    Ensure();
    import_pending_ = false;  // whatever LongTermMemory::Copy moves into the staging area is not ours to import
    Check("gmx_bank_copy", gmx_bank_copy(group_, 0, o.group_, 0));
  }

  ShortTermMemory& stm_;
  LongTermMemory& ltm_;
  gmx_group* group_ = nullptr;
  std::vector<gmx_mixer_desc> descs_;
  std::vector<GpuMixer*> mixers_;
  std::vector<int> memory_index_;
  std::vector<float> outputs_;
  std::vector<uint32_t> contexts_;
  std::vector<char> short_cache_, short_in_;
  bool import_pending_ = false, staged_ = false;
};

class GpuMixer : public Model {
 public:
  // mixer/mixer.h:17-19, argument for argument.
  GpuMixer(ShortTermMemory& short_term_memory, LongTermMemory& long_term_memory, unsigned int& context,
           float learning_rate, int layer_number, unsigned int table_size, std::string description,
           bool enable_analysis)
      : context_(context), bank_(GpuMixerBank::For(short_term_memory, long_term_memory)) {
    // mixer.cpp:12-15: the registrations Mixer::Mixer makes
    short_term_memory.AddMixer(description, layer_number, enable_analysis, this);
    int memory_index = (int)long_term_memory.mixers.size();
    long_term_memory.mixers.push_back(MixerMemory(table_size));
    index_ = bank_->Register(this, layer_number, table_size, learning_rate, memory_index);
  }
  void Predict(ShortTermMemory& short_term_memory, const LongTermMemory&) override {
    if (index_ == 0) bank_->PredictAll(short_term_memory);
  }
  void Learn(const ShortTermMemory& short_term_memory, LongTermMemory&) override {
    if (index_ != 0) return;
    bank_->LearnAll(short_term_memory);
    bank_->Unstage();
  }
  // mixer.cpp:178-182: steps_, max_steps_, contexts_seen_
  void WriteToDisk(std::ofstream* s) override {
    if (index_ == 0) bank_->Stage();
    s->write(&bank_->short_cache_[24 * (size_t)index_], 24);
  }
  // mixer.cpp:184-188
  void ReadFromDisk(std::ifstream* s) override {
    bank_->Ensure();
    s->read(&bank_->short_in_[24 * (size_t)index_], 24);
    bank_->import_pending_ = true;
  }
  // mixer.cpp:190-195 (+ the mixers' share of LongTermMemory::Copy, long-term-memory.cpp:201-214)
  void Copy(const MemoryInterface* m) override {
    const GpuMixer* orig = static_cast<const GpuMixer*>(m);
    if (index_ == 0) bank_->CopyFrom(*orig->bank_);
  }
  // mixer.cpp:197-205
  unsigned long long GetMemoryUsage(const ShortTermMemory&, const LongTermMemory&) override {
    bank_->Settle();
    uint64_t v = 0;
    GpuMixerBank::Check("gmx_bank_memory_usage", gmx_bank_memory_usage(bank_->group_, 0, index_, &v));
    return v;
  }
  unsigned int context() const { return context_; }

 private:
  unsigned int& context_;  // aliases a field of the Predictor's blackboard (mixer.h:31)
  std::shared_ptr<GpuMixerBank> bank_;
  int index_;  // construction order within the bank
};

inline void GpuMixerBank::PredictAll(ShortTermMemory& stm) {
  Settle();
  for (size_t j = 0; j < mixers_.size(); ++j) contexts_[j] = mixers_[j]->context();  // read at call time
  static_assert(sizeof(int) == sizeof(int32_t), "active_models is passed as it stands");
  float p = 0.5f;
  Check("gmx_bank_forward",
        gmx_bank_forward(group_, 0, &stm.predictions[0], stm.active_models.data(), (int)stm.active_models.size(),
                         contexts_.data(), &p, outputs_.data()));
  // mixer.cpp:99-105: where each Mixer::Predict leaves its result
  size_t j = 0;
  for (int k = 0; k < stm.num_layer0_mixers; ++k) stm.mixer_layer0_outputs[k] = outputs_[j++];
  for (int k = 0; k < stm.num_layer1_mixers; ++k) stm.mixer_layer1_outputs[k] = outputs_[j++];
  if (j < outputs_.size()) stm.final_mixer_output = outputs_[j];
}

}  // namespace gmx

#endif  // GMX_MODEL_ADAPTER_H_
