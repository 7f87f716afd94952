// gmx_mixer.h -- host-side C++ mirror of the reference's mixer plug-in surface, over the C ABI
// of libgmxmix.so (include/gmxmix.h).  Header-only, C++17, no HIP types.
//
// What it mirrors (names, argument meaning and order, void/no-throw error behaviour):
//   gmx::ShortTermMemory  the mixer-facing slice of the blackboard  (memory/short-term-memory.h:19-58,
//                         134-142; .cpp:178-213): predictions, active_models, SetPrediction,
//                         SetLogitPrediction, AddPrediction, AddMixer, the three output fields.
//   gmx::Mixer            `Mixer(stm, ltm, unsigned& context, float lr, int layer, unsigned
//                         table_size, std::string description, bool enable_analysis)`
//                         (mixer/mixer.h:17-19) with LongTermMemory replaced by gmx::MixerBank,
//                         the object that owns what LongTermMemory::mixers owns -- on the GPU.
//                         Predict/Learn/WriteToDisk/ReadFromDisk/Copy/GetMemoryUsage keep the
//                         Model signatures (model.h:22-37).
//   gmx::MixerBank        all mixers of one Predictor: one gmx_group with one stream.  The
//                         reference runs 33 Mixer::Predict calls back to back
//                         (predictor.cpp:366-368); here the FIRST mixer's Predict launches the
//                         whole bank and the others find their output already on the blackboard,
//                         likewise Learn -- so a Predictor that iterates its models_ vector in
//                         order needs no change.
//
// A maintainer switches the reference over by (1) including this header instead of
// mixer/mixer.h in predictor.cpp, (2) adding a `gmx::MixerBank mixer_bank_;` member next to
// long_term_memory_, (3) passing it where AddMixers passes long_term_memory_, and (4) calling
// mixer_bank_.Finalize() at the end of AddMixers.  INTEGRATION.md shows the diff.
#ifndef GMX_MIXER_H_
#define GMX_MIXER_H_

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <valarray>
#include <vector>

#include "../../include/gmxmix.h"

namespace gmx {

class Mixer;
class MixerBank;

// Sigmoid::Logit (mixer/sigmoid.cpp:7-13) stays on the host: only feature models call it.
inline float Logit(float p) {
  if (p < 0.0001)
    p = 0.0001;
  else if (p > 0.9999)
    p = 0.9999;
  return __builtin_logf(p / (1 - p));
}

struct ShortTermMemory {
  // short-term-memory.cpp:178-185
  int AddPrediction(std::string description, bool enable_analysis, void* ptr) {
    (void)enable_analysis;
    (void)ptr;
    ++num_predictions;
    model_descriptions.push_back(description);
    return num_predictions - 1;
  }
  // short-term-memory.cpp:187-191: a prediction of exactly 0.5 is stored but not active
  void SetPrediction(float prediction, int index) {
    predictions[index] = Logit(prediction);
    if (prediction == 0.5) return;
    active_models.push_back(index);
  }
  // short-term-memory.cpp:193-197
  void SetLogitPrediction(float prediction, int index) {
    predictions[index] = prediction;
    if (prediction == 0) return;
    active_models.push_back(index);
  }
  // short-term-memory.cpp:199-213
  int AddMixer(std::string description, int layer_number, bool enable_analysis, Mixer* ptr) {
    (void)enable_analysis;
    int index = 0;
    if (layer_number == 0) {
      index = num_layer0_mixers++;
    } else if (layer_number == 1) {
      index = num_layer1_mixers++;
    } else {
      index = num_layer0_mixers + num_layer1_mixers + 1;
    }
    model_descriptions.push_back(description);
    mixer_index_to_model_ptr.push_back(ptr);
    return index;
  }

  std::valarray<float> predictions;  // logit space
  std::vector<int> active_models;
  int num_predictions = 0;
  std::vector<std::string> model_descriptions;
  std::vector<int> models_with_skip_connection;
  int new_bit = 0;
  std::valarray<float> mixer_layer0_outputs;
  int num_layer0_mixers = 0;
  std::valarray<float> mixer_layer1_outputs;
  int num_layer1_mixers = 0;
  float final_mixer_output = 0;
  std::vector<Mixer*> mixer_index_to_model_ptr;
  // what the Indirect and LSTM mirrors of gmx_models.h read and write (short-term-memory.h:67-74,
  // :127, :161); BasicContexts / ModPPMD keep them up to date in the reference
  int recent_bits = 1;
  unsigned int bit_context = 0;
  unsigned int last_byte = 0;
  std::valarray<float> ppm_predictions = std::valarray<float>(1.0f / 256, 256);
  unsigned int lstm_prediction_context = 0;
};

// Owns the device-resident replacement of LongTermMemory::mixers for one Predictor.
class MixerBank {
 public:
  explicit MixerBank(int device = 0) : device_(device) {}
  ~MixerBank() {
    if (group_) gmx_group_destroy(group_);
  }
  MixerBank(const MixerBank&) = delete;
  MixerBank& operator=(const MixerBank&) = delete;

  // Called once after the last Mixer has been constructed (end of Predictor::AddMixers):
  // creates the device bank.  Returns a gmx_status; the void reference-style methods below
  // remember the first failure in status().
  int Finalize(const ShortTermMemory& stm) {
    if (group_) return GMX_OK;
    std::vector<int32_t> skip(stm.models_with_skip_connection.begin(),
                              stm.models_with_skip_connection.end());
    gmx_topology t;
    t.n_inputs = stm.num_predictions;
    t.n_skip = (int32_t)skip.size();
    t.skip_index = skip.data();
    t.n_mixers = (int32_t)descs_.size();
    t.mixers = descs_.data();
    n_inputs_ = stm.num_predictions;
    n_skip_ = skip.size();
    status_ = gmx_group_create(&group_, &t, 1, device_);
    outputs_.assign(descs_.size(), 0.f);
    contexts_.assign(descs_.size(), 0u);
    return status_;
  }
  bool ready() const { return group_ != nullptr; }
  // What Predictor::Predict returns for the forward pass just made (predictor.cpp:369-375).
  float last_probability() const { return last_p_; }
  int status() const { return status_; }
  gmx_group* group() { return group_; }
  size_t size() const { return descs_.size(); }

  // Mixer section of LongTermMemory::WriteToDisk / ReadFromDisk (long-term-memory.cpp:35-55,
  // 134-149): same bytes, so a .long written by the reference restores here and vice versa.
  void WriteToDisk(std::ofstream* s) {
    std::vector<char> l, sh;
    if (!Export(&l, &sh)) return;
    s->write(l.data(), l.size());
  }
  void ReadFromDisk(std::ifstream* s);
  // LongTermMemory::Copy for the mixers (long-term-memory.cpp:201-214) + Mixer::Copy x M.
  void Copy(const MixerBank* orig) {
    if (!group_ || !orig->group_) return;
    Note(gmx_bank_copy(group_, 0, orig->group_, 0));
  }

 private:
  friend class Mixer;
  int Register(Mixer* m, int layer, unsigned table_size, float lr) {
    gmx_mixer_desc d;
    d.layer = layer;
    d.table_size = table_size;
    d.learning_rate = lr;
    descs_.push_back(d);
    mixers_.push_back(m);
    return (int)descs_.size() - 1;
  }
  void Note(int rc) {
    if (rc != GMX_OK && status_ == GMX_OK) {
      status_ = rc;
      fprintf(stderr, "gmx::MixerBank: %s %s\n", gmx_strerror(rc), gmx_last_error());
    }
  }
  bool Export(std::vector<char>* l, std::vector<char>* sh) {
    if (!group_) return false;
    size_t nl = 0, ns = 0;
    int rc = gmx_bank_export(group_, 0, nullptr, &nl, nullptr, &ns);
    if (rc) { Note(rc); return false; }
    l->resize(nl ? nl : 1);
    sh->resize(ns ? ns : 1);
    rc = gmx_bank_export(group_, 0, l->data(), &nl, sh->data(), &ns);
    if (rc) { Note(rc); return false; }
    l->resize(nl);
    sh->resize(ns);
    return true;
  }
  void PredictAll(ShortTermMemory& stm);
  void LearnAll(const ShortTermMemory& stm);

  int device_;
  gmx_group* group_ = nullptr;
  int status_ = GMX_OK;
  std::vector<gmx_mixer_desc> descs_;
  std::vector<Mixer*> mixers_;
  std::vector<float> outputs_;
  std::vector<uint32_t> contexts_;
  std::vector<int32_t> active_;
  // Mixer::WriteToDisk streams 3 x u64 per mixer into the .short file one mixer at a time;
  // the bank fetches them once per checkpoint and hands them out.
  std::vector<char> short_cache_;
  std::vector<char> short_in_;
  float last_p_ = 0.5f;
  int n_inputs_ = 0;
  size_t n_skip_ = 0;
};

class Mixer {
 public:
  // mixer/mixer.h:17-19; `bank` stands where the reference passes long_term_memory.
  Mixer(ShortTermMemory& short_term_memory, MixerBank& bank, unsigned int& context,
        float learning_rate, int layer_number, unsigned int table_size, std::string description,
        bool enable_analysis)
      : context_(context), bank_(bank), layer_number_(layer_number) {
    output_index_ = short_term_memory.AddMixer(description, layer_number, enable_analysis, this);
    memory_index_ = bank.Register(this, layer_number, table_size, learning_rate);
  }
  // Model interface (model.h:22-37).  The first mixer of the bank launches all of them.
  void Predict(ShortTermMemory& short_term_memory, const MixerBank&) {
    if (memory_index_ == 0) bank_.PredictAll(short_term_memory);
  }
  void Learn(const ShortTermMemory& short_term_memory, MixerBank&) {
    if (memory_index_ == 0) bank_.LearnAll(short_term_memory);
  }
  // mixer.cpp:178-188: steps_, max_steps_, contexts_seen_ as 3 x u64
  void WriteToDisk(std::ofstream* s) {
    if (memory_index_ == 0) {
      std::vector<char> l;
      if (!bank_.Export(&l, &bank_.short_cache_)) return;
    }
    if (bank_.short_cache_.size() >= (size_t)(memory_index_ + 1) * 24)
      s->write(bank_.short_cache_.data() + (size_t)memory_index_ * 24, 24);
  }
  void ReadFromDisk(std::ifstream* s) {
    // The three counters of all mixers are restored together with the tables by
    // MixerBank::ReadFromDisk, which the reference order reaches later (.short before .long,
    // predictor.cpp:412-416): keep the bytes until then.
    char b[24];
    s->read(b, 24);
    if (memory_index_ == 0) bank_.short_in_.clear();
    bank_.short_in_.insert(bank_.short_in_.end(), b, b + 24);
  }
  void Copy(const Mixer*) {}  // the counters travel with MixerBank::Copy
  // mixer.cpp:197-205
  unsigned long long GetMemoryUsage(const ShortTermMemory&, const MixerBank&) {
    uint64_t v = 0;
    if (bank_.group_) bank_.Note(gmx_bank_memory_usage(bank_.group_, 0, memory_index_, &v));
    return v;
  }
  int layer_number() const { return layer_number_; }
  int output_index() const { return output_index_; }
  unsigned int context() const { return context_; }

 private:
  unsigned int& context_;  // aliases a field of the caller's blackboard (mixer.h:31)
  MixerBank& bank_;
  int output_index_, memory_index_;
  int layer_number_;
};

inline void MixerBank::PredictAll(ShortTermMemory& stm) {
  if (!group_) {
    if (Finalize(stm) != GMX_OK) return;
  }
  for (size_t j = 0; j < mixers_.size(); ++j) contexts_[j] = mixers_[j]->context();  // read at call time
  active_.assign(stm.active_models.begin(), stm.active_models.end());
  Note(gmx_bank_forward(group_, 0, &stm.predictions[0], active_.data(), (int)active_.size(),
                        contexts_.data(), &last_p_, outputs_.data()));
  // mixer.cpp:99-105: where each Mixer::Predict leaves its result
  size_t j = 0;
  for (int k = 0; k < stm.num_layer0_mixers; ++k) stm.mixer_layer0_outputs[k] = outputs_[j++];
  for (int k = 0; k < stm.num_layer1_mixers; ++k) stm.mixer_layer1_outputs[k] = outputs_[j++];
  if (j < outputs_.size()) stm.final_mixer_output = outputs_[j];
}

inline void MixerBank::LearnAll(const ShortTermMemory& stm) {
  if (!group_) return;
  Note(gmx_bank_learn(group_, 0, stm.new_bit));
}

inline void MixerBank::ReadFromDisk(std::ifstream* s) {
  if (!group_) return;
  // Parse the section to find its length (long-term-memory.cpp:134-149), then import it with
  // the 3 x u64 per mixer that Mixer::ReadFromDisk collected from the .short file.
  std::vector<char> buf;
  int k0 = 0, k1 = 0, l0 = 0, l1 = 0;
  for (const gmx_mixer_desc& d : descs_) (d.layer == 0 ? l0 : l1) += d.layer < 2;
  for (size_t j = 0; j < descs_.size(); ++j) {
    // a header is trusted only as far as the topology allows: at most table_size rows, each of exactly
    // this mixer's weight_size (mixer.cpp:17-26) -- a truncated or corrupt file is refused, not allocated for
    uint32_t hdr[2] = {0, 0};
    s->read(reinterpret_cast<char*>(hdr), 8);
    const gmx_mixer_desc& d = descs_[j];
    const int n_skip = (int)n_skip_;
    const uint32_t wsize = d.layer == 0 ? (uint32_t)(n_inputs_ + k0++)
                           : d.layer == 1 ? (uint32_t)(l0 + k1++ + n_skip) : (uint32_t)(l0 + l1 + n_skip);
    if (!s->good() || hdr[0] > d.table_size || (hdr[0] > 0 && hdr[1] != wsize)) {
      Note(GMX_ERR_FORMAT);
      return;
    }
    buf.insert(buf.end(), reinterpret_cast<char*>(hdr), reinterpret_cast<char*>(hdr) + 8);
    size_t body = (size_t)hdr[0] * (12 + 4 * (size_t)hdr[1]);
    size_t at = buf.size();
    buf.resize(at + body);
    s->read(buf.data() + at, body);
    if ((size_t)s->gcount() != body) {
      Note(GMX_ERR_FORMAT);
      return;
    }
  }
  Note(gmx_bank_import(group_, 0, buf.data(), buf.size(), short_in_.data(), short_in_.size()));
}

// The mixer slice of Predictor (predictor.h:20-38): the protocol Predict -> Perceive ->
// (optional) Learn, with the feature models' work -- filling `stm.predictions`,
// `stm.active_models` and the context variables -- left to the caller.
class MixerPredictor {
 public:
  ShortTermMemory stm;
  MixerBank bank;
  std::vector<std::unique_ptr<Mixer>> mixers;

  explicit MixerPredictor(int device = 0) : bank(device) {}
  // One AddModel(new Mixer(...)) of Predictor::AddMixers (predictor.cpp:254-357).
  void AddMixer(unsigned int& context, float lr, int layer, unsigned int table_size,
                const std::string& description) {
    mixers.emplace_back(new Mixer(stm, bank, context, lr, layer, table_size, description, false));
  }
  // The tail of Predictor::Predictor (predictor.cpp:29-36) + device allocation.
  int Finalize() {
    stm.predictions.resize(stm.num_predictions);
    stm.predictions = 0;
    stm.mixer_layer0_outputs.resize(stm.num_layer0_mixers);
    stm.mixer_layer0_outputs = 0;
    stm.mixer_layer1_outputs.resize(stm.num_layer1_mixers);
    stm.mixer_layer1_outputs = 0;
    return bank.Finalize(stm);
  }
  // Feature models call stm.SetPrediction / SetLogitPrediction between BeginBit and Predict.
  void BeginBit() { stm.active_models.clear(); }  // predictor.cpp:361
  float Predict() {                               // predictor.cpp:366-375
    for (auto& m : mixers) m->Predict(stm, bank);
    // Logistic(final_mixer_output) clamped to [1e-4, 1-1e-4] is part of the path and was
    // computed on the device with the outputs.
    return bank.last_probability();
  }
  void Perceive(int bit) { stm.new_bit = bit; }   // predictor.cpp:378-381
  void Learn() {                                  // predictor.cpp:383-387
    for (auto& m : mixers) m->Learn(stm, bank);
  }
  // predictor.cpp:389-420 for the mixer slice: `.short` = every Mixer::WriteToDisk in order,
  // `.long` = the mixer section.
  void WriteCheckpoint(const std::string& path) {
    std::ofstream s(path + ".short", std::ios::out | std::ios::binary);
    if (!s.is_open()) return;
    std::ofstream l(path + ".long", std::ios::out | std::ios::binary);
    if (!l.is_open()) return;
    for (auto& m : mixers) m->WriteToDisk(&s);
    bank.WriteToDisk(&l);
  }
  void ReadCheckpoint(const std::string& path) {
    std::ifstream s(path + ".short", std::ios::in | std::ios::binary);
    if (!s.is_open()) return;
    std::ifstream l(path + ".long", std::ios::in | std::ios::binary);
    if (!l.is_open()) return;
    for (auto& m : mixers) m->ReadFromDisk(&s);
    bank.ReadFromDisk(&l);
  }
  void Copy(const MixerPredictor& p) { bank.Copy(&p.bank); }  // predictor.cpp:42-48
};

}  // namespace gmx

#endif  // GMX_MIXER_H_
