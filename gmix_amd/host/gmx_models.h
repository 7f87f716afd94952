// gmx_models.h -- host-side C++ mirrors of the two feature models that live on the device next
// to the mixers: the reference's `Indirect` (models/indirect.{h,cpp}) and `LstmModel`
// (models/lstm-model.{h,cpp}), over the C ABI of libgmxmix.so (include/gmxmix.h).  Header-only,
// C++17, no HIP types.  Same pattern as gmx_mixer.h: the objects keep their place in
// Predictor::models_ and their Model signatures (model.h:22-37); what LongTermMemory owned for
// them is owned by a bank object on the GPU.
//
//   gmx::IndirectBank  stands where LongTermMemory::indirect stood (long-term-memory.h:11-25): all
//                      Indirect objects of one Predictor = one gmx_indirect with one stream.  The
//                      FIRST model's Predict runs all of them on the device; every model then puts
//                      its own two results on the blackboard in its own Predict call, so
//                      active_models keeps the reference's order.  Learn likewise.
//   gmx::Indirect      `Indirect(stm, bank, learning_rate, table_size, unsigned& context,
//                      description, enable_analysis)` (indirect.h:16-18).
//   gmx::LstmBank      stands where LongTermMemory::neuron_layer_weights / lstm_output_layer stood
//                      (long-term-memory.h:55-76); its constructor draws the initial weights from
//                      rand() exactly like LstmLayer's (lstm-layer.cpp:179-194), so a Predictor that
//                      calls srand(0xDEADBEEF) first (predictor.cpp:18) starts from the same model.
//   gmx::LstmModel     `LstmModel(stm, bank, enable_analysis)` (lstm-model.h:11-12): byte-level
//                      Lstm::Predict / Perceive on the device, the bit-level range coding of the byte
//                      distribution (lstm-model.cpp:34-48) on the host as in the reference.
#ifndef GMX_MODELS_H_
#define GMX_MODELS_H_

#include <cmath>
#include <cstdlib>
#include <numeric>

#include "gmx_mixer.h"

namespace gmx {

class Indirect;

class IndirectBank {
 public:
  explicit IndirectBank(int device = 0) : device_(device) {}
  ~IndirectBank() {
    if (h_) gmx_indirect_destroy(h_);
  }
  IndirectBank(const IndirectBank&) = delete;
  IndirectBank& operator=(const IndirectBank&) = delete;

  // Once, after the last Indirect has been constructed.  ns_next(state, bit) / rm_next(state, bit):
  // the Next() of the two state machines ShortTermMemory owns (short-term-memory.h:144-145;
  // called by Indirect::Learn, indirect.cpp:61-62, :67-68) -- tabulated here, walked on the device.
  template <class NsNext, class RmNext>
  int Finalize(NsNext ns_next, RmNext rm_next) {
    if (h_) return GMX_OK;
    uint8_t ns[512], rm[512];
    for (int s = 0; s < 256; ++s)
      for (int b = 0; b < 2; ++b) {
        ns[2 * s + b] = (uint8_t)ns_next(s, b);
        rm[2 * s + b] = (uint8_t)rm_next(s, b);
      }
    status_ = gmx_indirect_create(&h_, descs_.data(), (int)descs_.size(), ns, rm, 1, device_);
    contexts_.assign(descs_.size(), 0u);
    pred_.assign(2 * descs_.size(), 0.f);
    active_.assign(2 * descs_.size(), 0);
    return status_;
  }
  bool ready() const { return h_ != nullptr; }
  int status() const { return status_; }
  gmx_indirect* handle() { return h_; }
  size_t size() const { return descs_.size(); }

  // The indirect section of LongTermMemory::WriteToDisk / ReadFromDisk (long-term-memory.cpp:8-32,
  // :111-132), same bytes.
  void WriteToDisk(std::ofstream* s) {
    if (!h_) return;
    size_t n = 0;
    if (Note(gmx_indirect_export(h_, 0, nullptr, &n))) return;
    std::vector<char> buf(n ? n : 1);
    if (Note(gmx_indirect_export(h_, 0, buf.data(), &n))) return;
    s->write(buf.data(), n);
  }
  void ReadFromDisk(std::ifstream* s) {
    if (!h_) return;
    // per model: u32 count, then count x {u32 key, u8, u8} if sparse (count < size / 3) else the
    // two whole tables, then 2 x 256 floats
    std::vector<char> buf;
    auto take = [&](size_t n) {
      size_t at = buf.size();
      buf.resize(at + n);
      s->read(buf.data() + at, n);
    };
    for (const gmx_indirect_desc& d : descs_) {
      const uint64_t size = (uint64_t)d.table_size * 256u + 1u;
      take(4);
      uint32_t count;
      memcpy(&count, buf.data() + buf.size() - 4, 4);
      take(count < size / 3 ? (size_t)count * 6 : (size_t)size * 2);
      take(2 * 256 * 4);
    }
    Note(gmx_indirect_import(h_, 0, buf.data(), buf.size()));
  }
  void Copy(const IndirectBank* orig) {  // long-term-memory.cpp:193-199
    if (!h_ || !orig->h_) return;
    Note(gmx_indirect_copy(h_, 0, orig->h_, 0));
  }

 private:
  friend class Indirect;
  int Register(Indirect* m, unsigned table_size, float lr, int slot_indirect, int slot_run_map) {
    gmx_indirect_desc d;
    d.table_size = table_size;
    d.learning_rate = lr;
    d.slot_indirect = slot_indirect;
    d.slot_run_map = slot_run_map;
    descs_.push_back(d);
    models_.push_back(m);
    return (int)descs_.size() - 1;
  }
  bool Note(int rc) {
    if (rc != GMX_OK && status_ == GMX_OK) {
      status_ = rc;
      fprintf(stderr, "gmx::IndirectBank: %s %s\n", gmx_strerror(rc), gmx_last_error());
    }
    return rc != GMX_OK;
  }
  void PredictAll(const ShortTermMemory& stm);
  void LearnAll(const ShortTermMemory& stm) {
    if (h_) Note(gmx_indirect_learn(h_, 0, stm.new_bit));
  }

  int device_;
  gmx_indirect* h_ = nullptr;
  int status_ = GMX_OK;
  std::vector<gmx_indirect_desc> descs_;
  std::vector<Indirect*> models_;
  std::vector<uint32_t> contexts_;
  std::vector<float> pred_;
  std::vector<uint8_t> active_;
};

class Indirect {
 public:
  // indirect.h:16-18; `bank` stands where the reference passes long_term_memory.
  Indirect(ShortTermMemory& short_term_memory, IndirectBank& bank, float learning_rate,
           unsigned int table_size, unsigned int& context, std::string description, bool enable_analysis)
      : context_(context), bank_(bank) {
    prediction_index_indirect_ = short_term_memory.AddPrediction(description + "-indirect", enable_analysis, this);
    prediction_index_run_map_ = short_term_memory.AddPrediction(description + "-run_map", enable_analysis, this);
    memory_index_ = bank.Register(this, table_size, learning_rate, prediction_index_indirect_,
                                  prediction_index_run_map_);
  }
  // indirect.cpp:28-46.  The bank's answer for slot k is what the slot holds after this model's
  // Predict (unchanged if the model stayed silent) and whether SetLogitPrediction marked it active
  // (short-term-memory.cpp:193-197).
  void Predict(ShortTermMemory& short_term_memory, const IndirectBank&) {
    if (memory_index_ == 0) bank_.PredictAll(short_term_memory);
    if (!bank_.h_) return;
    const int slots[2] = {prediction_index_indirect_, prediction_index_run_map_};
    for (int k = 0; k < 2; ++k) {
      short_term_memory.predictions[slots[k]] = bank_.pred_[2 * memory_index_ + k];
      if (bank_.active_[2 * memory_index_ + k]) short_term_memory.active_models.push_back(slots[k]);
    }
  }
  void Learn(const ShortTermMemory& short_term_memory, IndirectBank&) {  // indirect.cpp:48-69
    if (memory_index_ == 0) bank_.LearnAll(short_term_memory);
  }
  void WriteToDisk(std::ofstream*) {}  // indirect.h:23-25: the state is all in the long-term section
  void ReadFromDisk(std::ifstream*) {}
  void Copy(const Indirect*) {}
  unsigned long long GetMemoryUsage(const ShortTermMemory&, const IndirectBank&) {  // indirect.cpp:71-78
    uint64_t v = 0;
    if (bank_.h_) bank_.Note(gmx_indirect_memory_usage(bank_.h_, memory_index_, &v));
    return v;
  }
  unsigned int context() const { return context_; }

 private:
  unsigned int& context_;  // aliases a field of the caller's blackboard (indirect.h:31)
  IndirectBank& bank_;
  int prediction_index_indirect_, prediction_index_run_map_, memory_index_;
};

inline void IndirectBank::PredictAll(const ShortTermMemory& stm) {
  if (!h_) return;
  for (size_t i = 0; i < models_.size(); ++i) contexts_[i] = models_[i]->context();  // read at call time
  Note(gmx_indirect_forward(h_, 0, contexts_.data(), stm.bit_context, pred_.data(), active_.data()));
}

// ---- LSTM ------------------------------------------------------------------------------------

class LstmBank {
 public:
  // The constructor chain of Lstm(256, 256, 50, 1, 100, 0.03, 10) (lstm-model.cpp:7) as far as
  // it touches LongTermMemory: three 50 x 563 gate matrices drawn from rand() in the
  // interleaved order of lstm-layer.cpp:179-194, the forget gate's bias column set to 1.
  explicit LstmBank(int device = 0) {
    status_ = gmx_lstm_create(&h_, 1, device);
    if (status_ != GMX_OK) return;
    const int kCells = 50, kInputs = 563;
    std::vector<float> w((size_t)3 * kCells * kInputs);
    const float val = std::sqrt(6.0f / float(256 + 256));
    const float low = -val, range = 2 * val;
    for (int i = 0; i < kCells; ++i) {
      for (int j = 0; j < kInputs; ++j)
        for (int g = 0; g < 3; ++g)
          w[((size_t)g * kCells + i) * kInputs + j] = low + (static_cast<float>(rand()) / static_cast<float>(RAND_MAX)) * range;
      w[((size_t)0 * kCells + i) * kInputs + kInputs - 1] = 1;
    }
    Note(gmx_lstm_set_weights(h_, 0, w.data()));
  }
  ~LstmBank() {
    if (h_) gmx_lstm_destroy(h_);
  }
  LstmBank(const LstmBank&) = delete;
  LstmBank& operator=(const LstmBank&) = delete;
  int status() const { return status_; }
  gmx_lstm* handle() { return h_; }

  // The LSTM section of LongTermMemory::WriteToDisk / ReadFromDisk (long-term-memory.cpp:57-67,
  // :151-160).  Reading completes the import LstmModel::ReadFromDisk began (.short is read before
  // .long, predictor.cpp:412-416).
  void WriteToDisk(std::ofstream* s) {
    std::vector<char> l, sh;
    if (Export(&l, &sh)) s->write(l.data(), l.size());
  }
  void ReadFromDisk(std::ifstream* s) {
    if (!h_) return;
    size_t nl = 0, ns = 0;
    if (Note(gmx_lstm_export(h_, 0, nullptr, &nl, nullptr, &ns))) return;
    std::vector<char> l(nl);
    s->read(l.data(), nl);
    Note(gmx_lstm_import(h_, 0, l.data(), nl, short_in_.data(), short_in_.size()));
  }
  void Copy(const LstmBank* orig) {  // long-term-memory.cpp:216-219 + LstmModel::Copy
    if (h_ && orig->h_) Note(gmx_lstm_copy(h_, 0, orig->h_, 0));
  }

 private:
  friend class LstmModel;
  bool Note(int rc) {
    if (rc != GMX_OK && status_ == GMX_OK) {
      status_ = rc;
      fprintf(stderr, "gmx::LstmBank: %s %s\n", gmx_strerror(rc), gmx_last_error());
    }
    return rc != GMX_OK;
  }
  bool Export(std::vector<char>* l, std::vector<char>* sh) {
    if (!h_) return false;
    size_t nl = 0, ns = 0;
    if (Note(gmx_lstm_export(h_, 0, nullptr, &nl, nullptr, &ns))) return false;
    l->resize(nl);
    sh->resize(ns);
    return !Note(gmx_lstm_export(h_, 0, l->data(), &nl, sh->data(), &ns));
  }
  gmx_lstm* h_ = nullptr;
  int status_ = GMX_OK;
  std::vector<char> short_in_;
};

class LstmModel {
 public:
  // lstm-model.cpp:5-15; `bank` stands where the reference passes long_term_memory.
  LstmModel(ShortTermMemory& short_term_memory, LstmBank& bank, bool enable_analysis)
      : bank_(bank), top_(255), mid_(127), bot_(0), probs_(1.0f / 256, 256) {
    prediction_index_ = short_term_memory.AddPrediction("LSTM", enable_analysis, this);
    short_term_memory.models_with_skip_connection.push_back(prediction_index_);
  }
  // lstm-model.cpp:17-49.  At a byte boundary the device runs Lstm::SetInput + Lstm::Predict and
  // returns the byte distribution and its arg max (lstm_prediction_context); between boundaries
  // the interval [bot_, top_] is halved by the coded bit.  The bit prediction is the share of the
  // interval's upper half, both sums started and continued exactly as std::accumulate does there.
  void Predict(ShortTermMemory& short_term_memory, const LstmBank&) {
    if (short_term_memory.recent_bits == 1) {
      uint32_t ctx = 0;
      if (bank_.h_)
        bank_.Note(gmx_lstm_forward(bank_.h_, 0, (int)short_term_memory.last_byte,
                                    &short_term_memory.ppm_predictions[0], &probs_[0], &ctx));
      short_term_memory.lstm_prediction_context = ctx;
      top_ = 255;
      bot_ = 0;
    } else if (short_term_memory.new_bit) {
      bot_ = mid_ + 1;
    } else {
      top_ = mid_;
    }
    mid_ = bot_ + ((top_ - bot_) / 2);
    const float num = std::accumulate(&probs_[mid_ + 1], &probs_[top_ + 1], 0.0f);
    const float denom = std::accumulate(&probs_[bot_], &probs_[mid_ + 1], num);
    if (denom != 0) short_term_memory.SetPrediction(num / denom, prediction_index_);
  }
  // lstm-model.cpp:51-60: the last bit of a byte hands the byte to Lstm::Perceive
  void Learn(const ShortTermMemory& short_term_memory, LstmBank&) {
    const int current_byte = short_term_memory.recent_bits * 2 + short_term_memory.new_bit;
    if (current_byte >= 256 && bank_.h_) bank_.Note(gmx_lstm_perceive(bank_.h_, 0, current_byte - 256));
  }
  // lstm-model.cpp:62-76 (at a byte boundary: top_/mid_/bot_/probs_ are part of the device's bytes)
  void WriteToDisk(std::ofstream* s) {
    std::vector<char> l, sh;
    if (bank_.Export(&l, &sh)) s->write(sh.data(), sh.size());
  }
  void ReadFromDisk(std::ifstream* s) {
    size_t nl = 0, ns = 0;
    if (!bank_.h_ || bank_.Note(gmx_lstm_export(bank_.h_, 0, nullptr, &nl, nullptr, &ns))) return;
    bank_.short_in_.resize(ns);
    s->read(bank_.short_in_.data(), ns);
    if (ns >= 12 + 1024) {
      memcpy(&top_, bank_.short_in_.data(), 4);
      memcpy(&mid_, bank_.short_in_.data() + 4, 4);
      memcpy(&bot_, bank_.short_in_.data() + 8, 4);
      memcpy(&probs_[0], bank_.short_in_.data() + 12, 1024);
    }
  }
  void Copy(const LstmModel* orig) {  // lstm-model.cpp:78-85 (the device part: LstmBank::Copy)
    top_ = orig->top_;
    mid_ = orig->mid_;
    bot_ = orig->bot_;
    probs_ = orig->probs_;
  }
  unsigned long long GetMemoryUsage(const ShortTermMemory&, const LstmBank&) {  // lstm-model.cpp:87-101
    uint64_t v = 0;
    if (bank_.h_) bank_.Note(gmx_lstm_memory_usage(bank_.h_, &v));
    return v;
  }
  const std::valarray<float>& probs() const { return probs_; }

 private:
  LstmBank& bank_;
  int top_, mid_, bot_, prediction_index_;
  std::valarray<float> probs_;
};

}  // namespace gmx

#endif  // GMX_MODELS_H_
