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
// are (dropin/Makefile builds gmix and the reference's tester that way, from a patched
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
#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <memory>
#include <string>
#include <vector>

#include "model.h"  // the reference's (src/model.h): Model, ShortTermMemory, LongTermMemory

#include "gmxmix.h"  // include/gmxmix.h

namespace gmx {

class GpuMixer;
class GpuMixerBank;
class GpuIndirect;
class GpuIndirectBank;
class GpuLstmModel;

// Process-wide tables of the adapters.  The reference lets several Predictors live in one process
// (runner-utils.cpp:291-292); here they may also be built and run on different threads (one file per thread,
// MixerPool below), so every access goes through AdapterMutex().
inline std::recursive_mutex& AdapterMutex() {
  static std::recursive_mutex m;
  return m;
}
inline bool ChainFused() {
  static const bool on = [] {
    const char* e = getenv("GMX_CHAIN_FUSED");
    return !(e && e[0] == '0');
  }();
  return on;
}

// ------------------------------------------------------------------------------------------------
// Run-ahead.  In compression and training the coded bits are known in advance and no feature model ever reads a
// mixer output (predictor.cpp:360-387 only hands them to later mixers, the final squash and the analysis), so
// the host-side feature models of a Predictor can run T bits ahead while the device works on the T bits before:
// the device-side models' Predict then only RECORDS its inputs and Learn the coded bit, straight into the pinned
// arrays of the banks' batches -- the mixers {predictions, active_models, 33 contexts, bit}, the Indirect models
// {41 contexts, bit_context, bit}, the LSTM {PPM byte distribution, byte} -- a full chunk is uploaded and run
// behind the chunk before it (a ring of four sets of batches, MixerPool::kRing) and its
// probabilities come back one chunk later, in order, to whoever consumes them (the arithmetic coder:
// gmx_batched.h).
// ------------------------------------------------------------------------------------------------
struct RunAheadView {
  // n bits of one stream, oldest first
  uint64_t n = 0;
  const float* p = nullptr;         // [n] what Predictor::Predict() would have returned
  const uint8_t* bits = nullptr;    // [n] the bit that was then perceived
  const float* outputs = nullptr;   // [n][n_mixers] every mixer's output (logit domain) -- only when the sink asked for
                                    // them (RunAheadSink::WantsAllOutputs); else null, and
  const float* last_outputs = nullptr;  // [n_mixers] the outputs of the newest of the n bits (what the blackboard holds)
  int n_mixers = 0;
  // with the Indirect models / the LSTM on the device too, and only when asked for (RunAheadSink::WantsModels):
  const float* ind_pred = nullptr;     // [n][2 * n_ind] what the models' blackboard slots held ([2i] indirect, [2i+1] run map)
  const uint8_t* ind_active = nullptr; // [n][2 * n_ind] whether SetLogitPrediction marked them active
  int n_ind = 0;
  const float* lstm_pred = nullptr;    // [n / 8][8] the LSTM's slot, bit by bit
  const uint8_t* lstm_active = nullptr;
  const uint32_t* lstm_context = nullptr;  // [n / 8] ShortTermMemory::lstm_prediction_context of every byte (whenever the
                                           // LSTM is on the device: the blackboard keeps the newest)
};
struct RunAheadSink {
  virtual ~RunAheadSink() {}
  virtual void Drain(const RunAheadView& v) = 0;
  virtual bool WantsModels() const { return false; }  // also bring the device-side feature models' predictions back
  virtual bool WantsAllOutputs() const { return false; }  // every mixer's output of every bit (else: the newest bit's)
  // Predictor::Predict clears the blackboard's predictions before the models run when analysis is on
  // (predictor.cpp:362-365): a device-side model that stays silent then leaves 0 in its slot, not its last value
  virtual bool SilentSlotsAreZero() const { return false; }
  virtual bool WantsMemoryUsage() const { return false; }  // Mixer::GetMemoryUsage will be asked while running ahead: the
                                                           // bank then counts the rows it has seen on the host, bit by bit
};

// The device banks of up to n_streams Predictors -- ONE gmx_group, and one gmx_indirect / gmx_lstm where the
// Predictors' Indirect models / LSTM are on the device as well -- one stream per Predictor, and the rings their
// run-ahead chunks travel through.  Banks constructed while a pool is installed take a stream of it; otherwise
// every Predictor owns a pool of one stream, which is the plain drop-in.  The Predictors may live on different
// threads (one file per thread): every C-ABI call on the shared objects happens under the pool's mutex, the
// chunk of a round is submitted by whichever stream arrives last, and a stream only ever touches its own
// stretch of the batches' arrays.
class MixerPool {
 public:
  explicit MixerPool(int n_streams, int device = -1) : S_(n_streams < 1 ? 1 : n_streams), device_(device), streams_(S_) {
    if (device_ < 0) {
      const char* dev = getenv("GMX_DEVICE");
      device_ = dev ? atoi(dev) : 0;
    }
  }
  ~MixerPool() {
    Uninstall();
    if (trace_)  // GMX_POOL_TRACE=1: where the submitting thread's time went, call by call
      for (auto& kv : step_seconds_) fprintf(stderr, "[gmx pool] %8.3f s  %s\n", kv.second, kv.first.c_str());
    for (int k = 0; k < kRing; ++k) {
      if (ring_[k]) gmx_batch_destroy(ring_[k]);
      if (iring_[k]) gmx_ind_batch_destroy(iring_[k]);
      if (lring_[k]) gmx_lstm_batch_destroy(lring_[k]);
    }
    if (cs_) gmx_chainstep_destroy(cs_);
    if (group_) gmx_group_destroy(group_);
    if (ind_) gmx_indirect_destroy(ind_);
    if (lstm_) gmx_lstm_destroy(lstm_);
  }
  MixerPool(const MixerPool&) = delete;
  MixerPool& operator=(const MixerPool&) = delete;

  static MixerPool*& Installed() {  // guarded by AdapterMutex()
    static MixerPool* p = nullptr;
    return p;
  }
  // A thread may name the pool ITS Predictors go to (InstallForThisThread; nullptr: the process-wide one again): several
  // pools side by side, each with worker threads of its own (gmx::BatchedDecompressFiles' groups).
  static MixerPool*& ThreadPool() {
    static thread_local MixerPool* p = nullptr;
    return p;
  }
  static MixerPool* Current() {  // AdapterMutex() held
    MixerPool* t = ThreadPool();
    return t ? t : Installed();
  }
  void InstallForThisThread() {
    {
      std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
      shared_ = true;
      DrawLstmInit();
    }
    ThreadPool() = this;
  }
  static void UninstallForThisThread() { ThreadPool() = nullptr; }
  // From now on the banks of every Predictor constructed (on any thread) live in this pool.  The pool must
  // outlive them.
  void Install() {
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    Installed() = this;
    shared_ = true;
    DrawLstmInit();
  }
  void Uninstall() {
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    if (Installed() == this) Installed() = nullptr;
  }
  // The LSTM's initial gate weights are a constant of the reference: every Predictor constructor begins with
  // srand(0xDEADBEEF) (predictor.cpp:18) and the LSTM is the first model that draws (lstm-layer.h:41; BasicContexts,
  // the interval contexts and PPMd, built before it, draw nothing).  A pool that hosts many Predictors draws the
  // 3 x 50 x 563 numbers ONCE, here, from the same rand() while nothing else runs, and gmx::GpuLstmModel's constructor
  // takes them from the pool instead of the process-wide generator -- so that Predictors whose LSTM is on the device
  // can be built side by side on many threads (the reference's constructor, 65 ms of host work each, was 4 of the
  // 5 seconds a 64-file run of 30 KB files took: VERDICT r3 #1).  The draw is checked against its known FNV-1a
  // (a stray rand() of another thread in between would shift it): on a mismatch the pool keeps no cache and
  // constructions stay serial, drawing for themselves as before.
  static constexpr int kLstmDraws = 3 * 50 * 563;
  void DrawLstmInit() {
    if (!lstm_init_.empty() || lstm_init_tried_) return;
    lstm_init_tried_ = true;
    std::vector<int> draws(kLstmDraws);
    for (int attempt = 0; attempt < 3; ++attempt) {
      srand(0xDEADBEEF);
      uint64_t h = 1469598103934665603ull;
      for (int i = 0; i < kLstmDraws; ++i) {
        draws[i] = rand();
        h = (h ^ (uint64_t)(uint32_t)draws[i]) * 1099511628211ull;
      }
      if (h == kLstmDrawsFnv) {
        lstm_init_.swap(draws);
        return;
      }
    }
    fprintf(stderr, "gmx::MixerPool: rand() after srand(0xDEADBEEF) is not the sequence this build was made with; "
                    "Predictors will be constructed one at a time\n");
  }
  // 0, or the kLstmDraws values rand() gives after srand(0xDEADBEEF), in drawing order
  const int* lstm_init() const { return lstm_init_.empty() ? nullptr : lstm_init_.data(); }
  // A Predictor of this pool has its LSTM on the device and took the initial weights from the pool: its constructor
  // drew nothing from rand(), so constructors may run side by side
  bool parallel_construction() const { return lstm_from_cache_.load(); }
  int n_streams() const { return S_; }
  int device() const { return device_; }
  gmx_group* group() const { return group_; }
  gmx_indirect* indirect() const { return ind_; }
  gmx_lstm* lstm() const { return lstm_; }
  // 0, or the status of the first C-ABI call that failed on this pool (shared pools report instead of aborting)
  int status() const { return status_.load(); }
  std::string error() const {
    std::lock_guard<std::mutex> lk(mu_);
    return error_;
  }
  uint64_t chunk_bits() const { return T_; }
  uint64_t bits_submitted() const { return bits_submitted_; }  // bits of all streams queued on the device so far
  uint64_t rounds() const { return round_; }                   // chunks (of all streams together) handed in
  // host time the submitting stream spent queueing chunks / waiting for the chunk before (the device's share of
  // the wall time when it, not the hosts' feature models, sets the pace)
  double submit_seconds() const { return submit_seconds_; }
  double wait_seconds() const { return wait_seconds_; }

  enum Parts { kMixers = 1, kIndirect = 2, kLstm = 4 };

  // ---- lock step (decoding).  A decoder learns each bit from Predict's own result (coder/decoder.cpp:19-39), so
  // the Predictors of a pool that decode advance together, one device step (gmx_chainstep) per coded bit: every
  // stream's Predict records its inputs and WAITS (`yield`: the caller's scheduler runs the other streams meanwhile --
  // gmx::LockstepRunner puts each decoder on a fibre); when every stream has got there the scheduler calls StepAll and
  // resumes them with their probabilities.  Without a scheduler (one Predictor, one thread) a stream's Predict runs
  // the step itself.
  void SetLockstepYield(std::function<void(int)> yield) { yield_ = std::move(yield); }
  // One step for every stream that asked for one since the last; mu_ NOT held; only ever one thread at a time.
  int StepAll() {
    if (!cs_ || status_.load()) return status_.load();
    int rc;
    {
      std::lock_guard<std::mutex> lk(mu_);
      rc = gmx_chainstep_step(cs_);
      if (rc) Fail("gmx_chainstep_step", rc);
    }
    if (rc == GMX_OK) {
      memset(ls_what_, 0, (size_t)S_);
      ++round_;
    }
    return rc;
  }
  // The same in two halves (a scheduler with two pools: one's host work beside the other's device step).
  int StepLaunch() {
    if (!cs_ || status_.load()) return status_.load();
    std::lock_guard<std::mutex> lk(mu_);
    const int rc = gmx_chainstep_launch(cs_);
    if (rc) {
      Fail("gmx_chainstep_launch", rc);
      return rc;
    }
    memset(ls_what_, 0, (size_t)S_);
    ++round_;
    return GMX_OK;
  }
  int StepWait() {
    if (!cs_ || status_.load()) return status_.load();
    std::lock_guard<std::mutex> lk(mu_);
    const int rc = gmx_chainstep_wait(cs_);
    if (rc) Fail("gmx_chainstep_wait", rc);
    return rc;
  }
  int lockstep_streams() const { return ls_participants_; }

 private:
  friend class GpuMixerBank;
  friend class GpuIndirectBank;
  friend class GpuLstmModel;

  struct Stream {  // one Predictor's place in the pool
    const LongTermMemory* owner = nullptr;
    int refs = 0;
    GpuMixerBank* mixers = nullptr;
    GpuIndirectBank* indirect = nullptr;
    GpuLstmModel* lstm = nullptr;
    // run-ahead: where the chunk being filled is recorded, and how many bits of it are
    bool ra = false;
    uint64_t t = 0;
    float* pred = nullptr;
    uint32_t *mask = nullptr, *ctx = nullptr;
    uint8_t* bits = nullptr;
    uint32_t *ictx = nullptr, *ibc = nullptr;
    uint8_t* ibits = nullptr;
    float* ppm = nullptr;
    uint8_t* bytes = nullptr;
    bool ls = false;  // lock step: Predict records into the chainstep's arrays and waits for the step
  };

  // The pool and stream of the Predictor that owns `ltm`: the installed pool, else a pool of its own (shared by
  // the mixers', the Indirect models' and the LSTM's adapters of that Predictor).
  static std::shared_ptr<MixerPool> Attach(const LongTermMemory* ltm, int* slot) {
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    std::shared_ptr<MixerPool> sp;
    auto& priv = Private();
    auto it = priv.find(ltm);
    if (it != priv.end()) sp = it->second.lock();
    if (!sp) {
      if (MixerPool* inst = Current()) {
        sp = std::shared_ptr<MixerPool>(inst, [](MixerPool*) {});
      } else {
        sp.reset(new MixerPool(1));
      }
      priv[ltm] = sp;
    }
    std::lock_guard<std::mutex> lk2(sp->mu_);
    int free_slot = -1;
    for (int i = 0; i < sp->S_; ++i) {
      if (sp->streams_[i].owner == ltm) {
        ++sp->streams_[i].refs;
        *slot = i;
        return sp;
      }
      if (!sp->streams_[i].owner && free_slot < 0) free_slot = i;
    }
    if (free_slot < 0) {
      fprintf(stderr, "\ngmx: the installed MixerPool has no free stream (%d in use)\n", sp->S_);
      abort();
    }
    sp->streams_[free_slot] = Stream();
    sp->streams_[free_slot].owner = ltm;
    sp->streams_[free_slot].refs = 1;
    *slot = free_slot;
    return sp;
  }
  void Detach(int slot) {
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    std::lock_guard<std::mutex> lk2(mu_);
    Stream& st = streams_[slot];
    if (--st.refs <= 0) {
      Private().erase(st.owner);
      st = Stream();
    }
  }
  static std::map<const LongTermMemory*, std::weak_ptr<MixerPool>>& Private() {
    static std::map<const LongTermMemory*, std::weak_ptr<MixerPool>> r;
    return r;
  }
  int Fail(const char* what, int rc) {  // mu_ held
    if (status_.load() == 0) {
      error_ = std::string(what) + ": " + gmx_strerror(rc) + (rc == GMX_ERR_HIP ? std::string(" ") + gmx_last_error() : "");
      status_.store(rc);
    }
    cv_.notify_all();
    return rc;
  }
  // A failed call: a pool that hosts many compressors reports (status(), error()); a Predictor's own pool
  // aborts, the convention of the reference's tester (tester.cpp:318-321).  mu_ NOT held.
  void Check(const char* who, const char* what, int rc) {
    if (rc == GMX_OK) return;
    if (shared_) {
      std::lock_guard<std::mutex> lk(mu_);
      Fail(what, rc);
      return;
    }
    fprintf(stderr, "\n%s: %s: %s %s\n(this model runs on an MI355X through libgmxmix.so; there is no CPU fallback)\n", who,
            what, gmx_strerror(rc), rc == GMX_ERR_HIP ? gmx_last_error() : "");
    abort();
  }
  // a C-ABI call on the pool's (possibly shared) objects
  template <class F>
  void Call(const char* who, const char* what, F f) {
    if (status_.load()) return;
    int rc;
    {
      std::lock_guard<std::mutex> lk(mu_);
      rc = f();
    }
    Check(who, what, rc);
  }
  // The objects are made from the first Predictor's description (all constructors of it have run by then);
  // the other Predictors must describe the same ones.  mu_ held.
  int EnsureGroup(const gmx_topology& t) {
    if (group_) {
      bool same = t.n_inputs == n_inputs_ && (size_t)t.n_mixers == descs_.size() && (size_t)t.n_skip == skip_.size();
      for (int j = 0; same && j < t.n_mixers; ++j)
        same = t.mixers[j].layer == descs_[j].layer && t.mixers[j].table_size == descs_[j].table_size &&
               t.mixers[j].learning_rate == descs_[j].learning_rate;
      for (int i = 0; same && i < t.n_skip; ++i) same = t.skip_index[i] == skip_[i];
      return same ? GMX_OK : GMX_ERR_INVALID;
    }
    n_inputs_ = t.n_inputs;
    descs_.assign(t.mixers, t.mixers + t.n_mixers);
    skip_.assign(t.skip_index, t.skip_index + t.n_skip);
    return gmx_group_create(&group_, &t, S_, device_);
  }
  int EnsureIndirect(const std::vector<gmx_indirect_desc>& d, const uint8_t* ns, const uint8_t* rm) {
    if (ind_) {
      bool same = d.size() == idescs_.size();
      for (size_t i = 0; same && i < d.size(); ++i)
        same = d[i].table_size == idescs_[i].table_size && d[i].learning_rate == idescs_[i].learning_rate &&
               d[i].slot_indirect == idescs_[i].slot_indirect && d[i].slot_run_map == idescs_[i].slot_run_map;
      return same ? GMX_OK : GMX_ERR_INVALID;
    }
    idescs_ = d;
    return gmx_indirect_create(&ind_, d.data(), (int)d.size(), ns, rm, S_, device_);
  }
  int EnsureLstm() { return lstm_ ? GMX_OK : gmx_lstm_create(&lstm_, S_, device_); }

  // ---- the ring ----
  // parts: which banks of the stream's Predictor are on the device (all streams of a pool alike);
  // lstm_slot / mixer_ctx_col / ind_ctx_col: where the LSTM's prediction and lstm_prediction_context go
  // (gmx_lstm_feed); models: the sink wants the feature models' predictions back as well
  int Join(int slot, uint64_t chunk_bits, int parts, int lstm_slot, int mixer_ctx_col, int ind_ctx_col, bool models,
           bool all_outputs) {
    std::unique_lock<std::mutex> lk(mu_);
    if (status_.load()) return status_.load();
    if (!ring_[0]) {
      T_ = chunk_bits < 8 ? 8 : chunk_bits & ~7ull;
      parts_ = parts;
      lstm_slot_ = lstm_slot;
      mixer_ctx_col_ = mixer_ctx_col;
      ind_ctx_col_ = ind_ctx_col;
      models_back_ = models;
      all_outputs_ = all_outputs;
      for (int k = 0; k < kRing; ++k) {
        // the probabilities, and of the mixers' outputs only the newest bit's (the blackboard) unless the sink
        // analyses mixers other than the final one: the kernel's throughput build stores nothing else
        int rc = all_outputs_ ? GMX_ERR_INVALID
                              : gmx_batch_create(&ring_[k], group_, T_, GMX_BATCH_LAST_OUTPUTS | GMX_BATCH_MASK);
        if (rc == GMX_ERR_INVALID) {  // (shapes whose kernels keep all outputs or none)
          all_outputs_ = true;
          rc = gmx_batch_create(&ring_[k], group_, T_, GMX_BATCH_OUTPUTS | GMX_BATCH_MASK);
        }
        if (rc) return Fail("gmx_batch_create", rc);
        if (!gmx_batch_predictions(ring_[k]) || !gmx_batch_active_mask(ring_[k]) || !gmx_batch_contexts(ring_[k]) ||
            !gmx_batch_bits(ring_[k]) || !gmx_batch_p(ring_[k]) ||
            !(all_outputs_ ? gmx_batch_outputs(ring_[k]) : gmx_batch_last_outputs(ring_[k])))
          return Fail("gmx_batch (pinned arrays)", GMX_ERR_NOMEM);
        if (parts_ & kIndirect) {
          if ((rc = gmx_ind_batch_create(&iring_[k], ind_, T_))) return Fail("gmx_ind_batch_create", rc);
          if (!gmx_ind_batch_contexts(iring_[k]) || !gmx_ind_batch_bit_contexts(iring_[k]) || !gmx_ind_batch_bits(iring_[k]))
            return Fail("gmx_ind_batch (pinned arrays)", GMX_ERR_NOMEM);
        }
        if (parts_ & kLstm) {
          if ((rc = gmx_lstm_batch_create(&lring_[k], lstm_, T_ / 8))) return Fail("gmx_lstm_batch_create", rc);
          if (!gmx_lstm_batch_ppm(lring_[k]) || !gmx_lstm_batch_bytes(lring_[k]))
            return Fail("gmx_lstm_batch (pinned arrays)", GMX_ERR_NOMEM);
        }
      }
      n_pad_ = gmx_batch_n_pad(ring_[0]);
      mask_words_ = gmx_batch_mask_words(ring_[0]);
      M_ = gmx_group_n_mixers(group_);
      K_ = (parts_ & kIndirect) ? gmx_indirect_n_models(ind_) : 0;
      n_cur_.assign(S_, 0);
      for (int k = 0; k < kRing; ++k) n_in_[k].assign(S_, 0);
      n_bytes_.assign(S_, 0);
    } else if (parts != parts_ || lstm_slot != lstm_slot_ || mixer_ctx_col != mixer_ctx_col_ || ind_ctx_col != ind_ctx_col_ ||
               (all_outputs && !all_outputs_)) {
      return GMX_ERR_INVALID;  // Predictors of different make in one pool (or a sink that needs what the ring does not carry)
    }
    models_back_ = models_back_ || models;
    n_cur_[slot] = 0;
    for (int k = 0; k < kRing; ++k) n_in_[k][slot] = 0;
    streams_[slot].ra = true;
    streams_[slot].t = 0;
    Records(slot);
    ++participants_;
    return GMX_OK;
  }
  // where `slot` records the chunk it is filling.  mu_ held.
  void Records(int slot) {
    Stream& st = streams_[slot];
    gmx_batch* b = ring_[cur_];
    st.pred = gmx_batch_predictions(b) + (size_t)slot * T_ * n_pad_;
    st.mask = gmx_batch_active_mask(b) + (size_t)slot * T_ * mask_words_;
    st.ctx = gmx_batch_contexts(b) + (size_t)slot * T_ * M_;
    st.bits = gmx_batch_bits(b) + (size_t)slot * T_;
    if (parts_ & kIndirect) {
      st.ictx = gmx_ind_batch_contexts(iring_[cur_]) + (size_t)slot * T_ * K_;
      st.ibc = gmx_ind_batch_bit_contexts(iring_[cur_]) + (size_t)slot * T_;
      st.ibits = gmx_ind_batch_bits(iring_[cur_]) + (size_t)slot * T_;
    }
    if (parts_ & kLstm) {
      st.ppm = gmx_lstm_batch_ppm(lring_[cur_]) + (size_t)slot * (T_ / 8) * 256;
      st.bytes = gmx_lstm_batch_bytes(lring_[cur_]) + (size_t)slot * (T_ / 8);
    }
  }
  // Every stream has handed in its chunk: queue it behind the chunk before, wait for THAT one.  mu_ held.
  void Lead() {
    const auto lead_t0 = std::chrono::steady_clock::now();
    const int c = cur_;
    uint64_t maxn = 0, sum = 0;
    for (int i = 0; i < S_; ++i) {
      maxn = std::max(maxn, n_cur_[i]);
      sum += n_cur_[i];
      n_bytes_[i] = n_cur_[i] / 8;
    }
    int rc = GMX_OK;
    const char* what = "";
#define GMX_POOL_STEP(call)                                                                              \
  if (rc == GMX_OK) {                                                                                    \
    const auto step_t0 = std::chrono::steady_clock::now();                                               \
    if ((rc = (call))) what = #call;                                                                     \
    if (trace_) {                                                                                        \
      const double step_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - step_t0).count(); \
      step_seconds_[#call] += step_s;                                                                    \
      if (round_ < 2) fprintf(stderr, "[gmx pool] round %llu: %8.3f ms  %.40s\n", (unsigned long long)round_, step_s * 1e3, #call); \
    }                                                                                                    \
  }
    // The oldest chunk in flight first: wait for its kernels (the mixers' are the last of the chain) and queue its
    // downloads -- of batches nothing else is queued on, so they run at once on the banks' download streams -- BEFORE
    // this chunk's uploads, which are tens of megabytes with many streams and would be in front of them in the copy
    // engines' queues.  They are waited for at the end of this call.
    const int other = (c + 1) % kRing;  // its arrays are the ones the streams fill next
    if (busy_[other]) {
      const uint64_t on = maxn_[other];
      GMX_POOL_STEP(gmx_batch_wait(ring_[other]));
      GMX_POOL_STEP(gmx_batch_download(ring_[other], on));
      if (models_back_ && (parts_ & kIndirect)) {
        GMX_POOL_STEP(gmx_ind_batch_wait(iring_[other]));
        GMX_POOL_STEP(gmx_ind_batch_download(iring_[other], on));
      }
      if (parts_ & kLstm) {
        GMX_POOL_STEP(gmx_lstm_batch_wait(lring_[other]));
        GMX_POOL_STEP(gmx_lstm_batch_download(lring_[other], on / 8));  // (44 bytes per byte: the context at least is wanted)
      }
    }
    const auto lead_tw = std::chrono::steady_clock::now();
    if (maxn > 0) {
      if (parts_ & kLstm) {
        // LstmModel::Predict x 8 / Learn for every byte of the chunk, then its prediction into the mixers' records
        // (slot, active flag) and lstm_prediction_context into the gate context / Indirect context that read it
        GMX_POOL_STEP(gmx_lstm_batch_upload(lring_[c], maxn / 8));
        GMX_POOL_STEP(gmx_lstm_run_ragged(lstm_, lring_[c], n_bytes_.data(), 1));
      }
      if (parts_ & kIndirect) GMX_POOL_STEP(gmx_ind_batch_upload(iring_[c], maxn));
      GMX_POOL_STEP(gmx_batch_upload(ring_[c], maxn));
      if (parts_ & kLstm) {
        GMX_POOL_STEP(gmx_lstm_feed(lstm_, lring_[c], maxn / 8, ring_[c], lstm_slot_, mixer_ctx_col_,
                                    (parts_ & kIndirect) && ind_ctx_col_ >= 0 ? iring_[c] : nullptr, ind_ctx_col_));
      }
      if (parts_ & kIndirect) {
        GMX_POOL_STEP(gmx_indirect_run_ragged(ind_, iring_[c], n_cur_.data(), 1, ring_[c]));
      }
      GMX_POOL_STEP(gmx_group_run_ragged(group_, ring_[c], n_cur_.data(), 1));
      // (no download queued here: a copy that waits milliseconds for its kernel holds up the uploads of the chunks
      // behind it in the copy engines' queues -- the results are fetched below, when their kernels are done)
      if (rc == GMX_OK) {
        busy_[c] = true;
        bits_submitted_ += sum;
      }
    }
    n_in_[c] = n_cur_;
    maxn_[c] = maxn;
    std::fill(n_cur_.begin(), n_cur_.end(), 0);
    const auto lead_t1 = std::chrono::steady_clock::now();
    if (rc == GMX_OK && busy_[other]) {
      busy_[other] = false;
      GMX_POOL_STEP(gmx_batch_wait(ring_[other]));  // (the downloads queued at the top of this call)
      if (models_back_ && (parts_ & kIndirect)) GMX_POOL_STEP(gmx_ind_batch_wait(iring_[other]));
      if (parts_ & kLstm) GMX_POOL_STEP(gmx_lstm_batch_wait(lring_[other]));
    }
#undef GMX_POOL_STEP
    if (rc) Fail(what, rc);
    submit_seconds_ += std::chrono::duration<double>(lead_t1 - lead_tw).count();
    wait_seconds_ += std::chrono::duration<double>((lead_tw - lead_t0) + (std::chrono::steady_clock::now() - lead_t1)).count();
    cur_ = other;
    arrived_ = 0;
    ++round_;
    cv_.notify_all();
  }
  // `slot` has recorded n bits into the chunk being filled.  Returns when that chunk is queued on the device
  // and the chunk before it is back: *v is the stream's stretch of THAT one (the arrays it fills next).
  int Arrive(int slot, uint64_t n, RunAheadView* v) {
    std::unique_lock<std::mutex> lk(mu_);
    if (status_.load()) return status_.load();
    if ((parts_ & kLstm) && (n & 7)) return GMX_ERR_STATE;  // the LSTM's records are bytes
    n_cur_[slot] = n;
    const uint64_t my_round = round_;
    if (trace_) {  // GMX_POOL_TRACE: a stream that took unusually long over its chunk (every other stream waits for it)
      const auto now = std::chrono::steady_clock::now();
      if (left_at_.size() == (size_t)S_ && my_round > 3) {
        const double ms = std::chrono::duration<double>(now - left_at_[slot]).count() * 1e3;
        if (ms > 5.0) fprintf(stderr, "[gmx pool] round %llu: stream %d filled its chunk in %.1f ms\n", (unsigned long long)my_round, slot, ms);
      }
    }
    if (++arrived_ >= participants_)
      Lead();
    else
      cv_.wait(lk, [&] { return round_ != my_round || status_.load() != 0; });
    if (status_.load()) return status_.load();
    gmx_batch* b = ring_[cur_];
    v->n = n_in_[cur_][slot];
    v->p = gmx_batch_p(b) + (size_t)slot * T_;
    if (all_outputs_) {
      v->outputs = gmx_batch_outputs(b) + (size_t)slot * T_ * M_;
      v->last_outputs = v->n ? v->outputs + (size_t)(v->n - 1) * M_ : nullptr;
    } else {
      v->last_outputs = gmx_batch_last_outputs(b) + (size_t)slot * M_;
    }
    v->n_mixers = M_;
    v->bits = gmx_batch_bits(b) + (size_t)slot * T_;
    if (models_back_ && (parts_ & kIndirect)) {
      v->ind_pred = gmx_ind_batch_predictions(iring_[cur_]) + (size_t)slot * T_ * 2 * K_;
      v->ind_active = gmx_ind_batch_active(iring_[cur_]) + (size_t)slot * T_ * 2 * K_;
      v->n_ind = K_;
    }
    if (parts_ & kLstm) {
      v->lstm_pred = gmx_lstm_batch_predictions(lring_[cur_]) + (size_t)slot * (T_ / 8) * 8;
      v->lstm_active = gmx_lstm_batch_active(lring_[cur_]) + (size_t)slot * (T_ / 8) * 8;
      v->lstm_context = gmx_lstm_batch_contexts(lring_[cur_]) + (size_t)slot * (T_ / 8);
    }
    n_in_[cur_][slot] = 0;
    streams_[slot].t = 0;
    Records(slot);
    if (trace_) {
      if (left_at_.size() != (size_t)S_) left_at_.assign(S_, std::chrono::steady_clock::now());
      left_at_[slot] = std::chrono::steady_clock::now();
    }
    return GMX_OK;
  }
  void Leave(int slot) {
    std::unique_lock<std::mutex> lk(mu_);
    streams_[slot].ra = false;
    if (participants_ > 0) --participants_;
    if (!n_cur_.empty()) n_cur_[slot] = 0;
    if (participants_ > 0 && arrived_ >= participants_ && status_.load() == 0) Lead();
  }

  // ---- lock step ----
  int JoinLockstep(int slot, int parts, int lstm_slot, int mixer_ctx_col, int ind_ctx_col) {
    std::unique_lock<std::mutex> lk(mu_);
    if (status_.load()) return status_.load();
    if (ring_[0]) return GMX_ERR_STATE;  // (a pool runs ahead or steps, not both)
    if (!cs_) {
      int rc = gmx_chainstep_create(&cs_, group_, (parts & kIndirect) ? ind_ : nullptr, (parts & kLstm) ? lstm_ : nullptr,
                                    lstm_slot, mixer_ctx_col, ind_ctx_col);
      if (rc) return Fail("gmx_chainstep_create", rc);
      parts_ = parts;
      lstm_slot_ = lstm_slot;
      mixer_ctx_col_ = mixer_ctx_col;
      ind_ctx_col_ = ind_ctx_col;
      M_ = gmx_group_n_mixers(group_);
      K_ = (parts_ & kIndirect) ? gmx_indirect_n_models(ind_) : 0;
      n_pad_ = (n_inputs_ + 3) / 4 * 4;
      mask_words_ = (n_inputs_ + 31) / 32;
      ls_what_ = gmx_chainstep_what(cs_);
    } else if (parts != parts_ || lstm_slot != lstm_slot_ || mixer_ctx_col != mixer_ctx_col_ || ind_ctx_col != ind_ctx_col_) {
      return GMX_ERR_INVALID;
    }
    streams_[slot].ls = true;
    ++ls_participants_;
    return GMX_OK;
  }
  void LeaveLockstep(int slot) {
    std::unique_lock<std::mutex> lk(mu_);
    if (streams_[slot].ls) {
      streams_[slot].ls = false;
      if (ls_participants_ > 0) --ls_participants_;
    }
  }
  // a stream has recorded what its next step needs: until the step has run
  void LockstepWait(int slot) {
    // the stream's records for the coming step are complete: into the device now, by this thread (S decoders on a few
    // worker threads move theirs side by side; what is left for the thread that steps is the control words)
    (void)gmx_chainstep_commit(cs_, slot);
    if (yield_)
      yield_(slot);
    else
      StepAll();
  }

  const int S_;
  int device_;
  bool shared_ = false;
  gmx_chainstep* cs_ = nullptr;
  uint8_t* ls_what_ = nullptr;
  int ls_participants_ = 0;
  std::function<void(int)> yield_;
  static constexpr uint64_t kLstmDrawsFnv = 0xcb25b734d7bec78full;  // FNV-1a over the draws (glibc's rand(): TYPE_3, r[i] = r[i-3] + r[i-31])
  std::vector<int> lstm_init_;
  bool lstm_init_tried_ = false;
  std::atomic<bool> lstm_from_cache_{false};
  gmx_group* group_ = nullptr;
  gmx_indirect* ind_ = nullptr;
  gmx_lstm* lstm_ = nullptr;
  int n_inputs_ = 0;
  std::vector<gmx_mixer_desc> descs_;
  std::vector<int32_t> skip_;
  std::vector<gmx_indirect_desc> idescs_;
  std::vector<Stream> streams_;
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<int> status_{0};
  std::string error_;
  // Three sets of batches at first: the hosts fill one while the device works on the two before it.  With two, the
  // hosts could only start on chunk k+2 when chunk k had come back whole, and the longest stage of the chain (the
  // LSTM) stood still meanwhile: one compressor ran at 4.0 us per bit, with three at the LSTM stage's own 3.4.
  // ... and four since the stages of the chain are of equal length (LSTM 4.7-4.9 ms, mixers 4.8-5.0 ms per chunk): with
  // three the hosts' own turn (wake up, drain, fill 2.3 ms, hand in) sat right on the critical path, and every bit of
  // jitter in it opened a gap in both stages (16 files: 5.9 ms per chunk against kernels of 5.0).  The library stages
  // what a launch reads beside its batch in as many slots (kStageSlots, gmx_capi.cpp).
  static constexpr int kRing = 4;
  gmx_batch* ring_[kRing] = {};
  gmx_ind_batch* iring_[kRing] = {};
  gmx_lstm_batch* lring_[kRing] = {};
  bool busy_[kRing] = {};
  uint64_t maxn_[kRing] = {};   // bits of the longest stream of the chunk in each ring slot
  int cur_ = 0, parts_ = 0, lstm_slot_ = -1, mixer_ctx_col_ = -1, ind_ctx_col_ = -1;
  bool models_back_ = false, all_outputs_ = false;
  uint64_t T_ = 0, round_ = 0, bits_submitted_ = 0;
  double submit_seconds_ = 0, wait_seconds_ = 0;
  const bool trace_ = getenv("GMX_POOL_TRACE") != nullptr;
  std::map<std::string, double> step_seconds_;
  std::vector<std::chrono::steady_clock::time_point> left_at_;  // (trace) when each stream last left Arrive
  int n_pad_ = 0, mask_words_ = 0, M_ = 0, K_ = 0;
  int participants_ = 0, arrived_ = 0;
  std::vector<uint64_t> n_cur_, n_in_[kRing], n_bytes_;
};

// All mixers of one Predictor: one stream of a gmx_group (the Predictor's own, or a MixerPool's).
class GpuMixerBank {
 public:
  // The bank of the Predictor that owns `ltm` (created on first use, gone with its last mixer).
  static std::shared_ptr<GpuMixerBank> For(ShortTermMemory& stm, LongTermMemory& ltm) {
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    auto& reg = Registry();
    auto it = reg.find(&ltm);
    if (it != reg.end())
      if (auto sp = it->second.lock()) return sp;
    std::shared_ptr<GpuMixerBank> sp(new GpuMixerBank(stm, ltm));
    reg[&ltm] = sp;
    return sp;
  }
  // The bank of the Predictor object at [p, p + size): its LongTermMemory is a member of it (predictor.h:41).
  static std::shared_ptr<GpuMixerBank> Of(const void* p, size_t size) {
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    const char* lo = static_cast<const char*>(p);
    for (auto& kv : Registry()) {
      const char* k = reinterpret_cast<const char*>(kv.first);
      if (k >= lo && k < lo + size)
        if (auto sp = kv.second.lock()) return sp;
    }
    return nullptr;
  }
  ~GpuMixerBank() {
    if (st().ra) pool_->Leave(slot_);
    if (st().ls) pool_->LeaveLockstep(slot_);
    {
      std::lock_guard<std::mutex> lk(pool_->mu_);
      st().mixers = nullptr;
    }
    pool_->Detach(slot_);
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    Registry().erase(&ltm_);
  }
  GpuMixerBank(const GpuMixerBank&) = delete;
  GpuMixerBank& operator=(const GpuMixerBank&) = delete;

  gmx_group* group() { return group_; }
  int stream() const { return slot_; }
  MixerPool* pool() { return pool_.get(); }
  ShortTermMemory& stm() { return stm_; }
  LongTermMemory& ltm() { return ltm_; }
  int n_mixers() const { return (int)descs_.size(); }
  bool running_ahead() const { return pool_->streams_[slot_].ra; }
  // 0, or the status of the C-ABI call that failed (a bank of a shared pool reports; a bank of its own aborts)
  int status() const { return pool_->status(); }
  // the device-side feature models of this Predictor, for whoever interprets a RunAheadView:
  // prediction slots of Indirect model i ([2i] indirect, [2i+1] run map), the LSTM's slot (-1: on the host)
  std::vector<int> IndirectSlots() const;
  int LstmSlot() const;
  int SlotsHome();  // the device-side models' prediction slots, back onto the blackboard

  // From the next Predict on, the device-side models record instead of compute; results reach `sink` one chunk
  // later.  Every bit must be Predict -> Perceive -> Learn (the paths that know their bits:
  // runner-utils.cpp:43-67, :223-322); with the LSTM on the device, start and end on byte boundaries.
  int BeginRunAhead(RunAheadSink* sink, uint64_t chunk_bits);
  // Hands in what is recorded, brings every outstanding result home (through the sink) and leaves the
  // blackboard's mixer outputs as per-bit calls would have.
  int EndRunAhead() {
    if (!st().ra) return GMX_OK;
    int rc = SyncRunAhead();
    pool_->Leave(slot_);
    sink_ = nullptr;
    return rc;
  }
  // Lock step (decoding, MixerPool above): from the next Predict on, this Predictor's device-side models step with the
  // other Predictors of its pool.  Starts at a byte boundary; every bit Predict -> Perceive -> Learn.
  int BeginLockstep();
  // The last bit's Learn goes out (a step of its own) and the stream leaves; the device-side models' blackboard slots
  // come home (a device call).  FinishLockstep is the first half alone -- no device call: a fibre may run it.
  int EndLockstep();
  void FinishLockstep();
  // The same without leaving run-ahead mode (a checkpoint in the middle of a file).
  int SyncRunAhead() {
    if (!st().ra) return GMX_OK;
    int rc = GMX_OK;  // what is recorded goes out, and every chunk in flight comes home
    for (int k = 0; k < MixerPool::kRing && rc == GMX_OK; ++k) rc = Flush();
    if (rc == GMX_OK) rc = SlotsHome();
    return rc;
  }

 private:
  friend class GpuMixer;
  friend class GpuIndirectBank;
  friend class GpuLstmModel;
  GpuMixerBank(ShortTermMemory& stm, LongTermMemory& ltm) : stm_(stm), ltm_(ltm) {
    pool_ = MixerPool::Attach(&ltm, &slot_);
    std::lock_guard<std::mutex> lk(pool_->mu_);
    st().mixers = this;
  }
  static std::map<const LongTermMemory*, std::weak_ptr<GpuMixerBank>>& Registry() {
    static std::map<const LongTermMemory*, std::weak_ptr<GpuMixerBank>> r;
    return r;
  }
  MixerPool::Stream& st() { return pool_->streams_[slot_]; }
  void Check(const char* what, int rc) { pool_->Check("gmx::GpuMixer", what, rc); }
  template <class F>
  void Call(const char* what, F f) {
    pool_->Call("gmx::GpuMixer", what, f);
  }
  int Register(GpuMixer* m, int layer, unsigned table_size, float lr, int memory_index, int weight_size) {
    gmx_mixer_desc d;
    d.layer = layer;
    d.table_size = table_size;
    d.learning_rate = lr;
    descs_.push_back(d);
    mixers_.push_back(m);
    memory_index_.push_back(memory_index);
    weight_size_.push_back(weight_size);
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
    Call("gmx_group_create", [&] { return pool_->EnsureGroup(t); });
    group_ = pool_->group();
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
    if (st().ra) SyncRunAhead();
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
    Call("gmx_bank_import",
         [&] { return gmx_bank_import(group_, slot_, buf.data(), buf.size(), short_in_.data(), short_in_.size()); });
    staged_ = true;
    ever_ran_ = true;
  }
  // Bank -> LongTermMemory::mixers (+ the 3 x u64 of every mixer), for the reference's writers.
  void Stage() {
    Settle();
    if (st().ra) SyncRunAhead();
    size_t nl = 0, ns = 0;
    Call("gmx_bank_export", [&] { return gmx_bank_export(group_, slot_, nullptr, &nl, nullptr, &ns); });
    std::vector<char> l(nl ? nl : 1);
    short_cache_.assign(ns ? ns : 1, 0);
    Call("gmx_bank_export", [&] { return gmx_bank_export(group_, slot_, l.data(), &nl, short_cache_.data(), &ns); });
    if (status()) return;
    short_cache_.resize(ns);
    const char* p = l.data();
    for (size_t j = 0; j < descs_.size(); ++j) {
      auto& table = ltm_.mixers[memory_index_[j]].mixer_table;
      table.clear();  // (rows of an earlier staging go; the vector is sized for the reference's writers)
      table.resize(descs_[j].table_size);
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
  // LongTermMemory::mixers[..].mixer_table holds one pointer per gate-table row (the stock tables: ~10^8 rows, most
  // of a gigabyte of pointers) which the reference's Mixer fills as contexts are seen.  Here the rows live on the
  // device, so the vectors stay EMPTY except around the reference's own serialisers -- sized before
  // LongTermMemory::ReadFromDisk (which keeps a table's size, long-term-memory.cpp:133-137) and while staged for
  // WriteToDisk.  (LongTermMemory::Copy then has nothing to clear, allocate and walk for them: a Predictor::Copy
  // took 0.9 s, most of it there.)
  void HostTables(bool on) {
    for (size_t j = 0; j < descs_.size(); ++j) {
      auto& table = ltm_.mixers[memory_index_[j]].mixer_table;
      if (on) {
        if (table.size() != descs_[j].table_size) {
          table.clear();
          table.resize(descs_[j].table_size);
        }
      } else {
        std::vector<std::unique_ptr<MixerData>>().swap(table);
      }
    }
  }
  void Unstage() {
    if (import_pending_) return;
    if (!staged_ && (descs_.empty() || ltm_.mixers[memory_index_[0]].mixer_table.empty())) return;
    staged_ = false;
    HostTables(false);
  }
  void PredictAll(ShortTermMemory& stm);
  void LearnAll(const ShortTermMemory& stm) {
    Settle();
    MixerPool::Stream& s = st();
    if (s.ls) {  // Mixer::Learn x 33 (and the other device-side models' Learn): asked of the next step
      if (!ls_predicted_ || status()) return;
      ls_predicted_ = false;
      gmx_chainstep_bits(pool_->cs_)[slot_] = (uint8_t)stm.new_bit;
      pool_->ls_what_[slot_] |= GMX_STEP_LEARN;
      return;
    }
    if (s.ra) {
      if (!recorded_ || status()) return;  // (a Learn without its Predict has nothing to learn from)
      recorded_ = false;
      s.bits[s.t] = (uint8_t)stm.new_bit;
      if (track_seen_) {  // Mixer::FindOrCreateMixerData's ++contexts_seen_ (mixer.cpp:39-49)
        const uint32_t* c = s.ctx + (size_t)s.t * descs_.size();
        for (size_t j = 0; j < descs_.size(); ++j) {
          if ((int)j == lstm_ctx_col_) continue;    // (that context is the device's: counted when its chunk returns)
          if (!first_mark_ && c[j] == last_ctx_[j]) continue;  // (most contexts stand for a whole byte)
          last_ctx_[j] = c[j];
          MarkSeen(j, c[j]);
        }
        first_mark_ = false;
      }
      if (++s.t == T_) Flush();
      return;
    }
    ever_ran_ = true;
    Call("gmx_bank_learn", [&] { return gmx_bank_learn(group_, slot_, stm.new_bit); });
  }
  void MarkSeen(size_t j, uint32_t context) {
    const uint32_t row = context % descs_[j].table_size;
    uint64_t& w = seen_[j][row >> 6];
    const uint64_t bit = 1ull << (row & 63);
    if (!(w & bit)) {
      w |= bit;
      ++seen_count_[j];
    }
  }
  void CopyFrom(GpuMixerBank& o) {
    if (o.st().ra) o.SyncRunAhead();
    if (st().ra) SyncRunAhead();
    o.Settle();
    o.Unstage();  // (LongTermMemory::Copy, which follows, would copy staged rows one by one)
    Ensure();
    import_pending_ = false;  // whatever LongTermMemory::Copy moves into the staging area is not ours to import
    ever_ran_ = true;
    if (o.pool_ == pool_) {
      Call("gmx_bank_copy", [&] { return gmx_bank_copy(group_, slot_, o.group_, o.slot_); });
    } else {  // two groups: both pools' calls held off, always in address order
      MixerPool* a = pool_.get() < o.pool_.get() ? pool_.get() : o.pool_.get();
      MixerPool* b = pool_.get() < o.pool_.get() ? o.pool_.get() : pool_.get();
      int rc;
      {
        std::lock_guard<std::mutex> la(a->mu_);
        std::lock_guard<std::mutex> lb(b->mu_);
        rc = gmx_bank_copy(group_, slot_, o.group_, o.slot_);
      }
      Check("gmx_bank_copy", rc);
    }
    if (st().ra && !o.seen_.empty()) {  // the copied bank's rows are this bank's now
      seen_ = o.seen_;
      seen_count_ = o.seen_count_;
    }
  }
  unsigned long long MemoryUsage(int index) {
    Settle();
    if (st().ra && track_seen_)  // mixer.cpp:197-205 from the host's own count of rows (the device runs a chunk behind)
      return 29ull + seen_count_[index] * (unsigned long long)(weight_size_[index] * 4 + 12) +
             8ull * descs_[index].table_size;
    if (st().ra) SyncRunAhead();  // (a sink that had not asked for it: every chunk home first, then the device's count)
    uint64_t v = 0;
    Call("gmx_bank_memory_usage", [&] { return gmx_bank_memory_usage(group_, slot_, index, &v); });
    return v;
  }
  // One round of the ring: the chunk recorded so far goes to the device, the chunk before it comes back.
  int Flush();

  ShortTermMemory& stm_;
  LongTermMemory& ltm_;
  std::shared_ptr<MixerPool> pool_;
  int slot_ = 0;
  gmx_group* group_ = nullptr;
  std::vector<gmx_mixer_desc> descs_;
  std::vector<GpuMixer*> mixers_;
  std::vector<int> memory_index_, weight_size_;
  std::vector<float> outputs_;
  std::vector<uint32_t> contexts_;
  std::vector<float> chain_pred_;
  std::vector<uint8_t> chain_act_;
  std::vector<char> short_cache_, short_in_;
  bool import_pending_ = false, staged_ = false, ever_ran_ = false;
  bool ls_predicted_ = false;  // lock step: a Predict waits for its Learn
  // run-ahead
  bool recorded_ = false;
  RunAheadSink* sink_ = nullptr;
  uint64_t T_ = 0;
  int n_pad_ = 0, mask_words_ = 0, lstm_ctx_col_ = -1;
  std::vector<std::vector<uint64_t>> seen_;
  std::vector<uint32_t> last_ctx_;   // the context each mixer's row was last marked for
  std::vector<uint8_t> last_ind_active_;  // the Indirect models' active flags of the newest bit that came back
  bool track_seen_ = false, first_mark_ = true;
  std::vector<uint64_t> seen_count_;
};

class GpuMixer : public Model {
 public:
  // mixer/mixer.h:17-19, argument for argument.
  GpuMixer(ShortTermMemory& short_term_memory, LongTermMemory& long_term_memory, unsigned int& context,
           float learning_rate, int layer_number, unsigned int table_size, std::string description,
           bool enable_analysis)
      : context_(context), bank_(GpuMixerBank::For(short_term_memory, long_term_memory)) {
    // mixer.cpp:12-15: the registrations Mixer::Mixer makes
    const int output_index = short_term_memory.AddMixer(description, layer_number, enable_analysis, this);
    int memory_index = (int)long_term_memory.mixers.size();
    long_term_memory.mixers.push_back(MixerMemory(0));  // (no host rows: GpuMixerBank::HostTables)
    int weight_size;  // mixer.cpp:17-26
    if (layer_number == 0)
      weight_size = short_term_memory.num_predictions + output_index;
    else if (layer_number == 1)
      weight_size = short_term_memory.num_layer0_mixers + output_index +
                    (int)short_term_memory.models_with_skip_connection.size();
    else
      weight_size = short_term_memory.num_layer0_mixers + short_term_memory.num_layer1_mixers +
                    (int)short_term_memory.models_with_skip_connection.size();
    index_ = bank_->Register(this, layer_number, table_size, learning_rate, memory_index, weight_size);
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
    if (index_ == 0) bank_->HostTables(true);  // LongTermMemory::ReadFromDisk comes next (predictor.cpp:412-416) and fills them
  }
  // mixer.cpp:190-195 (+ the mixers' share of LongTermMemory::Copy, long-term-memory.cpp:201-214)
  void Copy(const MemoryInterface* m) override {
    const GpuMixer* orig = static_cast<const GpuMixer*>(m);
    if (index_ == 0) bank_->CopyFrom(*orig->bank_);
  }
  // mixer.cpp:197-205
  unsigned long long GetMemoryUsage(const ShortTermMemory&, const LongTermMemory&) override {
    return bank_->MemoryUsage(index_);
  }
  unsigned int context() const { return context_; }
  const unsigned int* context_address() const { return &context_; }

 private:
  unsigned int& context_;  // aliases a field of the Predictor's blackboard (mixer.h:31)
  std::shared_ptr<GpuMixerBank> bank_;
  int index_;  // construction order within the bank
};

// ================================================================================================
// The same for the two feature models that live on the device next to the mixers (SURVEY.md
// section 8f ranks 3 and 4): `gmx::GpuIndirect` with Indirect's constructor signature
// (models/indirect.h:16-18) and `gmx::GpuLstmModel` with LstmModel's (models/lstm-model.h:11-12).
// Switched in like the mixers (`new Indirect(` -> `new gmx::GpuIndirect(` at its 33 places, `new LstmModel(`
// -> `new gmx::GpuLstmModel(` once), the reference's Predictor keeps LSTM -> 41 Indirect -> 33 mixers
// on the MI355X while PPMd, the match models, the context hashes, the coder and the runners stay the
// reference's own host code (dropin/Makefile builds it as gmix_chain / ref_tester_chain).
// ================================================================================================

// All Indirect models of one Predictor: one stream of a gmx_indirect.  The last feature model in front of the
// mixers is an Indirect model (predictor.cpp:24-28), so per bit its bank does not run at its own Predict but
// hands contexts to the first mixer's: gmx_chain_forward then takes both banks through ONE host round trip (the
// Indirect wave rings the mixers' wave itself).  GMX_CHAIN_FUSED=0 switches back to one call per bank.
class GpuIndirectBank {
 public:
  static std::shared_ptr<GpuIndirectBank> For(ShortTermMemory& stm, LongTermMemory& ltm) {
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    auto& reg = Registry();
    auto it = reg.find(&ltm);
    if (it != reg.end())
      if (auto sp = it->second.lock()) return sp;
    std::shared_ptr<GpuIndirectBank> sp(new GpuIndirectBank(stm, ltm));
    reg[&ltm] = sp;
    return sp;
  }
  ~GpuIndirectBank() {
    {
      std::lock_guard<std::mutex> lk(pool_->mu_);
      st().indirect = nullptr;
    }
    pool_->Detach(slot_);
    std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
    Registry().erase(&ltm_);
  }
  GpuIndirectBank(const GpuIndirectBank&) = delete;
  GpuIndirectBank& operator=(const GpuIndirectBank&) = delete;

 private:
  friend class GpuIndirect;
  friend class GpuMixerBank;
  GpuIndirectBank(ShortTermMemory& stm, LongTermMemory& ltm) : stm_(stm), ltm_(ltm) {
    pool_ = MixerPool::Attach(&ltm, &slot_);
    std::lock_guard<std::mutex> lk(pool_->mu_);
    st().indirect = this;
  }
  static std::map<const LongTermMemory*, std::weak_ptr<GpuIndirectBank>>& Registry() {
    static std::map<const LongTermMemory*, std::weak_ptr<GpuIndirectBank>> r;
    return r;
  }
  MixerPool::Stream& st() { return pool_->streams_[slot_]; }
  template <class F>
  void Call(const char* what, F f) {
    pool_->Call("gmx::GpuIndirect", what, f);
  }
  int Register(GpuIndirect* m, unsigned table_size, float lr, int slot_a, int slot_b, int memory_index) {
    gmx_indirect_desc d;
    d.table_size = table_size;
    d.learning_rate = lr;
    d.slot_indirect = slot_a;
    d.slot_run_map = slot_b;
    descs_.push_back(d);
    models_.push_back(m);
    memory_index_.push_back(memory_index);
    return (int)descs_.size() - 1;
  }
  void Ensure() {
    if (h_) return;
    // the two state machines are ShortTermMemory's data (short-term-memory.h:144-145): tabulated
    // here by calling the reference's own Next(), walked on the device
    uint8_t ns[512], rm[512];
    for (int s = 0; s < 256; ++s)
      for (int b = 0; b < 2; ++b) {
        ns[2 * s + b] = (uint8_t)stm_.nonstationary.Next(s, b);
        rm[2 * s + b] = (uint8_t)stm_.run_map.Next(s, b);
      }
    Call("gmx_indirect_create", [&] { return pool_->EnsureIndirect(descs_, ns, rm); });
    h_ = pool_->indirect();
    contexts_.assign(descs_.size(), 0u);
    pred_.assign(2 * descs_.size(), 0.f);
    active_.assign(2 * descs_.size(), 0);
  }
  // LongTermMemory::indirect is only the staging area of the reference's serialisers here: its tables (790 MB for
  // the stock models) exist while a checkpoint is being written or read, and not otherwise.
  void StagingTables(bool on) {
    for (size_t i = 0; i < descs_.size(); ++i) {
      auto& mem = ltm_.indirect[memory_index_[i]];
      if (on) {
        const size_t size = (size_t)descs_[i].table_size * 256 + 1;  // indirect.cpp:14-19
        mem.nonstationary_table.assign(size, 255);
        mem.run_map_table.assign(size, 0);
      } else {
        std::vector<unsigned char>().swap(mem.nonstationary_table);
        std::vector<unsigned char>().swap(mem.run_map_table);
      }
    }
    tables_staged_ = on;
  }
  // Tables the reference's LongTermMemory::ReadFromDisk has read since ReadFromDisk was called on the
  // models go to the device (the indirect section's own format, long-term-memory.cpp:8-32).
  void Settle() {
    Ensure();
    if (!import_pending_) return;
    import_pending_ = false;
    std::vector<char> buf;
    auto put = [&buf](const void* p, size_t n) {
      const char* c = static_cast<const char*>(p);
      buf.insert(buf.end(), c, c + n);
    };
    for (size_t i = 0; i < descs_.size(); ++i) {
      auto& mem = ltm_.indirect[memory_index_[i]];
      std::vector<unsigned int> keys;
      for (unsigned int k = 0; k < mem.nonstationary_table.size(); ++k)
        if (mem.nonstationary_table[k] != 255) keys.push_back(k);
      unsigned int size = (unsigned int)keys.size();
      put(&size, 4);
      if (size < mem.nonstationary_table.size() / 3) {
        for (unsigned int key : keys) {
          put(&key, 4);
          put(&mem.nonstationary_table[key], 1);
          put(&mem.run_map_table[key], 1);
        }
      } else {
        put(mem.nonstationary_table.data(), mem.nonstationary_table.size());
        put(mem.run_map_table.data(), mem.run_map_table.size());
      }
      put(mem.nonstationary_predictions.data(), 256 * 4);
      put(mem.run_map_predictions.data(), 256 * 4);
    }
    Call("gmx_indirect_import", [&] { return gmx_indirect_import(h_, slot_, buf.data(), buf.size()); });
    StagingTables(false);
  }
  // Device -> LongTermMemory::indirect, for the reference's writer.
  void Stage() {
    Settle();
    size_t n = 0;
    Call("gmx_indirect_export", [&] { return gmx_indirect_export(h_, slot_, nullptr, &n); });
    std::vector<char> buf(n ? n : 1);
    Call("gmx_indirect_export", [&] { return gmx_indirect_export(h_, slot_, buf.data(), &n); });
    if (pool_->status()) return;
    StagingTables(true);
    const char* p = buf.data();
    for (size_t i = 0; i < descs_.size(); ++i) {
      auto& mem = ltm_.indirect[memory_index_[i]];
      const size_t size = mem.nonstationary_table.size();
      unsigned int count;
      memcpy(&count, p, 4);
      p += 4;
      if (count < size / 3) {
        for (unsigned int k = 0; k < count; ++k) {
          unsigned int key;
          memcpy(&key, p, 4);
          mem.nonstationary_table[key] = (unsigned char)p[4];
          mem.run_map_table[key] = (unsigned char)p[5];
          p += 6;
        }
      } else {
        memcpy(mem.nonstationary_table.data(), p, size);
        memcpy(mem.run_map_table.data(), p + size, size);
        p += 2 * size;
      }
      memcpy(mem.nonstationary_predictions.data(), p, 256 * 4);
      memcpy(mem.run_map_predictions.data(), p + 1024, 256 * 4);
      p += 2048;
    }
  }
  void Unstage() {
    if (tables_staged_ && !import_pending_) StagingTables(false);
  }
  void PredictAll(ShortTermMemory& stm);
  void ToBlackboard(ShortTermMemory& stm);
  // The models' blackboard slots between ShortTermMemory::predictions and the bank (gmx_indirect_slots_set / _get):
  // while batches run the bank's copy is the current one, and the reference writes these with the blackboard.
  void SlotsToDevice(const ShortTermMemory& stm) {
    for (size_t i = 0; i < descs_.size(); ++i) {
      pred_[2 * i] = stm.predictions[descs_[i].slot_indirect];
      pred_[2 * i + 1] = stm.predictions[descs_[i].slot_run_map];
    }
    Call("gmx_indirect_slots_set", [&] { return gmx_indirect_slots_set(h_, slot_, pred_.data()); });
  }
  void SlotsFromDevice(ShortTermMemory& stm) {
    Call("gmx_indirect_slots_get", [&] { return gmx_indirect_slots_get(h_, slot_, pred_.data()); });
    if (pool_->status()) return;
    for (size_t i = 0; i < descs_.size(); ++i) {
      stm.predictions[descs_[i].slot_indirect] = pred_[2 * i];
      stm.predictions[descs_[i].slot_run_map] = pred_[2 * i + 1];
    }
  }
  void Deliver(ShortTermMemory& stm, const float* pred, const uint8_t* active) {  // from gmx_chain_forward
    pending_ = false;
    std::copy(pred, pred + pred_.size(), pred_.begin());
    std::copy(active, active + active_.size(), active_.begin());
    ToBlackboard(stm);
  }
  void LearnAll(const ShortTermMemory& stm) {
    Settle();
    Unstage();
    MixerPool::Stream& s = st();
    if (s.ls) return;  // (lock step: the mixers' Learn asks for the step's learn, which is every device-side model's)
    if (s.ra) {
      s.ibits[s.t] = (uint8_t)stm.new_bit;  // Indirect::Learn x 41, recorded
      return;
    }
    Call("gmx_indirect_learn", [&] { return gmx_indirect_learn(h_, slot_, stm.new_bit); });
  }
  void CopyFrom(GpuIndirectBank& o);

  ShortTermMemory& stm_;
  LongTermMemory& ltm_;
  std::shared_ptr<MixerPool> pool_;
  int slot_ = 0;
  gmx_indirect* h_ = nullptr;
  std::vector<gmx_indirect_desc> descs_;
  std::vector<GpuIndirect*> models_;
  std::vector<int> memory_index_;
  std::vector<uint32_t> contexts_;
  std::vector<float> pred_;
  std::vector<uint8_t> active_;
  bool import_pending_ = false, tables_staged_ = false;
  bool pending_ = false;  // this bit's Predict waits for the mixers' (gmx_chain_forward)
  uint32_t pending_bit_context_ = 0;
};

class GpuIndirect : public Model {
 public:
  // models/indirect.h:16-18, argument for argument.
  GpuIndirect(ShortTermMemory& short_term_memory, LongTermMemory& long_term_memory, float learning_rate,
              unsigned int table_size, unsigned int& context, std::string description, bool enable_analysis)
      : context_(context), bank_(GpuIndirectBank::For(short_term_memory, long_term_memory)) {
    // indirect.cpp:10-26: the registrations Indirect::Indirect makes.  The tables in LongTermMemory are only the
    // staging area of the reference's serialisers here (as for the mixers) and stay empty until one runs.
    const int a = short_term_memory.AddPrediction(description + "-indirect", enable_analysis, this);
    const int b = short_term_memory.AddPrediction(description + "-run_map", enable_analysis, this);
    const int memory_index = (int)long_term_memory.indirect.size();
    long_term_memory.indirect.push_back(IndirectMemory(0));
    for (int i = 0; i < 256; ++i) {
      long_term_memory.indirect.back().nonstationary_predictions[i] = 0;
      long_term_memory.indirect.back().run_map_predictions[i] = 0;
    }
    index_ = bank_->Register(this, table_size, learning_rate, a, b, memory_index);
  }
  // The context of a model may come from the model just in front of it (IndirectHash and SkipContext
  // sit between the Indirect models, predictor.cpp:78-250), so the bank runs when the LAST of them is
  // called: every context is final by then, and nothing in between reads an Indirect prediction.
  void Predict(ShortTermMemory& short_term_memory, const LongTermMemory&) override {
    if (index_ + 1 == (int)bank_->models_.size()) bank_->PredictAll(short_term_memory);
  }
  void Learn(const ShortTermMemory& short_term_memory, LongTermMemory&) override {
    if (index_ == 0) bank_->LearnAll(short_term_memory);
  }
  void WriteToDisk(std::ofstream*) override {  // indirect.h:23: nothing in .short; stage for the .long writer
    if (index_ == 0) bank_->Stage();
  }
  void ReadFromDisk(std::ifstream*) override {  // the reference reads the tables next (predictor.cpp:412-416)
    bank_->Ensure();
    if (index_ == 0) bank_->StagingTables(true);
    bank_->import_pending_ = true;
  }
  void Copy(const MemoryInterface* m) override {
    const GpuIndirect* orig = static_cast<const GpuIndirect*>(m);
    if (index_ == 0) bank_->CopyFrom(*orig->bank_);
  }
  unsigned long long GetMemoryUsage(const ShortTermMemory&, const LongTermMemory&) override {  // indirect.cpp:71-78
    bank_->Ensure();
    uint64_t v = 0;
    bank_->Call("gmx_indirect_memory_usage", [&] { return gmx_indirect_memory_usage(bank_->h_, index_, &v); });
    return v;
  }
  unsigned int context() const { return context_; }
  const unsigned int* context_address() const { return &context_; }

 private:
  unsigned int& context_;  // aliases a field of the Predictor's blackboard (indirect.h:31)
  std::shared_ptr<GpuIndirectBank> bank_;
  int index_;
};

inline void GpuIndirectBank::CopyFrom(GpuIndirectBank& o) {
  o.Settle();
  Ensure();
  import_pending_ = false;
  if (o.pool_ == pool_) {
    Call("gmx_indirect_copy", [&] { return gmx_indirect_copy(h_, slot_, o.h_, o.slot_); });
  } else {
    MixerPool* a = pool_.get() < o.pool_.get() ? pool_.get() : o.pool_.get();
    MixerPool* b = pool_.get() < o.pool_.get() ? o.pool_.get() : pool_.get();
    int rc;
    {
      std::lock_guard<std::mutex> la(a->mu_);
      std::lock_guard<std::mutex> lb(b->mu_);
      rc = gmx_indirect_copy(h_, slot_, o.h_, o.slot_);
    }
    pool_->Check("gmx::GpuIndirect", "gmx_indirect_copy", rc);
  }
}

inline void GpuIndirectBank::PredictAll(ShortTermMemory& stm) {
  Settle();
  Unstage();
  if (pool_->status()) return;
  MixerPool::Stream& s = st();
  if (s.ls) {  // lock step: the contexts into the step's records; the mixers' Predict, next in line, waits for the step
    uint32_t* c = gmx_chainstep_ind_contexts(pool_->cs_) + (size_t)slot_ * models_.size();
    for (size_t i = 0; i < models_.size(); ++i) c[i] = models_[i]->context();
    gmx_chainstep_bit_contexts(pool_->cs_)[slot_] = stm.bit_context;
    return;
  }
  if (s.ra) {
    // Indirect::Predict x 41, recorded: the contexts as they stand when the last model is called, bit_context
    uint32_t* c = s.ictx + (size_t)s.t * models_.size();
    for (size_t i = 0; i < models_.size(); ++i) c[i] = models_[i]->context();
    s.ibc[s.t] = stm.bit_context;
    return;
  }
  for (size_t i = 0; i < models_.size(); ++i) contexts_[i] = models_[i]->context();
  if (ChainFused() && s.mixers) {
    // this Predictor's mixers are on the device too and theirs is the next Predict (predictor.cpp:24-28):
    // they take this bank along (gmx_chain_forward)
    pending_ = true;
    pending_bit_context_ = stm.bit_context;
    return;
  }
  Call("gmx_indirect_forward", [&] {
    return gmx_indirect_forward(h_, slot_, contexts_.data(), stm.bit_context, pred_.data(), active_.data());
  });
  ToBlackboard(stm);
}

inline void GpuIndirectBank::ToBlackboard(ShortTermMemory& stm) {
  // What 41 x Indirect::Predict leave on the blackboard (indirect.cpp:35-44 through SetLogitPrediction,
  // short-term-memory.cpp:193-197): an active model stores its logit and joins active_models, a zero
  // logit is stored but not active, a model that has never seen its state stores nothing -- the bank
  // then reports the slot's old value, which is only written back when it is that stored zero.
  for (size_t i = 0; i < descs_.size(); ++i) {
    const int slot[2] = {descs_[i].slot_indirect, descs_[i].slot_run_map};
    for (int k = 0; k < 2; ++k) {
      const float v = pred_[2 * i + k];
      if (active_[2 * i + k]) {
        stm.predictions[slot[k]] = v;
        stm.active_models.push_back(slot[k]);
      } else if (v == 0) {
        stm.predictions[slot[k]] = v;
      }
    }
  }
  // the reference's models push their (ascending) indices as they are called; the match models sit
  // between the Indirect groups, so restore that order
  std::sort(stm.active_models.begin(), stm.active_models.end());
}

// LstmModel (models/lstm-model.cpp) with Lstm(256, 256, 50, 1, 100, 0.03, 10) on the device: the byte-level
// network once per byte through gmx_lstm_forward / gmx_lstm_perceive, the bit-level range coding of
// the byte distribution (lstm-model.cpp:34-48) on the host as in the reference -- or, running ahead, all of it in
// gmx_lstm_run on the bytes' recorded PPM distributions.
class GpuLstmModel : public Model {
 public:
  static constexpr int kCells = 50, kInputs = 563, kHorizon = 100, kOut = 256, kHidden = 51;
  // models/lstm-model.h:11-12
  GpuLstmModel(ShortTermMemory& short_term_memory, LongTermMemory& long_term_memory, bool enable_analysis)
      : stm_(short_term_memory), ltm_(long_term_memory), top_(255), mid_(127), bot_(0), probs_(1.0 / 256, 256) {
    // What the constructors behind Lstm(256, 256, 50, 1, 100, 0.03, 10, ltm) do to LongTermMemory
    // (lstm.cpp:26-29, lstm-layer.cpp:57-59, :179-194): the output-layer ring, three gate matrices,
    // their initial values drawn from rand() in the reference's interleaved order.
    ltm_.lstm_output_layer.resize(
        kHorizon, std::valarray<std::valarray<float>>(std::valarray<float>(kHidden), kOut));
    first_layer_ = (int)ltm_.neuron_layer_weights.size();
    for (int g = 0; g < 3; ++g) ltm_.neuron_layer_weights.push_back(NeuronLayerWeights(kInputs, kCells));
    const float val = std::sqrt(6.0f / float(256 + 256));
    const float low = -val, range = 2 * val;
    // (a pool of many Predictors has drawn these numbers once: MixerPool::DrawLstmInit)
    const int* drawn = nullptr;
    {
      std::lock_guard<std::recursive_mutex> lk(AdapterMutex());
      if (MixerPool* inst = MixerPool::Current()) {
        drawn = inst->lstm_init();
        if (drawn) inst->lstm_from_cache_.store(true);
      }
    }
    for (int i = 0; i < kCells; ++i) {
      for (int j = 0; j < kInputs; ++j)
        for (int g = 0; g < 3; ++g)
          ltm_.neuron_layer_weights[first_layer_ + g].weights[i][j] =
              low + (static_cast<float>(drawn ? *drawn++ : rand()) / static_cast<float>(RAND_MAX)) * range;
      ltm_.neuron_layer_weights[first_layer_].weights[i][kInputs - 1] = 1;
    }
    prediction_index_ = short_term_memory.AddPrediction("LSTM", enable_analysis, this);
    short_term_memory.models_with_skip_connection.push_back(prediction_index_);
    pool_ = MixerPool::Attach(&long_term_memory, &slot_);
    std::lock_guard<std::mutex> lk(pool_->mu_);
    st().lstm = this;
  }
  ~GpuLstmModel() override {
    {
      std::lock_guard<std::mutex> lk(pool_->mu_);
      st().lstm = nullptr;
    }
    pool_->Detach(slot_);
  }
  void Predict(ShortTermMemory& short_term_memory, const LongTermMemory&) override {  // lstm-model.cpp:17-49
    MixerPool::Stream& s = st();
    if (s.ls) {  // lock step: the PPM byte distribution of a byte that opens; the device walks the bits itself
      if (short_term_memory.recent_bits == 1)
        memcpy(gmx_chainstep_ppm(pool_->cs_) + (size_t)slot_ * 256, &short_term_memory.ppm_predictions[0], 1024);
      range_on_device_ = true;
      return;
    }
    if (s.ra) {
      // Lstm::SetInput + Lstm::Predict at a byte boundary, recorded: the PPM byte distribution as it stands
      if (short_term_memory.recent_bits == 1)
        memcpy(s.ppm + (size_t)(s.t / 8) * 256, &short_term_memory.ppm_predictions[0], 1024);
      return;
    }
    if (short_term_memory.recent_bits == 1) {
      Settle();
      uint32_t ctx = 0;
      Call("gmx_lstm_forward", [&] {
        return gmx_lstm_forward(h_, slot_, (int)short_term_memory.last_byte, &short_term_memory.ppm_predictions[0],
                                &probs_[0], &ctx);
      });
      short_term_memory.lstm_prediction_context = ctx;
      top_ = 255;
      bot_ = 0;
      range_on_device_ = false;
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
  void Learn(const ShortTermMemory& short_term_memory, LongTermMemory&) override {  // lstm-model.cpp:51-60
    const int current_byte = short_term_memory.recent_bits * 2 + short_term_memory.new_bit;
    if (current_byte < 256) return;
    MixerPool::Stream& s = st();
    if (s.ls) return;  // (lock step: the device puts the byte together from the bits it is told)
    if (s.ra) {
      s.bytes[s.t / 8] = (uint8_t)(current_byte - 256);  // Lstm::Perceive, recorded
      return;
    }
    Settle();
    Call("gmx_lstm_perceive", [&] { return gmx_lstm_perceive(h_, slot_, current_byte - 256); });
  }
  // lstm-model.cpp:62-68 and everything behind it: the device writes the model's stretch of the .short
  // file; the range state is this object's own.  The LSTM section of the .long file is the reference's
  // to write (long-term-memory.cpp:57-67), from the arrays staged here.
  void WriteToDisk(std::ofstream* s) override {
    Settle();
    SyncIfAhead();
    size_t nl = 0, ns = 0;
    Call("gmx_lstm_export", [&] { return gmx_lstm_export(h_, slot_, nullptr, &nl, nullptr, &ns); });
    std::vector<char> l(nl ? nl : 1), sh(ns ? ns : 1);
    Call("gmx_lstm_export", [&] { return gmx_lstm_export(h_, slot_, l.data(), &nl, sh.data(), &ns); });
    if (pool_->status()) return;
    if (!range_on_device_) {  // (after a stretch of run-ahead the device's own copy of these is the current one)
      memcpy(sh.data(), &top_, 4);
      memcpy(sh.data() + 4, &mid_, 4);
      memcpy(sh.data() + 8, &bot_, 4);
      memcpy(sh.data() + 12, &probs_[0], 1024);
    }
    s->write(sh.data(), ns);
    const float* f = reinterpret_cast<const float*>(l.data());
    for (auto& x : ltm_.lstm_output_layer)
      for (auto& y : x) {
        memcpy(&y[0], f, 4 * y.size());
        f += y.size();
      }
    for (int g = 0; g < 3; ++g)
      for (auto& y : ltm_.neuron_layer_weights[first_layer_ + g].weights) {
        memcpy(&y[0], f, 4 * y.size());
        f += y.size();
      }
  }
  void ReadFromDisk(std::ifstream* s) override {  // lstm-model.cpp:70-76
    Ensure();
    SyncIfAhead();
    size_t nl = 0, ns = 0;
    Call("gmx_lstm_export", [&] { return gmx_lstm_export(h_, slot_, nullptr, &nl, nullptr, &ns); });
    short_in_.resize(ns);
    s->read(short_in_.data(), ns);
    if (ns >= 12 + 1024) {
      memcpy(&top_, short_in_.data(), 4);
      memcpy(&mid_, short_in_.data() + 4, 4);
      memcpy(&bot_, short_in_.data() + 8, 4);
      memcpy(&probs_[0], short_in_.data() + 12, 1024);
    }
    range_on_device_ = false;
    import_pending_ = true;  // the weights follow when LongTermMemory::ReadFromDisk has run
  }
  void Copy(const MemoryInterface* m) override {  // lstm-model.cpp:78-85
    GpuLstmModel* orig = const_cast<GpuLstmModel*>(static_cast<const GpuLstmModel*>(m));
    orig->Settle();
    orig->SyncIfAhead();
    Ensure();
    SyncIfAhead();
    import_pending_ = false;
    if (orig->pool_ == pool_) {
      Call("gmx_lstm_copy", [&] { return gmx_lstm_copy(h_, slot_, orig->h_, orig->slot_); });
    } else {
      MixerPool* a = pool_.get() < orig->pool_.get() ? pool_.get() : orig->pool_.get();
      MixerPool* b = pool_.get() < orig->pool_.get() ? orig->pool_.get() : pool_.get();
      int rc;
      {
        std::lock_guard<std::mutex> la(a->mu_);
        std::lock_guard<std::mutex> lb(b->mu_);
        rc = gmx_lstm_copy(h_, slot_, orig->h_, orig->slot_);
      }
      pool_->Check("gmx::GpuLstmModel", "gmx_lstm_copy", rc);
    }
    top_ = orig->top_;
    mid_ = orig->mid_;
    bot_ = orig->bot_;
    probs_ = orig->probs_;
    range_on_device_ = orig->range_on_device_;
  }
  unsigned long long GetMemoryUsage(const ShortTermMemory&, const LongTermMemory&) override {
    Ensure();
    uint64_t v = 0;
    Call("gmx_lstm_memory_usage", [&] { return gmx_lstm_memory_usage(h_, &v); });
    return v;
  }
  int prediction_index() const { return prediction_index_; }

 private:
  friend class GpuMixerBank;
  MixerPool::Stream& st() { return pool_->streams_[slot_]; }
  template <class F>
  void Call(const char* what, F f) {
    pool_->Call("gmx::GpuLstmModel", what, f);
  }
  void SyncIfAhead();
  std::vector<char> LongBytes() const {  // the LSTM section as LongTermMemory writes it
    std::vector<char> l;
    auto put = [&l](const std::valarray<float>& y) {
      const char* c = reinterpret_cast<const char*>(&y[0]);
      l.insert(l.end(), c, c + 4 * y.size());
    };
    for (auto& x : ltm_.lstm_output_layer)
      for (auto& y : x) put(y);
    for (int g = 0; g < 3; ++g)
      for (auto& y : ltm_.neuron_layer_weights[first_layer_ + g].weights) put(y);
    return l;
  }
  void Ensure() {
    if (h_) return;
    Call("gmx_lstm_create", [&] { return pool_->EnsureLstm(); });
    h_ = pool_->lstm();
    std::vector<float> w((size_t)3 * kCells * kInputs);
    for (int g = 0; g < 3; ++g)
      for (int i = 0; i < kCells; ++i)
        memcpy(&w[((size_t)g * kCells + i) * kInputs], &ltm_.neuron_layer_weights[first_layer_ + g].weights[i][0],
               4 * kInputs);
    Call("gmx_lstm_set_weights", [&] { return gmx_lstm_set_weights(h_, slot_, w.data()); });
  }
  void Settle() {
    Ensure();
    if (!import_pending_) return;
    import_pending_ = false;
    std::vector<char> l = LongBytes();
    Call("gmx_lstm_import",
         [&] { return gmx_lstm_import(h_, slot_, l.data(), l.size(), short_in_.data(), short_in_.size()); });
  }

  ShortTermMemory& stm_;
  LongTermMemory& ltm_;
  std::shared_ptr<MixerPool> pool_;
  int slot_ = 0;
  gmx_lstm* h_ = nullptr;
  int first_layer_ = 0;
  int top_, mid_, bot_, prediction_index_;
  std::valarray<float> probs_;
  std::vector<char> short_in_;
  bool import_pending_ = false;
  bool range_on_device_ = false;  // top_/mid_/bot_/probs_ were last advanced by the device (run-ahead)
};

// ---- the parts of the mixers' bank that know the other two ---------------------------------------------------
inline void GpuLstmModel::SyncIfAhead() {
  MixerPool::Stream& s = st();
  if (s.ra && s.mixers) s.mixers->SyncRunAhead();
}

inline std::vector<int> GpuMixerBank::IndirectSlots() const {
  std::vector<int> v;
  const GpuIndirectBank* ib = pool_->streams_[slot_].indirect;
  if (ib)
    for (auto& d : ib->descs_) {
      v.push_back(d.slot_indirect);
      v.push_back(d.slot_run_map);
    }
  return v;
}
inline int GpuMixerBank::LstmSlot() const {
  const GpuLstmModel* l = pool_->streams_[slot_].lstm;
  return l ? l->prediction_index() : -1;
}

inline int GpuMixerBank::BeginRunAhead(RunAheadSink* sink, uint64_t chunk_bits) {
  if (st().ra) return GMX_ERR_STATE;
  Settle();
  MixerPool::Stream& s = st();
  int parts = MixerPool::kMixers, lstm_slot = -1, mixer_ctx_col = -1, ind_ctx_col = -1;
  if (s.indirect) {
    s.indirect->Settle();
    s.indirect->pending_ = false;
    s.indirect->SlotsToDevice(stm_);
    parts |= MixerPool::kIndirect;
  }
  if (s.lstm) {
    // the run-ahead LSTM works on whole bytes, and hands over what hangs on ShortTermMemory::lstm_prediction_context:
    // a gate context of the mixers (predictor.cpp:321) and the context of one Indirect model (predictor.cpp:117-119)
    // (between a byte's last Learn and the next Predict recent_bits still holds the byte's first seven bits,
    // basic-contexts.cpp:27-33: the blackboard runs a bit behind)
    if (!(stm_.recent_bits >= 128 || (stm_.recent_bits == 1 && stm_.bits_seen == 0))) return GMX_ERR_STATE;
    s.lstm->Settle();
    parts |= MixerPool::kLstm;
    lstm_slot = s.lstm->prediction_index();
    for (size_t j = 0; j < mixers_.size(); ++j)
      if (mixers_[j]->context_address() == &stm_.lstm_prediction_context) mixer_ctx_col = (int)j;
    if (s.indirect)
      for (size_t i = 0; i < s.indirect->models_.size(); ++i)
        if (s.indirect->models_[i]->context_address() == &stm_.lstm_prediction_context) ind_ctx_col = (int)i;
  }
  if (status()) return status();
  lstm_ctx_col_ = mixer_ctx_col;
  // rows this bank has seen (Mixer::contexts_seen_ / GetMemoryUsage while the device runs behind), when the sink
  // will ask: 33 modulos and bitmap words per bit otherwise spent for nothing
  track_seen_ = sink && sink->WantsMemoryUsage();
  seen_.assign(descs_.size(), std::vector<uint64_t>());
  seen_count_.assign(descs_.size(), 0);
  last_ctx_.assign(descs_.size(), 0);
  first_mark_ = true;
  for (size_t j = 0; track_seen_ && j < descs_.size(); ++j) seen_[j].assign(((size_t)descs_[j].table_size + 63) / 64, 0);
  if (track_seen_ && ever_ran_) {
    Stage();
    for (size_t j = 0; j < descs_.size(); ++j) {
      auto& table = ltm_.mixers[memory_index_[j]].mixer_table;
      for (size_t c = 0; c < table.size(); ++c)
        if (table[c]) {
          seen_[j][c >> 6] |= 1ull << (c & 63);
          ++seen_count_[j];
        }
    }
    Unstage();
  }
  int rc = pool_->Join(slot_, chunk_bits, parts, lstm_slot, mixer_ctx_col, ind_ctx_col, sink && sink->WantsModels(),
                       sink && sink->WantsAllOutputs());
  if (rc) return rc;
  T_ = pool_->chunk_bits();
  n_pad_ = pool_->n_pad_;
  mask_words_ = pool_->mask_words_;
  recorded_ = false;
  sink_ = sink;
  ever_ran_ = true;
  return GMX_OK;
}

inline int GpuMixerBank::BeginLockstep() {
  if (st().ra || st().ls) return GMX_ERR_STATE;
  Settle();
  MixerPool::Stream& s = st();
  int parts = MixerPool::kMixers, lstm_slot = -1, mixer_ctx_col = -1, ind_ctx_col = -1;
  if (s.indirect) {
    s.indirect->Settle();
    s.indirect->pending_ = false;
    s.indirect->SlotsToDevice(stm_);
    parts |= MixerPool::kIndirect;
  }
  if (s.lstm) {
    if (!(stm_.recent_bits >= 128 || (stm_.recent_bits == 1 && stm_.bits_seen == 0))) return GMX_ERR_STATE;  // (cf. BeginRunAhead)
    s.lstm->Settle();
    parts |= MixerPool::kLstm;
    lstm_slot = s.lstm->prediction_index();
    for (size_t j = 0; j < mixers_.size(); ++j)
      if (mixers_[j]->context_address() == &stm_.lstm_prediction_context) mixer_ctx_col = (int)j;
    if (s.indirect)
      for (size_t i = 0; i < s.indirect->models_.size(); ++i)
        if (s.indirect->models_[i]->context_address() == &stm_.lstm_prediction_context) ind_ctx_col = (int)i;
  }
  if (status()) return status();
  ls_predicted_ = false;
  ever_ran_ = true;
  return pool_->JoinLockstep(slot_, parts, lstm_slot, mixer_ctx_col, ind_ctx_col);
}

inline void GpuMixerBank::FinishLockstep() {
  if (!st().ls) return;
  if (pool_->ls_what_[slot_] & GMX_STEP_LEARN) pool_->LockstepWait(slot_);  // the last bit's Learn: a step without a Predict
  pool_->LeaveLockstep(slot_);
}

inline int GpuMixerBank::EndLockstep() {
  if (!st().ls) return GMX_OK;
  FinishLockstep();
  return SlotsHome();
}

inline int GpuMixerBank::SlotsHome() {
  GpuIndirectBank* ib = st().indirect;
  if (!ib) return status();
  ib->SlotsFromDevice(stm_);
  if (sink_ && sink_->SilentSlotsAreZero() && last_ind_active_.size() == 2 * ib->descs_.size())
    for (size_t i = 0; i < ib->descs_.size(); ++i) {
      if (!last_ind_active_[2 * i]) stm_.predictions[ib->descs_[i].slot_indirect] = 0;  // (silent, or a zero logit)
      if (!last_ind_active_[2 * i + 1]) stm_.predictions[ib->descs_[i].slot_run_map] = 0;
    }
  return status();
}

inline int GpuMixerBank::Flush() {
  RunAheadView v;
  int rc = pool_->Arrive(slot_, st().t, &v);
  if (rc) return rc;
  if (v.n) {
    if (sink_) sink_->Drain(v);
    // mixer.cpp:99-105: where the Mixer::Predict calls of the newest bit left their results
    const int M = v.n_mixers;
    const float* o = v.last_outputs;
    size_t j = 0;
    for (int k = 0; k < stm_.num_layer0_mixers; ++k) stm_.mixer_layer0_outputs[k] = o[j++];
    for (int k = 0; k < stm_.num_layer1_mixers; ++k) stm_.mixer_layer1_outputs[k] = o[j++];
    if (j < (size_t)M) stm_.final_mixer_output = o[j];
    if (v.ind_active) last_ind_active_.assign(v.ind_active + (v.n - 1) * 2 * (size_t)v.n_ind, v.ind_active + v.n * 2 * (size_t)v.n_ind);
    if (st().lstm) {
      st().lstm->range_on_device_ = true;
      // lstm-model.cpp:25-33: what the newest byte's LstmModel::Predict left on the blackboard
      if (v.lstm_context && v.n >= 8) stm_.lstm_prediction_context = v.lstm_context[v.n / 8 - 1];
      if (v.lstm_pred)  // (a silent bit repeats the slot -- or leaves the zero Predictor::Predict put there)
        stm_.predictions[st().lstm->prediction_index()] =
            (sink_ && sink_->SilentSlotsAreZero() && !v.lstm_active[v.n - 1]) ? 0.0f : v.lstm_pred[v.n - 1];
    }
  }
  return GMX_OK;
}

inline void GpuMixerBank::PredictAll(ShortTermMemory& stm) {
  Settle();
  if (status()) return;
  MixerPool::Stream& s = st();
  if (s.ls) {
    // Mixer::Predict x 33 in lock step: the raw blackboard, active_models as a mask and the contexts read now go into
    // the step's records; then every other decoder of the pool gets its turn, the step runs, and the outputs are here
    if (ls_predicted_) {
      Check("lock step needs Predict -> Perceive -> Learn for every bit", GMX_ERR_STATE);
      return;
    }
    gmx_chainstep* cs = pool_->cs_;
    const int N = stm.num_predictions;
    memcpy(gmx_chainstep_predictions(cs) + (size_t)slot_ * pool_->n_pad_, &stm.predictions[0], 4 * (size_t)N);
    uint32_t* m = gmx_chainstep_active_mask(cs) + (size_t)slot_ * pool_->mask_words_;
    for (int w = 0; w < pool_->mask_words_; ++w) m[w] = 0;
    for (int idx : stm.active_models) m[idx >> 5] |= 1u << (idx & 31);
    uint32_t* c = gmx_chainstep_contexts(cs) + (size_t)slot_ * mixers_.size();
    for (size_t j = 0; j < mixers_.size(); ++j) c[j] = mixers_[j]->context();
    pool_->ls_what_[slot_] |= GMX_STEP_PREDICT;
    ls_predicted_ = true;
    ever_ran_ = true;
    pool_->LockstepWait(slot_);
    if (status()) return;
    const float* o = gmx_chainstep_outputs(cs) + (size_t)slot_ * mixers_.size();
    size_t j = 0;
    for (int k = 0; k < stm.num_layer0_mixers; ++k) stm.mixer_layer0_outputs[k] = o[j++];
    for (int k = 0; k < stm.num_layer1_mixers; ++k) stm.mixer_layer1_outputs[k] = o[j++];
    if (j < mixers_.size()) stm.final_mixer_output = o[j];
    return;
  }
  if (s.ra) {
    // Mixer::Predict x 33, recorded: the raw blackboard, active_models as a mask, the contexts read now
    if (recorded_) {  // a Predict whose bit was never learned: not a path that runs ahead
      Check("run-ahead needs Predict -> Perceive -> Learn for every bit", GMX_ERR_STATE);
      return;
    }
    const int N = stm.num_predictions;
    memcpy(s.pred + (size_t)s.t * n_pad_, &stm.predictions[0], 4 * (size_t)N);
    uint32_t* m = s.mask + (size_t)s.t * mask_words_;
    for (int w = 0; w < mask_words_; ++w) m[w] = 0;
    for (int idx : stm.active_models) m[idx >> 5] |= 1u << (idx & 31);
    uint32_t* c = s.ctx + (size_t)s.t * mixers_.size();
    for (size_t j = 0; j < mixers_.size(); ++j) c[j] = mixers_[j]->context();
    recorded_ = true;
    return;
  }
  for (size_t j = 0; j < mixers_.size(); ++j) contexts_[j] = mixers_[j]->context();  // read at call time
  static_assert(sizeof(int) == sizeof(int32_t), "active_models is passed as it stands");
  float p = 0.5f;
  GpuIndirectBank* ind = (s.indirect && s.indirect->pending_) ? s.indirect : nullptr;
  ever_ran_ = true;
  if (ind) {
    // the Indirect models' Predict of this bit is still to run: both banks in one round trip
    chain_pred_.resize(128);  // two slots for each of at most 64 models (gmx_indirect_create's limit)
    chain_act_.resize(128);
    Call("gmx_chain_forward", [&] {
      return gmx_chain_forward(ind->h_, group_, slot_, ind->contexts_.data(), ind->pending_bit_context_,
                               &stm.predictions[0], stm.active_models.data(), (int)stm.active_models.size(),
                               contexts_.data(), &p, outputs_.data(), chain_pred_.data(), chain_act_.data());
    });
    ind->Deliver(stm, chain_pred_.data(), chain_act_.data());
  } else {
    Call("gmx_bank_forward", [&] {
      return gmx_bank_forward(group_, slot_, &stm.predictions[0], stm.active_models.data(),
                              (int)stm.active_models.size(), contexts_.data(), &p, outputs_.data());
    });
  }
  // mixer.cpp:99-105: where each Mixer::Predict leaves its result
  size_t j = 0;
  for (int k = 0; k < stm.num_layer0_mixers; ++k) stm.mixer_layer0_outputs[k] = outputs_[j++];
  for (int k = 0; k < stm.num_layer1_mixers; ++k) stm.mixer_layer1_outputs[k] = outputs_[j++];
  if (j < outputs_.size()) stm.final_mixer_output = outputs_[j];
}

}  // namespace gmx

#endif  // GMX_MODEL_ADAPTER_H_
