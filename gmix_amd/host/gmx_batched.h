// gmx_batched.h -- the run-ahead compressor: runner_utils::Compress (src/runner/runner-utils.cpp:43-67) with the
// 33 mixers of the Predictor batched on an MI355X.
//
// The reference codes one bit at a time: Predict -> Encode -> Perceive -> Learn, all 121 models in turn.  But in
// compression every bit is known beforehand, and no feature model ever reads a mixer output
// (predictor.cpp:360-387), so the loop splits:
//
//     host, bits t .. t+T      Predictor::Predict / Perceive / Learn as in the reference -- the 88 feature models
//                              run, the mixers (gmx::GpuMixer in run-ahead mode, gmx_model_adapter.h) only record
//                              {predictions, active_models, 33 contexts, bit} into the pinned arrays of a gmx_batch
//     device, bits t-T .. t    gmx_batch_upload / gmx_group_run / gmx_batch_download on the chunk recorded before
//                              (a ring of four batches, MixerPool::kRing: BASELINE configs[3]'s double-buffered
//                              batches with one more in flight) -- and, when the Predictor's Indirect models and LSTM
//                              are gmx::GpuIndirect / gmx::GpuLstmModel too, gmx_lstm_run -> gmx_lstm_feed ->
//                              gmx_indirect_run(into) in front of it, each bank on a stream of its own
//     host, bits t-2T .. t-T   Encoder::Encode (coder/encoder.cpp:10-25) drains the probabilities that came back
//
// Same Predictor, same feature models, same coder, same bytes as `gmix -c` -- tests/test_gpu_batched.py compares
// them with the stock build's.  Compiled against the reference like gmx_model_adapter.h (-I<gmix>/src); the switch
// in the reference is one call in RunCompression (runner-utils.cpp:118):
//     -  Compress(*input_bytes, &data_in, &data_out, output_bytes, &p);
//     +  gmx::BatchedCompress(*input_bytes, &data_in, &data_out, output_bytes, &p);
// dropin/Makefile builds the reference's CLI that way (gmix_batched).
//
// Decompression cannot run ahead (the decoder learns each bit from Predict's own result, decoder.cpp:19-39): it
// keeps the per-bit path of gmx_model_adapter.h.
//
// Analysis (runner-utils.cpp:47 switches it on for every compression): the Predictor keeps its per-bit
// entropy averages of the feature models; the averages of the mixers it analyses (the final mixer,
// predictor.cpp:354-357) need outputs that are a chunk away, so this file computes them as
// Predictor::UpdateEntropy does (predictor.cpp:439-469) when the chunk returns, and writes the rows of
// analysis/entropy.tsv / memory.tsv (predictor.cpp:471-504) itself, each from the values captured at its
// sample bit.  The tables come out identical to the stock build's.
//
// Many files: BatchedCompressFiles runs one Predictor per file on a thread of its own, all mixers in ONE gmx_group
// (gmx::MixerPool), one launch per chunk for all files.
#ifndef GMX_BATCHED_H_
#define GMX_BATCHED_H_

#include <sched.h>
#include <dirent.h>
#include <sys/mman.h>

#include <chrono>
#include <climits>
#include <cmath>
#include <deque>
#include <filesystem>
#include <functional>
#include <iomanip>
#include <thread>

#include "coder/decoder.h"          // the reference's
#include "coder/encoder.h"          // the reference's
#include "gmx_model_adapter.h"
#include "predictor.h"              // the reference's
#include "runner/runner-utils.h"    // the reference's (WriteHeader)

namespace gmx {

struct BatchedOptions {
  uint64_t chunk_bits = 2048;  // bits per stream and launch (a multiple of 8: chunks end on byte boundaries)
  bool analysis = true;        // runner-utils.cpp:47
  bool progress = true;        // runner-utils.cpp:59-63
  int device = -1;             // BatchedCompressFiles: the pool's device (-1: $GMX_DEVICE, else 0) -- one process per GPU
  bool pin_threads = true;     // BatchedCompressFiles: threads onto the cores of the device's NUMA node
  int groups = 1;              // BatchedDecompressFiles: the files go in this many pools, each a lock step of its own, taken in
                               // turn by the same worker threads (one pool's device step beside the other's host turn).
                               // A wash, so one by default: a step's device time hardly depends on its stream count (two
                               // launches of latency chains, 21 us at 64 streams, 27 at 256), so two pools are twice the
                               // device time per bit of all files for the host time they hide -- 256 files: 3.0-3.4e6
                               // bits/s either way (profiles/r04_exp_decode_pools.txt).  Kept as an option (tested).
  bool destroy_predictors = true;  // BatchedCompressFiles / BatchedDecompressFiles: false = leave the Predictors standing when
                               // the call returns -- for a command-line driver that exits next: 64 destructors give back
                               // 64 x 2 GB of address space page table by page table (1.8 s for 64 files), the kernel
                               // reclaims the same in one sweep at exit
  int max_cpus = -1;           // ... onto at most this many of them.  -1: twice what the container's CPU quota is worth
                               // (threads spread over every core of the node spend the quota in a burst and are then
                               // stopped together for the rest of each scheduler period, the device idle meanwhile; a
                               // thread here waits for its round about half of the time, so twice the quota's cores
                               // keep inside it -- profiles/r03_exp_cpus.txt); 0: all of the node's
};

// The cores of the NUMA node the device hangs on (sysfs), or nothing when that cannot be told.
inline std::vector<int> DeviceNodeCpus(int device) {
  std::vector<int> cpus;
  char bus[64] = {0};
  if (gmx_device_pci_bus_id(device, bus, sizeof bus) != GMX_OK) return cpus;
  std::string id(bus);
  for (auto& ch : id) ch = (char)tolower(ch);
  std::ifstream nf("/sys/bus/pci/devices/" + id + "/numa_node");
  int node = -1;
  if (!(nf >> node) || node < 0) return cpus;
  std::ifstream lf("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
  std::string list;
  if (!std::getline(lf, list)) return cpus;
  size_t pos = 0;
  while (pos < list.size()) {  // "0-15,128-143"
    size_t end = list.find(',', pos);
    if (end == std::string::npos) end = list.size();
    const std::string part = list.substr(pos, end - pos);
    const size_t dash = part.find('-');
    const int a = atoi(part.c_str());
    const int b = dash == std::string::npos ? a : atoi(part.c_str() + dash + 1);
    for (int c = a; c <= b; ++c) cpus.push_back(c);
    pos = end + 1;
  }
  return cpus;
}

// Where the device stands among the GPUs of its NUMA node (*rank of *count; both from sysfs: every AMD display /
// accelerator function with that numa_node, in bus order -- also the ones this process cannot open).  A host with
// several GPUs per node has a process per GPU, often a container each that calls its own "device 0": the rank is what
// tells their core windows apart.  false when sysfs does not say.
inline bool DeviceRankOnNode(int device, int* rank, int* count) {
  char bus[64] = {0};
  if (gmx_device_pci_bus_id(device, bus, sizeof bus) != GMX_OK) return false;
  std::string id(bus);
  for (auto& ch : id) ch = (char)tolower(ch);
  int node = -1;
  {
    std::ifstream nf("/sys/bus/pci/devices/" + id + "/numa_node");
    if (!(nf >> node) || node < 0) return false;
  }
  std::vector<std::string> gpus;
  if (DIR* d = opendir("/sys/bus/pci/devices")) {
    while (struct dirent* e = readdir(d)) {
      if (e->d_name[0] == '.') continue;
      const std::string base = std::string("/sys/bus/pci/devices/") + e->d_name;
      std::string vendor, cls;
      int n = -1;
      {
        std::ifstream f(base + "/vendor");
        f >> vendor;
      }
      if (vendor != "0x1002") continue;
      {
        std::ifstream f(base + "/class");
        f >> cls;  // 0x0300xx VGA, 0x0302xx 3D, 0x0380xx display, 0x1200xx processing accelerator
      }
      if (cls.compare(0, 4, "0x03") != 0 && cls.compare(0, 6, "0x1200") != 0) continue;
      {
        std::ifstream f(base + "/numa_node");
        f >> n;
      }
      if (n == node) gpus.push_back(e->d_name);
    }
    closedir(d);
  }
  std::sort(gpus.begin(), gpus.end());
  for (size_t i = 0; i < gpus.size(); ++i) {
    std::string g = gpus[i];
    for (auto& ch : g) ch = (char)tolower(ch);
    if (g == id) {
      *rank = (int)i;
      *count = (int)gpus.size();
      return true;
    }
  }
  return false;
}

// What the container's CPU quota is worth in cores (cgroup v2 cpu.max, else v1), rounded up; 0: no quota / not known.
inline int QuotaCpus() {
  double quota = 0, period = 0;
  {
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string q;
    if (f >> q >> period && q != "max") quota = atof(q.c_str());
  }
  if (quota <= 0) {
    std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (!(fq >> quota) || !(fp >> period)) return 0;
  }
  if (quota <= 0 || period <= 0) return 0;
  return (int)((quota + period - 1) / period);
}

// Keeps the calling thread on those cores of the device's node that it is allowed on anyway (a container's share) -- on
// max_cpus of them when that is given: a window of the node's list (which names one hardware thread of every core
// before the second) that starts device * max_cpus in, so that processes driving other devices of the same node take
// other cores.  No-op when sysfs does not say.
inline bool PinThreadToDeviceNode(int device, int max_cpus = 0, int window = -1) {
  const std::vector<int> cpus = DeviceNodeCpus(device);
  if (cpus.empty()) return false;
  cpu_set_t now, want;
  CPU_ZERO(&now);
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof now, &now) != 0) return false;
  std::vector<int> allowed;
  for (int c : cpus)
    if (c < CPU_SETSIZE && CPU_ISSET(c, &now)) allowed.push_back(c);
  if (allowed.empty()) return false;
  const size_t n = allowed.size(), k = (max_cpus <= 0 || (size_t)max_cpus > n) ? n : (size_t)max_cpus;
  // (window >= 0: the window-th stretch of k cores instead of the device's)  The device's own stretch starts where its
  // share of the node does -- the node's cores divided among the node's GPUs: four containers with a GPU each on one
  // node would all take "device 0"'s first cores otherwise, and their spinning workers each other's time slices.
  size_t first = ((size_t)(window >= 0 ? window : (device < 0 ? 0 : device)) * k) % n;
  int rank = 0, count = 0;
  if (window < 0 && DeviceRankOnNode(device, &rank, &count) && count > 0) {
    size_t cores = n;  // (the list names one hardware thread of every core before the second: divide the first ones)
    int smt = 0;
    std::ifstream f("/sys/devices/system/cpu/smt/active");
    if ((f >> smt) && smt == 1 && n >= 2) cores = n / 2;
    first = (size_t)rank * (cores / (size_t)count) % n;
  }
  for (size_t i = 0; i < k; ++i) CPU_SET(allowed[(first + i) % n], &want);
  return sched_setaffinity(0, sizeof want, &want) == 0;
}

// Keeps the calling thread on the device's SHARE of its node: the node's cores divided among the node's GPUs (both
// hardware threads of each) -- 16 cores = two L3 domains on an 8-GPU MI355X host.  For threads that spin (the lock-step
// decoder's workers): fewer of them than cores in the share, so that whoever else wakes up there -- the runtime's own
// threads, a neighbour's -- finds an idle sibling instead of taking a worker off its core; and close together, so that the
// step's barrier stays inside two L3s.  The whole node where sysfs does not say how many GPUs share it.
inline bool PinThreadToDeviceShare(int device) {
  const std::vector<int> cpus = DeviceNodeCpus(device);
  if (cpus.empty()) return false;
  cpu_set_t now, want;
  CPU_ZERO(&now);
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof now, &now) != 0) return false;
  std::vector<int> allowed;
  for (int c : cpus)
    if (c < CPU_SETSIZE && CPU_ISSET(c, &now)) allowed.push_back(c);
  if (allowed.empty()) return false;
  const size_t n = allowed.size();
  int rank = 0, count = 0, smt = 0;
  {
    std::ifstream f("/sys/devices/system/cpu/smt/active");
    if (!(f >> smt)) smt = 0;
  }
  const size_t cores = (smt == 1 && n >= 2 && n % 2 == 0) ? n / 2 : n;
  if (!DeviceRankOnNode(device, &rank, &count) || count <= 0 || cores / (size_t)count == 0) {
    for (int c : allowed) CPU_SET(c, &want);
  } else {
    const size_t share = cores / (size_t)count, first = (size_t)rank * share;
    for (size_t i = 0; i < share; ++i) {
      CPU_SET(allowed[first + i], &want);
      if (cores < n) CPU_SET(allowed[cores + first + i], &want);  // (the list names the second hardware threads behind the first)
    }
  }
  return sched_setaffinity(0, sizeof want, &want) == 0;
}

// One stream's compression: the loop of runner_utils::Compress, the coder behind the device.
class BatchedCompressor : public RunAheadSink {
 public:
  BatchedCompressor(Predictor* p, std::ofstream* os, const BatchedOptions& opt)
      : p_(p), enc_(os), opt_(opt), bank_(GpuMixerBank::Of(p, sizeof(Predictor))) {}

  // 0, or a gmx_status (nothing is coded on the CPU instead: a Predictor whose mixers are not gmx::GpuMixer is
  // refused).  *output_bytes as runner_utils::Compress leaves it.
  int Run(unsigned long long input_bytes, std::ifstream* is, std::ofstream* os, unsigned long long* output_bytes) {
    int rc = Begin(input_bytes);
    return rc ? rc : Code(input_bytes, is, os, output_bytes);
  }
  // The two halves of Run.  Several compressors that share a pool all Begin before any of them Codes: a stream
  // that joined the pool's ring late would hand in its chunks out of step with the others for the whole run, and
  // every round would fall apart into launches of a few streams each.
  int Begin(unsigned long long input_bytes) {
    if (!bank_) {
      fprintf(stderr, "gmx::BatchedCompress: this Predictor's mixers are not gmx::GpuMixer\n");
      return GMX_ERR_INVALID;
    }
    ShortTermMemory& stm = bank_->stm();
    const int sample_frequency = (int)(8 * input_bytes / 1000);  // runner-utils.cpp:47 (the call narrows to int)
    sample_frequency_ = sample_frequency;
    if (opt_.analysis) {
      p_->EnableAnalysis(sample_frequency);
      F_ = sample_frequency > 0 ? sample_frequency : 0;
    }
    if (F_ > 0) {
      // the Predictor keeps averaging per bit but leaves the rows to this object (they need the device's outputs)
      p_->SetAnalysisFrequency(kNever);
      const int N = stm.num_predictions;
      const std::vector<int> ind_slots = bank_->IndirectSlots();
      const int lstm_slot = bank_->LstmSlot();
      for (int i = 0; i < (int)stm.model_enable_analysis.size(); ++i) {
        if (!stm.model_enable_analysis[i]) continue;
        analysed_.push_back(i);
        Source src;
        src.column = (int)analysed_.size() - 1;
        if (i >= N) {
          src.kind = kMixer;
          src.index = i - N;
          // (the final mixer's average follows from the probability itself, below; any other mixer's needs its output)
          if (i != (int)stm.model_enable_analysis.size() - 1) wants_all_outputs_ = true;
        } else if (i == lstm_slot) {
          src.kind = kLstm;
          wants_models_ = true;
        } else {
          auto it = std::find(ind_slots.begin(), ind_slots.end(), i);
          if (it == ind_slots.end()) continue;  // a feature model on the host: the Predictor's own average is right
          src.kind = kIndirect;
          src.index = (int)(it - ind_slots.begin());
          wants_models_ = true;
        }
        src.ema = stm.entropy[i];
        on_device_.push_back(src);
      }
    }
    uint64_t chunk = opt_.chunk_bits < 8 ? 8 : opt_.chunk_bits & ~7ull;
    return bank_->BeginRunAhead(this, chunk);
  }
  int Code(unsigned long long input_bytes, std::ifstream* is, std::ofstream* os, unsigned long long* output_bytes) {
    const unsigned long long percent = 1 + (input_bytes / 10000);
    if (opt_.progress) {
      fprintf(stderr, "\r                     \r");
      fflush(stderr);
    }
    for (unsigned long long pos = 0; pos < input_bytes && bank_->status() == 0; ++pos) {
      CodeByte(is->get());
      if (opt_.progress && pos % percent == 0) {
        fprintf(stderr, "\rprogress: %.2f%%", 100.0 * pos / input_bytes);
        fflush(stderr);
      }
    }
    return Finish(os, output_bytes);
  }
  // Code in steps, for a caller with a loop of its own: eight bits of one byte ...
  void CodeByte(int c) {
    ShortTermMemory& stm = bank_->stm();
    for (int j = 7; j >= 0; --j) {
      const int bit = (c >> j) & 1;
      p_->Predict();   // the feature models predict; the mixers record their inputs
      if (F_ > 0 && stm.bits_seen > 0 && stm.bits_seen % (unsigned long long)kNever == 0) {
        p_->SetAnalysisFrequency(kNever - 1);  // (never a row of the Predictor's own)
        p_->Perceive(bit);
        p_->SetAnalysisFrequency(kNever);
      } else {
        p_->Perceive(bit);
      }
      if (F_ > 0 && stm.bits_seen % (unsigned long long)F_ == 0 && stm.bits_seen > 0) Capture();
      ++recorded_;
      p_->Learn();     // the feature models learn; the mixers record the bit (a full chunk goes to the device)
    }
  }
  // ... every bit handed in so far coded and every result home, run-ahead mode kept (the Predictor can then be copied
  // or written out, and what was asked to be summed is whole) ...
  int Sync() {
    int rc = bank_->SyncRunAhead();
    return rc ? rc : bank_->status();
  }
  // ... and the end: run-ahead mode left, the coder flushed
  int Finish(std::ofstream* os, unsigned long long* output_bytes) {
    ShortTermMemory& stm = bank_->stm();
    int rc = bank_->EndRunAhead();
    if (rc == GMX_OK) rc = bank_->status();
    if (F_ > 0) {
      for (const Source& src : on_device_) stm.entropy[analysed_[src.column]] = src.ema;
      p_->SetAnalysisFrequency(sample_frequency_);
    }
    if (rc) return rc;
    enc_.Flush();
    *output_bytes = os->tellp();
    return GMX_OK;
  }
  // RunAheadSink: a chunk is back
  bool WantsModels() const override { return wants_models_ || F_ > 0; }  // (with analysis on at least the last bit's flags)
  bool SilentSlotsAreZero() const override { return F_ > 0; }
  bool WantsAllOutputs() const override { return wants_all_outputs_; }
  bool WantsMemoryUsage() const override { return F_ > 0; }  // the rows of analysis/memory.tsv (predictor.cpp:471-504)
  void Drain(const RunAheadView& v) override {
    for (uint64_t i = 0; i < v.n; ++i) {
      enc_.Encode(v.bits[i], v.p[i]);
      if (F_ > 0) {
        for (Source& src : on_device_) {  // Predictor::UpdateEntropy (predictor.cpp:439-469) on what the device produced
          float x;
          if (src.kind == kMixer && !v.outputs) {
            // The final mixer, from the probability: Predictor::Predict clamped Logistic(final_mixer_output) to
            // [1e-4, 1 - 1e-4] (predictor.cpp:369-375), UpdateEntropy clamps the same Logistic to [0.01, 0.99]
            // (predictor.cpp:453-457) -- the tighter clamp of the wider one is the tighter clamp.
            src.ema = AverageOfProb(src.ema, v.p[i], v.bits[i]);
            continue;
          }
          if (src.kind == kMixer) {
            x = v.outputs[i * v.n_mixers + src.index];
          } else if (src.kind == kIndirect) {
            // (a model that stayed silent left the zero Predictor::Predict had put there, predictor.cpp:362-365)
            const size_t q = i * 2 * (size_t)v.n_ind + src.index;
            x = v.ind_active[q] ? v.ind_pred[q] : 0.0f;
          } else {
            x = v.lstm_active[i] ? v.lstm_pred[i] : 0.0f;  // [byte][8] is bit order
          }
          src.ema = Average(src.ema, x, v.bits[i]);
        }
        if (!rows_.empty() && rows_.front().bit == drained_) {
          WriteRow(rows_.front());
          rows_.pop_front();
        }
      }
      ++drained_;
    }
  }

 private:
  static constexpr int kNever = INT_MAX;
  struct Row {
    uint64_t bit;  // position in this run
    unsigned long long bits_seen;
    std::vector<double> entropy;
    std::vector<unsigned long long> memory;
    size_t history;
  };
  static double Average(double e, float x, int bit) { return AverageOfProb(e, Sigmoid::Logistic(x), bit); }
  static double AverageOfProb(double e, float prob, int bit) {
    float eps = 0.01;
    if (prob < eps)
      prob = eps;
    else if (prob > 1 - eps)
      prob = 1 - eps;
    float entropy;
    if (bit)
      entropy = std::log2(prob);
    else
      entropy = std::log2(1 - prob);
    double alpha = 0.00001;
    return (1 - alpha) * e + alpha * entropy;
  }
  // What Predictor::RunAnalysis (predictor.cpp:471-504) reads at a sample bit, taken at that bit.
  void Capture() {
    ShortTermMemory& stm = bank_->stm();
    LongTermMemory& ltm = bank_->ltm();
    Row r;
    r.bit = recorded_;
    r.bits_seen = stm.bits_seen;
    for (int i : analysed_) {
      r.entropy.push_back(stm.entropy[i]);
      if (i < stm.num_predictions)
        r.memory.push_back(stm.prediction_index_to_model_ptr[i]->GetMemoryUsage(stm, ltm));
      else
        r.memory.push_back(stm.mixer_index_to_model_ptr[i - stm.num_predictions]->GetMemoryUsage(stm, ltm));
    }
    r.history = ltm.history.size();
    rows_.push_back(std::move(r));
  }
  void WriteRow(Row& r) {
    for (const Source& src : on_device_) r.entropy[src.column] = src.ema;
    std::ofstream entropy_file("analysis/entropy.tsv", std::ios::app);
    std::ofstream memory_file("analysis/memory.tsv", std::ios::app);
    entropy_file << r.bits_seen;
    memory_file << r.bits_seen;
    for (size_t a = 0; a < analysed_.size(); ++a) {
      entropy_file << std::fixed << std::setprecision(5) << "\t" << -r.entropy[a];
      memory_file << "\t" << r.memory[a];
    }
    memory_file << "\t" << r.history;
    entropy_file << std::endl;
    memory_file << std::endl;
  }

  Predictor* p_;
  Encoder enc_;
  BatchedOptions opt_;
  std::shared_ptr<GpuMixerBank> bank_;
  int F_ = 0, sample_frequency_ = 0;
  enum Kind { kMixer, kIndirect, kLstm };
  struct Source {   // an analysed entry whose values come from the device
    int column = 0;   // position in analysed_
    Kind kind = kMixer;
    int index = 0;    // mixer number / position in GpuMixerBank::IndirectSlots()
    double ema = 0;   // its average, up to the last bit that came back
  };
  std::vector<int> analysed_;       // entropy indices with analysis on, ascending (the tables' columns)
  std::vector<Source> on_device_;
  bool wants_models_ = false, wants_all_outputs_ = false;
  std::deque<Row> rows_;
  uint64_t recorded_ = 0, drained_ = 0;
};

// runner_utils::Compress (runner-utils.cpp:43-67), argument for argument; returns 0 or a gmx_status.
inline int BatchedCompress(unsigned long long input_bytes, std::ifstream* is, std::ofstream* os,
                           unsigned long long* output_bytes, Predictor* p, const BatchedOptions& opt = BatchedOptions()) {
  BatchedCompressor c(p, os, opt);
  int rc = c.Run(input_bytes, is, os, output_bytes);
  if (rc) fprintf(stderr, "\ngmx::BatchedCompress: %s\n", gmx_strerror(rc));
  return rc;
}

// ---- many files, one device group ---------------------------------------------------------------------------
struct BatchedJob {
  std::string input_path, output_path;
  unsigned long long input_bytes = 0, output_bytes = 0;
  int status = 0;        // 0 or a gmx_status / -100 for a file that would not open
  double seconds = 0;    // the coding loop alone (Predictor construction not included)
};
struct BatchedStats {
  double total_seconds = 0;    // the whole call: pool, Predictors, device banks, coding, teardown -- the like-for-like figure
                               // against runner_utils::RunCompression (runner-utils.cpp:88-121), which builds its Predictor too
  double wall_seconds = 0;     // first coding loop's start to the last one's end (every Predictor and bank standing)
  double build_seconds = 0;    // everything before that: Predictors (side by side when they draw nothing from rand(), else
                               // one after the other: predictor.cpp:18), device banks, the ring's pinned arrays
  double first_predictor_seconds = 0;  // ... of which the first Predictor, which is always built alone
  double teardown_seconds = 0; // last coding loop's end to the return
  bool parallel_construction = false;
  uint64_t launches = 0;       // chunks (steps, when decoding) of all streams queued on the device
  uint64_t bits = 0;           // bits of all streams mixed there
  int pinned_threads = 0;      // threads kept on the cores of the device's NUMA node
  int pinned_cpus = 0;         // ... on how many of them (0: all of the node's)
  double submit_seconds = 0;   // host time spent queueing chunks on the device ...
  double wait_seconds = 0;     // ... and waiting for the chunk before (all streams stand still meanwhile)
};

// What runs on the thread of one job between the Predictors' construction and their destruction.
struct ManyFilesHooks {
  // the job's Predictor stands, its thread is pinned: join the pool (0 or a gmx_status)
  std::function<int(int s, Predictor* p)> begin;
  // every job has begun: the timed loop
  std::function<int(int s, Predictor* p)> run;
  // the loop is over (the Predictor still stands)
  std::function<void(int s)> end;
};

// One Predictor and one thread per job, all Predictors in one MixerPool of jobs.size() streams: the frame shared by
// BatchedCompressFiles and BatchedDecompressFiles.  `open(s)` opens job s's files (false: the job fails with -100).
inline void RunManyFiles(std::vector<BatchedJob>& jobs, MixerPool& pool, const BatchedOptions& opt,
                         const std::function<bool(int)>& open, const ManyFilesHooks& hooks, BatchedStats* stats) {
  using clock = std::chrono::steady_clock;
  const clock::time_point tb = clock::now();
  const int S = (int)jobs.size();
  const bool trace = getenv("GMX_POOL_TRACE") != nullptr;
  pool.Install();  // (draws the LSTM's constant initial weights once, MixerPool::DrawLstmInit)
  std::mutex construct;  // Predictor::Predictor draws the LSTM's weights from rand() after srand() (predictor.cpp:18)
  std::mutex start_mu;
  std::condition_variable start_cv;
  int built = 0, ready = 0;
  bool first_built = false;
  std::atomic<int> pinned{0};
  const int max_cpus = opt.max_cpus < 0 ? 2 * QuotaCpus() : opt.max_cpus;
  clock::time_point t0 = tb, t_first = tb;
  std::vector<clock::time_point> ends(S, tb);
  std::vector<std::thread> threads;
  for (int s = 0; s < S; ++s) {
    threads.emplace_back([&, s] {
      BatchedJob& job = jobs[s];
      std::unique_ptr<Predictor> p;
      bool pinned_early = false;
      const bool opened = open(s);
      if (!opened) job.status = -100;
      // The FIRST Predictor is built alone.  If its LSTM turned out to be gmx::GpuLstmModel taking its initial
      // weights from the pool (the chain builds), no constructor draws from the process-wide rand() and the others
      // are built side by side, while the first one's thread already brings up the device banks.  If not (the host's
      // own LstmModel draws, lstm-layer.h:41), constructions stay serial and nothing else runs until every Predictor
      // stands: whatever a thread next to a constructor calls -- the HIP runtime coming up, a pinned allocation --
      // may draw from the same generator (seen: one output of several differing from `gmix -c`, always the same one).
      if (s == 0) {
        if (opened) p.reset(new Predictor());
        std::lock_guard<std::mutex> lk(start_mu);
        first_built = true;
        t_first = clock::now();
        start_cv.notify_all();
      } else {
        {
          std::unique_lock<std::mutex> lk(start_mu);
          start_cv.wait(lk, [&] { return first_built; });
        }
        if (opened) {
          if (pool.parallel_construction()) {
            // (on its cores first: the constructor touches ~120 MB, which then lie on the device's node, and 64 threads
            // let loose on every core of the host spend a container's CPU quota in one burst)
            if (opt.pin_threads && PinThreadToDeviceNode(pool.device(), max_cpus)) {
              ++pinned;
              pinned_early = true;
            }
            p.reset(new Predictor());
          } else {
            std::lock_guard<std::mutex> lk(construct);
            p.reset(new Predictor());
          }
        }
      }
      if (!pool.parallel_construction()) {
        std::unique_lock<std::mutex> lk(start_mu);
        if (++built == S)
          start_cv.notify_all();
        else
          start_cv.wait(lk, [&] { return built == S; });
      }
      if (!pinned_early && opt.pin_threads && PinThreadToDeviceNode(pool.device(), max_cpus)) ++pinned;
      bool begun = false;
      if (p) {
        const clock::time_point tb0 = clock::now();
        job.status = hooks.begin(s, p.get());  // every stream is in the pool before the first one records or codes
        begun = job.status == 0;
        if (trace && (s == 0 || s == S - 1))
          fprintf(stderr, "[gmx many] stream %d: Predictor stood at %.3f s, joined the pool at %.3f s (begin took %.3f s)\n", s,
                  std::chrono::duration<double>(tb0 - tb).count(), std::chrono::duration<double>(clock::now() - tb).count(),
                  std::chrono::duration<double>(clock::now() - tb0).count());
      }
      {  // every Predictor and every bank stands before the first loop starts
        std::unique_lock<std::mutex> lk(start_mu);
        if (++ready == S) {
          t0 = clock::now();
          start_cv.notify_all();
        } else {
          start_cv.wait(lk, [&] { return ready == S; });
        }
      }
      const clock::time_point a = clock::now();
      if (begun) job.status = hooks.run(s, p.get());
      ends[s] = clock::now();
      job.seconds = std::chrono::duration<double>(ends[s] - a).count();
      hooks.end(s);
      if (!opt.destroy_predictors) {
        p.release();
      } else if (pool.parallel_construction()) {
        p.reset();
      } else {
        std::lock_guard<std::mutex> lk(construct);  // (a Predictor gives back gigabytes: one at a time)
        p.reset();
      }
    });
  }
  for (auto& t : threads) t.join();
  pool.Uninstall();
  if (stats) {
    clock::time_point t1 = t0;
    for (auto& e : ends) t1 = std::max(t1, e);
    const clock::time_point tz = clock::now();
    stats->total_seconds = std::chrono::duration<double>(tz - tb).count();
    stats->wall_seconds = std::chrono::duration<double>(t1 - t0).count();
    stats->build_seconds = std::chrono::duration<double>(t0 - tb).count();
    stats->first_predictor_seconds = std::chrono::duration<double>(t_first - tb).count();
    stats->teardown_seconds = std::chrono::duration<double>(tz - t1).count();
    stats->parallel_construction = pool.parallel_construction();
    stats->launches = pool.rounds();
    stats->bits = pool.bits_submitted();
    stats->pinned_threads = pinned.load();
    stats->pinned_cpus = max_cpus;
    stats->submit_seconds = pool.submit_seconds();
    stats->wait_seconds = pool.wait_seconds();
  }
}

// runner_utils::RunCompression (runner-utils.cpp:88-121) for every job at once: a Predictor and a thread per
// file, their mixers in one gmx_group of jobs.size() streams.  Returns the number of jobs that failed.
inline int BatchedCompressFiles(std::vector<BatchedJob>& jobs, const BatchedOptions& opt_in = BatchedOptions(),
                                BatchedStats* stats = nullptr) {
  BatchedOptions opt = opt_in;
  opt.analysis = false;  // (analysis/*.tsv are one pair of files per process: not with several Predictors at once)
  opt.progress = false;
  const int S = (int)jobs.size();
  if (S == 0) return 0;
  MixerPool pool(S, opt.device);
  std::vector<std::ifstream> in(S);
  std::vector<std::ofstream> out(S);
  std::vector<std::unique_ptr<BatchedCompressor>> c(S);
  ManyFilesHooks hooks;
  hooks.begin = [&](int s, Predictor* p) {
    c[s].reset(new BatchedCompressor(p, &out[s], opt));
    int rc = c[s]->Begin(jobs[s].input_bytes);
    if (rc) c[s].reset();
    return rc;
  };
  hooks.run = [&](int s, Predictor*) { return c[s]->Code(jobs[s].input_bytes, &in[s], &out[s], &jobs[s].output_bytes); };
  hooks.end = [&](int s) {
    c[s].reset();
    out[s].close();
  };
  RunManyFiles(jobs, pool, opt, [&](int s) {
    BatchedJob& job = jobs[s];
    in[s].open(job.input_path, std::ios::in | std::ios::binary);
    if (!in[s].is_open()) return false;
    in[s].seekg(0, std::ios::end);
    job.input_bytes = in[s].tellg();
    in[s].seekg(0, std::ios::beg);
    out[s].open(job.output_path, std::ios::out | std::ios::binary);
    if (!out[s].is_open()) return false;
    runner_utils::WriteHeader(job.input_bytes, &out[s]);
    return true;
  }, hooks, stats);
  int failed = 0;
  for (auto& j : jobs) failed += j.status != 0;
  if (pool.status() != 0) fprintf(stderr, "gmx::BatchedCompressFiles: %s\n", pool.error().c_str());
  return failed;
}

// ---- many files restored side by side: S of the reference's Decoders in lock step ------------------------------
// Decoder::Decode (coder/decoder.cpp:19-39) calls Predictor::Predict, takes the bit from the code stream and the
// probability, then Perceive and Learn: the bit is only known when Predict has returned, so nothing runs ahead.  But S
// files can be restored TOGETHER: every Decoder runs on a fibre of its own; its Predict records the device-side models'
// inputs (gmx_model_adapter.h, lock-step mode) and hands the thread to the next fibre; when every fibre of every
// worker thread waits, ONE device step (gmx_chainstep: LSTM, Indirect models and mixers of all S streams, one hipGraph)
// produces all S probabilities, and the fibres go on -- Perceive, Learn, the next Predict.  The Decoders, the Predictors
// and the feature models on the host are the reference's own, unmodified; a stream whose file has ended sits the
// remaining steps out.
extern "C" void gmx_fiber_switch(void** save_sp, void* load_sp);
#if defined(__x86_64__)
// (callee-saved registers of the System V ABI on the old stack, stack pointers swapped, the same registers off the
// new one; a fresh fibre's stack holds six zeros and the address of its entry function)
asm(".text\n"
    ".weak gmx_fiber_switch\n"
    ".type gmx_fiber_switch,@function\n"
    "gmx_fiber_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size gmx_fiber_switch,.-gmx_fiber_switch\n");
#else
#error "gmx::LockstepRunner's fibres are written for x86-64 (the reference's hosts)"
#endif

class LockstepRunner {
 public:
  struct Fiber {
    std::function<void()> fn;
    void* sp = nullptr;
    void* sched_sp = nullptr;  // where the worker that runs it waits
    char* stack = nullptr;
    size_t stack_bytes = 0;
    bool done = false;
    int phase = 0;             // the set (pool) it belongs to
  };
  // n_workers threads; fibre i runs on worker i % n_workers.  n_phases: the fibres are in that many sets (pools), each
  // with a lock step of its own, taken in turn by the SAME workers -- while one set's device step runs, the workers do
  // the other set's host work (with one set they would spin through every step: at 256 files the host's turn and the
  // device's are about as long as each other).
  LockstepRunner(int n_workers, int n_phases = 1)
      : W_(n_workers < 1 ? 1 : n_workers), P_(n_phases < 1 ? 1 : n_phases), live_(P_), inflight_(P_, 0) {
    for (auto& l : live_) l.store(0);
  }
  ~LockstepRunner() {
    for (auto& f : fibers_)
      if (f && f->stack) munmap(f->stack, f->stack_bytes);
  }
  // Before Run: a body for fibre `index` of set `phase` (bodies run in index order within a worker).
  void Add(int index, int phase, std::function<void()> fn) {
    if ((int)fibers_.size() <= index) fibers_.resize(index + 1);
    std::unique_ptr<Fiber> f(new Fiber());
    f->fn = std::move(fn);
    f->phase = phase;
    f->stack_bytes = kStackBytes;
    void* m = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (m == MAP_FAILED) {
      fprintf(stderr, "gmx::LockstepRunner: no memory for a fibre's stack\n");
      abort();
    }
    mprotect(m, 4096, PROT_NONE);  // a guard page at the far end
    f->stack = (char*)m;
    uintptr_t top = ((uintptr_t)m + kStackBytes) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;               // (what the entry function would return to: it never does)
    *--sp = (void*)&Entry;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f->sp = sp;
    fibers_[index] = std::move(f);
  }
  // The calling thread's fibre waits for the next step (MixerPool::SetLockstepYield).
  static void Yield() {
    Fiber* f = Current();
    gmx_fiber_switch(&f->sp, f->sched_sp);
  }
  // `setup(worker)` on every worker thread first (Predictor construction), then -- all of them done -- `join(worker)`
  // (device banks, on the threads' own stacks), then the fibres, set by set, until all have ended: whenever every live
  // fibre of a set waits, ONE thread, on its own stack, calls `launch(set)` (the set's device step, queued) and then
  // `wait(next set)` (that one's answers are there), and the workers go on with the next set's fibres.  Returns the
  // steps taken.
  uint64_t Run(const std::function<void(int)>& setup, const std::function<void(int)>& join,
               const std::function<int(int)>& launch, const std::function<int(int)>& wait) {
    std::vector<std::thread> threads;
    lists_ = std::vector<List>((size_t)P_ * W_);
    for (size_t i = 0; i < fibers_.size(); ++i) {
      if (!fibers_[i]) continue;
      live_[fibers_[i]->phase].fetch_add(1);
      lists_[(size_t)fibers_[i]->phase * W_ + i % W_].f.push_back(fibers_[i].get());  // fibre i's home is worker i % W
    }
    for (int w = 0; w < W_; ++w) threads.emplace_back([&, w] { Work(w, setup, join, launch, wait); });
    for (auto& t : threads) t.join();
    return steps_;
  }
  std::chrono::steady_clock::time_point setup_done() const { return t_setup_; }
  std::chrono::steady_clock::time_point join_done() const { return t_join_; }

 private:
  static constexpr size_t kStackBytes = 1u << 20;
  static Fiber*& Current() {
    static thread_local Fiber* f = nullptr;
    return f;
  }
  static void Entry() {
    Fiber* f = Current();
    f->fn();
    f->done = true;
    gmx_fiber_switch(&f->sp, f->sched_sp);
    abort();  // (a finished fibre is never resumed)
  }
  void Resume(Fiber* f) {
    Current() = f;
    gmx_fiber_switch(&f->sched_sp, f->sp);
    Current() = nullptr;
  }
  // every worker arrives; the last one runs fn; all leave together
  template <class F>
  void Barrier(F fn) {
    const unsigned gen = gen_.load(std::memory_order_acquire);
    if (arrived_.fetch_add(1, std::memory_order_acq_rel) + 1 == W_) {
      fn();
      arrived_.store(0, std::memory_order_relaxed);
      gen_.store(gen + 1, std::memory_order_release);
    } else {
      unsigned spins = 0;
      while (gen_.load(std::memory_order_acquire) == gen) {
        if (++spins < 4096) {
          __builtin_ia32_pause();
        } else {
          std::this_thread::yield();
        }
      }
    }
  }
  void Work(int w, const std::function<void(int)>& setup, const std::function<void(int)>& join,
            const std::function<int(int)>& launch, const std::function<int(int)>& wait) {
    setup(w);
    Barrier([&] { t_setup_ = std::chrono::steady_clock::now(); });
    join(w);
    Barrier([&] { t_join_ = std::chrono::steady_clock::now(); });
    for (;;) {
      const int ph = cur_.load(std::memory_order_acquire);
      // this worker's own fibres of the set first, then whatever the others have not got to yet: a worker whose core
      // it shares with somebody else's thread (a busy host) would otherwise hold every step up with its whole list
      for (int k = 0; k < W_; ++k) {
        List& l = lists_[(size_t)ph * W_ + (w + k) % W_];
        for (;;) {
          const int i = l.next.fetch_add(1, std::memory_order_acq_rel);
          if (i >= (int)l.f.size()) break;
          Fiber* f = l.f[i];
          if (f->done) continue;
          Resume(f);
          if (f->done) live_[ph].fetch_sub(1);
        }
      }
      Barrier([&] {
        // every live fibre of this set, on every worker, waits for its step (or has just ended): queue the step; then
        // the next set with anybody left -- this one again if it is the only one -- once ITS step's answers are there
        for (int k = 0; k < W_; ++k) lists_[(size_t)ph * W_ + k].next.store(0, std::memory_order_relaxed);
        bool failed = failed_.load() != 0;
        if (!failed && live_[ph].load() > 0) {
          if (launch(ph) != 0) failed = true;
          inflight_[ph] = 1;
          ++steps_;
        }
        int next = -1;
        for (int k = 1; k <= P_ && next < 0; ++k)
          if (live_[(ph + k) % P_].load() > 0) next = (ph + k) % P_;
        if (!failed && next >= 0 && inflight_[next]) {
          if (wait(next) != 0) failed = true;
          inflight_[next] = 0;
        }
        if (failed) failed_.store(1);
        finished_.store(failed || next < 0);
        if (next >= 0) cur_.store(next, std::memory_order_release);
      });
      if (finished_.load()) break;
    }
  }

  struct List {  // the fibres of one set whose home is one worker, and how far into it the workers have got this turn
    std::vector<Fiber*> f;
    std::atomic<int> next{0};
    List() {}
    List(const List& o) : f(o.f), next(0) {}
  };
  const int W_, P_;
  std::chrono::steady_clock::time_point t_setup_, t_join_;  // when every worker had finished `setup` / `join`
  std::vector<std::unique_ptr<Fiber>> fibers_;
  std::vector<List> lists_;
  std::vector<std::atomic<int>> live_;   // per set: fibres that have not ended
  std::vector<char> inflight_;           // per set: a step is queued (the barrier's one thread only)
  std::atomic<int> arrived_{0}, failed_{0}, cur_{0};
  std::atomic<unsigned> gen_{0};
  std::atomic<bool> finished_{false};
  uint64_t steps_ = 0;
};

// runner_utils::Decompress (runner-utils.cpp:69-86) for a Predictor whose device-side models step in lock step with
// the other Predictors of its pool: the reference's Decoder, bit by bit.  Returns 0 or a gmx_status.
inline int LockstepDecompress(unsigned long long output_length, std::ifstream* is, std::ofstream* os, Predictor* p) {
  std::shared_ptr<GpuMixerBank> bank = GpuMixerBank::Of(p, sizeof(Predictor));
  if (!bank) {
    fprintf(stderr, "gmx::LockstepDecompress: this Predictor's mixers are not gmx::GpuMixer\n");
    return GMX_ERR_INVALID;
  }
  int rc = bank->BeginLockstep();
  if (rc) return rc;
  {
    Decoder d(is, p);
    for (unsigned long long pos = 0; pos < output_length && bank->status() == 0; ++pos) {
      int byte = 1;
      while (byte < 256) byte += byte + d.Decode();
      os->put(byte);
    }
  }
  rc = bank->EndLockstep();
  return rc ? rc : bank->status();
}

// runner_utils::RunDecompression (runner-utils.cpp:123-156) for every job at once: a Predictor and a Decoder per file,
// all device-side models of a pool of them one device step per coded bit.  opt.max_cpus worker threads (default: two
// fewer than the container's CPU quota is worth, else the hardware's threads) carry the files' fibres.  opt.groups
// pools (default one) are stepped in turn by the same workers: while one pool's device step runs, the workers do the
// other pool's host work.  Returns the number of jobs that failed.
inline int BatchedDecompressFiles(std::vector<BatchedJob>& jobs, const BatchedOptions& opt_in = BatchedOptions(),
                                  BatchedStats* stats = nullptr) {
  using clock = std::chrono::steady_clock;
  const clock::time_point tb = clock::now();
  BatchedOptions opt = opt_in;
  const int S = (int)jobs.size();
  if (S == 0) return 0;
  // The workers spin at the step's barrier, so each is a core: two fewer than a container's CPU quota is worth leave
  // the runtime's own threads their share (with workers = quota the run was throttled a dozen periods,
  // profiles/r04_exp_decode_workers.txt; 8 .. 14 workers do the same within a few per cent)
  int W = opt.max_cpus > 0 ? opt.max_cpus : (QuotaCpus() > 3 ? QuotaCpus() - 2 : QuotaCpus());
  if (W <= 0) W = (int)std::thread::hardware_concurrency();
  if (W <= 0) W = 1;
  if (W > S) W = S;
  int G = opt.groups > 0 ? opt.groups : 1;
  if (G > S) G = S;
  // file s belongs to pool s * G / S (whole stretches: the pools' streams are their files in order)
  std::vector<int> pool_of(S);
  std::vector<std::unique_ptr<MixerPool>> pools;
  for (int g = 0; g < G; ++g) {
    const int s0 = (int)((long long)S * g / G), s1 = (int)((long long)S * (g + 1) / G);
    pools.emplace_back(new MixerPool(s1 - s0, opt.device));
    pools.back()->DrawLstmInit();  // (the draw IS rand(): here, before any constructor runs)
    for (int s = s0; s < s1; ++s) pool_of[s] = g;
  }
  const clock::time_point t_pools = clock::now();
  std::vector<std::ifstream> in(S);
  std::vector<std::ofstream> out(S);
  std::vector<std::unique_ptr<Predictor>> preds(S);
  std::mutex first_mu, serial_mu;
  std::mutex device_mu;  // several pools: their banks are brought up one thread at a time -- a chainstep is CAPTURED into
                         // graphs on its group's (blocking) stream, and any synchronous copy of another pool's thread
                         // meanwhile -- the runtime's legacy stream -- "would make the legacy stream depend on a
                         // capturing blocking stream" and fails both
  std::condition_variable first_cv;
  bool first_built = false;
  std::atomic<int> pinned{0};
  clock::time_point t_first = tb, t0 = tb;
  LockstepRunner runner(W, G);
  for (auto& p : pools) p->SetLockstepYield([](int) { LockstepRunner::Yield(); });
  std::vector<char> worker_pinned(W, 0);
  auto pin = [&](int w) {
    if (worker_pinned[w]) return;
    worker_pinned[w] = 1;
    // (the device's share of the node, not W cores of it: W spinning workers on exactly W cores have nowhere to go when
    // somebody else's thread wakes up on one of them -- on a shared host that was 10 000 preemptions a second and
    // worker, and every one of them holds the step up: profiles/r04_exp_decode_busy_host.txt)
    static const char* mode = getenv("GMX_PIN_MODE");  // (experiments: "narrow" = W cores, "node" = the whole node)
    bool ok = false;
    if (!opt.pin_threads) return;
    if (mode && !strcmp(mode, "narrow"))
      ok = PinThreadToDeviceNode(pools[0]->device(), W);
    else if (mode && !strcmp(mode, "node"))
      ok = PinThreadToDeviceNode(pools[0]->device(), 0);
    else
      ok = PinThreadToDeviceShare(pools[0]->device());
    if (ok) ++pinned;
  };
  // (known once the first Predictor stands: whether a constructor draws from rand(), MixerPool::parallel_construction)
  std::atomic<bool> side_by_side{false};
  // A fibre runs the reference's Decoder and nothing that calls the device: the banks are brought up and joined
  // (BeginLockstep) on the worker threads' own stacks before the fibres start, every step is taken there too, and a
  // fibre's last act is the wait for its last Learn.  (Runtime and profiler code on a 1 MB foreign stack: rocprofv3
  // around this very loop died with a segmentation fault while BeginLockstep still ran inside the fibres.)
  std::vector<std::shared_ptr<GpuMixerBank>> banks(S);
  for (int s = 0; s < S; ++s) {
    runner.Add(s, pool_of[s], [&, s] {
      BatchedJob& job = jobs[s];
      if (job.status || !banks[s]) return;
      const clock::time_point a = clock::now();
      {
        Decoder d(&in[s], preds[s].get());  // runner_utils::Decompress (runner-utils.cpp:69-86), bit by bit
        for (unsigned long long pos = 0; pos < job.output_bytes && banks[s]->status() == 0; ++pos) {
          int byte = 1;
          while (byte < 256) byte += byte + d.Decode();
          out[s].put(byte);
        }
      }
      banks[s]->FinishLockstep();
      job.status = banks[s]->status();
      job.seconds = std::chrono::duration<double>(clock::now() - a).count();
      out[s].close();
    });
  }
  auto setup = [&](int w) {
    for (int s = w; s < S; s += W) {
      BatchedJob& job = jobs[s];
      pools[pool_of[s]]->InstallForThisThread();  // the Predictor about to be built is that pool's
      in[s].open(job.input_path, std::ios::in | std::ios::binary);
      bool ok = in[s].is_open();
      if (ok) {
        in[s].seekg(0, std::ios::end);
        job.input_bytes = in[s].tellg();
        in[s].seekg(0, std::ios::beg);
        runner_utils::ReadHeader(&in[s], &job.output_bytes);
        out[s].open(job.output_path, std::ios::out | std::ios::binary);
        ok = out[s].is_open();
      }
      if (!ok) job.status = -100;
      // (construction as in RunManyFiles: the first Predictor alone, the others side by side when no constructor draws
      // from rand(), else one at a time)
      if (s == 0) {
        if (ok) preds[0].reset(new Predictor());
        side_by_side.store(pools[0]->parallel_construction());
        std::lock_guard<std::mutex> lk(first_mu);
        first_built = true;
        t_first = clock::now();
        first_cv.notify_all();
        continue;
      }
      {
        std::unique_lock<std::mutex> lk(first_mu);
        first_cv.wait(lk, [&] { return first_built; });
      }
      if (!ok) continue;
      if (side_by_side.load()) {
        pin(w);
        preds[s].reset(new Predictor());
      } else {
        std::lock_guard<std::mutex> lk(serial_mu);
        preds[s].reset(new Predictor());
      }
    }
    MixerPool::UninstallForThisThread();
  };
  // ... and, every Predictor standing: the device banks (the first to get here creates a pool's for all), each stream's join
  auto join = [&](int w) {
    pin(w);  // (where constructors draw from rand() the threads are only pinned now)
    // (one pool: its own mutex orders what has to be; the streams' uploads run side by side -- one after the other they
    // were 15 s for 256 files)
    std::unique_lock<std::mutex> one_at_a_time(device_mu, std::defer_lock);
    if (G > 1) one_at_a_time.lock();
    for (int s = w; s < S; s += W) {
      if (!preds[s] || jobs[s].status) continue;
      banks[s] = GpuMixerBank::Of(preds[s].get(), sizeof(Predictor));
      if (!banks[s]) {
        fprintf(stderr, "gmx::BatchedDecompressFiles: this Predictor's mixers are not gmx::GpuMixer\n");
        jobs[s].status = GMX_ERR_INVALID;
        continue;
      }
      jobs[s].status = banks[s]->BeginLockstep();
      if (jobs[s].status) banks[s].reset();
    }
  };
  bool started = false;
  const uint64_t steps = runner.Run(
      setup, join,
      [&](int g) {
        if (!started) {
          started = true;
          t0 = clock::now();  // (the first step: every fibre of the first pool has built nothing more than its first record)
        }
        return pools[g]->StepLaunch();
      },
      [&](int g) { return pools[g]->StepWait(); });
  const clock::time_point t1 = clock::now();
  if (getenv("GMX_POOL_TRACE")) {
    auto sec = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    fprintf(stderr, "[gmx decode] %d files, %d workers, %d pool(s): pools %.2f s, Predictors (first alone: %.2f s) %.2f s, "
            "device banks and joins %.2f s, first records %.2f s, %llu steps %.2f s\n", S, W, G, sec(tb, t_pools),
            sec(tb, t_first), sec(t_pools, runner.setup_done()), sec(runner.setup_done(), runner.join_done()),
            sec(runner.join_done(), t0), (unsigned long long)steps, sec(t0, t1));
  }
  for (auto& p : pools) p->SetLockstepYield(nullptr);
  if (opt.destroy_predictors) {
    // side by side, like their construction: a Predictor gives back gigabytes of touched pages, and one thread's munmap
    // after another's is all a process's exit would do about them either
    std::vector<std::thread> th;
    for (int w = 0; w < W; ++w)
      th.emplace_back([&, w] {
        for (int s = w; s < S; s += W) preds[s].reset();
      });
    for (auto& t : th) t.join();
  } else {
    for (auto& p : preds) p.release();
    for (auto& p : pools) p.release();  // (their banks live in the pools)
  }
  int failed = 0;
  for (auto& j : jobs) failed += j.status != 0;
  for (auto& p : pools)
    if (p && p->status() != 0) fprintf(stderr, "gmx::BatchedDecompressFiles: %s\n", p->error().c_str());
  if (stats) {
    const clock::time_point tz = clock::now();
    stats->launches += steps;
    stats->pinned_threads += pinned.load();
    stats->pinned_cpus += W;
    stats->total_seconds = std::chrono::duration<double>(tz - tb).count();
    stats->wall_seconds = std::chrono::duration<double>(t1 - t0).count();
    stats->build_seconds = std::chrono::duration<double>(t0 - tb).count();
    stats->first_predictor_seconds = std::chrono::duration<double>(t_first - tb).count();
    stats->teardown_seconds = std::chrono::duration<double>(tz - t1).count();
    stats->parallel_construction = side_by_side.load();
    for (auto& j : jobs) stats->bits += 8ull * j.output_bytes;
  }
  return failed;
}

}  // namespace gmx

#endif  // GMX_BATCHED_H_
