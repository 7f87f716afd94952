// gmx_capi.cpp -- host side of libgmxmix.so: the C ABI of include/gmxmix.h.
//
// Owns device memory, the bank layout in HBM, record batches, the per-bit surface, and the
// reference-compatible (de)serialisation.  No compute happens here: every Predict/Learn goes
// to the gfx950 kernels of gmx_kernels.hip, and when no device is usable the calls fail with
// GMX_ERR_NO_DEVICE -- there is no CPU fallback in the product path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/gmxmix.h"
#include "gmx_buildhash.h"
#include "gmx_internal.h"

struct GmxSynthArgs {
  float* pred;
  uint32_t* mask;
  uint32_t* ctx;
  uint8_t* bits;
  uint64_t* rng;
  uint64_t* tcount;
  float* pstate;
  uint32_t* cstate;
  uint64_t rec_stride, n_bits, seed;
  int32_t n, n_pad, m, mask_words, n_streams, restart, ctx_mode, bit_mode;
  uint32_t ctx_mod, zero_mod;
};

extern "C" {
hipError_t gmx_launch_bank_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args, int n_streams,
                                  unsigned lds_bytes, int has_mask, int l0, int l1, int ns, int fin,
                                  unsigned stride0, hipStream_t stream);
hipError_t gmx_bank_kernel_set_lds(unsigned lds_bytes);
hipError_t gmx_launch_stock_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args, int n_streams,
                                   unsigned lds_bytes, int has_mask, int staged, int pow2_tables, hipStream_t stream);
hipError_t gmx_launch_single_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args, int n_inputs,
                                    int variant, hipStream_t stream);
hipError_t gmx_launch_wide_kernel(const GmxTopoDev* tp_dev, const GmxRunArgs* args, int n_streams, int has_mask,
                                  int n_inputs, hipStream_t stream);
hipError_t gmx_launch_synth_kernel(const GmxSynthArgs* args, hipStream_t stream);
struct GmxDecayArgs {
  const uint64_t* steps0;
  float* table;
  uint32_t* amb;
  uint32_t amb_cap, U;
  uint64_t T;
};
hipError_t gmx_launch_decay_kernel(const GmxDecayArgs* args, hipStream_t stream);
hipError_t gmx_launch_init_scal(uint8_t* banks, uint64_t bank_bytes, uint64_t scal_off, int m,
                                int n_streams, hipStream_t stream);
hipError_t gmx_launch_math_probe(const float* x, float* y, uint64_t n, int what, hipStream_t stream);
hipError_t gmx_launch_math_range(uint64_t lo, uint64_t count, int what, unsigned long long* out,
                                 hipStream_t stream);
}

static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  g_last_error = buf;
  (void)hipGetLastError();
  return GMX_ERR_HIP;
}
#define HIPCHK(call)                                   \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #call);  \
  } while (0)

static const size_t kBatchOwnStreamMin = 256u * 1024u;  // bytes from which a transfer takes a stream of its own
// ---- transfers of a record batch beside the kernels (BASELINE configs[3]: "pinned hipMemcpyAsync
// double-buffered host->device probability batches"): uploads and downloads of big batches run on
// streams of their own; three events per batch order them against the kernels that use it ------
struct GmxXfer {
  hipEvent_t ev_up = nullptr;    // behind the newest upload
  hipEvent_t ev_dev = nullptr;   // behind the newest device-side use, on the bank's stream
  hipEvent_t ev_down = nullptr;  // behind the newest download
  hipEvent_t ev_wr = nullptr;    // behind the newest write by ANOTHER bank's kernel, on that bank's stream
  bool up_rec = false, dev_rec = false, down_rec = false, wr_rec = false;
  bool idle = false;             // the host has waited for everything recorded above and nothing was queued since
};
static int xfer_init(GmxXfer& x) {
  HIPCHK(hipEventCreateWithFlags(&x.ev_up, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&x.ev_dev, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&x.ev_down, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&x.ev_wr, hipEventDisableTiming));
  return GMX_OK;
}
static void xfer_free(GmxXfer& x) {
  hipEvent_t evs[] = {x.ev_up, x.ev_dev, x.ev_down, x.ev_wr};
  for (hipEvent_t e : evs)
    if (e) (void)hipEventDestroy(e);
  x = GmxXfer();
}
static int xfer_note_device_use(GmxXfer& x, hipStream_t main) {
  HIPCHK(hipEventRecord(x.ev_dev, main));
  x.dev_rec = true;
  x.idle = false;
  return GMX_OK;
}
// A writer on ANOTHER bank's stream (the LSTM's scatter, the Indirect models' `into`) is about to write this batch's
// device arrays: it waits for what last touched THEM -- the batch's upload, the last device-side use -- not for
// everything queued on the batch's own stream.  (Waiting for the whole stream put the LSTM of chunk k+1 behind the
// mixers of chunk k-1 through the scatter between them: 1 ms of every 6 idle in the chain's longest stage.)
static int xfer_writer_waits(GmxXfer& x, hipStream_t writer) {
  if (x.up_rec) HIPCHK(hipStreamWaitEvent(writer, x.ev_up, 0));
  if (x.dev_rec) HIPCHK(hipStreamWaitEvent(writer, x.ev_dev, 0));
  if (x.wr_rec) HIPCHK(hipStreamWaitEvent(writer, x.ev_wr, 0));  // (the writer before it: the LSTM's scatter before `into`)
  return GMX_OK;
}
// ... and when its kernel is queued: the batch's own stream waits for it, and so does whoever writes the batch next
// (another such writer, the next upload).  The mark is an event on the WRITER's stream: recorded on the batch's own
// stream it would stand behind that stream's running kernel -- the mixers of the chunk before -- and the next writer
// (the Indirect models of chunk k+1) waited for the mixers of chunk k with nothing to wait for (seen in the kernel
// timeline: the chain's period was mixers + Indirect models, 5.6 ms, instead of the longest stage's 4.8).
static int xfer_writer_done(GmxXfer& x, hipStream_t writer, hipStream_t main) {
  HIPCHK(hipEventRecord(x.ev_wr, writer));
  x.wr_rec = true;
  x.idle = false;
  HIPCHK(hipStreamWaitEvent(main, x.ev_wr, 0));
  return GMX_OK;
}
// Hardware queues.  The runtime maps a process's streams onto FOUR hardware queues per priority level, round robin,
// and work on one queue runs in order whatever streams it came from.  The chain LSTM -> Indirect models -> mixers keeps
// three compute streams busy at once (the LSTM of chunk k+1 beside the mixers of chunk k) plus their transfer
// streams: on one level those ten streams share four queues, and an LSTM launch or an upload sits behind a 3 ms mixer
// kernel it has nothing to do with (measured: 6.2 us per bit for one compressor instead of 4.0; GPU_MAX_HW_QUEUES=8
// hid it, which is how it was found).  So every bank type lives on a priority level of its own -- the mixers normal,
// the Indirect models least, the LSTM (the longest stage) greatest -- and a bank's transfer streams follow its compute
// stream's level: at most four streams per level.
static hipError_t bank_stream_create(hipStream_t* out, int level /* 0 normal, 1 least, 2 greatest */) {
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) return e;
  const int mid = (least + greatest) / 2;
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, level == 1 ? least : level == 2 ? greatest : mid);
}
static hipError_t sibling_stream_create(hipStream_t* out, hipStream_t main) {
  int p = 0;
  hipError_t e = hipStreamGetPriority(main, &p);
  if (e != hipSuccess) return e;
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, p);
}
// The stream an upload of `bytes` runs on: one of its own (created on demand) behind the last
// device-side use of the batch, or the bank's stream for small ones.
static int xfer_begin_upload(GmxXfer& x, hipStream_t main, hipStream_t* own, size_t bytes, hipStream_t* use) {
  if (bytes < kBatchOwnStreamMin) {
    *use = main;
    return GMX_OK;
  }
  if (!*own) HIPCHK(sibling_stream_create(own, main));
  *use = *own;
  if (x.dev_rec) HIPCHK(hipStreamWaitEvent(*use, x.ev_dev, 0));
  if (x.wr_rec) HIPCHK(hipStreamWaitEvent(*use, x.ev_wr, 0));
  return GMX_OK;
}
// The two transfer streams of a bank, made when its first batch is (creating a stream takes milliseconds: not in the
// first upload of a run).
static int xfer_streams_ready(hipStream_t main, hipStream_t* up, hipStream_t* down) {
  if (!*up) HIPCHK(sibling_stream_create(up, main));
  if (!*down) HIPCHK(sibling_stream_create(down, main));
  return GMX_OK;
}
static int xfer_end_upload(GmxXfer& x, hipStream_t main, hipStream_t use) {
  HIPCHK(hipEventRecord(x.ev_up, use));
  x.up_rec = true;
  x.idle = false;
  if (use != main) HIPCHK(hipStreamWaitEvent(main, x.ev_up, 0));
  return GMX_OK;
}
static int xfer_begin_download(GmxXfer& x, hipStream_t main, hipStream_t* own, size_t bytes, hipStream_t* use) {
  if (x.idle) {
    // The host has already waited for the batch's work (gmx_batch_wait) and queued nothing on it since: the copy
    // depends on nothing and runs at once on the download stream -- NOT behind whatever else the bank's stream holds
    // by now.  This is how a pipelined caller should fetch results: a download queued right behind its kernel is a
    // copy that waits for milliseconds, and the copy engines take copies in order -- the uploads of the NEXT chunks
    // queued behind it waited with it (seen in the chain's timeline with 16 files: every chunk's upload ran only
    // when the mixers of the chunk before had finished, and the LSTM's scatter and both long kernels behind it: 6.4 ms
    // per chunk instead of 5.0).
    if (!*own) HIPCHK(sibling_stream_create(own, main));
    *use = *own;
    return GMX_OK;
  }
  if (bytes < kBatchOwnStreamMin) {
    *use = main;
    return GMX_OK;
  }
  if (!*own) HIPCHK(sibling_stream_create(own, main));
  *use = *own;
  int rc = xfer_note_device_use(x, main);
  if (rc) return rc;
  HIPCHK(hipStreamWaitEvent(*use, x.ev_dev, 0));
  return GMX_OK;
}
static int xfer_end_download(GmxXfer& x, hipStream_t use) {
  HIPCHK(hipEventRecord(x.ev_down, use));
  x.down_rec = true;
  x.idle = false;
  return GMX_OK;
}
// before a kernel overwrites the batch's result arrays
static int xfer_before_run(GmxXfer& x, hipStream_t main) {
  if (x.down_rec) HIPCHK(hipStreamWaitEvent(main, x.ev_down, 0));
  return GMX_OK;
}
static int xfer_wait(GmxXfer& x) {
  if (x.up_rec) HIPCHK(hipEventSynchronize(x.ev_up));
  if (x.wr_rec) HIPCHK(hipEventSynchronize(x.ev_wr));
  if (x.dev_rec) HIPCHK(hipEventSynchronize(x.ev_dev));
  if (x.down_rec) HIPCHK(hipEventSynchronize(x.ev_down));
  x.idle = true;
  return GMX_OK;
}


// Per-stream counts of a launch over streams of different lengths (GmxRunArgs::T_list and its cousins): staging
// slots used in turn, a slot reused only when the launch that read it is done.
// Staging slots of what a launch reads beside its batch (the per-stream counts of a ragged launch, the mixers' decay
// tables): as many as launches may be queued before the host has to wait for one to finish -- a caller that keeps a ring
// of kStageSlots batches in flight (gmx::MixerPool::kRing) never waits here.
static constexpr int kStageSlots = 4;
struct GmxCountList {
  uint64_t* dev[kStageSlots] = {};
  uint64_t* host[kStageSlots] = {};  // pinned
  hipEvent_t done[kStageSlots] = {};
  bool busy[kStageSlots] = {};
  size_t cap = 0;
  unsigned seq = 0;
};
static void count_list_free(GmxCountList& c) {
  for (int k = 0; k < kStageSlots; ++k) {
    if (c.dev[k]) (void)hipFree(c.dev[k]);
    if (c.host[k]) (void)hipHostFree(c.host[k]);
    if (c.done[k]) (void)hipEventDestroy(c.done[k]);
  }
  c = GmxCountList();
}
// counts [n] -> device, queued on `st` in front of the kernel that reads them; *dev_out is that kernel's argument
static int count_list_stage(GmxCountList& c, const uint64_t* counts, int n, hipStream_t st, const uint64_t** dev_out) {
  const int k = (int)(c.seq++ % (unsigned)kStageSlots);
  if (c.busy[k]) {
    HIPCHK(hipEventSynchronize(c.done[k]));
    c.busy[k] = false;
  }
  if ((size_t)n > c.cap) {
    for (int j = 0; j < kStageSlots; ++j) {
      if (c.busy[j]) {
        HIPCHK(hipEventSynchronize(c.done[j]));
        c.busy[j] = false;
      }
      if (c.dev[j]) (void)hipFree(c.dev[j]);
      if (c.host[j]) (void)hipHostFree(c.host[j]);
      c.dev[j] = nullptr;
      c.host[j] = nullptr;
    }
    c.cap = 0;
    const size_t cap = (size_t)n + 64;
    for (int j = 0; j < kStageSlots; ++j) {
      HIPCHK(hipMalloc((void**)&c.dev[j], cap * sizeof(uint64_t)));
      HIPCHK(hipHostMalloc((void**)&c.host[j], cap * sizeof(uint64_t), hipHostMallocDefault));
      if (!c.done[j]) HIPCHK(hipEventCreateWithFlags(&c.done[j], hipEventDisableTiming));
    }
    c.cap = cap;
  }
  memcpy(c.host[k], counts, (size_t)n * sizeof(uint64_t));
  HIPCHK(hipMemcpyAsync(c.dev[k], c.host[k], (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  *dev_out = c.dev[k];
  return GMX_OK;
}
// ... and behind that kernel
static int count_list_used(GmxCountList& c, hipStream_t st) {
  const int k = (int)((c.seq - 1u) % (unsigned)kStageSlots);
  HIPCHK(hipEventRecord(c.done[k], st));
  c.busy[k] = true;
  return GMX_OK;
}

struct gmx_group {
  int device = 0;
  int S = 0;
  GmxTopoDev topo;                 // host copy
  GmxTopoDev* topo_dev = nullptr;  // device copy
  uint8_t* banks = nullptr;        // S * bank_bytes
  float* latch_out = nullptr;      // [S][m]
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  unsigned lds_bytes = 0;
  std::vector<uint64_t> steps;     // host mirror of Mixer::steps_ (identical for all mixers of a stream)
  std::vector<uint8_t> fwd_done;   // per-bit protocol: forward seen, learn allowed
  // decay tables: staging slots used in turn, so the host prepares the launches ahead while
  // launch k still reads its tables
  struct DecaySlot {
    float* dev = nullptr;
    uint32_t* idx_dev = nullptr;
    float* host = nullptr;         // pinned
    uint32_t* idx_host = nullptr;
    size_t cap = 0;                // floats of dev / host
    size_t idx_cap = 0;            // entries of idx_dev / idx_host (one per stream of a launch)
    hipEvent_t done = nullptr;     // recorded behind the kernel that read this slot
    hipEvent_t ready = nullptr;    // recorded behind the upload on copy_stream
    bool busy = false;
    // tables made on the device (many different bit counts in one launch)
    uint64_t* st_dev = nullptr;    // [st_cap] bit counts of the table rows
    uint64_t* st_host = nullptr;   // pinned
    uint32_t* amb_dev = nullptr;   // [1 + kDecayAmbCap] unsettled entries
    uint32_t* amb_host = nullptr;  // pinned
    float* patch_host = nullptr;   // pinned [kDecayAmbCap]
    size_t st_cap = 0;
  } decay[kStageSlots];
  GmxCountList counts;                // per-stream bit counts of ragged launches
  hipStream_t copy_stream = nullptr;  // uploads of the decay tables, beside the running kernel
  // record batches travel on streams of their own, beside the running kernel (BASELINE configs[3]:
  // "pinned hipMemcpyAsync double-buffered host->device probability batches"); events order them
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  unsigned run_seq = 0;
  hipEvent_t tm0 = nullptr, tm1 = nullptr;  // gmx_group_timer_*
  gmx_batch* one = nullptr;        // 1-bit batch (one record per stream) of the per-bit surface
  std::vector<gmx_batch*> batches; // live batches; orphaned (b->g = nullptr) when the group dies
  std::vector<struct GmxSession*> sessions;  // per stream, lazily: persistent per-bit kernels
  std::vector<struct gmx_lockstep*> locksteps;  // live lock-step objects; orphaned (ls->g = nullptr) when the group dies
  bool banks_in_use_by_dead_kernel = false;  // a persistent wave that stopped answering may still hold them: never freed
  bool use_sessions = true;        // tests: per-bit calls as two launches instead of a session
  bool mailbox_on_device = true;   // tests: false keeps the sessions' command blocks in pinned host memory
  bool force_general = false;      // tests: route everything through the general kernel
  bool stock_exact = false;        // tests: stock kernels use their masked forward chains only
  int stock_staged = -1;           // stock kernel: rows through the LDS images (1), lane-private (0), by stream count (-1)
  bool decay_on_host = false;      // tests: the decay tables always from the host's libm loop
  bool stock_pairs = false;        // batched runs of the stock shape through gmx_wide_kernel<90, 64> (lane pairs)
  int single_variant = 0;          // tests/tuning: lanes per stream of the single-mixer kernel (0 = default)
};

struct gmx_batch {
  gmx_group* g = nullptr;
  int S = 0;                       // streams covered (g->S, or 1 for the per-bit batch)
  uint64_t max_bits = 0;
  unsigned flags = 0;
  // device
  float* d_pred = nullptr;
  uint32_t* d_mask = nullptr;
  uint32_t* d_ctx = nullptr;
  uint8_t* d_bits = nullptr;
  float* d_p = nullptr;
  float* d_out = nullptr;
  float* d_last = nullptr;         // [S][m] with GMX_BATCH_LAST_OUTPUTS
  // synthetic generator state
  uint64_t* d_rng = nullptr;
  uint64_t* d_tcount = nullptr;
  float* d_pstate = nullptr;
  uint32_t* d_cstate = nullptr;
  // pinned host staging (lazy)
  float* h_pred = nullptr;
  uint32_t* h_mask = nullptr;
  uint32_t* h_ctx = nullptr;
  uint8_t* h_bits = nullptr;
  float* h_p = nullptr;
  float* h_out = nullptr;
  float* h_last = nullptr;
  GmxXfer x;  // ordering of this batch's transfers against the kernels that use its device arrays
};

// lock-step stepping of all streams, one hipGraph per half step (gmx_lockstep.inc)
struct gmx_lockstep {
  gmx_group* g = nullptr;
  gmx_batch* b = nullptr;        // one record per stream; owned
  float* dec_host = nullptr;     // pinned [S]: float(0.9 / pow(1e-7 * steps_ + 0.8, 0.8)) of every stream's next learn
  uint8_t* bits_host = nullptr;  // pinned [S]: the coded bits as they were when gmx_lockstep_learn was called
  float* dec_dev = nullptr;      // [S]
  uint32_t* idx_dev = nullptr;   // [S] identity: stream s uses table row s
  hipGraph_t g_predict = nullptr, g_learn = nullptr, g_step = nullptr;
  hipGraphExec_t x_predict = nullptr, x_learn = nullptr, x_step = nullptr;  // step = learn, then predict
  bool predicted = false;
  // GMX_LOCKSTEP_PERSISTENT (gmx_stock_lockstep_kernel)
  bool persistent = false, ps_running = false, ps_dead = false, ps_counted = false;
  hipStream_t ps_stream = nullptr;
  GmxLsDoor* door = nullptr;     // device memory the host writes through the BAR
  uint32_t* done = nullptr;      // pinned host
  uint32_t* blk_seen = nullptr;  // device [S]
  uint32_t* blk_replay = nullptr;
  uint32_t* relay = nullptr;     // device [64]: doorbell relay and arrival count (GmxLsArgs)
  float* live = nullptr;         // device [S][GMX_LS_LIVE_FLOATS]
  uint32_t seq = 0, word = 0, completed = 0;  // newest command published / known complete
  bool learn_inflight = false;   // the newest command is a learn nobody has waited for
};

enum GmxKernelKind : int;
static GmxKernelKind kernel_for(const gmx_group* g, unsigned mode);
// per-bit sessions (gmx_session.inc); sessions_close also stops the persistent lock-step waves of the group
static int sessions_close(gmx_group* g, bool keep_forward);
static int locksteps_stop(gmx_group* g, gmx_lockstep* except, bool keep_forward);
static void sessions_free(gmx_group* g);

extern "C" const char* gmx_strerror(int status) {
  switch (status) {
    case GMX_OK: return "ok";
    case GMX_ERR_INVALID: return "invalid argument or unsupported topology";
    case GMX_ERR_NOMEM: return "out of memory";
    case GMX_ERR_HIP: return "HIP runtime error";
    case GMX_ERR_NO_DEVICE: return "no usable gfx950 device (this library has no CPU fallback)";
    case GMX_ERR_STATE: return "call order violated";
    case GMX_ERR_FORMAT: return "malformed checkpoint";
    default: return "unknown status";
  }
}

extern "C" const char* gmx_last_error(void) { return g_last_error.c_str(); }

extern "C" const char* gmx_build_info(void) {
  return "libgmxmix gfx950 (CDNA4) hipcc -O3 -ffp-contract=off; abi 1; src " GMX_SRC_HASH "; kernels " GMX_KERN_HASH;
}

extern "C" int gmx_device_count(int* count) {
  if (!count) return GMX_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *count = 0;
    return GMX_OK;
  }
  *count = n;
  return GMX_OK;
}

// "0000:c1:00.0": the key of /sys/bus/pci/devices/<id>/numa_node (host threads that feed a device belong on its node)
extern "C" int gmx_device_pci_bus_id(int device, char* buf, size_t len) {
  if (!buf || len < 13) return GMX_ERR_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return GMX_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) return GMX_ERR_INVALID;
  HIPCHK(hipDeviceGetPCIBusId(buf, (int)len, device));
  return GMX_OK;
}

static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }
static uint64_t round_up64(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Predictor::AddMixers bookkeeping (predictor.cpp:251-358, mixer.cpp:3-27,
// short-term-memory.cpp:199-213) turned into a device layout.
static int build_topology(const gmx_topology* t, GmxTopoDev* o) {
  if (!t || !t->mixers || t->n_mixers < 1 || t->n_mixers > GMX_MAX_MIXERS) return GMX_ERR_INVALID;
  if (t->n_inputs < 1 || t->n_inputs > GMX_MAX_INPUTS) return GMX_ERR_INVALID;
  if (t->n_skip < 0 || t->n_skip > GMX_MAX_SKIP || (t->n_skip > 0 && !t->skip_index))
    return GMX_ERR_INVALID;
  memset(o, 0, sizeof *o);
  o->n = t->n_inputs;
  o->n_pad = (int32_t)round_up((uint32_t)t->n_inputs, 4);
  o->n_skip = t->n_skip;
  o->m = t->n_mixers;
  o->mask_words = (t->n_inputs + 31) / 32;
  for (int i = 0; i < t->n_skip; ++i) {
    if (t->skip_index[i] < 0 || t->skip_index[i] >= t->n_inputs) return GMX_ERR_INVALID;
    o->skip_idx[i] = t->skip_index[i];
  }
  int prev_layer = 0;
  for (int j = 0; j < t->n_mixers; ++j) {
    const gmx_mixer_desc& md = t->mixers[j];
    if (md.layer < prev_layer || md.layer > 2 || md.table_size == 0) return GMX_ERR_INVALID;
    if (md.layer == 2 && o->has_final) return GMX_ERR_INVALID;  // a second final would overwrite the first
    prev_layer = md.layer;
    GmxMixerDev& x = o->mx[j];
    x.table_size = md.table_size;
    x.layer = md.layer;
    x.lr = md.learning_rate;
    if (md.layer == 0) {
      x.out_index = o->l0++;
      x.weight_size = (uint32_t)(o->n + x.out_index);
    } else if (md.layer == 1) {
      x.out_index = o->l1++;
      x.weight_size = (uint32_t)(o->l0 + x.out_index + o->n_skip);
    } else {
      x.out_index = 0;
      x.weight_size = (uint32_t)(o->l0 + o->l1 + o->n_skip);
      o->has_final = 1;
    }
    if (x.weight_size == 0) return GMX_ERR_INVALID;
    x.stride = round_up(x.weight_size, 32);
    x.pitch = x.stride + 4;
  }
  if (o->l0 < 1) return GMX_ERR_INVALID;
  for (int j = o->l0; j < o->m; ++j)
    if (o->mx[j].weight_size > 64) return GMX_ERR_INVALID;  // layer-1/final rows: one lane per weight
  // Stored row lengths are uniform per layer group, so the update passes run without
  // per-row lane masks: every layer-0 row as long as the longest one (the cascade makes
  // them differ by at most l0-1 weights), every layer-1/final row 64 floats (one per lane).
  for (int j = 0; j < o->l0; ++j) o->mx[j].stride = o->mx[o->l0 - 1].stride;
  for (int j = o->l0; j < o->m; ++j) o->mx[j].stride = 64;
  // bank layout in HBM: weight tables, then row-step tables, then per-mixer scalars.
  // The reference's own shape keeps each row's step counter INSIDE the row instead, in the last 8
  // bytes of its zero padding (every row there ends in at least one whole spare quad: 113 of 128,
  // 33 of 64 floats): a separate counter costs a 64-byte sector each way per row change -- 19 % of
  // that shape's traffic when every row changes every bit -- while the row's last sector travels
  // anyway.  The spare quad is never moved as weights by any kernel of that shape.
  const bool fold = o->n == 90 && o->l0 == 24 && o->l1 == 8 && o->n_skip == 1 && o->has_final &&
                    o->mx[23].stride == 128;
  // The 256-input 24/8/1 shape (gmx_wide.hip) keeps the counter inside the row too, right BEHIND the weights:
  // the 16-byte piece after the last one that holds weights (float round_up(weight_size, 4) of a layer-0 row,
  // float 36 of a layer-1 / final row) is padding no kernel of that shape moves -- gmx_wide.hip fetches and
  // writes back only pieces with weights, the general kernel only quads below weight_size -- and it lies in the
  // 64-byte sector the row's last weights travel in anyway, for 28 of the 33 rows (layer-0 rows 0 and 13..16 end
  // on a sector boundary: those five still pay a sector each way).
  const bool fold2 = o->n == 256 && o->l0 == 24 && o->l1 == 8 && o->n_skip == 1 && o->has_final &&
                     o->mx[23].stride == 288;
  uint64_t off = 0;
  for (int j = 0; j < o->m; ++j) {
    o->mx[j].w_off = off;
    off += (uint64_t)o->mx[j].table_size * o->mx[j].stride * 4u;
  }
  for (int j = 0; j < o->m; ++j) {
    GmxMixerDev& x = o->mx[j];
    x.rs_folded = fold ? 1u : (fold2 ? 2u : 0u);
    if (fold) {
      if (x.stride < round_up(x.weight_size, 4) + 4u) return GMX_ERR_INVALID;
      x.rs_off = x.w_off + (uint64_t)x.stride * 4u - 8u;
      x.rs_pitch = x.stride * 4u;
    } else if (fold2) {
      const uint32_t at = j < o->l0 ? round_up(x.weight_size, 4) : 36u;  // float index of the counter in the row
      if (x.weight_size > at || at + 2u > x.stride) return GMX_ERR_INVALID;
      x.rs_off = x.w_off + (uint64_t)at * 4u;
      x.rs_pitch = x.stride * 4u;
    } else {
      x.rs_off = off;
      x.rs_pitch = 8u;
      off += (uint64_t)x.table_size * 8u;
      off = round_up64(off, 128);
    }
  }
  o->scal_off = off;
  off += (uint64_t)o->m * 24u;
  o->bank_bytes = round_up64(off, 256);
  // LDS image of one wave (float offsets)
  uint32_t l = 0;
  o->lds_in0 = l;
  o->in0_sz = round_up((uint32_t)o->n_pad + (uint32_t)o->l0 + 4u, 4);
  l += 2 * o->in0_sz;
  o->lds_o1 = l;
  l += round_up((uint32_t)(o->l1 ? o->l1 : 1), 4);
  o->lds_skip = l;
  l += GMX_MAX_SKIP;
  o->lds_misc = l;
  l += 256;
  // Row cache: per layer group (layer 0 | layers 1+2) all slot-0 rows, then all slot-1 rows,
  // every row of a group D = longest stride + 4 floats apart.  D/4 is odd and the slot-1
  // image starts a multiple of 64 floats later, so the 16 lanes of a ds_read_b128 group that
  // read "their mixer's current row at element j" always hit 16 different bank quads.
  for (int grp = 0; grp < 2; ++grp) {
    const int lo = grp == 0 ? 0 : o->l0, hi = grp == 0 ? o->l0 : o->m;
    if (lo >= hi) continue;
    uint32_t dmax = 0;
    for (int j = lo; j < hi; ++j) dmax = std::max(dmax, o->mx[j].stride);
    const uint32_t D = dmax + 4;
    const uint32_t S = round_up((uint32_t)(hi - lo) * D, 64);
    for (int j = lo; j < hi; ++j) {
      o->mx[j].lds_off = l + (uint32_t)(j - lo) * D;
      o->mx[j].pitch = S;  // distance from a mixer's slot 0 to its slot 1
    }
    l += 2 * S;
  }
  o->lds_total = l;
  if ((uint64_t)o->lds_total * 4u > 160u * 1024u) return GMX_ERR_INVALID;
  return GMX_OK;
}

static void batch_free(gmx_batch* b);
static int batch_alloc(gmx_batch** out, gmx_group* g, int S, uint64_t max_bits, unsigned flags);
static int batch_note_device_use(gmx_batch* b);

extern "C" int gmx_group_create(gmx_group** out, const gmx_topology* topo, int n_streams, int device) {
  if (!out || n_streams < 1) return GMX_ERR_INVALID;
  *out = nullptr;
  GmxTopoDev td;
  int rc = build_topology(topo, &td);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return GMX_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(device));
  gmx_group* g = new (std::nothrow) gmx_group();
  if (!g) return GMX_ERR_NOMEM;
  g->device = device;
  g->S = n_streams;
  g->topo = td;
  g->lds_bytes = td.lds_total * 4u;
  g->steps.assign(n_streams, 0);
  g->fwd_done.assign(n_streams, 0);
#define GCHK(call)                                 \
  do {                                             \
    hipError_t e_ = (call);                        \
    if (e_ != hipSuccess) {                        \
      int r_ = hip_fail(e_, #call);                \
      gmx_group_destroy(g);                        \
      return e_ == hipErrorOutOfMemory ? GMX_ERR_NOMEM : r_; \
    }                                              \
  } while (0)
  GCHK(bank_stream_create(&g->stream, 0));
  GCHK(hipEventCreate(&g->ev0));
  GCHK(hipEventCreate(&g->ev1));
  GCHK(hipMalloc((void**)&g->topo_dev, sizeof(GmxTopoDev)));
  GCHK(hipMemcpy(g->topo_dev, &g->topo, sizeof(GmxTopoDev), hipMemcpyHostToDevice));
  GCHK(hipMalloc((void**)&g->banks, (size_t)n_streams * td.bank_bytes));
  GCHK(hipMalloc((void**)&g->latch_out, (size_t)n_streams * td.m * sizeof(float)));
  GCHK(hipMemsetAsync(g->latch_out, 0, (size_t)n_streams * td.m * sizeof(float), g->stream));
  if (g->lds_bytes > 48u * 1024u) GCHK(gmx_bank_kernel_set_lds(g->lds_bytes));
#undef GCHK
  rc = gmx_group_reset(g);
  if (rc) {
    gmx_group_destroy(g);
    return rc;
  }
  *out = g;
  return GMX_OK;
}

extern "C" void gmx_group_destroy(gmx_group* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  sessions_free(g);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->one) {
    batch_free(g->one);
    g->one = nullptr;
  }
  // Batches outlive their group only as empty shells: they keep their buffers until
  // gmx_batch_destroy, but every call on them fails with GMX_ERR_INVALID from now on.
  for (gmx_lockstep* ls : g->locksteps) ls->g = nullptr;  // their calls fail from now on; gmx_lockstep_destroy frees them
  g->locksteps.clear();
  for (gmx_batch* b : g->batches) b->g = nullptr;
  g->batches.clear();
  // (a persistent wave that stopped answering and is still resident reads the topology and writes rows and latch:
  // leaked rather than freed under a running kernel)
  if (!g->banks_in_use_by_dead_kernel) {
    if (g->banks) (void)hipFree(g->banks);
    if (g->latch_out) (void)hipFree(g->latch_out);
    if (g->topo_dev) (void)hipFree(g->topo_dev);
  }
  for (auto& d : g->decay) {
    if (d.st_dev) (void)hipFree(d.st_dev);
    if (d.st_host) (void)hipHostFree(d.st_host);
    if (d.amb_dev) (void)hipFree(d.amb_dev);
    if (d.amb_host) (void)hipHostFree(d.amb_host);
    if (d.patch_host) (void)hipHostFree(d.patch_host);
    if (d.dev) (void)hipFree(d.dev);
    if (d.idx_dev) (void)hipFree(d.idx_dev);
    if (d.host) (void)hipHostFree(d.host);
    if (d.idx_host) (void)hipHostFree(d.idx_host);
    if (d.done) (void)hipEventDestroy(d.done);
    if (d.ready) (void)hipEventDestroy(d.ready);
  }
  count_list_free(g->counts);
  if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
  if (g->up_stream) {
    (void)hipStreamSynchronize(g->up_stream);
    (void)hipStreamDestroy(g->up_stream);
  }
  if (g->down_stream) {
    (void)hipStreamSynchronize(g->down_stream);
    (void)hipStreamDestroy(g->down_stream);
  }
  if (g->tm0) (void)hipEventDestroy(g->tm0);
  if (g->tm1) (void)hipEventDestroy(g->tm1);
  if (g->ev0) (void)hipEventDestroy(g->ev0);
  if (g->ev1) (void)hipEventDestroy(g->ev1);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

extern "C" int gmx_group_n_streams(const gmx_group* g) { return g ? g->S : GMX_ERR_INVALID; }
extern "C" int gmx_group_n_mixers(const gmx_group* g) { return g ? g->topo.m : GMX_ERR_INVALID; }
extern "C" int gmx_group_n_inputs(const gmx_group* g) { return g ? g->topo.n : GMX_ERR_INVALID; }
extern "C" uint64_t gmx_group_bank_bytes(const gmx_group* g) { return g ? g->topo.bank_bytes : 0; }

extern "C" int gmx_group_reset(gmx_group* g) {
  if (!g) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, false);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(g->banks, 0, (size_t)g->S * g->topo.bank_bytes, g->stream));
  HIPCHK(gmx_launch_init_scal(g->banks, g->topo.bank_bytes, g->topo.scal_off, g->topo.m, g->S,
                              g->stream));
  HIPCHK(hipStreamSynchronize(g->stream));
  std::fill(g->steps.begin(), g->steps.end(), 0);
  std::fill(g->fwd_done.begin(), g->fwd_done.end(), 0);
  for (gmx_lockstep* ls : g->locksteps) ls->predicted = false;
  return GMX_OK;
}

extern "C" int gmx_group_sync(gmx_group* g) {
  if (!g) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, true);  // per-bit work is part of "everything submitted so far"
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(g->stream));
  return GMX_OK;
}

// HIP events on the group's stream around any number of queued calls (bench.py brackets its
// timed launches with them: GPU time without a host synchronisation per launch).
extern "C" int gmx_group_timer_start(gmx_group* g) {
  if (!g) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  if (!g->tm0) {
    HIPCHK(hipEventCreate(&g->tm0));
    HIPCHK(hipEventCreate(&g->tm1));
  }
  HIPCHK(hipEventRecord(g->tm0, g->stream));
  return GMX_OK;
}

extern "C" int gmx_group_timer_stop(gmx_group* g, float* ms) {
  if (!g || !ms || !g->tm0) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(hipEventRecord(g->tm1, g->stream));
  HIPCHK(hipEventSynchronize(g->tm1));
  HIPCHK(hipEventElapsedTime(ms, g->tm0, g->tm1));
  return GMX_OK;
}

// First factor of the learning-rate decay, float(0.9 / pow(1e-7 * steps_ + 0.8, 0.8))
// (mixer.cpp:111).  It depends on the bit count only, so the host computes it once per bit
// with the same libm pow the reference calls and ships it with the records; the per-row
// second factor (mixer.cpp:112) is IEEE double arithmetic and stays on the device.
static float decay_base(uint64_t steps) { return (float)(0.9 / pow(0.0000001 * steps + 0.8, 0.8)); }

// Every gate table a power of two (the reference's are): the stock kernel's compile-time "plain" build indexes with a mask.
static int topo_tables_pow2(const GmxTopoDev& t) {
  for (int j = 0; j < t.m; ++j)
    if (t.mx[j].table_size == 0 || (t.mx[j].table_size & (t.mx[j].table_size - 1u)) != 0) return 0;
  return 1;
}

#include "gmx_session.inc"

static const uint32_t kDecayAmbCap = 4096;
static const size_t kDecayDeviceMin = 8192;  // table entries from which the device makes the table

// The decay table of `uniq` rows x T made on the device (gmx_decay_kernel), the entries it could
// not settle computed here with libm.  Returns GMX_ERR_STATE if there were too many of those
// (the caller then fills the table on the host).
static int decay_table_on_device(gmx_group* g, gmx_group::DecaySlot& d, const std::map<uint64_t, uint32_t>& uniq,
                                 uint64_t T) {
  const size_t U = uniq.size();
  if (U > d.st_cap) {
    if (d.st_dev) (void)hipFree(d.st_dev);
    if (d.st_host) (void)hipHostFree(d.st_host);
    d.st_dev = nullptr;
    d.st_host = nullptr;
    d.st_cap = 0;
    const size_t cap = U + U / 2 + 64;
    HIPCHK(hipMalloc((void**)&d.st_dev, cap * sizeof(uint64_t)));
    HIPCHK(hipHostMalloc((void**)&d.st_host, cap * sizeof(uint64_t), hipHostMallocDefault));
    d.st_cap = cap;
  }
  if (!d.amb_dev) {
    HIPCHK(hipMalloc((void**)&d.amb_dev, (1 + kDecayAmbCap) * sizeof(uint32_t)));
    HIPCHK(hipHostMalloc((void**)&d.amb_host, (1 + kDecayAmbCap) * sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&d.patch_host, kDecayAmbCap * sizeof(float), hipHostMallocDefault));
  }
  for (auto& kv : uniq) d.st_host[kv.second] = kv.first;
  hipStream_t cs = g->copy_stream;
  HIPCHK(hipMemcpyAsync(d.st_dev, d.st_host, U * sizeof(uint64_t), hipMemcpyHostToDevice, cs));
  HIPCHK(hipMemsetAsync(d.amb_dev, 0, sizeof(uint32_t), cs));
  GmxDecayArgs a;
  a.steps0 = d.st_dev;
  a.table = d.dev;
  a.amb = d.amb_dev;
  a.amb_cap = kDecayAmbCap;
  a.U = (uint32_t)U;
  a.T = T;
  HIPCHK(gmx_launch_decay_kernel(&a, cs));
  HIPCHK(hipMemcpyAsync(d.amb_host, d.amb_dev, (1 + kDecayAmbCap) * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
  HIPCHK(hipStreamSynchronize(cs));
  const uint32_t n = d.amb_host[0];
  if (n > kDecayAmbCap) return GMX_ERR_STATE;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t i = d.amb_host[1 + k];
    d.patch_host[k] = decay_base(d.st_host[i / T] + i % T);
    HIPCHK(hipMemcpyAsync(d.dev + i, d.patch_host + k, sizeof(float), hipMemcpyHostToDevice, cs));
  }
  return GMX_OK;
}

// Fill the group's decay tables for a run of T learning bits over streams [s0, s0+ns).
static int prepare_decay(gmx_group* g, int s0, int ns, uint64_t T, int learn, gmx_group::DecaySlot** out) {
  std::map<uint64_t, uint32_t> uniq;
  std::vector<uint32_t> idx(ns);
  for (int i = 0; i < ns; ++i) {
    uint64_t st = learn ? g->steps[s0 + i] : 0;
    auto it = uniq.find(st);
    if (it == uniq.end()) it = uniq.emplace(st, (uint32_t)uniq.size()).first;
    idx[i] = it->second;
  }
  gmx_group::DecaySlot& d = g->decay[g->run_seq++ % (unsigned)kStageSlots];
  *out = &d;
  // the launch kStageSlots back read this slot: it must be done before the staging is rewritten
  if (d.busy) {
    HIPCHK(hipEventSynchronize(d.done));
    d.busy = false;
  }
  // (what one slot lacks, every slot gets now: allocations belong to a run's first launch, not to its first four)
  const size_t need = (size_t)uniq.size() * T;
  if (need > d.cap || (size_t)ns > d.idx_cap || !d.done) {
    for (gmx_group::DecaySlot& e : g->decay) {
      if (e.busy) {
        HIPCHK(hipEventSynchronize(e.done));
        e.busy = false;
      }
      if (need > e.cap) {
        if (e.dev) (void)hipFree(e.dev);
        if (e.host) (void)hipHostFree(e.host);
        e.dev = nullptr;
        e.host = nullptr;
        e.cap = 0;
        size_t cap = need + need / 2 + 1024;
        HIPCHK(hipMalloc((void**)&e.dev, cap * sizeof(float)));
        HIPCHK(hipHostMalloc((void**)&e.host, cap * sizeof(float), hipHostMallocDefault));
        e.cap = cap;
      }
      if ((size_t)ns > e.idx_cap) {  // one row index per stream of the launch
        if (e.idx_dev) (void)hipFree(e.idx_dev);
        if (e.idx_host) (void)hipHostFree(e.idx_host);
        e.idx_dev = nullptr;
        e.idx_host = nullptr;
        e.idx_cap = 0;
        const size_t icap = (size_t)ns + 64;
        HIPCHK(hipMalloc((void**)&e.idx_dev, icap * sizeof(uint32_t)));
        HIPCHK(hipHostMalloc((void**)&e.idx_host, icap * sizeof(uint32_t), hipHostMallocDefault));
        e.idx_cap = icap;
      }
      if (!e.done) {
        HIPCHK(hipEventCreateWithFlags(&e.done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming));
      }
    }
  }
  if (!g->copy_stream) HIPCHK(sibling_stream_create(&g->copy_stream, g->stream));
  // Streams at the same bit count share a row; with many different counts the device makes the
  // table (see gmx_decay_kernel), else the host's libm loop is shorter than the detour.
  bool on_device = false;
  // (the device's list of unsettled entries holds 32-bit flat indices: larger tables take the host loop)
  if (learn && uniq.size() > 1 && uniq.size() * T >= kDecayDeviceMin && uniq.size() * T < (1ull << 32) &&
      !g->decay_on_host) {
    const int rcd = decay_table_on_device(g, d, uniq, T);
    if (rcd != GMX_OK && rcd != GMX_ERR_STATE) return rcd;
    on_device = rcd == GMX_OK;
  }
  if (!on_device) {
    for (auto& kv : uniq) {
      float* tab = d.host + (size_t)kv.second * T;
      if (learn)
        for (uint64_t t = 0; t < T; ++t) tab[t] = decay_base(kv.first + t);
      else
        for (uint64_t t = 0; t < T; ++t) tab[t] = 0.f;
    }
    HIPCHK(hipMemcpyAsync(d.dev, d.host, uniq.size() * T * sizeof(float), hipMemcpyHostToDevice, g->copy_stream));
  }
  memcpy(d.idx_host, idx.data(), ns * sizeof(uint32_t));
  HIPCHK(hipMemcpyAsync(d.idx_dev, d.idx_host, ns * sizeof(uint32_t), hipMemcpyHostToDevice, g->copy_stream));
  HIPCHK(hipEventRecord(d.ready, g->copy_stream));
  HIPCHK(hipStreamWaitEvent(g->stream, d.ready, 0));
  return GMX_OK;
}

// Which kernel a launch takes.
enum GmxKernelKind : int { GMX_K_SINGLE, GMX_K_WIDE, GMX_K_STOCK, GMX_K_BANK };
static GmxKernelKind kernel_for(const gmx_group* g, unsigned mode) {
  // Banks that are a single layer-0 mixer take the register-resident throughput kernel
  // (gmx_single.hip); everything else, and every per-bit call, the general kernel.
  const bool single = g->topo.m == 1 && g->topo.n <= 256 && (mode & GMX_MODE_PREDICT) &&
                      !(mode & GMX_MODE_LATCH) && !g->force_general;
  // The reference's own shape (90 inputs, 24/8/1, one skip input) runs with its rows in
  // registers (gmx_stock.hip); it only needs the small part of the LDS image.
  const bool stock = g->topo.n == 90 && g->topo.l0 == 24 && g->topo.l1 == 8 && g->topo.n_skip == 1 &&
                     g->topo.has_final && g->topo.mx[23].stride == 128 && !g->force_general;
  // The 256-input 24/8/1 bank (BASELINE configs[2]) likewise, two lanes per layer-0 row
  // (gmx_wide.hip); batched Predict(+Learn) only.
  const bool wide = g->topo.l0 == 24 && g->topo.l1 == 8 && g->topo.n_skip == 1 && g->topo.has_final &&
                    ((g->topo.n == 256 && g->topo.mx[23].stride == 288) ||
                     (g->topo.n == 90 && g->topo.mx[23].stride == 128 && g->stock_pairs && !g->stock_exact)) &&
                    (mode & GMX_MODE_PREDICT) && !(mode & GMX_MODE_LATCH) && !g->force_general;
  return single ? GMX_K_SINGLE : wide ? GMX_K_WIDE : stock ? GMX_K_STOCK : GMX_K_BANK;
}

static int launch_run(gmx_group* g, gmx_batch* b, int s0, int rec0, int ns, uint64_t T,
                      unsigned mode, float* kernel_ms, const uint64_t* n_list = nullptr) {
  // n_list (host, [ns]): bits of each stream of the range, at most T; only for the kernels that take a count per
  // block (gmx_stock_kernel, gmx_bank_kernel)
  if (T == 0) return GMX_OK;
  for (gmx_lockstep* ls : g->locksteps) ls->predicted = false;  // the latch no longer belongs to their Predict
  gmx_group::DecaySlot* dec = nullptr;
  int rc = prepare_decay(g, s0, ns, T, (mode & GMX_MODE_LEARN) ? 1 : 0, &dec);
  if (rc) return rc;
  GmxRunArgs a;
  memset(&a, 0, sizeof a);
  a.banks = g->banks;
  a.pred = b->d_pred;
  a.mask = (b->flags & GMX_BATCH_MASK) ? b->d_mask : nullptr;
  a.ctx = b->d_ctx;
  a.bits = b->d_bits;
  a.decay = dec->dev;
  a.decay_idx = dec->idx_dev;
  a.p_out = b->d_p;
  a.out_all = (b->flags & GMX_BATCH_OUTPUTS) ? b->d_out : nullptr;
  a.out_last = (b->flags & GMX_BATCH_LAST_OUTPUTS) ? b->d_last : nullptr;
  a.latch_out = g->latch_out;
  a.rec_stride = b->max_bits;
  a.T = T;
  if (n_list) {
    int rcl = count_list_stage(g->counts, n_list, ns, g->stream, &a.T_list);
    if (rcl) return rcl;
  }
  a.mode = mode | (g->stock_exact ? GMX_MODE_EXACT : 0u);
  a.stream_base = s0;
  a.rec_base = rec0;
  a.n_streams = ns;
  {
    int rcx = xfer_before_run(b->x, g->stream);  // p of the previous run is out
    if (rcx) return rcx;
  }
  if (kernel_ms) HIPCHK(hipEventRecord(g->ev0, g->stream));
  const GmxKernelKind kind = kernel_for(g, mode);
  const bool single = kind == GMX_K_SINGLE, wide = kind == GMX_K_WIDE, stock = kind == GMX_K_STOCK;
  if (a.out_last && (single || wide)) return GMX_ERR_INVALID;  // (gmx_batch_create refuses the flag for these shapes)
  if (single)
    HIPCHK(gmx_launch_single_kernel(g->topo_dev, &a, g->topo.n, g->single_variant, g->stream));
  else if (wide)
    HIPCHK(gmx_launch_wide_kernel(g->topo_dev, &a, ns, a.mask != nullptr, g->topo.n, g->stream));
  else if (stock)
    HIPCHK(gmx_launch_stock_kernel(g->topo_dev, &a, ns, GMX_STK_LDS_BYTES(g->topo.lds_misc),
                                   a.mask != nullptr, g->stock_staged, topo_tables_pow2(g->topo), g->stream));
  else
    HIPCHK(gmx_launch_bank_kernel(g->topo_dev, &a, ns, g->lds_bytes, a.mask != nullptr, g->topo.l0,
                                  g->topo.l1, g->topo.n_skip, g->topo.has_final,
                                  g->topo.mx[g->topo.l0 - 1].stride, g->stream));
  HIPCHK(hipEventRecord(dec->done, g->stream));
  dec->busy = true;
  if (n_list) {
    int rcl = count_list_used(g->counts, g->stream);
    if (rcl) return rcl;
  }
  {
    int rcn = batch_note_device_use(b);
    if (rcn) return rcn;
  }
  if (kernel_ms) {
    HIPCHK(hipEventRecord(g->ev1, g->stream));
    HIPCHK(hipEventSynchronize(g->ev1));
    HIPCHK(hipEventElapsedTime(kernel_ms, g->ev0, g->ev1));
  }
  if (mode & GMX_MODE_LEARN)
    for (int i = 0; i < ns; ++i) g->steps[s0 + i] += n_list ? n_list[i] : T;
  return GMX_OK;
}

// ---- batches ---------------------------------------------------------------------------
static void batch_free(gmx_batch* b) {
  if (!b) return;
  if (b->g) {
    (void)hipSetDevice(b->g->device);
    if (b->g->stream) (void)hipStreamSynchronize(b->g->stream);
    if (b->g->up_stream) (void)hipStreamSynchronize(b->g->up_stream);
    if (b->g->down_stream) (void)hipStreamSynchronize(b->g->down_stream);
    auto& v = b->g->batches;
    v.erase(std::remove(v.begin(), v.end(), b), v.end());
  }
  void* dv[] = {b->d_pred, b->d_mask, b->d_ctx, b->d_bits, b->d_p, b->d_out, b->d_last,
                b->d_rng, b->d_tcount, b->d_pstate, b->d_cstate};
  for (void* p : dv)
    if (p) (void)hipFree(p);
  void* hv[] = {b->h_pred, b->h_mask, b->h_ctx, b->h_bits, b->h_p, b->h_out, b->h_last};
  for (void* p : hv)
    if (p) (void)hipHostFree(p);
  xfer_free(b->x);
  delete b;
}

static int batch_alloc(gmx_batch** out, gmx_group* g, int S, uint64_t max_bits, unsigned flags) {
  if (!out || !g || max_bits == 0) return GMX_ERR_INVALID;
  *out = nullptr;
  HIPCHK(hipSetDevice(g->device));
  gmx_batch* b = new (std::nothrow) gmx_batch();
  if (!b) return GMX_ERR_NOMEM;
  b->g = g;
  b->S = S;
  b->max_bits = max_bits;
  b->flags = flags;
  const GmxTopoDev& t = g->topo;
  const size_t R = (size_t)S * max_bits;
#define BCHK(call)                                 \
  do {                                             \
    hipError_t e_ = (call);                        \
    if (e_ != hipSuccess) {                        \
      int r_ = hip_fail(e_, #call);                \
      batch_free(b);                               \
      return e_ == hipErrorOutOfMemory ? GMX_ERR_NOMEM : r_; \
    }                                              \
  } while (0)
  BCHK(hipMalloc((void**)&b->d_pred, R * t.n_pad * sizeof(float)));
  if (flags & GMX_BATCH_MASK) BCHK(hipMalloc((void**)&b->d_mask, R * t.mask_words * sizeof(uint32_t)));
  // (+ one record: gmx_stock_kernel's plain build reads, and ignores, the context word behind its last bit)
  BCHK(hipMalloc((void**)&b->d_ctx, (R + 1) * t.m * sizeof(uint32_t)));
  BCHK(hipMalloc((void**)&b->d_bits, R));
  BCHK(hipMalloc((void**)&b->d_p, R * sizeof(float)));
  if (flags & GMX_BATCH_OUTPUTS) BCHK(hipMalloc((void**)&b->d_out, R * t.m * sizeof(float)));
  if (flags & GMX_BATCH_LAST_OUTPUTS) {
    BCHK(hipMalloc((void**)&b->d_last, (size_t)S * t.m * sizeof(float)));
    BCHK(hipMemset(b->d_last, 0, (size_t)S * t.m * sizeof(float)));
  }
#undef BCHK
  {
    int rcx = xfer_init(b->x);
    if (rcx) {
      batch_free(b);
      return rcx;
    }
  }
  g->batches.push_back(b);
  *out = b;
  return GMX_OK;
}

extern "C" int gmx_batch_create(gmx_batch** out, gmx_group* g, uint64_t max_bits, unsigned flags) {
  if (!g) return GMX_ERR_INVALID;
  if (flags & GMX_BATCH_LAST_OUTPUTS) {  // the one-mixer and lane-pair kernels do not keep them
    const GmxKernelKind kind = kernel_for(g, GMX_MODE_PREDICT | GMX_MODE_LEARN);
    if (kind == GMX_K_SINGLE || kind == GMX_K_WIDE) return GMX_ERR_INVALID;
  }
  int rc = batch_alloc(out, g, g->S, max_bits, flags);
  if (rc) return rc;
  return xfer_streams_ready(g->stream, &g->up_stream, &g->down_stream);
}

extern "C" void gmx_batch_destroy(gmx_batch* b) { batch_free(b); }
extern "C" int gmx_batch_n_pad(const gmx_batch* b) { return (b && b->g) ? b->g->topo.n_pad : GMX_ERR_INVALID; }
extern "C" int gmx_batch_mask_words(const gmx_batch* b) { return (b && b->g) ? b->g->topo.mask_words : GMX_ERR_INVALID; }
extern "C" uint64_t gmx_batch_max_bits(const gmx_batch* b) { return b ? b->max_bits : 0; }

template <typename T>
static T* lazy_host(gmx_batch* b, T** slot, size_t count) {
  if (!*slot) {
    if (hipSetDevice(b->g->device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, count * sizeof(T), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    memset(p, 0, count * sizeof(T));
    *slot = (T*)p;
  }
  return *slot;
}

extern "C" float* gmx_batch_predictions(gmx_batch* b) {
  return (b && b->g) ? lazy_host(b, &b->h_pred, (size_t)b->S * b->max_bits * b->g->topo.n_pad) : nullptr;
}
extern "C" uint32_t* gmx_batch_active_mask(gmx_batch* b) {
  if (!b || !b->g || !(b->flags & GMX_BATCH_MASK)) return nullptr;
  return lazy_host(b, &b->h_mask, (size_t)b->S * b->max_bits * b->g->topo.mask_words);
}
extern "C" uint32_t* gmx_batch_contexts(gmx_batch* b) {
  return (b && b->g) ? lazy_host(b, &b->h_ctx, (size_t)b->S * b->max_bits * b->g->topo.m) : nullptr;
}
extern "C" uint8_t* gmx_batch_bits(gmx_batch* b) {
  return (b && b->g) ? lazy_host(b, &b->h_bits, (size_t)b->S * b->max_bits) : nullptr;
}
extern "C" const float* gmx_batch_p(gmx_batch* b) {
  return (b && b->g) ? lazy_host(b, &b->h_p, (size_t)b->S * b->max_bits) : nullptr;
}
extern "C" const float* gmx_batch_last_outputs(gmx_batch* b) {
  if (!b || !b->g || !(b->flags & GMX_BATCH_LAST_OUTPUTS)) return nullptr;
  return lazy_host(b, &b->h_last, (size_t)b->S * b->g->topo.m);
}
extern "C" const float* gmx_batch_outputs(gmx_batch* b) {
  if (!b || !b->g || !(b->flags & GMX_BATCH_OUTPUTS)) return nullptr;
  return lazy_host(b, &b->h_out, (size_t)b->S * b->max_bits * b->g->topo.m);
}

// Copy the first n_bits records of every stream (rows of a [S][max_bits][w] array).
static hipError_t copy_rows(void* dst, const void* src, size_t elem_bytes, size_t w, gmx_batch* b,
                            uint64_t n_bits, hipMemcpyKind kind, hipStream_t st) {
  const size_t pitch = (size_t)b->max_bits * w * elem_bytes;
  const size_t width = (size_t)n_bits * w * elem_bytes;
  if (n_bits == b->max_bits || b->S == 1)
    return hipMemcpyAsync(dst, src, b->S == 1 ? width : pitch * b->S, kind, st);
  return hipMemcpy2DAsync(dst, pitch, src, pitch, width, (size_t)b->S, kind, st);
}

// Device-side work on the group's stream has just been queued that reads or writes b's arrays.
static int batch_note_device_use(gmx_batch* b) { return xfer_note_device_use(b->x, b->g->stream); }

// The copies run on the group's upload stream: behind the last kernel that used this batch's
// arrays, beside whatever else the group's stream is running (the kernel of ANOTHER batch, in a
// double-buffered loop).  Everything queued on the group's stream after this call sees the records.
extern "C" int gmx_batch_upload(gmx_batch* b, uint64_t n_bits) {
  if (!b || !b->g || n_bits > b->max_bits) return GMX_ERR_INVALID;
  if (n_bits == 0) return GMX_OK;
  gmx_group* g = b->g;
  const GmxTopoDev& t = g->topo;
  HIPCHK(hipSetDevice(g->device));
  if (!gmx_batch_predictions(b) || !gmx_batch_contexts(b) || !gmx_batch_bits(b)) return GMX_ERR_NOMEM;
  if ((b->flags & GMX_BATCH_MASK) && !gmx_batch_active_mask(b)) return GMX_ERR_NOMEM;
  // (small transfers stay on the group's stream: the cross-stream hand-shakes would cost more than
  // the copies could ever overlap)
  const size_t bytes = (size_t)b->S * n_bits * ((size_t)t.n_pad * 4 + (size_t)t.m * 4 + 1 +
                                                ((b->flags & GMX_BATCH_MASK) ? (size_t)t.mask_words * 4 : 0));
  hipStream_t st = nullptr;
  int rc = xfer_begin_upload(b->x, g->stream, &g->up_stream, bytes, &st);
  if (rc) return rc;
  HIPCHK(copy_rows(b->d_pred, b->h_pred, 4, t.n_pad, b, n_bits, hipMemcpyHostToDevice, st));
  if (b->flags & GMX_BATCH_MASK)
    HIPCHK(copy_rows(b->d_mask, b->h_mask, 4, t.mask_words, b, n_bits, hipMemcpyHostToDevice, st));
  HIPCHK(copy_rows(b->d_ctx, b->h_ctx, 4, t.m, b, n_bits, hipMemcpyHostToDevice, st));
  HIPCHK(copy_rows(b->d_bits, b->h_bits, 1, 1, b, n_bits, hipMemcpyHostToDevice, st));
  return xfer_end_upload(b->x, g->stream, st);
}

// Behind everything queued on the group's stream so far, on the group's download stream.
extern "C" int gmx_batch_download(gmx_batch* b, uint64_t n_bits) {
  if (!b || !b->g || n_bits > b->max_bits) return GMX_ERR_INVALID;
  if (n_bits == 0) return GMX_OK;
  gmx_group* g = b->g;
  const GmxTopoDev& t = g->topo;
  HIPCHK(hipSetDevice(g->device));
  if (!gmx_batch_p(b)) return GMX_ERR_NOMEM;
  if ((b->flags & GMX_BATCH_OUTPUTS) && !gmx_batch_outputs(b)) return GMX_ERR_NOMEM;
  if ((b->flags & GMX_BATCH_LAST_OUTPUTS) && !gmx_batch_last_outputs(b)) return GMX_ERR_NOMEM;
  const size_t bytes = (size_t)b->S * n_bits * (4 + ((b->flags & GMX_BATCH_OUTPUTS) ? (size_t)t.m * 4 : 0));
  hipStream_t st = nullptr;
  int rc = xfer_begin_download(b->x, g->stream, &g->down_stream, bytes, &st);
  if (rc) return rc;
  HIPCHK(copy_rows(b->h_p, b->d_p, 4, 1, b, n_bits, hipMemcpyDeviceToHost, st));
  if (b->flags & GMX_BATCH_OUTPUTS)
    HIPCHK(copy_rows(b->h_out, b->d_out, 4, t.m, b, n_bits, hipMemcpyDeviceToHost, st));
  if (b->flags & GMX_BATCH_LAST_OUTPUTS)
    HIPCHK(hipMemcpyAsync(b->h_last, b->d_last, (size_t)b->S * t.m * sizeof(float), hipMemcpyDeviceToHost, st));
  return xfer_end_download(b->x, st);
}

// The host waits for THIS batch's queued work: its upload, the device work that used it, its
// download.  (Work on other batches of the group keeps running.)
extern "C" int gmx_batch_wait(gmx_batch* b) {
  if (!b || !b->g) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(b->g->device));
  return xfer_wait(b->x);
}

extern "C" int gmx_batch_fill_synthetic(gmx_batch* b, uint64_t n_bits, uint64_t seed,
                                        uint64_t restart, int ctx_mode, uint32_t ctx_mod,
                                        uint32_t zero_mod, int bit_mode) {
  if (!b || !b->g || n_bits > b->max_bits) return GMX_ERR_INVALID;
  const GmxTopoDev& t = b->g->topo;
  HIPCHK(hipSetDevice(b->g->device));
  if (!b->d_rng) {
    if (!restart) return GMX_ERR_STATE;
    HIPCHK(hipMalloc((void**)&b->d_rng, (size_t)b->S * 8));
    HIPCHK(hipMalloc((void**)&b->d_tcount, (size_t)b->S * 8));
    HIPCHK(hipMalloc((void**)&b->d_pstate, (size_t)b->S * t.n_pad * 4));
    HIPCHK(hipMalloc((void**)&b->d_cstate, (size_t)b->S * t.m * 4));
  }
  GmxSynthArgs a;
  memset(&a, 0, sizeof a);
  a.pred = b->d_pred;
  a.mask = (b->flags & GMX_BATCH_MASK) ? b->d_mask : nullptr;
  a.ctx = b->d_ctx;
  a.bits = b->d_bits;
  a.rng = b->d_rng;
  a.tcount = b->d_tcount;
  a.pstate = b->d_pstate;
  a.cstate = b->d_cstate;
  a.rec_stride = b->max_bits;
  a.n_bits = n_bits;
  a.seed = seed ? seed : 0x9E3779B97F4A7C15ull;
  a.n = t.n;
  a.n_pad = t.n_pad;
  a.m = t.m;
  a.mask_words = t.mask_words;
  a.n_streams = b->S;
  a.restart = restart ? 1 : 0;
  a.ctx_mode = ctx_mode;
  a.bit_mode = bit_mode;
  a.ctx_mod = ctx_mod;
  a.zero_mod = zero_mod;
  HIPCHK(gmx_launch_synth_kernel(&a, b->g->stream));
  return batch_note_device_use(b);
}

extern "C" int gmx_group_run(gmx_group* g, gmx_batch* b, uint64_t n_bits, int learn, float* kernel_ms) {
  if (!g || !b || b->g != g || b->S != g->S || n_bits > b->max_bits) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, false);
  if (rc) return rc;
  std::fill(g->fwd_done.begin(), g->fwd_done.end(), 0);
  return launch_run(g, b, 0, 0, g->S, n_bits, GMX_MODE_PREDICT | (learn ? GMX_MODE_LEARN : 0u),
                    kernel_ms);
}

// Streams of different lengths in one call (many files compressed side by side end at different bits):
// maximal runs of neighbouring streams with the same count share a launch, a stream with 0 bits sits out.
extern "C" int gmx_group_run_ragged(gmx_group* g, gmx_batch* b, const uint64_t* n_bits, int learn) {
  if (!g || !b || b->g != g || b->S != g->S || !n_bits) return GMX_ERR_INVALID;
  uint64_t maxn = 0;
  bool same = true;
  for (int s = 0; s < g->S; ++s) {
    if (n_bits[s] > b->max_bits) return GMX_ERR_INVALID;
    maxn = std::max(maxn, n_bits[s]);
    same = same && n_bits[s] == n_bits[0];
  }
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, false);
  if (rc) return rc;
  std::fill(g->fwd_done.begin(), g->fwd_done.end(), 0);
  const unsigned mode = GMX_MODE_PREDICT | (learn ? GMX_MODE_LEARN : 0u);
  if (same) return launch_run(g, b, 0, 0, g->S, maxn, mode, nullptr);
  // The kernels with one stream per block take a count per block: ONE launch whatever the lengths (finished
  // files leave holes among the streams; a launch per run of equal neighbours would put the runs one behind the
  // other, each at a whole kernel's latency).  The one-mixer and lane-pair kernels run the runs.
  const GmxKernelKind kind = kernel_for(g, mode);
  static const bool split = getenv("GMX_RAGGED_SPLIT") != nullptr;  // debugging: a launch per run of equal neighbours
  if (!split && (kind == GMX_K_STOCK || kind == GMX_K_BANK)) return launch_run(g, b, 0, 0, g->S, maxn, mode, nullptr, n_bits);
  for (int s0 = 0; s0 < g->S;) {
    int s1 = s0 + 1;
    while (s1 < g->S && n_bits[s1] == n_bits[s0]) ++s1;
    rc = launch_run(g, b, s0, s0, s1 - s0, n_bits[s0], mode, nullptr);
    if (rc) return rc;
    s0 = s1;
  }
  return GMX_OK;
}

// ---- per-bit surface -------------------------------------------------------------------
// One 1-bit record per stream lives in g->one (a batch with max_bits = 1 over all streams):
// forward writes stream s's record and latches its outputs, learn re-reads both.
static int ensure_one(gmx_group* g) {
  if (g->one) return GMX_OK;
  int rc = batch_alloc(&g->one, g, g->S, 1, GMX_BATCH_OUTPUTS | GMX_BATCH_MASK);
  if (rc) return rc;
  if (!gmx_batch_predictions(g->one) || !gmx_batch_active_mask(g->one) || !gmx_batch_contexts(g->one) ||
      !gmx_batch_bits(g->one) || !gmx_batch_p(g->one) || !gmx_batch_outputs(g->one))
    return GMX_ERR_NOMEM;
  return GMX_OK;
}

extern "C" int gmx_bank_forward(gmx_group* g, int stream, const float* predictions,
                                const int32_t* active_models, int n_active, const uint32_t* contexts,
                                float* p_final, float* out_all) {
  if (!g || stream < 0 || stream >= g->S || !predictions || !contexts) return GMX_ERR_INVALID;
  if (n_active > 0 && !active_models) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  int rc;
  if (g->use_sessions && group_is_stock(g)) {
    rc = session_forward(g, stream, predictions, active_models, n_active, contexts, p_final, out_all);
    if (rc == GMX_OK) g->fwd_done[stream] = 2;
    if (rc != GMX_ERR_STATE) return rc;  // GMX_ERR_STATE: no session slot, two launches instead
  }
  rc = sessions_close(g, true);
  if (rc) return rc;
  rc = ensure_one(g);
  if (rc) return rc;
  gmx_batch* b = g->one;
  const GmxTopoDev& t = g->topo;
  const size_t s = (size_t)stream;
  HIPCHK(hipStreamSynchronize(g->stream));
  float* hp = b->h_pred + s * t.n_pad;
  uint32_t* hm = b->h_mask + s * t.mask_words;
  uint32_t* hc = b->h_ctx + s * t.m;
  memcpy(hp, predictions, t.n * sizeof(float));
  for (int i = t.n; i < t.n_pad; ++i) hp[i] = 0.f;
  if (n_active < 0) {
    for (int w = 0; w < t.mask_words; ++w) hm[w] = 0xffffffffu;
  } else {
    for (int w = 0; w < t.mask_words; ++w) hm[w] = 0;
    for (int i = 0; i < n_active; ++i) {
      int idx = active_models[i];
      if (idx < 0 || idx >= t.n) return GMX_ERR_INVALID;
      hm[idx >> 5] |= 1u << (idx & 31);
    }
  }
  memcpy(hc, contexts, t.m * sizeof(uint32_t));
  HIPCHK(hipMemcpyAsync(b->d_pred + s * t.n_pad, hp, t.n_pad * 4, hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipMemcpyAsync(b->d_mask + s * t.mask_words, hm, t.mask_words * 4, hipMemcpyHostToDevice,
                        g->stream));
  HIPCHK(hipMemcpyAsync(b->d_ctx + s * t.m, hc, t.m * 4, hipMemcpyHostToDevice, g->stream));
  rc = launch_run(g, b, stream, stream, 1, 1, GMX_MODE_PREDICT | GMX_MODE_LATCH, nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(b->h_p + s, b->d_p + s, 4, hipMemcpyDeviceToHost, g->stream));
  if (out_all)
    HIPCHK(hipMemcpyAsync(b->h_out + s * t.m, b->d_out + s * t.m, t.m * 4, hipMemcpyDeviceToHost,
                          g->stream));
  HIPCHK(hipStreamSynchronize(g->stream));
  if (p_final) *p_final = b->h_p[s];
  if (out_all) memcpy(out_all, b->h_out + s * t.m, t.m * sizeof(float));
  g->fwd_done[stream] = 1;
  return GMX_OK;
}

extern "C" int gmx_bank_learn(gmx_group* g, int stream, int bit) {
  if (!g || stream < 0 || stream >= g->S || (bit != 0 && bit != 1)) return GMX_ERR_INVALID;
  if (!g->fwd_done[stream]) return GMX_ERR_STATE;
  HIPCHK(hipSetDevice(g->device));
  if (g->fwd_done[stream] == 2) {
    int rcs = session_learn(g, stream, bit);
    if (rcs == GMX_OK) g->fwd_done[stream] = 0;
    return rcs;
  }
  if (!g->one) return GMX_ERR_STATE;
  gmx_batch* b = g->one;
  HIPCHK(hipStreamSynchronize(g->stream));
  b->h_bits[stream] = (uint8_t)bit;
  HIPCHK(hipMemcpyAsync(b->d_bits + stream, b->h_bits + stream, 1, hipMemcpyHostToDevice, g->stream));
  int rc = launch_run(g, b, stream, stream, 1, 1, GMX_MODE_LEARN, nullptr);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(g->stream));
  g->fwd_done[stream] = 0;
  return GMX_OK;
}

#include "gmx_lockstep.inc"

// ---- persistence -------------------------------------------------------------------------
static int fetch_bank(gmx_group* g, int stream, std::vector<uint8_t>& img) {
  img.resize(g->topo.bank_bytes);
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, true);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(g->stream));
  HIPCHK(hipMemcpy(img.data(), g->banks + (size_t)stream * g->topo.bank_bytes, img.size(),
                   hipMemcpyDeviceToHost));
  return GMX_OK;
}

extern "C" int gmx_bank_export(gmx_group* g, int stream, void* long_buf, size_t* long_bytes,
                               void* short_buf, size_t* short_bytes) {
  if (!g || stream < 0 || stream >= g->S || !long_bytes || !short_bytes) return GMX_ERR_INVALID;
  const GmxTopoDev& t = g->topo;
  std::vector<uint8_t> img;
  int rc = fetch_bank(g, stream, img);
  if (rc) return rc;
  // Mixer::WriteToDisk x m (mixer.cpp:178-182)
  const size_t need_short = (size_t)t.m * 24;
  // mixer section of LongTermMemory::WriteToDisk (long-term-memory.cpp:35-55)
  size_t need_long = 0;
  for (int j = 0; j < t.m; ++j) {
    need_long += 8;
    for (uint32_t r = 0; r < t.mx[j].table_size; ++r)
      if (*GMX_RS_PTR(img.data(), t.mx[j], r)) need_long += 12 + 4 * (size_t)t.mx[j].weight_size;
  }
  const bool fits = long_buf && short_buf && *long_bytes >= need_long && *short_bytes >= need_short;
  *long_bytes = need_long;
  *short_bytes = need_short;
  if (!long_buf && !short_buf) return GMX_OK;
  if (!fits) return GMX_ERR_INVALID;
  memcpy(short_buf, img.data() + t.scal_off, need_short);
  uint8_t* o = (uint8_t*)long_buf;
  for (int j = 0; j < t.m; ++j) {
    const GmxMixerDev& x = t.mx[j];
    const float* w = (const float*)(img.data() + x.w_off);
    uint32_t cnt = 0;
    for (uint32_t r = 0; r < x.table_size; ++r)
      if (*GMX_RS_PTR(img.data(), x, r)) ++cnt;
    uint32_t input_size = cnt ? x.weight_size : 0;
    memcpy(o, &cnt, 4);
    memcpy(o + 4, &input_size, 4);
    o += 8;
    for (uint32_t r = 0; r < x.table_size; ++r) {
      const uint64_t steps = *GMX_RS_PTR(img.data(), x, r);
      if (!steps) continue;
      memcpy(o, &r, 4);
      memcpy(o + 4, &steps, 8);
      memcpy(o + 12, w + (size_t)r * x.stride, 4 * (size_t)x.weight_size);
      o += 12 + 4 * (size_t)x.weight_size;
    }
  }
  return GMX_OK;
}

extern "C" int gmx_bank_import(gmx_group* g, int stream, const void* long_buf, size_t long_bytes,
                               const void* short_buf, size_t short_bytes) {
  if (!g || stream < 0 || stream >= g->S || !long_buf || !short_buf) return GMX_ERR_INVALID;
  const GmxTopoDev& t = g->topo;
  if (short_bytes != (size_t)t.m * 24) return GMX_ERR_FORMAT;
  std::vector<uint8_t> img(t.bank_bytes, 0);
  memcpy(img.data() + t.scal_off, short_buf, short_bytes);
  const uint64_t* sc = (const uint64_t*)(img.data() + t.scal_off);
  for (int j = 1; j < t.m; ++j)
    if (sc[3 * j] != sc[0]) return GMX_ERR_FORMAT;  // every Mixer learns on every bit: steps_ agree
  // LongTermMemory::ReadFromDisk, mixer section (long-term-memory.cpp:134-149)
  const uint8_t* p = (const uint8_t*)long_buf;
  const uint8_t* end = p + long_bytes;
  for (int j = 0; j < t.m; ++j) {
    const GmxMixerDev& x = t.mx[j];
    if (end - p < 8) return GMX_ERR_FORMAT;
    uint32_t cnt, input_size;
    memcpy(&cnt, p, 4);
    memcpy(&input_size, p + 4, 4);
    p += 8;
    if (cnt > x.table_size || (cnt && input_size != x.weight_size)) return GMX_ERR_FORMAT;
    float* w = (float*)(img.data() + x.w_off);
    for (uint32_t i = 0; i < cnt; ++i) {
      if ((size_t)(end - p) < 12 + 4 * (size_t)input_size) return GMX_ERR_FORMAT;
      uint32_t r;
      memcpy(&r, p, 4);
      if (r >= x.table_size) return GMX_ERR_FORMAT;
      uint64_t* const rs = GMX_RS_PTR(img.data(), x, r);
      memcpy(rs, p + 4, 8);
      if (*rs == 0) return GMX_ERR_FORMAT;  // a stored row has been learned at least once
      memcpy(w + (size_t)r * x.stride, p + 12, 4 * (size_t)input_size);
      p += 12 + 4 * (size_t)input_size;
    }
  }
  if (p != end) return GMX_ERR_FORMAT;
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, true);
  if (rc) return rc;
  if (stream < (int)g->sessions.size() && g->sessions[stream]) g->sessions[stream]->fwd_live = false;
  HIPCHK(hipStreamSynchronize(g->stream));
  HIPCHK(hipMemcpy(g->banks + (size_t)stream * t.bank_bytes, img.data(), img.size(),
                   hipMemcpyHostToDevice));
  g->steps[stream] = sc[0];
  g->fwd_done[stream] = 0;
  for (gmx_lockstep* ls : g->locksteps) ls->predicted = false;
  return GMX_OK;
}

extern "C" int gmx_bank_copy(gmx_group* dst, int dst_stream, gmx_group* src, int src_stream) {
  if (!dst || !src || dst_stream < 0 || dst_stream >= dst->S || src_stream < 0 || src_stream >= src->S)
    return GMX_ERR_INVALID;
  const GmxTopoDev &a = dst->topo, &b = src->topo;
  if (a.m != b.m || a.n != b.n || a.n_skip != b.n_skip || a.bank_bytes != b.bank_bytes)
    return GMX_ERR_INVALID;
  for (int j = 0; j < a.m; ++j)
    if (a.mx[j].table_size != b.mx[j].table_size || a.mx[j].weight_size != b.mx[j].weight_size ||
        a.mx[j].layer != b.mx[j].layer)
      return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(src->device));
  int rc = sessions_close(src, true);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(src->stream));
  HIPCHK(hipSetDevice(dst->device));
  rc = sessions_close(dst, true);
  if (rc) return rc;
  if (dst_stream < (int)dst->sessions.size() && dst->sessions[dst_stream])
    dst->sessions[dst_stream]->fwd_live = false;
  HIPCHK(hipStreamSynchronize(dst->stream));
  HIPCHK(hipMemcpy(dst->banks + (size_t)dst_stream * a.bank_bytes,
                   src->banks + (size_t)src_stream * b.bank_bytes, a.bank_bytes,
                   hipMemcpyDeviceToDevice));
  dst->steps[dst_stream] = src->steps[src_stream];
  dst->fwd_done[dst_stream] = 0;
  for (gmx_lockstep* ls : dst->locksteps) ls->predicted = false;
  return GMX_OK;
}

extern "C" int gmx_bank_memory_usage(gmx_group* g, int stream, int mixer, uint64_t* bytes) {
  if (!g || stream < 0 || stream >= g->S || mixer < 0 || mixer >= g->topo.m || !bytes)
    return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, true);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(g->stream));
  uint64_t sc[3];
  HIPCHK(hipMemcpy(sc, g->banks + (size_t)stream * g->topo.bank_bytes + g->topo.scal_off + 24u * mixer,
                   24, hipMemcpyDeviceToHost));
  // Mixer::GetMemoryUsage (mixer.cpp:197-205)
  const GmxMixerDev& x = g->topo.mx[mixer];
  *bytes = 29 + sc[2] * (uint64_t)(x.weight_size * 4 + 12) + 8ull * x.table_size;
  return GMX_OK;
}


#include "gmx_indirect.inc"
#include "gmx_lstm.inc"
#include "gmx_chainstep.inc"

// ---- test probes (device math against host math; not part of the product surface) --------
extern "C" int gmx_debug_single_variant(gmx_group* g, int lanes_per_stream) {
  if (!g || (lanes_per_stream != 0 && lanes_per_stream != 16 && lanes_per_stream != 32 &&
             lanes_per_stream != 64 && lanes_per_stream != 165 && lanes_per_stream != 325))
    return GMX_ERR_INVALID;
  g->single_variant = lanes_per_stream;
  return GMX_OK;
}

// Batched runs of the stock shape: 1 = the lane-pair kernel (gmx_wide.hip), 0 = the generated
// instruction streams (gmx_stock.hip).
extern "C" int gmx_debug_stock_pairs(gmx_group* g, int on) {
  if (!g) return GMX_ERR_INVALID;
  g->stock_pairs = on != 0;
  return GMX_OK;
}

// Decay tables always from the host's libm loop (on != 0), or from the device when a launch covers
// many different bit counts (the default).
extern "C" int gmx_debug_decay_on_host(gmx_group* g, int on) {
  if (!g) return GMX_ERR_INVALID;
  g->decay_on_host = on != 0;
  return GMX_OK;
}

// The device-made decay table for rows starting at steps0[0..U), T entries each, as the kernels
// would see it (unsettled entries already replaced by libm's); *n_unsettled = how many those were.
extern "C" int gmx_debug_decay_table(gmx_group* g, const uint64_t* steps0, int U, uint64_t T, float* out,
                                     uint32_t* n_unsettled) {
  if (!g || !steps0 || U < 1 || T < 1 || !out) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  if (!g->copy_stream) HIPCHK(sibling_stream_create(&g->copy_stream, g->stream));
  gmx_group::DecaySlot tmp;
  std::map<uint64_t, uint32_t> uniq;
  std::vector<uint32_t> row(U);
  for (int u = 0; u < U; ++u) {
    auto it = uniq.find(steps0[u]);
    if (it == uniq.end()) it = uniq.emplace(steps0[u], (uint32_t)uniq.size()).first;
    row[u] = it->second;
  }
  HIPCHK(hipMalloc((void**)&tmp.dev, uniq.size() * T * sizeof(float)));
  int rc = decay_table_on_device(g, tmp, uniq, T);
  if (rc == GMX_OK) {
    if (n_unsettled) *n_unsettled = tmp.amb_host[0];
    std::vector<float> h(uniq.size() * T);
    hipError_t e = hipMemcpyAsync(h.data(), tmp.dev, h.size() * sizeof(float), hipMemcpyDeviceToHost, g->copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->copy_stream);
    if (e != hipSuccess) rc = hip_fail(e, "gmx_debug_decay_table copy");
    else
      for (int u = 0; u < U; ++u) memcpy(out + (size_t)u * T, h.data() + (size_t)row[u] * T, T * sizeof(float));
  }
  void* dv[] = {tmp.dev, tmp.st_dev, tmp.amb_dev};
  for (void* p : dv)
    if (p) (void)hipFree(p);
  void* hv[] = {tmp.st_host, tmp.amb_host, tmp.patch_host};
  for (void* p : hv)
    if (p) (void)hipHostFree(p);
  return rc;
}

extern "C" int gmx_debug_force_general(gmx_group* g, int on) {
  if (!g) return GMX_ERR_INVALID;
  g->force_general = on != 0;
  return GMX_OK;
}

// Stock kernels: only the masked (exec-per-step) forward chains, which are otherwise the
// fallback for non-finite values.
// Tuning / tests: how gmx_stock_kernel moves layer-0 rows (1: staged through LDS, coalesced; 0: lane-private;
// -1: by the stream count of the launch).  Same floats either way.
extern "C" int gmx_debug_stock_staged(gmx_group* g, int mode) {
  if (!g || mode < -1 || mode > 1) return GMX_ERR_INVALID;
  g->stock_staged = mode;
  return GMX_OK;
}

extern "C" int gmx_debug_stock_exact(gmx_group* g, int on) {
  if (!g) return GMX_ERR_INVALID;
  int rc = sessions_close(g, true);
  if (rc) return rc;
  g->stock_exact = on != 0;
  return GMX_OK;
}

// Per-bit calls as two kernel launches (on != 0: sessions allowed, the default).
extern "C" int gmx_debug_use_sessions(gmx_group* g, int on) {
  if (!g) return GMX_ERR_INVALID;
  int rc = sessions_close(g, true);
  if (rc) return rc;
  for (GmxSession* se : g->sessions)
    if (se && se->fwd_live) return GMX_ERR_STATE;  // not between a forward and its learn
  g->use_sessions = on != 0;
  return GMX_OK;
}

extern "C" int gmx_debug_open_sessions(void) { return g_open_sessions.load(); }

// Where the sessions keep their command blocks (on != 0: device memory when the host can store
// there, the default; 0: pinned host memory).  Returns GMX_OK; *active (nullable) = 1 if the
// session of `stream` exists and has its command block on the device.
extern "C" int gmx_debug_mailbox_on_device(gmx_group* g, int on, int stream, int* active) {
  if (!g) return GMX_ERR_INVALID;
  if (active) {
    GmxSession* se = (stream >= 0 && stream < (int)g->sessions.size()) ? g->sessions[stream] : nullptr;
    *active = (se && se->mc_on_device) ? 1 : 0;
  }
  if ((on != 0) == g->mailbox_on_device) return GMX_OK;
  int rc = sessions_close(g, true);
  if (rc) return rc;
  for (GmxSession* se : g->sessions)
    if (se && se->fwd_live) return GMX_ERR_STATE;  // not between a forward and its learn
  sessions_free(g);
  g->mailbox_on_device = on != 0;
  return GMX_OK;
}

// Wall-clock cost of n Predict+Learn pairs on `stream` through gmx_bank_forward / gmx_bank_learn
// with made-up inputs (it trains the bank: use a scratch group).  ctx_hold = bits a context
// stays (8 = byte-boundary contexts).
extern "C" int gmx_debug_per_bit_latency(gmx_group* g, int stream, int n, int ctx_hold, double* us_per_bit) {
  if (!g || n <= 0 || ctx_hold <= 0 || !us_per_bit) return GMX_ERR_INVALID;
  const GmxTopoDev& t = g->topo;
  std::vector<float> pred(t.n);
  std::vector<uint32_t> ctx(t.m, 0);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < t.n; ++k) pred[k] = (float)((int64_t)(rnd() >> 40) - (1 << 23)) * (1.0f / (1 << 21));
    if (i % ctx_hold == 0)
      for (int k = 0; k < t.m; ++k) ctx[k] = (uint32_t)rnd();
    float p;
    int rc = gmx_bank_forward(g, stream, pred.data(), nullptr, -1, ctx.data(), &p, nullptr);
    if (rc) return rc;
    rc = gmx_bank_learn(g, stream, p > 0.5f ? 1 : 0);
    if (rc) return rc;
  }
  int rc = gmx_group_sync(g);
  if (rc) return rc;
  clock_gettime(CLOCK_MONOTONIC, &b);
  *us_per_bit = ((b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3) / n;
  return GMX_OK;
}

// Compute-unit share of a bank's kernels: its main stream is re-created with a CU mask
// (hipExtStreamCreateWithCUMask; n_words 32-bit words, bit i = CU i), so that kernels of different
// banks that cannot share a SIMD (the 512-register mixer waves, the LSTM workgroups) run side by side
// on disjoint CUs instead of one after the other.  n_words == 0: back to all CUs.
static int stream_with_cu_mask(hipStream_t* st, const uint32_t* mask, int n_words) {
  hipStream_t fresh = nullptr;
  if (n_words > 0)
    HIPCHK(hipExtStreamCreateWithCUMask(&fresh, (uint32_t)n_words, mask));
  else if (*st)
    HIPCHK(sibling_stream_create(&fresh, *st));  // (all CUs again: back on the bank's own priority level)
  else
    HIPCHK(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
  if (*st) {
    (void)hipStreamSynchronize(*st);
    (void)hipStreamDestroy(*st);
  }
  *st = fresh;
  return GMX_OK;
}
extern "C" int gmx_group_set_cu_mask(gmx_group* g, const uint32_t* mask, int n_words) {
  if (!g || n_words < 0 || (n_words > 0 && !mask)) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(g->device));
  int rc = sessions_close(g, false);
  if (rc) return rc;
  if (!g->locksteps.empty()) return GMX_ERR_STATE;  // their graphs are bound to the old stream
  return stream_with_cu_mask(&g->stream, mask, n_words);
}
extern "C" int gmx_indirect_set_cu_mask(gmx_indirect* ib, const uint32_t* mask, int n_words) {
  if (!ib || n_words < 0 || (n_words > 0 && !mask)) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(ib->device));
  {
    int rcs = ind_sessions_close(ib);
    if (rcs) return rcs;
  }
  return stream_with_cu_mask(&ib->stream, mask, n_words);
}

// Tests / tuning: the per-bit surface of an Indirect bank through persistent sessions (1, default) or a
// kernel launch per call (0).  Same floats either way.
extern "C" int gmx_debug_indirect_use_sessions(gmx_indirect* ib, int on) {
  if (!ib) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(ib->device));
  int rc = ind_sessions_close(ib);
  if (rc) return rc;
  ib->use_sessions = on != 0;
  return GMX_OK;
}
// ---- Indirect models -> mixers, one host round trip -------------------------------------------------
// gmx_indirect_forward followed by gmx_bank_forward with the Indirect models' predictions and active flags put
// into the mixers' inputs at the models' slots -- but when both banks answer through per-bit sessions on the
// same device, the wave of the Indirect models hands its results to the mixers' wave itself and rings its
// mailbox (GmxIndMbCmd::chain_*): the host waits once.  `predictions` / `active_models` are the blackboard as
// it stands BEFORE the Indirect models' Predict (their slots are overwritten, their indices ignored if listed).
// The Learn calls are the usual ones (gmx_indirect_learn, gmx_bank_learn): they are only noted and travel
// with the next forward anyway.  Same floats as the two calls.
extern "C" int gmx_chain_forward(gmx_indirect* ib, gmx_group* g, int stream, const uint32_t* ind_contexts,
                                 uint32_t bit_context, const float* predictions, const int32_t* active_models,
                                 int n_active, const uint32_t* contexts, float* p_final, float* out_all,
                                 float* ind_predictions, uint8_t* ind_active) {
  if (!ib || !g || stream < 0 || stream >= ib->S || stream >= g->S || !ind_contexts || !predictions || !contexts)
    return GMX_ERR_INVALID;
  if (n_active > 0 && !active_models) return GMX_ERR_INVALID;
  const GmxTopoDev& t = g->topo;
  const int K = ib->dev.k;
  if (n_active > t.n) return GMX_ERR_INVALID;
  for (int i = 0; i < n_active; ++i)  // before either bank has moved: a bad index must not leave the Indirect models mid-bit
    if (active_models[i] < 0 || active_models[i] >= t.n) return GMX_ERR_INVALID;
  uint32_t own[GMX_MAX_INPUTS / 32] = {0};  // the active-mask bits of the Indirect models' slots
  for (int i = 0; i < K; ++i) {
    const int a = ib->dev.m[i].slot_a, b = ib->dev.m[i].slot_b;
    if (a < 0 || a >= t.n || b < 0 || b >= t.n) return GMX_ERR_INVALID;
    own[a >> 5] |= 1u << (a & 31);
    own[b >> 5] |= 1u << (b & 31);
  }
  HIPCHK(hipSetDevice(g->device));
  float ip_[2 * GMX_IND_MAX_MODELS];  // (no allocation on a per-bit path)
  uint8_t ia_[2 * GMX_IND_MAX_MODELS];
  float* const ipd = ip_;
  uint8_t* const iad = ia_;
  const size_t n2 = (size_t)2 * K;
  if (ib->device == g->device && ib->use_sessions && g->use_sessions && group_is_stock(g) && n_active >= 0) {
    // the mixers' command first (payload in its slot, word decided, doorbell NOT rung) ...
    int rc = session_forward_prepare(g, stream, predictions, active_models, n_active, contexts, own);
    if (rc == GMX_OK) {
      GmxSession* se = g->sessions[stream];
      // ... then the Indirect models' forward, which rings it
      rc = ind_session_forward(ib, stream, ind_contexts, bit_context, ipd, iad, se->word, se->slot, se->mc);
      if (rc == GMX_OK) {
        ib->fwd_done[stream] = 2;
        rc = session_forward_finish(g, stream, p_final, out_all);
        if (rc == GMX_OK) g->fwd_done[stream] = 2;
      } else {
        // the Indirect side did not take the command: ring the mixers ourselves once its inputs are whole
        int rc2 = rc == GMX_ERR_STATE ? gmx_indirect_forward(ib, stream, ind_contexts, bit_context, ipd, iad) : rc;
        if (rc2 == GMX_OK) {
          GmxMbPayload* pay = &se->mc->slot[se->slot];
          for (int i = 0; i < K; ++i) {
            const int sl[2] = {ib->dev.m[i].slot_a, ib->dev.m[i].slot_b};
            for (int h = 0; h < 2; ++h) {
              pay->pred[sl[h]] = ipd[2 * i + h];
              if (iad[2 * i + h]) pay->mask[sl[h] >> 5] = pay->mask[sl[h] >> 5] | (1u << (sl[h] & 31));
            }
          }
          session_ring(se);
          rc2 = session_forward_finish(g, stream, p_final, out_all);
          if (rc2 == GMX_OK) g->fwd_done[stream] = 2;
        } else {
          // the prepared command still goes (it carries the learn gmx_bank_learn noted, and the mailbox
          // protocol expects an answer to the word it was given); its forward is not one to learn from
          session_ring(se);
          if (session_forward_finish(g, stream, nullptr, nullptr) == GMX_OK) se->fwd_live = false;
        }
        rc = rc2;
      }
      if (rc == GMX_OK) {
        if (ind_predictions) memcpy(ind_predictions, ipd, n2 * 4);
        if (ind_active) memcpy(ind_active, iad, n2);
      }
      return rc;
    }
    if (rc != GMX_ERR_STATE) return rc;  // GMX_ERR_STATE: no session slot for the mixers, the two calls instead
  }
  // the two calls, the host in between (fixed-size arrays: no allocation on a per-bit path)
  int rc = gmx_indirect_forward(ib, stream, ind_contexts, bit_context, ipd, iad);
  if (rc) return rc;
  float pr[GMX_MAX_INPUTS];
  int32_t act[GMX_MAX_INPUTS];
  int na = 0;
  memcpy(pr, predictions, (size_t)t.n * sizeof(float));
  uint32_t on[GMX_MAX_INPUTS / 32] = {0};
  if (n_active < 0) {
    for (int idx = 0; idx < t.n; ++idx) on[idx >> 5] |= 1u << (idx & 31);
  } else {
    for (int i = 0; i < n_active; ++i) on[active_models[i] >> 5] |= 1u << (active_models[i] & 31);
  }
  for (int w = 0; w < GMX_MAX_INPUTS / 32; ++w) on[w] &= ~own[w];
  for (int i = 0; i < K; ++i) {
    const int sl[2] = {ib->dev.m[i].slot_a, ib->dev.m[i].slot_b};
    for (int h = 0; h < 2; ++h) {
      pr[sl[h]] = ipd[2 * i + h];
      if (iad[2 * i + h]) on[sl[h] >> 5] |= 1u << (sl[h] & 31);
    }
  }
  for (int idx = 0; idx < t.n; ++idx)  // ascending, like ShortTermMemory::active_models
    if ((on[idx >> 5] >> (idx & 31)) & 1u) act[na++] = idx;
  rc = gmx_bank_forward(g, stream, pr, act, na, contexts, p_final, out_all);
  if (rc) return rc;
  if (ind_predictions) memcpy(ind_predictions, ipd, n2 * 4);
  if (ind_active) memcpy(ind_active, iad, n2);
  return GMX_OK;
}

extern "C" int gmx_lstm_set_cu_mask(gmx_lstm* l, const uint32_t* mask, int n_words) {
  if (!l || n_words < 0 || (n_words > 0 && !mask)) return GMX_ERR_INVALID;
  HIPCHK(hipSetDevice(l->device));
  LSTM_CLOSE_SESSIONS(l);
  int rc = stream_with_cu_mask(&l->stream, mask, n_words);
  if (rc) return rc;
  int n = 0;
  if (n_words > 0) {
    for (int w = 0; w < n_words; ++w) n += __builtin_popcount(mask[w]);
  } else {
    HIPCHK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, l->device));
  }
  l->cus = n;  // (which build of the kernel a launch takes: one workgroup per unit or two, gmx_lstm.hip)
  return GMX_OK;
}

// Wall-clock cost of n lock-step steps (Predict for all streams, the probabilities on the host, Learn
// for all streams) through gmx_lockstep_*, with new gate contexts every ctx_hold steps.
extern "C" int gmx_debug_lockstep_latency(gmx_group* g, int n, int ctx_hold, int fused, double* us_per_step) {
  if (!g || n <= 0 || ctx_hold <= 0 || !us_per_step) return GMX_ERR_INVALID;
  gmx_lockstep* ls = nullptr;
  const bool persistent = (fused & 2) != 0;  // bit 1: GMX_LOCKSTEP_PERSISTENT
  fused &= 1;
  int rc = gmx_lockstep_create(&ls, g, persistent ? GMX_LOCKSTEP_PERSISTENT : 0);
  if (rc) return rc;
  if (persistent && !gmx_lockstep_is_persistent(ls)) {
    gmx_lockstep_destroy(ls);
    return GMX_ERR_STATE;
  }
  gmx_batch* b = gmx_lockstep_batch(ls);
  const GmxTopoDev& t = g->topo;
  float* pred = gmx_batch_predictions(b);
  uint32_t* ctx = gmx_batch_contexts(b);
  uint8_t* bits = gmx_batch_bits(b);
  const float* p = gmx_batch_p(b);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int s = 0; s < g->S; ++s)
    for (int k = 0; k < t.n; ++k)
      pred[(size_t)s * t.n_pad + k] = (float)((int64_t)(rnd() >> 40) - (1 << 23)) * (1.0f / (1 << 21));
  timespec a, e;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < n && rc == GMX_OK; ++i) {
    if (i % ctx_hold == 0)
      for (size_t k = 0; k < (size_t)g->S * t.m; ++k) ctx[k] = (uint32_t)rnd();
    // the first step predicts; every later one learns the bits of the step before and predicts, as
    // one graph (fused == 0: two graphs per step)
    rc = (fused && i > 0) ? gmx_lockstep_learn_predict(ls) : gmx_lockstep_predict(ls);
    if (rc) break;
    for (int s = 0; s < g->S; ++s) bits[s] = p[s] > 0.5f ? 1 : 0;  // stands in for S arithmetic decoders
    if (!fused || i + 1 == n) rc = gmx_lockstep_learn(ls);
  }
  if (rc == GMX_OK) rc = gmx_group_sync(g);
  clock_gettime(CLOCK_MONOTONIC, &e);
  *us_per_step = ((e.tv_sec - a.tv_sec) * 1e6 + (e.tv_nsec - a.tv_nsec) * 1e-3) / n;
  gmx_lockstep_destroy(ls);
  return rc;
}

extern "C" int gmx_debug_math_probe(int device, const float* x, float* y, uint64_t n, int what) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return GMX_ERR_NO_DEVICE;
  }
  HIPCHK(hipSetDevice(device));
  float *dx = nullptr, *dy = nullptr;
  HIPCHK(hipMalloc((void**)&dx, n * 4));
  HIPCHK(hipMalloc((void**)&dy, n * 4));
  HIPCHK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
  HIPCHK(gmx_launch_math_probe(dx, dy, n, what, nullptr));
  HIPCHK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(dx);
  (void)hipFree(dy);
  return GMX_OK;
}

extern "C" int gmx_debug_math_range(int device, uint64_t lo, uint64_t count, int what,
                                    unsigned long long out[2]) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return GMX_ERR_NO_DEVICE;
  }
  HIPCHK(hipSetDevice(device));
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 16));
  HIPCHK(hipMemset(d, 0, 16));
  HIPCHK(gmx_launch_math_range(lo, count, what, d, nullptr));
  HIPCHK(hipMemcpy(out, d, 16, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return GMX_OK;
}
