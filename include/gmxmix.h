/* gmxmix.h -- C ABI of libgmxmix.so: gmix's mixer hot path on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of byronknoll/gmix: the per-bit mixer (the 33
 * `Mixer` objects that `Predictor::AddMixers` builds and that `Predictor::Predict()` /
 * `Perceive()` / `Learn()` run after the feature models).  Everything else in gmix -- the
 * context models, the arithmetic coder, the CLI -- stays on the host and is not part of this
 * library.  Citations are file:line in the reference tree (/root/reference).
 *
 * Object model
 *   gmx_group : S independent mixer banks ("streams") of one topology on one device.  One
 *               bank = everything the last 33 entries of Predictor::models_ own: the
 *               LongTermMemory::mixers tables (long-term-memory.h:27-40) plus each Mixer's
 *               steps_/max_steps_/contexts_seen_ (mixer.h:33-38).  S = 1 is the drop-in for a
 *               single Predictor; S > 1 is the embarrassingly parallel multi-file case.
 *   gmx_batch : pinned host staging + device buffers for up to max_bits bits of per-stream
 *               records {predictions[N], active mask, contexts[M], bit} and the results.
 *
 * Conventions
 *   - every function returns 0 (GMX_OK) or a negative gmx_status; nothing throws or aborts
 *     across this boundary (the reference's Model methods are void and never fail,
 *     model.h:22-37; HIP failures are mapped to GMX_ERR_HIP and the text kept for
 *     gmx_last_error()).
 *   - a group is bound to one HIP stream and is not thread-safe (the reference Predictor is
 *     single-threaded and not re-entrant); distinct groups are independent.
 *   - results are a pure function of (bank state, inputs): the batched and the per-bit entry
 *     points produce identical floats (the encoder/decoder symmetry the reference's tester
 *     relies on, tester.cpp:350-356).
 *   - arithmetic parity: every mixer output, probability, weight and counter equals what the
 *     reference's strict C++ build computes, bit for bit (fp32 left-to-right sums, separate
 *     multiply and add roundings, IEEE divide, glibc's expf restated in gmx_math.h).
 */
#ifndef GMXMIX_H_
#define GMXMIX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gmx_status {
  GMX_OK = 0,
  GMX_ERR_INVALID = -1,      /* bad argument / unsupported topology */
  GMX_ERR_NOMEM = -2,        /* host or device allocation failed */
  GMX_ERR_HIP = -3,          /* a HIP runtime call failed; see gmx_last_error() */
  GMX_ERR_NO_DEVICE = -4,    /* no usable gfx950 device: there is no CPU fallback */
  GMX_ERR_STATE = -5,        /* call order violated (e.g. learn without forward) */
  GMX_ERR_FORMAT = -6        /* malformed checkpoint bytes on import */
} gmx_status;

/* One Mixer constructor call (mixer.h:17-19; the literals of predictor.cpp:251-358):
 * layer 0/1/2, gate-table size (rows), learning rate (the reference narrows a double
 * literal to float at the call).  The gate context itself is per-bit data. */
typedef struct gmx_mixer_desc {
  int32_t layer;
  uint32_t table_size;
  float learning_rate;
} gmx_mixer_desc;

/* What Predictor's constructor fixes before AddMixers runs (predictor.cpp:17-40):
 * num_predictions, models_with_skip_connection (lstm-model.cpp:12-14) and the mixers in
 * construction order: all layer-0 mixers, then layer-1, then at most one final mixer.
 * Limits: n_mixers <= 64, n_skip <= 8, n_inputs <= 2048,
 * layer-1 and final rows <= 64 weights. */
typedef struct gmx_topology {
  int32_t n_inputs;
  int32_t n_skip;
  const int32_t* skip_index;     /* [n_skip] indices into predictions */
  int32_t n_mixers;
  const gmx_mixer_desc* mixers;  /* [n_mixers] */
} gmx_topology;

typedef struct gmx_group gmx_group;
typedef struct gmx_batch gmx_batch;

/* ---- library ------------------------------------------------------------------------ */
const char* gmx_strerror(int status);
const char* gmx_last_error(void);          /* text of the last GMX_ERR_HIP on this thread */
int gmx_device_count(int* count);          /* gfx950 devices visible */
const char* gmx_build_info(void);          /* arch, flags, version */
/* PCI address of a device ("0000:c1:00.0", len >= 13): the key of /sys/bus/pci/devices/<id>/numa_node, so that
 * the host threads feeding a device (the reference's feature models, one Predictor per stream) can be kept on
 * the cores next to it. */
int gmx_device_pci_bus_id(int device, char* buf, size_t len);

/* ---- group: replaces Predictor::AddMixers' 33 objects (predictor.cpp:251-358) ---------- */
int gmx_group_create(gmx_group** out, const gmx_topology* topo, int n_streams, int device);
void gmx_group_destroy(gmx_group* g);
int gmx_group_n_streams(const gmx_group* g);
int gmx_group_n_mixers(const gmx_group* g);
int gmx_group_n_inputs(const gmx_group* g);
uint64_t gmx_group_bank_bytes(const gmx_group* g);   /* device bytes per stream */
int gmx_group_reset(gmx_group* g);                   /* all banks back to the constructed state */
int gmx_group_sync(gmx_group* g);                    /* wait for everything queued on its stream */
/* HIP events on the group's stream: start is recorded behind what is queued so far, stop behind
 * what was queued since; *ms = device time between the two (stop waits for it). */
int gmx_group_timer_start(gmx_group* g);
int gmx_group_timer_stop(gmx_group* g, float* ms);

/* ---- per-bit surface: Predict / Perceive / Learn for one stream ------------------------ */
/* 33 x Mixer::Predict + the final squash of Predictor::Predict (mixer.cpp:51-106,
 * predictor.cpp:366-375).  predictions[n_inputs] is the raw ShortTermMemory::predictions
 * blackboard (stale slots allowed), active_models[n_active] the ascending model indices of
 * ShortTermMemory::active_models (NULL with n_active < 0 = every slot active), contexts[M]
 * the values of the context variables the mixers alias (mixer.h:31) read at call time.
 * *p_final receives the clamped probability; out_all (nullable) the M logit outputs
 * (mixer_layer0_outputs, mixer_layer1_outputs, final_mixer_output). Synchronous. */
int gmx_bank_forward(gmx_group* g, int stream, const float* predictions,
                     const int32_t* active_models, int n_active, const uint32_t* contexts,
                     float* p_final, float* out_all);
/* Predictor::Perceive(bit) + 33 x Mixer::Learn (predictor.cpp:378-387, mixer.cpp:108-176)
 * on the inputs latched by the preceding gmx_bank_forward of that stream.  Optional, like
 * Learn() in the reference (generation never calls it, runner-utils.cpp:199-209). */
int gmx_bank_learn(gmx_group* g, int stream, int bit);

/* ---- batched surface: T bits for every stream in one launch ---------------------------- */
#define GMX_BATCH_OUTPUTS 1u   /* also return all M mixer outputs per bit */
#define GMX_BATCH_MASK 2u      /* records carry an active mask; otherwise every slot is active */
#define GMX_BATCH_LAST_OUTPUTS 8u  /* return the M outputs of each stream's LAST bit of a run only -- what the
                                    * blackboard (mixer_layer0_outputs, ..., final_mixer_output) holds afterwards; a
                                    * compressor needs no more, and the throughput build of the kernel then stores
                                    * nothing else beside the probabilities.  GMX_ERR_INVALID for the shapes that run
                                    * through the one-mixer and lane-pair kernels (use GMX_BATCH_OUTPUTS there). */
int gmx_batch_create(gmx_batch** out, gmx_group* g, uint64_t max_bits, unsigned flags);
void gmx_batch_destroy(gmx_batch* b);
/* Layout of the staging arrays (all stream-major, then bit):
 *   predictions [S][max_bits][n_pad]   n_pad = n_inputs rounded up to 4 (gmx_batch_n_pad)
 *   active_mask [S][max_bits][mask_words] bit i of word i/32 = model i is in active_models
 *   contexts    [S][max_bits][M]
 *   bits        [S][max_bits]
 *   p           [S][max_bits]          clamped probabilities (what Predict() returns)
 *   outputs     [S][max_bits][M]       with GMX_BATCH_OUTPUTS
 *   last_outputs[S][M]                 with GMX_BATCH_LAST_OUTPUTS */
int gmx_batch_n_pad(const gmx_batch* b);
int gmx_batch_mask_words(const gmx_batch* b);
uint64_t gmx_batch_max_bits(const gmx_batch* b);
float* gmx_batch_predictions(gmx_batch* b);     /* pinned host pointers, caller fills/reads */
uint32_t* gmx_batch_active_mask(gmx_batch* b);  /* NULL without GMX_BATCH_MASK */
uint32_t* gmx_batch_contexts(gmx_batch* b);
uint8_t* gmx_batch_bits(gmx_batch* b);
const float* gmx_batch_p(gmx_batch* b);
const float* gmx_batch_outputs(gmx_batch* b);   /* NULL without GMX_BATCH_OUTPUTS */
const float* gmx_batch_last_outputs(gmx_batch* b);  /* [S][M] after gmx_batch_download; NULL without GMX_BATCH_LAST_OUTPUTS */
/* Transfers run on streams of their own, ordered by events against the kernels that use the
 * batch: an upload starts when the last run of THIS batch is done and overlaps whatever runs on
 * other batches (double buffering: run(A); upload(B); run(B); download(A); wait(A); refill A ...);
 * everything queued on the group after gmx_batch_upload returns sees the new records.  A download
 * follows everything queued on the group so far -- except for a batch the host has already waited for
 * (gmx_batch_wait) with nothing queued on it since: that one is fetched at once, whatever the group's stream
 * holds by then.  A caller with several batches in flight fetches that way -- wait(A); download(A); wait(A) -- when
 * A's turn comes, rather than queueing download(A) right behind run(A): a copy that waits for its kernel holds up
 * the copies queued behind it, the uploads of the batches after it among them.  wait = this batch's upload, runs
 * and download.  (The same holds for gmx_ind_batch_* and gmx_lstm_batch_*.)
 * The batch's host arrays are read (upload) and written (download) when the copies execute, not when
 * the calls return: leave them alone between gmx_batch_upload / gmx_batch_download and the
 * gmx_batch_wait that follows. */
int gmx_batch_upload(gmx_batch* b, uint64_t n_bits);    /* async H2D of the first n_bits of every stream */
int gmx_batch_download(gmx_batch* b, uint64_t n_bits);  /* async D2H of p (and outputs) */
int gmx_batch_wait(gmx_batch* b);                       /* host waits for this batch's queued work */
/* Fill the DEVICE record buffers with the next n_bits of the synthetic stream of
 * BASELINE.json configs[1] (xorshift64; logits on the [-4,4] grid, 32-bit contexts, random
 * bits), generated on the GPU.  restart != 0 re-seeds stream s with
 * seed + s * 0x9E3779B97F4A7C15; restart == 0 continues where this batch's previous fill
 * stopped.  ctx_mode/ctx_mod/zero_mod/bit_mode as in oracle/gmx_synth.h. */
int gmx_batch_fill_synthetic(gmx_batch* b, uint64_t n_bits, uint64_t seed, uint64_t restart,
                             int ctx_mode, uint32_t ctx_mod, uint32_t zero_mod, int bit_mode);

/* Predict (+ Perceive + Learn when learn != 0) for bits [0, n_bits) of every stream from the
 * batch's device records; results land in the batch's device buffers.  Asynchronous on the
 * group's stream; kernel_ms (nullable) receives the kernel's duration measured with HIP
 * events on that stream, which makes the call synchronous. */
int gmx_group_run(gmx_group* g, gmx_batch* b, uint64_t n_bits, int learn, float* kernel_ms);

/* The same for streams that stand at different lengths -- S files compressed side by side
 * (runner-utils.cpp:43-67 once per file) end at different bits: stream s runs bits [0, n_bits[s]) of its
 * records, 0 = the stream sits this launch out.  The kernels with one stream per block (the reference's own shape and
 * the general kernel) take the counts as a per-block list: ONE launch whatever the lengths; the one-mixer and lane-pair
 * shapes, which put several streams into a wave, run one launch per stretch of neighbouring streams with equal counts
 * (GMX_RAGGED_SPLIT=1 in the environment forces that for every shape: a debugging aid).  The decay tables of the launch
 * are staged at a pitch of max(n_bits) for every stream.  Asynchronous on the group's stream. */
int gmx_group_run_ragged(gmx_group* g, gmx_batch* b, const uint64_t* n_bits /* [S] */, int learn);

/* ---- persistence ------------------------------------------ */
/* Byte-compatible with the reference: *short_bytes = Mixer::WriteToDisk of every mixer in
 * order (3 x u64 each, mixer.cpp:178-182); *long_bytes = the mixer section of
 * LongTermMemory::WriteToDisk (long-term-memory.cpp:35-55).  Call with NULL buffers to size. */
int gmx_bank_export(gmx_group* g, int stream, void* long_buf, size_t* long_bytes,
                    void* short_buf, size_t* short_bytes);
int gmx_bank_import(gmx_group* g, int stream, const void* long_buf, size_t long_bytes,
                    const void* short_buf, size_t short_bytes);
/* Predictor::Copy for the mixer slice (mixer.cpp:190-195, long-term-memory.cpp:201-214). */
int gmx_bank_copy(gmx_group* dst, int dst_stream, gmx_group* src, int src_stream);
/* Mixer::GetMemoryUsage (mixer.cpp:197-205). */
int gmx_bank_memory_usage(gmx_group* g, int stream, int mixer, uint64_t* bytes);

/* ==== Lock-step surface: all S streams advance one bit per step ================================
 * S decoders on one GPU (coder/decoder.cpp:19-39: a decoder learns its bit from Predict's own
 * result) step together: Predict for all streams, S arithmetic decoders on the host, Learn for all
 * streams.  Each half -- and the pair learn + next predict -- is one hipGraph captured at creation (record uploads, the T = 1 kernel, the
 * download of the S probabilities), so a step costs two graph launches instead of a dozen runtime
 * calls.  Fill the host arrays of gmx_lockstep_batch (gmx_batch_predictions / _active_mask /
 * _contexts: one record per stream), call gmx_lockstep_predict, read gmx_batch_p (and
 * gmx_batch_outputs if created with GMX_BATCH_OUTPUTS), put the coded bits into gmx_batch_bits,
 * call gmx_lockstep_learn (asynchronous -- it works on a private copy of the bits, the array is the
 * caller's again when it returns; skip it for generation, runner-utils.cpp:199-209).
 * Same floats as every other surface.  Destroy before the group. */
/* GMX_LOCKSTEP_PERSISTENT (the reference's own mixer shape, up to 128 streams -- beyond, a graph's bulk copies
 * beat a thousand waves' small reads across the link): no graph launch and no stream synchronisation per step -- S persistent waves poll one doorbell,
 * fetch their records straight from the host arrays and write the probabilities straight back (the arrays
 * are read while gmx_lockstep_predict / _learn_predict run, never after they return).  Same calls, same
 * floats; where it does not apply the flag is ignored and the graphs are used. */
#define GMX_LOCKSTEP_PERSISTENT 4u
typedef struct gmx_lockstep gmx_lockstep;
int gmx_lockstep_create(gmx_lockstep** out, gmx_group* g, unsigned flags /* GMX_BATCH_OUTPUTS | GMX_LOCKSTEP_PERSISTENT */);
/* 1 if the object steps through persistent waves, 0 if through graphs */
int gmx_lockstep_is_persistent(const gmx_lockstep* ls);
void gmx_lockstep_destroy(gmx_lockstep* ls);
gmx_batch* gmx_lockstep_batch(gmx_lockstep* ls);
int gmx_lockstep_predict(gmx_lockstep* ls);
int gmx_lockstep_learn(gmx_lockstep* ls);
/* learn (bits in gmx_batch_bits) and the next predict (records in the host arrays) as one graph: the
 * step of S decoders once their first prediction is out.  Returns with the new probabilities. */
int gmx_lockstep_learn_predict(gmx_lockstep* ls);

/* ==== Indirect models =========================================
 * The producers of 82 of the mixers' 90 inputs: the reference's 41 `Indirect` objects
 * (models/indirect.h:11-34, constructed in predictor.cpp:78-120, :122-185, :210-250) and their
 * IndirectMemory (long-term-memory.h:11-25), for S streams on one device.  With a mixer batch
 * attached to gmx_indirect_run the predictions go from the models to the mixers inside HBM. */

/* One Indirect constructor call: table_size and learning_rate as in indirect.h:16-18; the two
 * prediction indices ShortTermMemory::AddPrediction returned for "<description>-indirect" and
 * "<description>-run_map" (indirect.cpp:10-13). */
typedef struct gmx_indirect_desc {
  uint32_t table_size;       /* the model owns 256 * table_size + 1 states (indirect.cpp:14-19) */
  float learning_rate;
  int32_t slot_indirect;
  int32_t slot_run_map;
} gmx_indirect_desc;

typedef struct gmx_indirect gmx_indirect;
typedef struct gmx_ind_batch gmx_ind_batch;

/* nonstationary_next / run_map_next: [256][2] next-state tables, state-major, of the two state
 * machines ShortTermMemory owns (short-term-memory.h `nonstationary`, `run_map`; the models call
 * their Next(state, bit), indirect.cpp:61-62, :67-68): fill them by calling Next() 512 times. */
int gmx_indirect_create(gmx_indirect** out, const gmx_indirect_desc* models, int n_models,
                        const uint8_t* nonstationary_next, const uint8_t* run_map_next,
                        int n_streams, int device);
void gmx_indirect_destroy(gmx_indirect* ib);
int gmx_indirect_n_streams(const gmx_indirect* ib);
int gmx_indirect_n_models(const gmx_indirect* ib);
uint64_t gmx_indirect_bank_bytes(const gmx_indirect* ib);   /* device bytes per stream */
int gmx_indirect_reset(gmx_indirect* ib);                   /* every table back to "never seen" */
int gmx_indirect_sync(gmx_indirect* ib);

/* Per-bit surface.  forward = n_models x Indirect::Predict (indirect.cpp:28-46): contexts[i] is
 * the value of model i's aliased context variable now, bit_context is
 * ShortTermMemory::bit_context; predictions[2*n_models] / active[2*n_models] (nullable) receive
 * what the blackboard slots of model i hold afterwards ([2i] indirect, [2i+1] run map: a model
 * that stays silent leaves its slot as it was) and whether SetLogitPrediction marked them
 * active (short-term-memory.cpp:193-197).  learn = n_models x Indirect::Learn
 * (indirect.cpp:48-69) with the contexts of the preceding forward.  Like gmx_bank_forward / gmx_bank_learn these
 * run through a persistent per-stream session (one mailbox command per bit, the learn travelling with the
 * next forward) where a session slot is free, else through a kernel launch per call; same floats. */
int gmx_indirect_forward(gmx_indirect* ib, int stream, const uint32_t* contexts, uint32_t bit_context,
                         float* predictions, uint8_t* active);
int gmx_indirect_learn(gmx_indirect* ib, int stream, int bit);

/* The Indirect models' Predict and the mixers' Predict of one bit as ONE call (predictor.cpp:366-368 runs them
 * back to back: the last feature models, then the 33 mixers): gmx_indirect_forward followed by gmx_bank_forward
 * with the models' predictions and active flags put into the mixers' inputs at slot_indirect / slot_run_map.
 * When both banks answer through per-bit sessions on the same device, the Indirect models' wave hands its
 * results to the mixers' wave itself: one host round trip instead of two.  `predictions` / `active_models`
 * (n_active >= 0) are the blackboard WITHOUT the Indirect models (their slots are overwritten, their indices
 * ignored); ind_predictions[2*n_models] / ind_active[2*n_models] (optional) return what gmx_indirect_forward
 * would.  Learn with gmx_indirect_learn and gmx_bank_learn as usual.  Same floats as the two calls. */
int gmx_chain_forward(gmx_indirect* ib, gmx_group* g, int stream, const uint32_t* ind_contexts,
                      uint32_t bit_context, const float* predictions, const int32_t* active_models, int n_active,
                      const uint32_t* contexts, float* p_final, float* out_all, float* ind_predictions,
                      uint8_t* ind_active);

/* Batched surface: records {contexts[n_models], bit_context, bit} of up to max_bits bits per
 * stream, results {predictions[2*n_models], active[2*n_models]} per bit. */
int gmx_ind_batch_create(gmx_ind_batch** out, gmx_indirect* ib, uint64_t max_bits);
void gmx_ind_batch_destroy(gmx_ind_batch* b);
uint64_t gmx_ind_batch_max_bits(const gmx_ind_batch* b);
uint32_t* gmx_ind_batch_contexts(gmx_ind_batch* b);       /* pinned host [S][max_bits][n_models] */
uint32_t* gmx_ind_batch_bit_contexts(gmx_ind_batch* b);   /* [S][max_bits] */
uint8_t* gmx_ind_batch_bits(gmx_ind_batch* b);            /* [S][max_bits] */
const float* gmx_ind_batch_predictions(gmx_ind_batch* b); /* [S][max_bits][2*n_models] */
const uint8_t* gmx_ind_batch_active(gmx_ind_batch* b);    /* [S][max_bits][2*n_models] */
int gmx_ind_batch_upload(gmx_ind_batch* b, uint64_t n_bits);
int gmx_ind_batch_download(gmx_ind_batch* b, uint64_t n_bits);
int gmx_ind_batch_wait(gmx_ind_batch* b);                  /* this batch's upload, runs, download (cf. gmx_batch_wait) */
/* Device-side generator of the synthetic stream of oracle/gmx_ind_synth.h (byte-structured
 * contexts, ctx_mod[4] moduli); stream s is seeded with seed + s * 0x9E3779B97F4A7C15. */
int gmx_ind_batch_fill_synthetic(gmx_ind_batch* b, uint64_t n_bits, uint64_t seed, uint64_t restart,
                                 const uint32_t* ctx_mod);
/* Predict (+ Learn when learn != 0) for bits [0, n_bits) of every stream.  `into` (nullable): a
 * batch of a mixer group with the same number of streams on the same device, created with
 * GMX_BATCH_MASK; the models' predictions are also written into its device prediction records
 * at their slot indices, their active bits replace those slots' bits of its mask records, and
 * the coded bits are copied into its bit records -- ordered after what was queued on the mixer
 * group before this call and before what is queued on it afterwards. */
int gmx_indirect_run(gmx_indirect* ib, gmx_ind_batch* b, uint64_t n_bits, int learn, gmx_batch* into,
                     float* kernel_ms);

/* ... for streams at different lengths (cf. gmx_group_run_ragged): stream s runs bits [0, n_bits[s]). */
int gmx_indirect_run_ragged(gmx_indirect* ib, gmx_ind_batch* b, const uint64_t* n_bits /* [S] */, int learn,
                            gmx_batch* into);

/* The indirect section of LongTermMemory::WriteToDisk / ReadFromDisk (long-term-memory.cpp:8-32,
 * :111-132), byte for byte; NULL buf to size.  Copy = long-term-memory.cpp:193-199;
 * memory_usage = Indirect::GetMemoryUsage (indirect.cpp:71-78). */
int gmx_indirect_export(gmx_indirect* ib, int stream, void* buf, size_t* bytes);
int gmx_indirect_import(gmx_indirect* ib, int stream, const void* buf, size_t bytes);
int gmx_indirect_copy(gmx_indirect* dst, int dst_stream, gmx_indirect* src, int src_stream);
int gmx_indirect_memory_usage(gmx_indirect* ib, int model, uint64_t* bytes);

/* What the two blackboard slots of each model hold ([2i] indirect, [2i+1] run map): ShortTermMemory::predictions at
 * slot_indirect / slot_run_map, which the reference writes with the blackboard (short-term-memory.cpp:4) and which a
 * silent model leaves as they were (indirect.cpp:35-44).  The bank carries them through batches and chain calls, where
 * the host's copy goes stale (no mixer reads a silent slot, so only what is WRITTEN OUT depends on them): a caller that
 * takes a stream from the per-bit surface to the batched one sets them before and gets them after. */
int gmx_indirect_slots_get(gmx_indirect* ib, int stream, float* values /* [2 * n_models] */);
int gmx_indirect_slots_set(gmx_indirect* ib, int stream, const float* values /* [2 * n_models] */);

/* ==== LSTM byte model ==========================================
 * The reference's LstmModel (models/lstm-model.h:12-31): Lstm(256, 256, 50, 1, 100, 0.03, 10) over
 * the PPM byte distribution, predicting the next byte once per byte and the 8 bits from that
 * distribution; Lstm::Perceive learns the output layer every byte and runs back-propagation
 * through time + Adam over the last 100 bytes every 100th byte.  Batches are whole bytes. */
typedef struct gmx_lstm gmx_lstm;
typedef struct gmx_lstm_batch gmx_lstm_batch;

int gmx_lstm_create(gmx_lstm** out, int n_streams, int device);   /* constructed state, gate weights zero */
void gmx_lstm_destroy(gmx_lstm* l);
int gmx_lstm_n_streams(const gmx_lstm* l);
uint64_t gmx_lstm_bank_bytes(const gmx_lstm* l);
int gmx_lstm_reset(gmx_lstm* l);
int gmx_lstm_sync(gmx_lstm* l);
/* Gate weights in the reference's layout, LongTermMemory::neuron_layer_weights[3][50][563]
 * (forget gate, input node, output gate): what LstmLayer's constructor draws from rand()
 * (lstm-layer.cpp:179-194) or a checkpoint holds.  get also returns lstm_output_layer
 * [100][256][51] (nullable): together the LSTM section of LongTermMemory::WriteToDisk
 * (long-term-memory.cpp:57-68). */
int gmx_lstm_set_weights(gmx_lstm* l, int stream, const float* weights);
int gmx_lstm_get_weights(gmx_lstm* l, int stream, float* weights, float* output_layer);

/* Records per byte: ppm[256] = ShortTermMemory::ppm_predictions at the byte boundary
 * (mod_ppmd.cpp:1655-1661), the byte itself; results per bit: what LstmModel::Predict left in its
 * blackboard slot (SetPrediction: the logit; stale when the model stays silent) and whether it was
 * marked active; per byte: ShortTermMemory::lstm_prediction_context (lstm-model.cpp:25-33). */
int gmx_lstm_batch_create(gmx_lstm_batch** out, gmx_lstm* l, uint64_t max_bytes);
void gmx_lstm_batch_destroy(gmx_lstm_batch* b);
float* gmx_lstm_batch_ppm(gmx_lstm_batch* b);                  /* pinned host [S][max_bytes][256] */
uint8_t* gmx_lstm_batch_bytes(gmx_lstm_batch* b);              /* [S][max_bytes] */
const float* gmx_lstm_batch_predictions(gmx_lstm_batch* b);    /* [S][max_bytes][8] */
const uint8_t* gmx_lstm_batch_active(gmx_lstm_batch* b);       /* [S][max_bytes][8] */
const uint32_t* gmx_lstm_batch_contexts(gmx_lstm_batch* b);    /* [S][max_bytes] */
int gmx_lstm_batch_upload(gmx_lstm_batch* b, uint64_t n_bytes);
int gmx_lstm_batch_download(gmx_lstm_batch* b, uint64_t n_bytes);
int gmx_lstm_batch_wait(gmx_lstm_batch* b);                /* this batch's upload, runs, download (cf. gmx_batch_wait) */
/* LstmModel::Predict x 8 bits (+ LstmModel::Learn when learn != 0) for bytes [0, n_bytes) of every
 * stream. */
int gmx_lstm_run(gmx_lstm* l, gmx_lstm_batch* b, uint64_t n_bytes, int learn, float* kernel_ms);
/* ... for streams at different lengths (cf. gmx_group_run_ragged): stream s runs bytes [0, n_bytes[s]). */
int gmx_lstm_run_ragged(gmx_lstm* l, gmx_lstm_batch* b, const uint64_t* n_bytes /* [S] */, int learn);
/* Per-byte surface, for decoding (the byte is not known when its prediction is needed).
 * forward = Lstm::SetInput + Lstm::Predict(last_byte) at a byte boundary (lstm-model.cpp:19-33,
 * last_byte = ShortTermMemory::last_byte): probs[256]
 * (nullable) = the byte distribution LstmModel keeps in probs_, *context (nullable) =
 * lstm_prediction_context; the 8 bit predictions follow from probs and the decoded bits exactly as
 * in LstmModel::Predict (lstm-model.cpp:34-48).  perceive = Lstm::Perceive(byte), i.e.
 * LstmModel::Learn at the last bit of that byte. */
/* Hand a batch's results to the models downstream, device to device, after gmx_lstm_run:
 * mixer_batch (nullable; created with GMX_BATCH_MASK, max_bits >= 8 * n_bytes): the prediction of
 * bit k of byte n goes to slot `slot` (the LSTM's prediction index) of record 8n+k, its active
 * flag into that record's mask, lstm_prediction_context into gate-context column mixer_ctx_col
 * (< 0: none) of the 8 records; ind_batch (nullable): the context into column ind_ctx_col of the
 * 8 Indirect records (the model built on lstm_prediction_context, predictor.cpp:117-119). */
int gmx_lstm_feed(gmx_lstm* l, gmx_lstm_batch* b, uint64_t n_bytes, gmx_batch* mixer_batch, int slot,
                  int mixer_ctx_col, gmx_ind_batch* ind_batch, int ind_ctx_col);
int gmx_lstm_forward(gmx_lstm* l, int stream, int last_byte, const float* ppm, float* probs, uint32_t* context);
int gmx_lstm_perceive(gmx_lstm* l, int stream, int byte);

/* Persistence, byte for byte the reference's: `short` = the model's stretch of the .short file,
 * LstmModel::WriteToDisk / ReadFromDisk (lstm-model.cpp:62-76) with Lstm::, LstmLayer:: and 3 x
 * NeuronLayer::WriteToDisk behind it (lstm.cpp:124-158, lstm-layer.cpp:356-394, :62-122; 1 458 256
 * bytes); `long` = the LSTM section of LongTermMemory::WriteToDisk / ReadFromDisk
 * (long-term-memory.cpp:57-67, :151-160; 5 560 200 bytes).  Both buffers NULL: sizes only.  Where a byte has
 * ended top_/mid_/bot_ are written as the eighth LstmModel::Predict of the last coded byte leaves them.  Between
 * gmx_lstm_forward and gmx_lstm_perceive -- LstmModel::WriteToDisk works at any bit, and the reference's
 * TestGeneration checkpoints after a Predict whose byte is never perceived (tester.cpp:284, :312) -- the network's
 * state is what the forward left, and the range state, which the caller advances on the host from there
 * (lstm-model.cpp:34-48), is written as that forward's own Predict leaves it (255 / 127 / 0): a caller further into
 * the byte puts its own top_/mid_/bot_ into the first 12 bytes.  import takes either kind; a file does not say
 * whether its newest forward has been perceived (the reference keeps no such flag), so after an import of anything
 * but an untouched model both gmx_lstm_perceive and gmx_lstm_forward are accepted next.  After an import the
 * bank's remembered last byte is the newest perceived entry of input_history_; a stream that did not learn from its
 * last byte passes it explicitly (gmx_lstm_forward).
 * copy = LstmModel::Copy (lstm-model.cpp:78-85) + the LSTM share of LongTermMemory::Copy
 * (long-term-memory.cpp:216-219); memory_usage = LstmModel::GetMemoryUsage (lstm-model.cpp:87-101). */
int gmx_lstm_export(gmx_lstm* l, int stream, void* long_buf, size_t* long_bytes, void* short_buf,
                    size_t* short_bytes);
int gmx_lstm_import(gmx_lstm* l, int stream, const void* long_buf, size_t long_bytes, const void* short_buf,
                    size_t short_bytes);
int gmx_lstm_copy(gmx_lstm* dst, int dst_stream, gmx_lstm* src, int src_stream);
int gmx_lstm_memory_usage(gmx_lstm* l, uint64_t* bytes);

/* ==== Lock step through the whole device chain: S decoders, one device step per coded bit ======
 * The reference's Decoder (coder/decoder.cpp:19-39) learns each bit from Predict's own result, so S files being
 * restored on one GPU advance together, a bit per step (gmx_lockstep_* above does this for the mixers alone).  A
 * gmx_chainstep steps the LSTM byte model and the Indirect models with the mixers -- `ib` and `l` may be NULL: the
 * caller's records then carry those models' predictions like any other -- so that, as in the batched chain
 * (gmx_lstm_feed, gmx_indirect_run's `into`), the LSTM's prediction (slot lstm_slot of the mixers' inputs, its active
 * bit, lstm_prediction_context in gate-context column mixer_ctx_col and Indirect-context column ind_ctx_col; < 0: none)
 * and the Indirect models' 2 x K predictions never leave the device.
 * Per step the caller fills, for every stream s that takes part, what[s] and the host arrays the step reads:
 *   GMX_STEP_LEARN    bits[s] = the bit the stream decoded from the last step's p[s]: Predictor::Learn -- Mixer::Learn
 *                     x M, Indirect::Learn x K, and Lstm::Perceive when the bit completes a byte
 *   GMX_STEP_PREDICT  predictions[s][n_pad] (the blackboard; the device-side models' slots are overwritten), active_mask
 *                     [s][mask_words] (their bits left clear), contexts[s][M], ind_contexts[s][K], bit_contexts[s], and
 *                     -- when the bit opens a byte -- ppm[s][256] (ShortTermMemory::ppm_predictions): Predictor::Predict
 * and calls gmx_chainstep_step, which returns with p[s] (and outputs[s][M]) of the streams that predicted.  A stream
 * whose what[s] is 0 sits the step out (its file has ended) -- but not between a Predict and its Learn: the Learn belongs
 * to the very next step (GMX_ERR_STATE otherwise); streams start at a byte boundary.  One hipGraph of kernels per step;
 * same floats as every other surface.  Destroy before the banks.
 * gmx_chainstep_commit(cs, s), optional: stream s has filled what[s], bits[s] and its records for the coming step --
 * where the host can store into device memory the library moves them there right away, in the calling thread (S decoders
 * on several threads: each commits its own streams, any number of threads at once, before the one that calls
 * gmx_chainstep_step); streams nobody committed are moved by gmx_chainstep_step itself.  Nothing of stream s may be
 * written between its commit and the step.
 * gmx_chainstep_launch / gmx_chainstep_wait: gmx_chainstep_step in two halves -- launch returns with the step queued
 * (and `what` free to be cleared), wait with p / outputs in place -- for a caller that drives two objects alternately,
 * one's host work beside the other's device step.  Between the two, nothing of the object may be written or committed
 * (GMX_ERR_STATE). */
#define GMX_STEP_LEARN 1u
#define GMX_STEP_PREDICT 2u
typedef struct gmx_chainstep gmx_chainstep;
int gmx_chainstep_create(gmx_chainstep** out, gmx_group* g, gmx_indirect* ib /* nullable */, gmx_lstm* l /* nullable */,
                         int lstm_slot, int mixer_ctx_col, int ind_ctx_col);
void gmx_chainstep_destroy(gmx_chainstep* cs);
int gmx_chainstep_n_streams(const gmx_chainstep* cs);
float* gmx_chainstep_predictions(gmx_chainstep* cs);     /* pinned host [S][n_pad] */
uint32_t* gmx_chainstep_active_mask(gmx_chainstep* cs);  /* [S][mask_words] */
uint32_t* gmx_chainstep_contexts(gmx_chainstep* cs);     /* [S][M] */
uint32_t* gmx_chainstep_ind_contexts(gmx_chainstep* cs); /* [S][K]; NULL without Indirect models */
uint32_t* gmx_chainstep_bit_contexts(gmx_chainstep* cs); /* [S] */
float* gmx_chainstep_ppm(gmx_chainstep* cs);             /* [S][256]; NULL without an LSTM */
uint8_t* gmx_chainstep_bits(gmx_chainstep* cs);          /* [S] */
uint8_t* gmx_chainstep_what(gmx_chainstep* cs);          /* [S] GMX_STEP_* */
const float* gmx_chainstep_p(gmx_chainstep* cs);         /* [S] */
const float* gmx_chainstep_outputs(gmx_chainstep* cs);   /* [S][M] */
int gmx_chainstep_commit(gmx_chainstep* cs, int stream);
int gmx_chainstep_step(gmx_chainstep* cs);
int gmx_chainstep_launch(gmx_chainstep* cs);
int gmx_chainstep_wait(gmx_chainstep* cs);

/* ==== Compute-unit shares ======================================================================
 * A bank's kernels normally spread over the whole chip.  Kernels of DIFFERENT banks that cannot share
 * a SIMD -- a mixer wave of the stock shape owns all 512 registers of its SIMD, an LSTM workgroup
 * half of every SIMD of its CU -- then run one after the other even from different streams.  With CU
 * masks (bit i of the 32-bit words = compute unit i; 8 words on MI355X, 32 CUs per XCD) the banks'
 * streams are re-created with hipExtStreamCreateWithCUMask and their kernels run side by side on
 * disjoint CUs: the LSTM on 16 CUs of every XCD, mixers and Indirect models on the other 16, is 24 %
 * faster end to end than all three on all CUs.  n_words == 0: all CUs again.  Call between launches
 * (the calls synchronise the bank); not with lock-step objects alive on a mixer group. */
int gmx_group_set_cu_mask(gmx_group* g, const uint32_t* mask, int n_words);
int gmx_indirect_set_cu_mask(gmx_indirect* ib, const uint32_t* mask, int n_words);
int gmx_lstm_set_cu_mask(gmx_lstm* l, const uint32_t* mask, int n_words);

#ifdef __cplusplus
}
#endif
#endif /* GMXMIX_H_ */
