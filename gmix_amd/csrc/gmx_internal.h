// gmx_internal.h -- structures shared by the host side of libgmxmix.so and its kernels.
#ifndef GMX_INTERNAL_H_
#define GMX_INTERNAL_H_

#include <stdint.h>

#define GMX_MAX_MIXERS 64
#define GMX_MAX_SKIP 8
#define GMX_MAX_INPUTS 2048

#define GMX_MODE_PREDICT 1u   // run the forward chain (else: outputs come from the latch)
#define GMX_MODE_LEARN 2u     // run Mixer::Learn after each bit
#define GMX_MODE_LATCH 4u     // forward-only call of the per-bit surface: keep outputs for learn
#define GMX_MODE_EXACT 8u     // stock kernels: masked forward chains only (tests; see gmx_stock.hip)

// Device-side description of one Mixer (mixer.h:33-38 + where its table lives in a bank).
struct GmxMixerDev {
  uint32_t table_size;   // rows of the gate table (mixer.cpp:15)
  uint32_t weight_size;  // weight_size_ (mixer.cpp:17-26)
  uint32_t stride;       // floats per stored row: weight_size rounded up to 32 (128-B rows)
  uint32_t lds_off;      // float offset of this mixer's slot 0 in the wave's LDS image
  uint32_t pitch;        // float distance between the mixer's two LDS slots (stride + 4)
  int32_t layer;         // 0, 1, 2
  int32_t out_index;     // index inside its layer (output_index_ for layers 0/1)
  float lr;              // learning_rate_
  uint64_t w_off;        // byte offset of the weight table inside a bank
  uint64_t rs_off;       // byte offset of row 0's step counter (MixerData::steps) ...
  uint32_t rs_pitch;     // ... and the byte distance to the next row's: 8 (a table of its own) or the row
                         // length (folded: the counter lives in the row's zero padding)
  uint32_t rs_folded;    // 0: table of its own; 1: the last 8 bytes of the row (the reference's 90-input shape);
                         // 2: right behind the weights (the 256-input 24/8/1 shape), see build_topology
};

// Address of the step counter of row `row` of mixer `mx` in the bank at `bank`.
#define GMX_RS_PTR(bank, mx, row) \
  ((uint64_t*)((uint8_t*)(bank) + (mx).rs_off + (uint64_t)(row) * (mx).rs_pitch))

struct GmxTopoDev {
  int32_t n, n_pad, n_skip, m, l0, l1, has_final, mask_words;
  int32_t skip_idx[GMX_MAX_SKIP];
  uint32_t lds_in0;      // float offset of in0[2][in0_sz]: [x (n) | layer-0 outputs (l0)]
  uint32_t in0_sz;
  uint32_t lds_o1;       // layer-1 outputs [l1]
  uint32_t lds_skip;     // raw skip inputs [n_skip]
  uint32_t lds_misc;     // per-mixer scratch: 4 x 64 dwords (row, dst slot, update, flags)
  uint32_t lds_total;    // floats
  uint64_t scal_off;     // byte offset of {steps_, max_steps_, contexts_seen_} x m
  uint64_t bank_bytes;   // bytes per stream
  GmxMixerDev mx[GMX_MAX_MIXERS];
};

// Kernel arguments of one run over T bits of `n_streams` streams.
struct GmxRunArgs {
  uint8_t* banks;             // device base of bank 0
  const float* pred;          // [S][*][n_pad]
  const uint32_t* mask;       // [S][*][mask_words] or null
  const uint32_t* ctx;        // [S][*][m]
  const uint8_t* bits;        // [S][*]
  const float* decay;         // [n_tabs][T] first factor of the decay (mixer.cpp:111), host pow
  const uint32_t* decay_idx;  // [S] which table a stream uses
  float* p_out;               // [S][*]
  float* out_all;             // [S][*][m] or null
  float* out_last;            // [S][m] or null: the m outputs of each stream's LAST bit of the launch only (what the
                              // blackboard holds afterwards), for callers that need no more: the throughput build of
                              // the stock kernel stores nothing else beside the probabilities
  float* latch_out;           // [S][m] outputs kept between forward and learn of the per-bit API
  uint64_t rec_stride;        // records per stream in the arrays above (max_bits of the batch)
  uint64_t T;                 // bits per stream (the decay tables' pitch) ...
  const uint64_t* T_list;     // ... or, not null: [n_streams] a count per block (<= T; 0: the stream sits the launch
                              // out) -- streams of different lengths in ONE launch (files end at different bits)
  uint32_t mode;
  int32_t stream_base;        // bank of block 0
  int32_t rec_base;           // record-array stream index of block 0
  int32_t n_streams;          // streams this launch covers
};

// Mailbox of a per-bit session (gmx_stock_session_kernel), one per open session, in two parts.
//   GmxMbCmd   host -> device.  Lives in fine-grained DEVICE memory when the host can store there
//              (large BAR: the wave polls and reads its payload locally, the host's stores cross
//              PCIe once, posted), else in host-coherent pinned memory (the wave polls across PCIe).
//   GmxMbReply device -> host, always host-coherent pinned memory (the host polls its own DRAM).
// The host publishes a command by writing a payload slot, then the command word
// (sequence number << 4 | payload slot << 3 | command, release); the device answers by writing
// the results, then the same word into done_seq.  At most one command is outstanding.  Two payload
// slots, used alternately: the payload of the forward whose learn is still to come stays intact
// while the next one is written (LEARN*_FWD carries both; a wave restarted in between replays the
// forward from the old slot).
// LDS of one wave of the stock kernels (gmx_stock.hip): inputs, expf's table, and the two staging images
// of the 33 rows (prefetch, write-back), rows 528 bytes apart
#define GMX_STK_LDS_BYTES(lds_misc) (((lds_misc) + 256u) * 4u + 2u * 33u * 528u)

#define GMX_MB_FORWARD 1u
#define GMX_MB_LEARN0 2u      // learn, coded bit 0
#define GMX_MB_LEARN1 3u      // learn, coded bit 1
#define GMX_MB_STOP 4u
#define GMX_MB_LEARN0_FWD 5u  // learn (bit 0) of the previous forward, then forward of the slot's payload
#define GMX_MB_LEARN1_FWD 6u
#define GMX_MB_STOP_KEEP 7u   // lock-step waves only: leave like GMX_MB_STOP, but a forward nobody has learned from yet
                              // is redone by the next instance from the block's own copy (as after an idle exit)
#define GMX_MB_CMD_MASK 7u
#define GMX_MB_SLOT_SHIFT 3
#define GMX_MB_SEQ_SHIFT 4
#define GMX_MB_RUNNING 0u
#define GMX_MB_EXIT_IDLE 1u   // left after idle_ticks without a command (rows written back)
#define GMX_MB_EXIT_STOP 2u   // left on GMX_MB_STOP
struct GmxMbPayload {
  uint32_t dec_bits;       // float bits of 0.9/pow(1e-7*steps_+0.8, 0.8) (mixer.cpp:111) for the learn
                           // that follows this forward
  uint32_t pad0[3];
  uint32_t mask[4];        // active_models as a bit mask
  uint32_t ctx[36];        // the 33 gate contexts
  uint32_t pad1[4];
  float pred[96];          // ShortTermMemory::predictions (90 used, zero padded)
};
struct GmxMbCmd {
  uint32_t cmd_seq;        // newest command word
  uint32_t pad0[15];
  GmxMbPayload slot[2];
};
struct GmxMbReply {
  uint32_t done_seq;       // last completed command word
  uint32_t state;          // GMX_MB_RUNNING / GMX_MB_EXIT_*
  float p;                 // forward result: clamped probability
  uint32_t pad[13];
  float outs[48];          // forward result: the 33 mixer outputs
};

// ---- lock step, persistent (gmx_stock_lockstep_kernel): S blocks, one doorbell ----------------------
// The host writes the records of all S streams into pinned host arrays (the lock-step batch's own), then one
// command word into `cmd_seq` (device memory behind the BAR, polled locally); block s reads ITS record
// straight from the host arrays, answers into the host arrays and arrives at `arrived`; the last one to
// arrive writes the word into *done (pinned host memory).  Commands: GMX_MB_FORWARD, GMX_MB_LEARN0 (= learn,
// the coded bits are per stream, in `bits`), GMX_MB_LEARN0_FWD, GMX_MB_STOP.
struct GmxLsDoor {          // device memory (fine-grained)
  uint32_t cmd_seq;
  uint32_t pad[15];
};
struct GmxLsArgs {
  uint8_t* banks;
  GmxLsDoor* door;
  uint32_t* done;           // pinned host memory: newest command word completed by ALL blocks
  const float* pred;        // pinned host [S][n_pad]
  const uint32_t* mask;     // pinned host [S][mask_words]
  const uint32_t* ctx;      // pinned host [S][m]
  const uint8_t* bits;      // pinned host [S]: coded bits of the learn
  const float* dec;         // pinned host [S]: decay factor of each stream's next learn, read at the forward
  float* p;                 // pinned host [S]
  float* outs;              // pinned host [S][m] or null
  // per block, device memory: what survives a block that leaves on its idle timer
  uint32_t* blk_seen;       // [S] last command word the block completed
  uint32_t* blk_replay;     // [S] 1: it left between a forward and its learn (the forward is redone from `live`)
  uint32_t* relay;          // device memory, L2-cached: [0] the doorbell's word as block 0 passes it on (a thousand
                            // waves polling one uncached word would starve the host's own store to it), [32] the
                            // count of blocks that have completed it
  float* live;              // [S][GMX_LS_LIVE_FLOATS] the newest forward's record as the block read it
  uint64_t idle_ticks;
  int32_t n_streams, exact;
};
#define GMX_LS_LIVE_FLOATS 160  // pred[96] | mask[4] | ctx[36] | dec | pad

// One step of the lock-step chain for the mixers of the reference's own shape (gmx_stock_step_kernel, gmx_chainstep.inc):
// Mixer::Learn x 33 on the records of the stream's last forward, then Mixer::Predict x 33 on this step's, in ONE launch.
struct GmxStkStepArgs {
  uint8_t* banks;
  const uint8_t* what;      // [S] bit 0: learn, bit 1: predict; 0: the stream sits the step out
  const uint8_t* bits;      // [S] the coded bit of each stream's previous forward
  const float* dec;         // [S] decay factor of the learn
  const float* pred_old;    // [S][92]  the records the learn works on: the step before's ...
  const uint32_t* mask_old; // [S][3]
  const uint32_t* ctx_old;  // [S][33]
  const float* pred_new;    // ... and this step's
  const uint32_t* mask_new;
  const uint32_t* ctx_new;
  float* p_out;             // [S]     pinned host memory
  float* out_all;           // [S][33] pinned host memory, or null
  float* latch;             // [S][33] the outputs a learn-only launch starts from (gmx_group::latch_out)
  const uint32_t* seq;      // [S] the step's number ...
  uint32_t* stamp;          // [S] pinned host memory: ... stored here by the stream's block when its answers are out
  int32_t n_streams, exact;
};

// The head of a step: every block fetches its own stream's slices of the step's pinned host block (control words and
// records: arrays of [S][bytes_per_stream]) and stores them at the same offsets of the device copy.
#define GMX_STEP_UP_MAX 12
struct GmxStepUpload {
  const uint8_t* src;       // the host block as the device sees it
  uint8_t* dst;             // this step's device copy
  uint32_t off[GMX_STEP_UP_MAX];   // where an array starts in the block
  uint32_t bps[GMX_STEP_UP_MAX];   // bytes per stream (1 or a multiple of 4)
  int32_t n;
};

// ---- Indirect models (models/indirect.cpp) -------------------
#define GMX_IND_MAX_MODELS 64

struct GmxIndModelDev {
  uint32_t size;        // 256 * table_size + 1 entries (indirect.cpp:15-19)
  float lr;             // learning_rate_
  int32_t slot_a;       // prediction index of "<name>-indirect" on the blackboard
  int32_t slot_b;       // prediction index of "<name>-run_map"
  uint64_t tab_off;     // byte offset of the model's u16 table inside a bank:
                        // low byte = nonstationary state (255 = never seen), high byte = run-map state
};

struct GmxIndDev {
  int32_t k, n_slots;   // models; highest slot index + 1
  uint64_t pred_off;    // float[k][512]: per model nonstationary_predictions then run_map_predictions
  uint64_t slots_off;   // float[2k]: what the two blackboard slots of each model hold (stale when silent)
  uint64_t bank_bytes;
  uint8_t ns_next[512]; // ShortTermMemory::nonstationary as a table: [state][bit]
  uint8_t rm_next[512]; // ShortTermMemory::run_map
  GmxIndModelDev m[GMX_IND_MAX_MODELS];
};

// Mailbox of a per-bit session of the Indirect models (gmx_indirect_session_kernel): same protocol as the
// mixers' (GmxMbCmd / GmxMbReply above: command word = sequence << 4 | command, answered by the same word
// in done_seq).
struct GmxIndMbCmd {       // host -> device: fine-grained device memory behind a large BAR, else pinned host memory
  uint32_t cmd_seq;        // sequence << 4 | payload slot << 3 | command
  // A chained forward (gmx_chain_forward): after its Predict the wave puts the 2k predictions into the `pred`
  // of payload slot `chain_slot` of the MIXERS' mailbox at `chain_mc` (at the models' slot indices), ORs the
  // active bits into its `mask`, and rings that mailbox with `chain_word` -- the mixers start without the
  // host having seen the predictions.  chain_word == 0: no chaining.
  uint32_t chain_word;
  uint32_t chain_slot;
  uint32_t pad0;
  uint64_t chain_mc;
  uint32_t pad1[8];
  uint32_t bit_context[2]; // ShortTermMemory::bit_context, per payload slot
  // the models' aliased context variables, read at the Predict call.  Two slots, used alternately: the one
  // of the forward whose learn is still to come stays intact while the next is written, so a wave
  // restarted in between can recompute that forward
  uint32_t ctx[2][GMX_IND_MAX_MODELS];
};
struct GmxIndMbReply {     // device -> host: pinned host memory
  uint32_t done_seq;
  uint32_t state;          // GMX_MB_RUNNING / GMX_MB_EXIT_*
  uint32_t pad0[2];
  uint64_t active_a, active_b;  // bit i: model i's "-indirect" / "-run_map" slot was marked active
  uint32_t pad1[8];
  float pred[2 * GMX_IND_MAX_MODELS];  // what the two blackboard slots of model i hold after its Predict
};

struct GmxIndRunArgs {
  uint8_t* banks;
  const uint32_t* ctx;     // [S][*][k]
  const uint32_t* bc;      // [S][*]  bit_context
  const uint8_t* bits;     // [S][*]
  float* pred_out;         // [S][*][2k] or null
  uint8_t* act_out;        // [S][*][2k] or null
  uint64_t rec_stride, T;
  const uint64_t* T_list;  // not null: [blocks] a bit count per block (<= T; 0: the stream sits the launch out)
  uint32_t learn;
  int32_t stream_base, rec_base;
  // optional: also write into the record arrays of a mixer batch (same streams, same bits)
  float* mx_pred;          // [S][*][mx_n_pad] or null
  uint32_t* mx_mask;       // [S][*][mx_mask_words]
  uint8_t* mx_bits;        // [S][*]
  uint64_t mx_rec_stride;
  int32_t mx_n_pad, mx_mask_words;
};

// One bit of every stream, the coded bit of a forward known a launch later (gmx_indirect_step_kernel, gmx_chainstep.inc)
struct GmxIndStepArgs {
  uint8_t* banks;
  const uint32_t* ctx;     // [S][k]  the models' contexts of this step's Predict
  const uint32_t* bc;      // [S]     bit_context
  const uint8_t* bits;     // [S]     the coded bit of each stream's previous forward
  const uint8_t* what;     // [S]     bit 0: learn, bit 1: predict; 0: the stream sits out
  uint32_t* latch;         // [S][64][4] per model: table index, entry, "a forward waits for its learn"
  float* mx_pred;          // [S][mx_n_pad] the mixers' records of the same step, or null
  uint32_t* mx_mask;       // [S][mx_mask_words]
  int32_t mx_n_pad, mx_mask_words;
  float* pred_out;         // [S][2k] or null (tests)
  uint8_t* act_out;        // [S][2k]
  GmxStepUpload up;        // up.n > 0: the launch is the step's first -- it brings the stream's inputs in itself
};

// ---- LSTM byte model (models/lstm*.cpp) -----------------------
// The reference builds Lstm(256, 256, 50, 1, 100, 0.03, 10) (lstm-model.cpp:7).  One stream's
// state, in floats (u32 where noted), every [cell] vector padded to 64:
#define GMX_L_NI 256
#define GMX_L_NO 256
#define GMX_L_NC 50
#define GMX_L_H 100
#define GMX_L_LIN 307     // input (256) + hidden (50) + bias
#define GMX_L_LINP 320
#define GMX_L_W 563       // one-hot symbol column (256) + layer input (307)
#define GMX_L_HID 51
#define GMX_L_CP 64
#define GMX_L_HP 100

// The [input][cell] matrices of a gate (weights, update_, m_, v_): the 256 symbol rows one row of CP
// cells each; the 307 layer-input rows in blocks of four -- [block][cell][4] -- so that a lane's
// chain over the inputs reads 16 bytes per load (4-byte loads kept the texture path busy for 16
// cycles per 200 bytes: 9 us of a byte's 32 went there).
#define GMX_L_LINB 77     // blocks of four layer-input rows (the last one holds three)
#define GMX_L_MAT_FLOATS (GMX_L_NO * GMX_L_CP + GMX_L_LINB * GMX_L_CP * 4)
__host__ __device__ inline uint64_t gmx_l_mat(int j, int c) {  // float offset of (input j, cell c)
  return j < GMX_L_NO ? (uint64_t)j * GMX_L_CP + (uint64_t)c
                      : (uint64_t)GMX_L_NO * GMX_L_CP +
                            ((uint64_t)((j - GMX_L_NO) >> 2) * GMX_L_CP + (uint64_t)c) * 4 + (uint64_t)((j - GMX_L_NO) & 3);
}

struct GmxLstmGateOff {   // float offsets inside a bank
  uint64_t weights, update, m, v;                 // GMX_L_MAT_FLOATS each, indexed by gmx_l_mat
  uint64_t gamma, gamma_u, gamma_m, gamma_v, beta, beta_u, beta_m, beta_v, error;  // [CP]
  uint64_t state, norm;                           // [H][CP]
  uint64_t ivar;                                  // [H]
  uint64_t transpose;                             // [HID][CP]: NeuronLayer::transpose_, the recurrent weights as
                                                  // the last backward pass saw them (lstm-layer.cpp:300-304)
};

struct GmxLstmDev {
  uint64_t out_layer;                             // [H][HID][NO]  (LongTermMemory::lstm_output_layer, transposed)
  GmxLstmGateOff gate[3];                         // forget gate, input node, output gate
  uint64_t state, state_error, stored_error;      // [CP]
  uint64_t tanh_state, input_gate_state, last_state;  // [H][CP]
  uint64_t hidden, hidden_error;                  // [CP]
  uint64_t layer_input;                           // [H][LINP]
  uint64_t lin_t;                                 // [LIN][GMX_L_HP]: the same, input-major (one row = one input over the epochs)
  uint64_t errs;                                  // [3][H][CP]: the gates' final errors of a backward pass
  uint64_t output;                                // [H][NO]
  uint64_t input_history;                         // u32 [H]
  uint64_t probs;                                 // [NO]  LstmModel::probs_
  uint64_t scal;                                  // u32 [16]: epoch, layer epoch, update_steps, last_byte, context,
                                                  // prediction bits, "a byte has been coded"
  uint64_t bank_floats;
};

struct GmxLstmRunArgs {
  float* banks;
  const float* ppm;        // [S][*][256]
  const uint8_t* bytes;    // [S][*]
  const float* adam;       // [S][max_bptt][4]: alpha, 1 - beta1^t, 1 - beta2^t for the stream's next backward passes
  float* pred_out;         // [S][*][8]
  uint8_t* act_out;        // [S][*][8]
  uint32_t* ctx_out;       // [S][*]
  uint64_t rec_stride, n_bytes;
  const uint64_t* n_list;  // not null: [blocks] a byte count per block (<= n_bytes; 0: the stream sits the launch out)
  uint32_t learn, max_bptt;
  uint32_t phases;         // 1: Lstm::Predict, 2: the 8 bit predictions, 4: Lstm::Perceive (when learn)
  int32_t stream_base;     // bank of block 0 (records of block 0 are always stream 0 of the arrays)
  int32_t last_byte;       // >= 0: ShortTermMemory::last_byte for the first Predict (else the bank remembers it)
  // per-byte session (gmx_lstm_kernel<.., .., true>, one block): commands instead of records
  struct GmxLstmMbCmd* mc;
  struct GmxLstmMbReply* mb;
  uint64_t idle_ticks;
};

// The LSTM's prediction of ONE bit of every stream in lock step (gmx_lstm_bitstep_kernel, gmx_chainstep.inc):
// LstmModel::Predict's range walk (lstm-model.cpp:34-48) with the byte's bits arriving one step apart.
struct GmxLstmBitArgs {
  float* banks;
  const uint8_t* bits;     // [S] the coded bit of each stream's previous step
  const uint8_t* what;     // [S] bit 1: predict this step; bit 2: the step opens a byte (Lstm::Predict has just run)
  float* mx_pred;          // [S][mx_n_pad]
  uint32_t* mx_mask;       // [S][mx_mask_words]
  uint32_t* mx_ctx;        // [S][mx_m]
  uint32_t* ind_ctx;       // [S][ind_k] or null
  int32_t mx_n_pad, mx_mask_words, mx_m, slot, mixer_ctx_col, ind_k, ind_ctx_col;
};

// Mailbox of a per-byte session of the LSTM byte model: the protocol of the mixers' and the Indirect models'
// sessions (command word = sequence << 4 | command, answered by the same word in done_seq).  Commands:
// GMX_MB_FORWARD = Lstm::Predict on `ppm` with `last_byte`; GMX_MB_LEARN0 = Lstm::Perceive(`byte`) (with
// `adam` when that byte's backward pass is due); GMX_MB_LEARN0_FWD = both, in that order.  One payload: the
// model's state is in its bank whenever no wave runs, so nothing ever has to be replayed.
struct GmxLstmMbCmd {      // host -> device: fine-grained device memory behind a large BAR, else pinned host memory
  uint32_t cmd_seq;
  uint32_t byte;           // the byte Lstm::Perceive learns
  uint32_t last_byte;      // ShortTermMemory::last_byte at the Predict
  uint32_t pad0;
  float adam[4];           // alpha, 1 - beta1^t, 1 - beta2^t of the backward pass this Perceive runs (if it does)
  uint32_t pad1[8];
  float ppm[256];          // the PPM byte distribution (ShortTermMemory::ppm_predictions)
};
struct GmxLstmMbReply {    // device -> host: pinned host memory
  uint32_t done_seq;
  uint32_t state;          // GMX_MB_RUNNING / GMX_MB_EXIT_*
  uint32_t context;        // lstm_prediction_context
  uint32_t pad[13];
  float probs[256];        // Lstm::Predict's distribution
};

#endif  // GMX_INTERNAL_H_
