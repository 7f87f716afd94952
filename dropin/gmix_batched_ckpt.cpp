// gmix_batched_ckpt.cpp -- TEST DRIVER: gmx::BatchedCompress on the first n bytes of a file (analysis off), then
// Predictor::WriteCheckpoint -- what the run-ahead compressor leaves BEHIND: every bank's state, the blackboard (mixer
// outputs, lstm_prediction_context), the adapters' own fields.  tests compare the two files with what the reference's
// tester writes at the same point of the same input through its per-bit loop (tester.cpp:30-59, CompressFirstHalf).
// usage: gmix_batched_ckpt <input> <n bytes> <checkpoint path> <compressed out> [chunk bits]
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "gmx_batched.h"

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s <input> <n bytes> <checkpoint path> <compressed out> [chunk bits]\n", argv[0]);
    return 2;
  }
  const unsigned long long n = strtoull(argv[2], 0, 0);
  std::ifstream in(argv[1], std::ios::binary);
  std::ofstream out(argv[4], std::ios::binary);
  if (!in.is_open() || !out.is_open()) return 2;
  srand(0xDEADBEEF);
  Predictor p;
  gmx::BatchedOptions opt;
  opt.analysis = false;
  opt.progress = false;
  if (argc > 5) opt.chunk_bits = strtoull(argv[5], 0, 0);
  gmx::BatchedCompressor c(&p, &out, opt);
  unsigned long long out_bytes = 0;
  int rc = c.Begin(n);
  if (rc == 0) rc = c.Code(n, &in, &out, &out_bytes);
  if (rc) {
    fprintf(stderr, "failed: %d\n", rc);
    return 1;
  }
  p.WriteCheckpoint(argv[3]);
  return 0;
}
