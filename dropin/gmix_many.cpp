// gmix_many.cpp -- TEST / BENCH DRIVER around gmx::BatchedCompressFiles (gmix_amd/host/gmx_batched.h): many files
// compressed side by side, one reference Predictor (and host thread) per file, all mixers in one device group.
// Linked by dropin/Makefile against the reference's own translation units (feature models, coder,
// runner-utils) with Predictor::AddMixers constructing gmx::GpuMixer -- the reference calls the product.
//
// usage: gmix_many [-d] [-T chunk_bits] [-n bytes] [--no-pin] [--cpus n] [--groups g] [--destroy] [--device d] <out dir> <input file>...
//   each input is compressed to <out dir>/<index>.gmix exactly as `gmix -c` would (runner-utils.cpp:88-121);
//   -n limits every input to its first n bytes (written to <out dir>/<index>.in first).
//   -d: each input is a file `gmix -c` wrote and is restored to <out dir>/<index>.out as `gmix -d` would
//   (runner-utils.cpp:123-156), all files together through gmx::BatchedDecompressFiles (Decoders in lock step).
//   One JSON line on stdout: per-file sizes and times, the wall time of the compression phase, bits, launches.
#include <sys/resource.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>

#include "gmx_batched.h"

int main(int argc, char** argv) {
  gmx::BatchedOptions opt;
  opt.destroy_predictors = false;  // (this process ends with the call: the kernel reclaims faster than 64 destructors)
  unsigned long long limit = 0;
  bool decode = false;
  int a = 1;
  for (; a < argc && argv[a][0] == '-'; ++a) {
    if (!strcmp(argv[a], "-d"))
      decode = true;
    else if (!strcmp(argv[a], "-T") && a + 1 < argc)
      opt.chunk_bits = strtoull(argv[++a], 0, 0);
    else if (!strcmp(argv[a], "-n") && a + 1 < argc)
      limit = strtoull(argv[++a], 0, 0);
    else if (!strcmp(argv[a], "--no-pin"))
      opt.pin_threads = false;
    else if (!strcmp(argv[a], "--cpus") && a + 1 < argc)
      opt.max_cpus = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--device") && a + 1 < argc)
      opt.device = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--groups") && a + 1 < argc)
      opt.groups = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--destroy"))
      opt.destroy_predictors = true;  // (experiments: the Predictors' destructors before the process ends)
    else
      break;
  }
  if (argc - a < 2) {
    fprintf(stderr, "usage: %s [-d] [-T chunk_bits] [-n bytes] [--no-pin] [--cpus n] [--groups g] [--destroy] [--device d] <out dir> <input file>...\n", argv[0]);
    return 2;
  }
  const std::string out_dir = argv[a++];
  std::filesystem::create_directories(out_dir);
  std::vector<gmx::BatchedJob> jobs;
  for (int k = 0; a < argc; ++a, ++k) {
    gmx::BatchedJob j;
    j.input_path = argv[a];
    if (limit) {
      std::ifstream in(argv[a], std::ios::binary);
      std::vector<char> head(limit);
      in.read(head.data(), limit);
      head.resize((size_t)in.gcount());
      j.input_path = out_dir + "/" + std::to_string(k) + ".in";
      std::ofstream(j.input_path, std::ios::binary).write(head.data(), head.size());
    }
    j.output_path = out_dir + "/" + std::to_string(k) + (decode ? ".out" : ".gmix");
    jobs.push_back(j);
  }
  gmx::BatchedStats st;
  const int failed = decode ? gmx::BatchedDecompressFiles(jobs, opt, &st) : gmx::BatchedCompressFiles(jobs, opt, &st);
  unsigned long long in_bytes = 0, out_bytes = 0;
  for (auto& j : jobs) {  // (in: the plain side, out: the coded side, whichever way the files went)
    in_bytes += decode ? j.output_bytes : j.input_bytes;
    out_bytes += decode ? j.input_bytes : j.output_bytes;
  }
  // "cold": the whole call -- pool, Predictors, device banks, coding, teardown -- as runner_utils::RunCompression
  // (runner-utils.cpp:88-121) builds its Predictor inside; "bits_per_second" is the coding loops alone.
  printf("{\"mode\": \"%s\", \"files\": %zu, \"failed\": %d, \"chunk_bits\": %llu, \"total_seconds\": %.6f, \"wall_seconds\": %.6f, "
         "\"build_seconds\": %.3f, \"first_predictor_seconds\": %.3f, \"teardown_seconds\": %.3f, "
         "\"parallel_construction\": %s, \"launches\": %llu, \"device_bits\": %llu, \"pinned_threads\": %d, "
         "\"pinned_cpus\": %d, \"submit_seconds\": %.4f, \"wait_seconds\": %.4f, \"jobs\": [",
         decode ? "decompress" : "compress", jobs.size(), failed, (unsigned long long)opt.chunk_bits, st.total_seconds, st.wall_seconds, st.build_seconds,
         st.first_predictor_seconds, st.teardown_seconds, st.parallel_construction ? "true" : "false",
         (unsigned long long)st.launches, (unsigned long long)st.bits, st.pinned_threads, st.pinned_cpus, st.submit_seconds,
         st.wait_seconds);
  for (size_t k = 0; k < jobs.size(); ++k)
    printf("%s{\"in\": %llu, \"out\": %llu, \"status\": %d, \"seconds\": %.6f}", k ? ", " : "", jobs[k].input_bytes,
           jobs[k].output_bytes, jobs[k].status, jobs[k].seconds);
  // (what the host gave the process: CPU seconds against wall time, and how often a thread was taken off its core --
  // on a busy host the lock step's spinning workers wait for whichever of them that happened to)
  struct rusage ru;
  memset(&ru, 0, sizeof ru);
  getrusage(RUSAGE_SELF, &ru);
  printf("], \"input_bytes\": %llu, \"output_bytes\": %llu, \"cpu_seconds\": %.3f, \"involuntary_switches\": %ld, "
         "\"bits_per_second\": %.1f, \"bits_per_second_cold\": %.1f}\n",
         in_bytes, out_bytes,
         ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec), ru.ru_nivcsw,
         st.wall_seconds > 0 ? 8.0 * in_bytes / st.wall_seconds : 0.0,
         st.total_seconds > 0 ? 8.0 * in_bytes / st.total_seconds : 0.0);
  fflush(stdout);
  fflush(stderr);
  // Every output file is closed and the line is out: the process ends HERE, without the destructors of 64 Predictors
  // (2 GB of address space each), of the pool (tens of gigabytes of device memory handed back piece by piece) and of the
  // runtime -- a second and a half for 64 files, which the kernel does in a fraction when the process just ends.
  _exit(failed ? 1 : 0);
}
