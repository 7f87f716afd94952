// gmix_many.cpp -- TEST / BENCH DRIVER around gmx::BatchedCompressFiles (gmix_amd/host/gmx_batched.h): many files
// compressed side by side, one reference Predictor (and host thread) per file, all mixers in one device group.
// Linked by dropin/Makefile against the reference's own translation units (feature models, coder,
// runner-utils) with Predictor::AddMixers constructing gmx::GpuMixer -- the reference calls the product.
//
// usage: gmix_many [-d] [-T chunk_bits] [-n bytes] [--no-pin] [--cpus n] [--groups g] [--destroy] [--plain-exit] [--device d] <out dir> <input file>...
//   each input is compressed to <out dir>/<index>.gmix exactly as `gmix -c` would (runner-utils.cpp:88-121);
//   -n limits every input to its first n bytes (written to <out dir>/<index>.in first).
//   -d: each input is a file `gmix -c` wrote and is restored to <out dir>/<index>.out as `gmix -d` would
//   (runner-utils.cpp:123-156), all files together through gmx::BatchedDecompressFiles (Decoders in lock step).
//   One JSON line on stdout: per-file sizes and times, the wall time of the compression phase, bits, launches.
#include <malloc.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>

#include "gmx_batched.h"

// ---- the process's end.  Every Predictor has TOUCHED about 2 GB (PPMd's arena is one `new byte[2000 << 20]`,
// mod_ppmd.cpp:127-128), and a process that just ends gives its pages back on ONE core: 1.3 s for 64 files, 5-6 s for
// 256 -- where the reference's CLI, a process per file, gives back 2 GB at a time on as many cores as it has processes.
// This driver therefore remembers its big allocations (a replaced global operator new, [replacement.functions]: the
// reference allocates with new / new[] / std::vector) and, when everything is written and closed, hands their pages
// back from all its threads at once -- madvise(MADV_DONTNEED) takes the address space's lock shared, so the threads'
// page freeing runs side by side -- before it ends.  Nothing is skipped: the same pages are freed, by more cores.
namespace {
constexpr size_t kBigAllocation = 32u << 20;
std::mutex g_big_mu;
std::vector<std::pair<char*, size_t>>* g_big = nullptr;  // (a pointer: alive until _exit, never destroyed)
bool g_release_pages = true;

void ReleaseBigAllocations(int n_threads) {
  std::vector<std::pair<char*, size_t>> slices;
  {
    std::lock_guard<std::mutex> lk(g_big_mu);
    if (!g_big) return;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE), slice = 64u << 20;
    for (auto& b : *g_big) {
      char* lo = (char*)(((uintptr_t)b.first + page - 1) / page * page);
      char* const hi = (char*)(((uintptr_t)b.first + b.second) / page * page);
      for (; lo < hi; lo += slice) slices.emplace_back(lo, (size_t)std::min<ptrdiff_t>((ptrdiff_t)slice, hi - lo));
    }
  }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&] {
      for (size_t i; (i = next.fetch_add(1)) < slices.size();) (void)madvise(slices[i].first, slices[i].second, MADV_DONTNEED);
    });
  for (auto& t : th) t.join();
}
}  // namespace

void* operator new(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  if (n >= kBigAllocation) {
    std::lock_guard<std::mutex> lk(g_big_mu);
    if (!g_big) g_big = new (malloc(sizeof *g_big)) std::vector<std::pair<char*, size_t>>();
    g_big->emplace_back((char*)p, n);
  }
  return p;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept {
  if (!p) return;
  if (g_big && malloc_usable_size(p) >= kBigAllocation) {  // (big blocks are few: a linear look is nothing beside their munmap)
    std::lock_guard<std::mutex> lk(g_big_mu);
    for (auto& b : *g_big)
      if (b.first == (char*)p) {
        b = g_big->back();
        g_big->pop_back();
        break;
      }
  }
  free(p);
}
void operator delete[](void* p) noexcept { operator delete(p); }
void operator delete(void* p, size_t) noexcept { operator delete(p); }
void operator delete[](void* p, size_t) noexcept { operator delete(p); }

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
    else if (!strcmp(argv[a], "--plain-exit"))
      g_release_pages = false;  // (experiments: end as any process does)
    else if (!strcmp(argv[a], "--destroy"))
      opt.destroy_predictors = true;  // (experiments: the Predictors' destructors before the process ends)
    else
      break;
  }
  if (argc - a < 2) {
    fprintf(stderr, "usage: %s [-d] [-T chunk_bits] [-n bytes] [--no-pin] [--cpus n] [--groups g] [--destroy] [--plain-exit] [--device d] <out dir> <input file>...\n", argv[0]);
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
  // runtime -- a second and a half for 64 files, which the kernel does in a fraction when the process just ends; the
  // Predictors' touched pages first, from all cores (above).
  if (g_release_pages && !opt.destroy_predictors) {
    int n = gmx::QuotaCpus();
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    ReleaseBigAllocations(n > 0 ? n : 1);
  }
  _exit(failed ? 1 : 0);
}
