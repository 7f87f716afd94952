// probe_bar_mailbox.hip -- where should the per-bit mailbox of the decode path live?
//
// Ping-pong between the host and one persistent wave: the host writes a 96-float payload and a
// sequence word, the wave waits for the word, sums the payload, writes a result and the word
// back.  Modes differ in where the command block lives:
//   0  pinned host memory, the wave polls across PCIe            (what gmx_session.inc does)
//   1  fine-grained device memory, the host writes through the BAR, the wave polls locally
//   2  plain hipMalloc memory, same
//   3  managed memory preferred on the device
// Replies always go to pinned host memory (the host polls its own DRAM).  Every mode runs in a
// child forked BEFORE any HIP call, so a host store that faults only ends that child; the wave's
// spin is bounded, so nothing can hang the GPU.
//
//   hipcc --offload-arch=gfx950 -O2 -o probe_bar_mailbox probe_bar_mailbox.hip && ./probe_bar_mailbox
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <immintrin.h>

struct Cmd {
  float payload[96];
  uint32_t ctx[36];
  uint32_t seq;
  uint32_t pad[27];
};
struct Reply {
  float result[48];
  uint32_t seq;
  uint32_t fail;
};

__global__ void __launch_bounds__(64) pingpong(Cmd* cmd, Reply* rep, int iters, long max_spin) {
  const int lane = threadIdx.x;
  for (int it = 1; it <= iters; ++it) {
    long spin = 0;
    while (__hip_atomic_load(&cmd->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != (uint32_t)it) {
      if (++spin > max_spin) {
        if (lane == 0) {
          rep->fail = (uint32_t)it;
          __hip_atomic_store(&rep->seq, 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
      }
    }
    float v = 0.f;
    if (lane < 48) {
      const float a = __hip_atomic_load(&cmd->payload[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const float b = __hip_atomic_load(&cmd->payload[48 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v = a + b;
      rep->result[lane] = v;
    }
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(&rep->seq, (uint32_t)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("{\"mode\": %d, \"error\": \"%s: %s\"}\n", mode, #x, hipGetErrorString(e)); \
      fflush(stdout);                                                          \
      _exit(3);                                                                \
    }                                                                          \
  } while (0)

static int run_mode(int mode, int iters) {
  Cmd* cmd = nullptr;
  Reply* rep = nullptr;
  CK(hipSetDevice(0));
  CK(hipHostMalloc((void**)&rep, sizeof(Reply), hipHostMallocCoherent | hipHostMallocMapped));
  memset((void*)rep, 0, sizeof *rep);
  if (mode == 0) {
    CK(hipHostMalloc((void**)&cmd, sizeof(Cmd), hipHostMallocCoherent | hipHostMallocMapped));
    memset((void*)cmd, 0, sizeof *cmd);
  } else if (mode == 1) {
    CK(hipExtMallocWithFlags((void**)&cmd, sizeof(Cmd), hipDeviceMallocFinegrained));
    CK(hipMemset(cmd, 0, sizeof *cmd));
  } else if (mode == 2) {
    CK(hipMalloc((void**)&cmd, sizeof(Cmd)));
    CK(hipMemset(cmd, 0, sizeof *cmd));
  } else {
    CK(hipMallocManaged((void**)&cmd, sizeof(Cmd)));
    CK(hipMemset(cmd, 0, sizeof *cmd));
    (void)hipMemAdvise(cmd, sizeof *cmd, hipMemAdviseSetPreferredLocation, 0);
    (void)hipMemAdvise(cmd, sizeof *cmd, hipMemAdviseSetAccessedBy, hipCpuDeviceId);
  }
  CK(hipDeviceSynchronize());
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, st, cmd, rep, iters, 40000000L);
  CK(hipGetLastError());
  volatile uint32_t* rseq = &rep->seq;
  double t0 = 0, worst = 0;
  int ok = 1;
  for (int it = 1; it <= iters; ++it) {
    if (it == iters / 10 + 1) t0 = now();
    const double a = now();
    for (int i = 0; i < 96; ++i) cmd->payload[i] = (float)(it + i);  // a faulting store ends this child
    for (int i = 0; i < 36; ++i) cmd->ctx[i] = (uint32_t)(it * 31 + i);
    _mm_sfence();
    __atomic_store_n(&cmd->seq, (uint32_t)it, __ATOMIC_RELEASE);
    _mm_sfence();
    const double deadline = a + 5.0;
    while (*rseq != (uint32_t)it) {
      if (*rseq == 0xFFFFFFFFu || now() > deadline) {
        ok = 0;
        break;
      }
    }
    if (!ok) break;
    const float want = (float)(it + 0) + (float)(it + 48);
    if (rep->result[0] != want) {
      printf("{\"mode\": %d, \"error\": \"stale payload at %d: %g != %g\"}\n", mode, it, rep->result[0], want);
      ok = 0;
      break;
    }
    const double d = now() - a;
    if (it > iters / 10 && d > worst) worst = d;
  }
  const double t1 = now();
  if (!ok) {
    // let the wave run into its spin bound
    __atomic_store_n(&cmd->seq, 0u, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(st);
    printf("{\"mode\": %d, \"ok\": false, \"fail_at\": %u}\n", mode, rep->fail);
    fflush(stdout);
    return 2;
  }
  CK(hipStreamSynchronize(st));
  const int timed = iters - iters / 10;
  printf("{\"mode\": %d, \"ok\": true, \"round_trip_us\": %.3f, \"worst_us\": %.1f, \"iters\": %d}\n", mode,
         (t1 - t0) / timed * 1e6, worst * 1e6, timed);
  fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  for (int mode = 0; mode < 4; ++mode) {
    fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) _exit(run_mode(mode, iters));
    int status = 0;
    waitpid(pid, &status, 0);
    if (WIFSIGNALED(status))
      printf("{\"mode\": %d, \"ok\": false, \"signal\": %d}\n", mode, WTERMSIG(status));
    fflush(stdout);
  }
  return 0;
}
