// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report for SCATTERED small accesses on gfx950.
// The guide's correction (FETCH_SIZE x 2) is calibrated on 16-byte-per-lane streaming reads; the Indirect models'
// kernel (gmx_indirect.hip) reads and writes single 16-bit table entries scattered over hundreds of megabytes per
// stream.  Each pattern below touches a KNOWN number of distinct 64-byte sectors / 128-byte lines of a buffer far
// larger than L2 + Infinity Cache, once, so the counter per access can be read off:
//   0  stream   every lane 16 B, coalesced                       (the guide's own case: expect bytes / 2)
//   1  u16@512  one 16-bit load per lane, 512 bytes apart        (one sector of one line per access)
//   2  u16@128  ... 128 bytes apart                              (every line once, one sector of it)
//   3  u16@64   ... 64 bytes apart                               (every sector once, two per line)
//   4  rmw@512  the same entry read, incremented, written back   (WRITE_SIZE per dirty sector)
//   5  rmw@64
// usage: fetch_calib <pattern> <n accesses (millions)>     prints one JSON line; run under
//   rocprofv3 --pmc FETCH_SIZE ...  /  --pmc WRITE_SIZE ...  /  --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_stream(const uint4* __restrict__ p, uint64_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_u16(const uint16_t* __restrict__ p, uint64_t n, uint64_t stride_elems, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    acc += p[i * stride_elems];
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_rmw(uint16_t* __restrict__ p, uint64_t n, uint64_t stride_elems) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i * stride_elems] = (uint16_t)(p[i * stride_elems] + 1);
}

int main(int argc, char** argv) {
  const int pat = argc > 1 ? atoi(argv[1]) : 0;
  const uint64_t n = (uint64_t)(argc > 2 ? atof(argv[2]) : 16.0) * 1000000ull;
  const uint64_t strides[] = {16, 512, 128, 64, 512, 64};
  const char* names[] = {"stream16", "u16@512", "u16@128", "u16@64", "rmw@512", "rmw@64"};
  const uint64_t span = n * strides[pat] + 4096;
  void* buf;
  CHK(hipMalloc(&buf, span));
  CHK(hipMemset(buf, 1, span));
  uint32_t* sink;
  CHK(hipMalloc(&sink, 4));
  CHK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0));
  const int blocks = 256 * 8, threads = 256;
  if (pat == 0)
    hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(threads), 0, 0, (const uint4*)buf, n, sink);
  else if (pat <= 3)
    hipLaunchKernelGGL(k_u16, dim3(blocks), dim3(threads), 0, 0, (const uint16_t*)buf, n, strides[pat] / 2, sink);
  else
    hipLaunchKernelGGL(k_rmw, dim3(blocks), dim3(threads), 0, 0, (uint16_t*)buf, n, strides[pat] / 2);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"pattern\": \"%s\", \"accesses\": %llu, \"span_bytes\": %llu, \"distinct_64B_sectors\": %llu, "
         "\"distinct_128B_lines\": %llu, \"kernel_ms\": %.3f}\n",
         names[pat], (unsigned long long)n, (unsigned long long)span,
         (unsigned long long)(pat == 0 ? n * 16 / 64 : n), (unsigned long long)(pat == 0 ? n * 16 / 128 : (strides[pat] >= 128 ? n : n / 2)), ms);
  return 0;
}
