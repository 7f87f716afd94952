// hbm_random_rows.hip -- measurement aid, not part of the product: what HBM3E on this MI355X
// sustains for the access pattern of BASELINE configs[1] -- read one random, 1 KiB-aligned
// 1 KiB row, write it back, nothing else -- so that gmx_single_kernel's achieved bandwidth can
// be put next to the ceiling of its pattern and not only next to the 8 TB/s spec figure.
//   hipcc --offload-arch=gfx950 -O3 scripts/hbm_random_rows.hip -o /tmp/hbm_random_rows
//   /tmp/hbm_random_rows [GiB=192] [rows per wave=2048] [mode: 0 read+write, 1 read only, 2 sequential r+w]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

template <int MODE, int DEPTH, bool NT = false>
__global__ void __launch_bounds__(256) rows_kernel(float4* buf, uint64_t n_rows, int iters, uint64_t seed) {
  const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) / 64;
  const int lane = threadIdx.x & 63;
  const uint64_t n_waves = (uint64_t)gridDim.x * blockDim.x / 64;
  for (int it = 0; it < iters; it += DEPTH) {
    float4 v[DEPTH];
    uint64_t r[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      r[d] = MODE == 2 ? ((wave + (uint64_t)(it + d) * n_waves) % n_rows)
                       : mix(seed + wave * 1000003ull + (uint64_t)(it + d)) % n_rows;
      if (NT) {
        const vf4 q = __builtin_nontemporal_load((const vf4*)&buf[r[d] * 64 + lane]);
        v[d] = make_float4(q.x, q.y, q.z, q.w);
      } else {
        v[d] = buf[r[d] * 64 + lane];
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      v[d].x += 1.0f;
      if (MODE != 1) {
        if (NT) __builtin_nontemporal_store((vf4){v[d].x, v[d].y, v[d].z, v[d].w}, (vf4*)&buf[r[d] * 64 + lane]);
        else buf[r[d] * 64 + lane] = v[d];
      }
      else if (v[d].x == 12345.678f) buf[lane] = v[d];
    }
  }
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 192.0;
  const int iters = argc > 2 ? atoi(argv[2]) : 2048;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;  // 3: random read+write with non-temporal hints
  const uint64_t bytes = (uint64_t)(gib * 1024.0 * 1024.0 * 1024.0) / 1024 * 1024;
  const uint64_t n_rows = bytes / 1024;
  float4* buf;
  CHECK(hipMalloc((void**)&buf, bytes));
  CHECK(hipMemset(buf, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block) x waves_per_simd
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      if (mode == 3) hipLaunchKernelGGL((rows_kernel<0, 8, true>), dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters, 77ull + rep);
      else if (mode == 0) hipLaunchKernelGGL((rows_kernel<0, 8>), dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters, 77ull + rep);
      else if (mode == 1) hipLaunchKernelGGL((rows_kernel<1, 8>), dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters, 77ull + rep);
      else hipLaunchKernelGGL((rows_kernel<2, 8>), dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters, 77ull + rep);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double moved = (double)blocks * 4 * iters * 1024.0 * (mode == 1 ? 1 : 2);
    printf("{\"pattern\": \"%s\", \"buffer_GiB\": %.0f, \"waves_per_simd\": %d, \"GB_per_s\": %.1f, \"ms\": %.3f}\n",
           mode == 3 ? "random 1KiB row read+write, nontemporal" : mode == 0 ? "random 1KiB row read+write" : mode == 1 ? "random 1KiB row read" : "sequential 1KiB rows read+write",
           gib, waves_per_simd, moved / (best * 1e-3) / 1e9, best);
  }
  return 0;
}
