// issue_ubench.hip -- what an instruction costs a wave that is alone on its SIMD (gfx950): the patterns
// gmx_stock_kernel's generated streams are made of, 256 repetitions each, timed with s_memtime by every block.
//   hipcc --offload-arch=gfx950 -O2 issue_ubench.hip -o /tmp/issue_ubench && /tmp/issue_ubench [blocks]
// Prints cycles per repetition (median over blocks) and the instruction count of the repetition.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

constexpr int kTests = 16;
#define R256(body) ".rept 256\n\t" body "\n\t.endr\n\t"

#define TIMED(id, body)                                                                      \
  {                                                                                          \
    uint64_t t0, t1;                                                                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t" body \
                 "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"                             \
                 : [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [p2] "+v"(p2), [q2] "+v"(q2) \
                 : [l] "v"(laddr), [s] "s"(sc)                                               \
                 : "memory", "vcc", "s40", "s41", "v40", "v41", "v42", "v43");               \
    if (threadIdx.x == 0) out[blockIdx.x * kTests + id] = (uint32_t)(t1 - t0);              \
  }

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64) ubench(uint32_t* out, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = seed * i;
  __syncthreads();
  float a = seed, b = seed * 2, c = seed * 3, d = seed * 5;
  f2 p2 = {seed, seed}, q2 = {seed * 7, seed * 9};
  const uint32_t laddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const float sc = seed * 11;
  TIMED(0, R256("v_add_f32 %[a], %[a], %[b]"))                                                  // dependent adds
  TIMED(1, R256("v_add_f32 %[a], %[a], %[b]\n\tv_add_f32 %[c], %[c], %[d]"))                    // two independent chains
  TIMED(2, R256("v_add_f32 %[a], %[a], %[b]\n\ts_nop 0"))                                       // add + nop
  TIMED(3, R256("v_add_f32 %[a], %[a], %[b]\n\ts_waitcnt lgkmcnt(7)"))                          // add + satisfied wait
  TIMED(4, R256("v_add_f32 %[a], %[a], %[b]\n\ts_mov_b32 s40, 5"))                              // add + SALU
  TIMED(5, R256("v_pk_mul_f32 %[p2], %[p2], %[q2]"))                                            // dependent packed multiply
  TIMED(6, R256("v_pk_mul_f32 v[40:41], %[p2], %[q2]\n\tv_add_f32 %[a], %[a], v40\n\tv_add_f32 %[a], %[a], v41"))  // the chain's step
  TIMED(7, R256("v_pk_mul_f32 v[40:41], %[p2], %[q2]\n\tv_pk_mul_f32 v[42:43], %[q2], %[p2]\n\tv_add_f32 %[a], %[a], v40\n\tv_add_f32 %[a], %[a], v41\n\tv_add_f32 %[a], %[a], v42\n\tv_add_f32 %[a], %[a], v43"))  // multiplies ahead of the adds
  TIMED(8, R256("v_readlane_b32 s40, %[a], 3\n\ts_nop 1\n\tv_mul_f32 %[c], s40, %[b]\n\tv_add_f32 %[a], %[a], %[c]"))  // the cascade's step
  TIMED(9, R256("v_mul_f32 %[c], %[s], %[b]\n\tv_add_f32 %[a], %[a], %[c]"))                    // mul + add
  TIMED(10, R256("v_add_f32_e64 %[a], %[a], %[b]"))                                             // 8-byte encoding
  TIMED(11, R256("v_fma_f64 v[40:41], v[40:41], v[42:43], v[40:41]"))                           // dependent fp64 fma
  TIMED(12, R256("v_add_f32 %[a], %[a], %[b]\n\tv_mov_b32 %[c], %[d]"))                         // add + independent move
  TIMED(13, R256("ds_read_b128 v[40:43], %[l]\n\tv_add_f32 %[a], %[a], %[b]\n\tv_add_f32 %[a], %[a], %[b]\n\tv_add_f32 %[a], %[a], %[b]"))  // one LDS read per 3 adds, never waited for
  TIMED(14, R256("s_mov_b32 s40, 5"))                                                           // SALU alone
  TIMED(15, R256("s_nop 0"))                                                                    // nops alone
  if (a + b + c + d + p2.x + q2.y == 12345.f) out[0] = 1;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1;
  uint32_t* d;
  hipMalloc(&d, (size_t)blocks * kTests * 4);
  std::vector<uint32_t> h((size_t)blocks * kTests);
  const char* names[kTests] = {"v_add dependent (1)", "2 independent v_add chains (2)", "v_add + s_nop 0 (2)", "v_add + s_waitcnt satisfied (2)",
                               "v_add + s_mov (2)", "v_pk_mul dependent (1)", "pk_mul + 2 dependent adds (3)", "2 pk_mul then 4 adds (6)",
                               "readlane, s_nop 1, v_mul, v_add (4)", "v_mul + v_add (2)", "v_add_e64 dependent (1)", "v_fma_f64 dependent (1)",
                               "v_add + v_mov (2)", "ds_read_b128 + 3 v_add (4)", "s_mov alone (1)", "s_nop 0 alone (1)"};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(ubench, dim3(blocks), dim3(64), 0, 0, d, 1.0f + rep);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  printf("%d block(s) of one wave; cycles per repetition (median over blocks)\n", blocks);
  for (int t = 0; t < kTests; ++t) {
    std::vector<uint32_t> v;
    for (int b = 0; b < blocks; ++b) v.push_back(h[(size_t)b * kTests + t]);
    std::sort(v.begin(), v.end());
    printf("  %-40s %6.2f\n", names[t], v[v.size() / 2] / 256.0);
  }
  return 0;
}
