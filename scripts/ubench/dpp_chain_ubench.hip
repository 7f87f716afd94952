// dpp_chain_ubench.hip -- what one element of a strict left-to-right sum costs when the row is spread over the lanes of
// a DPP row and the accumulator travels from lane to lane (acc = rot(acc) + p: gmx_single_kernel's prefix chain), for a
// wave that is alone on its SIMD (gfx950) -- the design DESIGN.md section 8.1 costs for the four bit-level mixers of
// gmx_stock_kernel.  256 repetitions each, timed with s_memtime.
//   hipcc --offload-arch=gfx950 -O2 dpp_chain_ubench.hip -o /tmp/dpp_chain_ubench && /tmp/dpp_chain_ubench [blocks]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

constexpr int kTests = 6;
#define R256(body) ".rept 256\n\t" body "\n\t.endr\n\t"
#define DPP_ADD "v_add_f32_dpp %[a], %[a], %[b] row_ror:1 row_mask:0xf bank_mask:0xf"

#define TIMED(id, body)                                                                      \
  {                                                                                          \
    uint64_t t0, t1;                                                                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t" body \
                 "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"                             \
                 : [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d) \
                 : [hi] "s"(0xffffffff00000000ull)                                           \
                 : "memory", "vcc", "s40", "s41");                                           \
    if (threadIdx.x == 0) out[blockIdx.x * kTests + id] = (uint32_t)(t1 - t0);              \
  }

__global__ void __launch_bounds__(64) ubench(uint32_t* out, float seed) {
  float a = seed, b = seed * 2, c = seed * 3, d = seed * 5;
  // the chain's step with nothing to fill the two wait states between a VALU write and the DPP read of it
  TIMED(0, R256(DPP_ADD "\n\ts_nop 1"))
  // ... filled by one instruction of another chain (what the main streams could offer if the registers allowed)
  TIMED(1, R256(DPP_ADD "\n\tv_add_f32 %[c], %[c], %[d]\n\ts_nop 0"))
  // ... filled by two
  TIMED(2, R256(DPP_ADD "\n\tv_add_f32 %[c], %[c], %[d]\n\tv_add_f32 %[d], %[d], %[b]"))
  // the spread rows live in lanes 32-63 only: exec = the upper half for the chain's step, all lanes for the filler
  TIMED(3, R256("s_mov_b64 s[40:41], exec\n\ts_mov_b64 exec, %[hi]\n\t" DPP_ADD "\n\ts_mov_b64 exec, s[40:41]\n\tv_add_f32 %[c], %[c], %[d]"))
  // four elements under one switch of exec (the rounds of section 8.1): 4 steps with their wait states, 2 fillers
  TIMED(4, R256("s_mov_b64 s[40:41], exec\n\ts_mov_b64 exec, %[hi]\n\t" DPP_ADD "\n\ts_nop 1\n\t" DPP_ADD "\n\ts_nop 1\n\t" DPP_ADD "\n\ts_nop 1\n\t" DPP_ADD
                "\n\ts_mov_b64 exec, s[40:41]\n\tv_add_f32 %[c], %[c], %[d]\n\tv_add_f32 %[d], %[d], %[b]"))
  // reference: a dependent v_add without DPP
  TIMED(5, R256("v_add_f32 %[a], %[a], %[b]"))
  if (a + b + c + d == 12345.f) out[0] = 1;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1;
  uint32_t* d;
  hipMalloc(&d, (size_t)blocks * kTests * 4);
  std::vector<uint32_t> h((size_t)blocks * kTests);
  const char* names[kTests] = {"dpp add + s_nop 1                (1 element)", "dpp add + 1 filler + s_nop 0      (1 element)",
                               "dpp add + 2 fillers               (1 element)", "exec hi, dpp add, exec all, filler (1 element)",
                               "exec hi, 4 x (dpp add, s_nop 1), exec all, 2 fillers (4 elements)", "plain dependent v_add"};
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
    printf("  %-72s %6.2f\n", names[t], v[v.size() / 2] / 256.0);
  }
  return 0;
}
