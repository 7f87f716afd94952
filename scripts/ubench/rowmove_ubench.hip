// rowmove_ubench.hip -- what one row-move instruction costs a wave that is alone on its SIMD (gfx950).
// Every candidate of gmx_stock_kernel's row moves, 29 (or 24) instructions back to back, timed with
// s_memtime by lane 0 of every block: cycles until the last one has ISSUED, and until all have COMPLETED.
//   hipcc --offload-arch=gfx950 -O2 rowmove_ubench.hip -o /tmp/rowmove_ubench && /tmp/rowmove_ubench [blocks]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define REP(n, ins) ".set o,0\n\t.rept " #n "\n\t" ins "\n\t.set o,o+16\n\t.endr\n\t"
#define REPP(n, ins) ".set o,0\n\t.rept " #n "\n\t" ins "\n\t.set o,o+528\n\t.endr\n\t"

constexpr int kTests = 22;

#define TIMED(id, setup_exec, body, wait)                                                     \
  {                                                                                           \
    __builtin_amdgcn_s_barrier();                                                             \
    uint64_t sv, t0, t1, t2;                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_mov_b64 %[sv], exec\n\t" setup_exec    \
                 "s_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t" body                         \
                 "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t" wait                         \
                 "s_memtime %[t2]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %[sv]\n\t"     \
                 : [sv] "=&s"(sv), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2)           \
                 : [p] "v"(prow), [pc] "s"(pbase), [voff] "v"(lane16), [l] "v"(lrow), [lc] "v"(lco), [m2] "s"(m2), \
                   [m33] "s"(m33), [m32] "s"(m32), [m24] "s"(m24), [z] "s"(0ull)             \
                 : "memory", "v40", "v41", "v42", "v43", "a40", "a41", "a42", "a43", "v36", "v37", "v38", "v39");         \
    if (threadIdx.x == 0) {                                                                   \
      out[(blockIdx.x * kTests + id) * 2 + 0] = (uint32_t)(t1 - t0);                          \
      out[(blockIdx.x * kTests + id) * 2 + 1] = (uint32_t)(t2 - t0);                          \
    }                                                                                         \
  }

__global__ void __launch_bounds__(64) ubench(uint8_t* rows, uint32_t* out, int rep) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x;
  // every lane its own 512-byte row, rows of a block far apart like gate-table rows are
  uint8_t* const prow = rows + ((uint64_t)blockIdx.x * 64 + lane) * 65536ull + (uint64_t)(rep & 15) * 4096;
  const uint64_t pbase = (uint64_t)(rows + ((uint64_t)blockIdx.x * 64) * 65536ull + (uint64_t)(rep & 15) * 4096 + 2048);
  const uint32_t lane16 = lane * 16u;
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const uint32_t lrow = lbase + lane * 528u;   // the lane's row of an image (transposed access)
  const uint32_t lco = lbase + lane16;         // coalesced access of one image row
  const uint64_t m2 = (1ull << 5) | (1ull << 17), m33 = (1ull << 33) - 1, m32 = 0xffffffffull, m24 = (1ull << 24) - 1;
  for (int i = lane; i < 24 * 528 / 4; i += 64) ((uint32_t*)lds)[i] = i;
  // 0/1: lane-private loads, 33 lanes / 2 lanes
  TIMED(0, "s_mov_b64 exec, %[m33]\n\t", REP(29, "global_load_dwordx4 a[40:43], %[p], off offset:o"), "s_waitcnt vmcnt(0)\n\t")
  TIMED(1, "s_mov_b64 exec, %[m2]\n\t", REP(29, "global_load_dwordx4 a[40:43], %[p], off offset:o"), "s_waitcnt vmcnt(0)\n\t")
  // 2/3: lane-private stores
  TIMED(2, "s_mov_b64 exec, %[m33]\n\t", REP(29, "global_store_dwordx4 %[p], v[40:43], off offset:o"), "s_waitcnt vmcnt(0)\n\t")
  TIMED(3, "s_mov_b64 exec, %[m2]\n\t", REP(29, "global_store_dwordx4 %[p], v[40:43], off offset:o"), "s_waitcnt vmcnt(0)\n\t")
  // 4/5: coalesced LDS-DMA of 24 rows (same global row, m0 fixed: issue cost only), 32 lanes / exec = 0
  TIMED(4, "s_mov_b64 exec, %[m32]\n\ts_mov_b32 m0, 0\n\t", REP(24, "global_load_lds_dwordx4 %[voff], %[pc] offset:0"), "s_waitcnt vmcnt(0)\n\t")
  TIMED(5, "s_mov_b64 exec, %[z]\n\ts_mov_b32 m0, 0\n\t", REP(24, "global_load_lds_dwordx4 %[voff], %[pc] offset:0"), "s_waitcnt vmcnt(0)\n\t")
  // 6/7: coalesced stores of 24 rows, 32 lanes / exec = 0
  TIMED(6, "s_mov_b64 exec, %[m32]\n\t", REP(24, "global_store_dwordx4 %[voff], v[40:43], %[pc] offset:0"), "s_waitcnt vmcnt(0)\n\t")
  TIMED(7, "s_mov_b64 exec, %[z]\n\t", REP(24, "global_store_dwordx4 %[voff], v[40:43], %[pc] offset:0"), "s_waitcnt vmcnt(0)\n\t")
  // 8/9: transposed ds_read_b128 of the lane's image row, 24 lanes / 2 lanes
  TIMED(8, "s_mov_b64 exec, %[m24]\n\t", REP(29, "ds_read_b128 v[40:43], %[l] offset:o"), "")
  TIMED(9, "s_mov_b64 exec, %[m2]\n\t", REP(29, "ds_read_b128 v[40:43], %[l] offset:o"), "")
  // 10/11: transposed ds_write_b128
  TIMED(10, "s_mov_b64 exec, %[m24]\n\t", REP(29, "ds_write_b128 %[l], v[40:43] offset:o"), "")
  TIMED(11, "s_mov_b64 exec, %[m2]\n\t", REP(29, "ds_write_b128 %[l], v[40:43] offset:o"), "")
  // 12: coalesced ds_read_b128 of 24 image rows into AGPRs, 32 lanes
  TIMED(12, "s_mov_b64 exec, %[m32]\n\t", REPP(24, "ds_read_b128 a[40:43], %[lc] offset:o"), "")
  // 13: 116 v_accvgpr_read under a 2-lane mask
  TIMED(13, "s_mov_b64 exec, %[m2]\n\t", ".rept 116\n\tv_accvgpr_read_b32 v40, a40\n\t.endr\n\t", "")
  // 14: one LDS-DMA + wait: latency of a single row fetch
  TIMED(14, "s_mov_b64 exec, %[m32]\n\ts_mov_b32 m0, 0\n\t", "global_load_lds_dwordx4 %[voff], %[pc] offset:512\n\t", "s_waitcnt vmcnt(0)\n\t")
  // 15: 29 s_nop 0 (the scale: 1 instruction / issue slot)
  TIMED(15, "", ".rept 29\n\ts_nop 0\n\t.endr\n\t", "")
  // 16..19: can VALU work hide the LDS write path?  232 packed multiplies alone / with a ds_write_b128 after every 8th,
  // 116 alone / with one after every 4th
  TIMED(16, "", ".rept 232\n\tv_pk_mul_f32 v[40:41], v[42:43], v[42:43]\n\t.endr\n\t", "")
  TIMED(17, "", ".set o,0\n\t.rept 29\n\t.rept 8\n\tv_pk_mul_f32 v[40:41], v[42:43], v[42:43]\n\t.endr\n\tds_write_b128 %[l], v[36:39] offset:o\n\t.set o,o+16\n\t.endr\n\t", "")
  TIMED(18, "", ".rept 116\n\tv_pk_mul_f32 v[40:41], v[42:43], v[42:43]\n\t.endr\n\t", "")
  TIMED(19, "", ".set o,0\n\t.rept 29\n\t.rept 4\n\tv_pk_mul_f32 v[40:41], v[42:43], v[42:43]\n\t.endr\n\tds_write_b128 %[l], v[36:39] offset:o\n\t.set o,o+16\n\t.endr\n\t", "")
  // 20/21: the same question for ds_read_b128 (transposed, 24 lanes) behind 4 multiplies each, and for 29 lane-private
  // stores of 2 lanes behind 8 multiplies each
  TIMED(20, "s_mov_b64 exec, %[m24]\n\t", ".set o,0\n\t.rept 29\n\t.rept 4\n\tv_pk_mul_f32 v[40:41], v[42:43], v[42:43]\n\t.endr\n\tds_read_b128 v[36:39], %[l] offset:o\n\t.set o,o+16\n\t.endr\n\t", "")
  TIMED(21, "", ".set o,0\n\t.rept 29\n\t.rept 8\n\tv_pk_mul_f32 v[40:41], v[42:43], v[42:43]\n\t.endr\n\ts_mov_b64 exec, %[m2]\n\tglobal_store_dwordx4 %[p], v[36:39], off offset:o\n\ts_mov_b64 exec, %[m33]\n\t.set o,o+16\n\t.endr\n\t", "s_waitcnt vmcnt(0)\n\t")
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024;
  uint8_t* rows;
  uint32_t* out;
  const size_t bytes = (size_t)blocks * 64 * 65536;
  if (hipMalloc((void**)&rows, bytes) != hipSuccess) return 1;
  hipMemset(rows, 0, bytes);
  hipMalloc((void**)&out, (size_t)blocks * kTests * 8);
  std::vector<uint32_t> h((size_t)blocks * kTests * 2);
  const char* names[kTests] = {"29 lane-private global_load_dwordx4, 33 lanes", "  ... 2 lanes",
                               "29 lane-private global_store_dwordx4, 33 lanes", "  ... 2 lanes",
                               "24 coalesced global_load_lds_dwordx4, 32 lanes", "  ... exec = 0",
                               "24 coalesced global_store_dwordx4, 32 lanes", "  ... exec = 0",
                               "29 transposed ds_read_b128, 24 lanes", "  ... 2 lanes",
                               "29 transposed ds_write_b128, 24 lanes", "  ... 2 lanes",
                               "24 coalesced ds_read_b128 -> AGPR, 32 lanes", "116 v_accvgpr_read_b32",
                               "1 LDS-DMA row + wait", "29 s_nop 0", "232 v_pk_mul_f32", "  ... + 29 ds_write_b128, one per 8",
                               "116 v_pk_mul_f32", "  ... + 29 ds_write_b128, one per 4", "116 v_pk_mul_f32 + 29 ds_read_b128, one per 4",
                               "232 v_pk_mul_f32 + 29 2-lane global_store_dwordx4"};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(ubench, dim3(blocks), dim3(64), 40960, 0, rows, out, rep);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
  }
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  printf("%d blocks (one wave each, 40 KB LDS: <= 4 per CU); cycles: issued / completed, mean over blocks\n", blocks);
  for (int t = 0; t < kTests; ++t) {
    double a = 0, b = 0;
    for (int k = 0; k < blocks; ++k) {
      a += h[((size_t)k * kTests + t) * 2];
      b += h[((size_t)k * kTests + t) * 2 + 1];
    }
    printf("  %-52s %8.0f %8.0f\n", names[t], a / blocks, b / blocks);
  }
  return 0;
}
