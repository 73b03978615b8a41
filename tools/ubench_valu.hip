// Micro-benchmark: issue throughput of the integer VALU instructions the Goldilocks kernels are
// built from (gfx950). One wave per SIMD x 4 waves/SIMD, N independent chains per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITERS 2048
#define CHAINS 8

#define DEF_KERNEL(NAME, DECL, BODY)                                                   \
  __global__ void NAME(uint32_t *out, uint32_t seed) {                                 \
    uint32_t a[CHAINS], b[CHAINS];                                                     \
    uint64_t q[CHAINS];                                                                \
    _Pragma("unroll") for (int i = 0; i < CHAINS; i++) {                               \
      a[i] = seed * (i + 3) + threadIdx.x; b[i] = seed ^ (i * 77 + threadIdx.x);       \
      q[i] = ((uint64_t)a[i] << 32) | b[i];                                            \
    }                                                                                  \
    DECL;                                                                              \
    for (int it = 0; it < ITERS; it++) {                                               \
      _Pragma("unroll") for (int i = 0; i < CHAINS; i++) { BODY; }                     \
    }                                                                                  \
    uint32_t acc = 0;                                                                  \
    _Pragma("unroll") for (int i = 0; i < CHAINS; i++) acc += a[i] + b[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
  }

DEF_KERNEL(k_add_u32, , asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_lshl_add_u32, , asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_add3_u32, , asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_mad_u32_u24, , asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_mul_lo_u32, , asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_mul_hi_u32, , asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_mad_u64_u32, , asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc"))
DEF_KERNEL(k_lshl_add_u64, , asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[i]) : "v"(q[(i + 1) % CHAINS])))
DEF_KERNEL(k_add_co_pair, , asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(seed) : "vcc"))
DEF_KERNEL(k_cndmask, , asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc"))
DEF_KERNEL(k_alignbit, , asm volatile("v_alignbit_b32 %0, %0, %1, 22" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_and_or, , asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_lshlrev_b64, , asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(q[i])))
DEF_KERNEL(k_cmp_lt_u64, , asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 1) % CHAINS]) : "vcc"))
DEF_KERNEL(k_mul_u64, , q[i] = q[i] * q[(i + 1) % CHAINS])
DEF_KERNEL(k_pk_add_u16, , asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_dot4_i32_i8, , asm volatile("v_dot4_i32_i8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_fma_f64, , asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(q[i])))
DEF_KERNEL(k_sub_co_subb, , asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(seed) : "vcc"))

// round 4: which encodings / operations issue at the full rate (2 cycles per wave64 on a SIMD-32) when several waves share a SIMD
DEF_KERNEL(k_and_b32_vop2, , asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_xor_b32_vop2, , asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_sub_u32_vop2, , asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_lshlrev_b32_vop2, , asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i])))
DEF_KERNEL(k_mov_b32, , asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_cndmask_vop2, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[0]), "v"(b[0]) : "vcc"), asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_cndmask_sgpr, uint64_t m; asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(a[0]), "v"(b[0])), asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(m)))
DEF_KERNEL(k_add_co_vop2, , asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc"))
DEF_KERNEL(k_add_co_sgpr, uint64_t m, asm volatile("v_add_co_u32 %0, %1, %0, %2" : "+v"(a[i]), "=s"(m) : "v"(b[i])))
DEF_KERNEL(k_mul_u32_u24_vop2, , asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_fma_f32_vop3, , asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_fmac_f32_vop2, , asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_mul_f32_vop2, , asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_add_f32_vop2, , asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_pk_fma_f32, , asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(q[i])))
DEF_KERNEL(k_pk_add_f32, , asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(q[i])))
DEF_KERNEL(k_add_f64, , asm volatile("v_add_f64 %0, %0, %0" : "+v"(q[i])))
DEF_KERNEL(k_mul_f64, , asm volatile("v_mul_f64 %0, %0, %0" : "+v"(q[i])))
DEF_KERNEL(k_fmac_f64_vop2, , asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(q[i]) : "v"(q[(i + 1) % CHAINS])))
DEF_KERNEL(k_cvt_f64_u32, , asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(q[i]) : "v"(a[i])))
DEF_KERNEL(k_cvt_u32_f64, , asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(q[i])))
DEF_KERNEL(k_add_u32_vop3_sgpr, , asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_mad_u64_u32_sgpr, uint64_t m, asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(q[i]), "=s"(m) : "v"(a[i]), "v"(b[i])))
DEF_KERNEL(k_mad_nop_cnd, uint64_t m, asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0\n s_nop 1\n v_cndmask_b32 %2, 0, 1, %1" : "+v"(q[i]), "=s"(m), "+v"(a[i]) : "v"(b[i])))
DEF_KERNEL(k_snop, , asm volatile("s_nop 0"))

template <typename K>
double run(K kern, const char *name, int instr_per_body, uint32_t *d_out, int blocks, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u + r);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = (double)reps * blocks * (threads / 64) * ITERS * CHAINS * instr_per_body;
  // cycles per wave-instruction per SIMD, assuming 1024 SIMDs and the clock below
  double clk = 2.4e9;
  double cyc = (ms * 1e-3) * clk * 1024.0 / wave_instr;
  printf("%-16s %8.3f ms  %6.2f cycles/wave-instr/SIMD @2.4GHz  (%.2f T lane-ops/s)\n", name, ms / reps, cyc,
         wave_instr * 64 / (ms * 1e-3) / 1e12);
  return cyc;
}

int main() {
  int blocks = 256 * 8, threads = 256;  // 8 blocks/CU -> 8 waves/SIMD
  uint32_t *d_out;
  hipMalloc(&d_out, (size_t)blocks * threads * 4);
#define RUN(K, N) run(K, #K, N, d_out, blocks, threads)
  RUN(k_add_u32, 1); RUN(k_lshl_add_u32, 1); RUN(k_add3_u32, 1); RUN(k_mad_u32_u24, 1);
  RUN(k_mul_lo_u32, 1); RUN(k_mul_hi_u32, 1); RUN(k_mad_u64_u32, 1); RUN(k_lshl_add_u64, 1);
  RUN(k_add_co_pair, 2); RUN(k_sub_co_subb, 2); RUN(k_cndmask, 1); RUN(k_alignbit, 1); RUN(k_and_or, 1);
  RUN(k_lshlrev_b64, 1); RUN(k_cmp_lt_u64, 1); RUN(k_mul_u64, 1); RUN(k_pk_add_u16, 1);
  RUN(k_dot4_i32_i8, 1); RUN(k_fma_f64, 1);
  printf("# round 4: encodings and operations, 8 waves per SIMD\n");
  RUN(k_and_b32_vop2, 1); RUN(k_xor_b32_vop2, 1); RUN(k_sub_u32_vop2, 1); RUN(k_lshlrev_b32_vop2, 1); RUN(k_mov_b32, 1);
  RUN(k_cndmask_vop2, 1); RUN(k_cndmask_sgpr, 1); RUN(k_add_co_vop2, 1); RUN(k_add_co_sgpr, 1); RUN(k_mul_u32_u24_vop2, 1);
  RUN(k_fma_f32_vop3, 1); RUN(k_fmac_f32_vop2, 1); RUN(k_mul_f32_vop2, 1); RUN(k_add_f32_vop2, 1); RUN(k_pk_fma_f32, 1); RUN(k_pk_add_f32, 1);
  RUN(k_add_f64, 1); RUN(k_mul_f64, 1); RUN(k_fmac_f64_vop2, 1); RUN(k_cvt_f64_u32, 1); RUN(k_cvt_u32_f64, 1); RUN(k_add_u32_vop3_sgpr, 1);
  RUN(k_mad_u64_u32_sgpr, 1); RUN(k_mad_nop_cnd, 2); RUN(k_snop, 1);
  for (int w : {1, 2, 4}) {  // fewer waves per SIMD
    printf("# %d wave(s) per SIMD\n", w);
    run(k_add_u32, "k_add_u32", 1, d_out, 256 * w, 256); run(k_and_or, "k_and_or", 1, d_out, 256 * w, 256);
    run(k_fma_f64, "k_fma_f64", 1, d_out, 256 * w, 256); run(k_mad_u64_u32, "k_mad_u64_u32", 1, d_out, 256 * w, 256);
    run(k_fma_f32_vop3, "k_fma_f32_vop3", 1, d_out, 256 * w, 256);
  }
  hipFree(d_out);
  return 0;
}
