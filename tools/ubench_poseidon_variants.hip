// Micro-benchmark: poseidon::permute as compiled (with whatever -DPOSEIDON_* switches the build line sets) against
// poseidon::permute_textbook — bit-exact comparison, then a throughput chain (no memory traffic in the timed loop).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPOSEIDON_...] -o ubench_poseidon_variants tools/ubench_poseidon_variants.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../city-rollup_amd/csrc/gl.h"
#include "../city-rollup_amd/csrc/poseidon_tables.h"
#include "../city-rollup_amd/csrc/poseidon.h"

using poseidon::W;

#ifdef WAVES  // occupancy experiment: -DWAVES=5 asks the register allocator for at most 512/5 VGPRs
#define OCC __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
#else
#define OCC
#endif
template <int VARIANT>  // 0: textbook rounds, 1: poseidon::permute
__global__ __launch_bounds__(256) OCC void k_chain(uint64_t *out, uint64_t seed, int reps) {
  uint64_t s[W];
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(seed * (i + 1) + t * 0x9E3779B97F4A7C15ull + i);
  if (t < 12) s[t] = gl::P - 1;  // a few extreme values
  if (t == 13)
    for (int i = 0; i < W; i++) s[i] = 0;
  for (int r = 0; r < reps; r++) {
    if (VARIANT == 1) poseidon::permute(s);
    else poseidon::permute_textbook(s);
  }
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < W; i++) acc ^= s[i] * (2 * i + 1);
  out[t] = acc;
}

int main() {
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RC), POSEIDON_RC, sizeof POSEIDON_RC);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RCD), POSEIDON_RCD, sizeof POSEIDON_RCD);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDK), POSEIDON_DOMD_K, sizeof POSEIDON_DOMD_K);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDLAST), POSEIDON_DOMD_LAST, sizeof POSEIDON_DOMD_LAST);
  const int blocks = 256 * 16, reps = 64;
  const size_t n = (size_t)blocks * 256;
  uint64_t *a, *b;
  hipMalloc(&a, n * 8);
  hipMalloc(&b, n * 8);
  hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, a, 12345ull, 5);
  hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, b, 12345ull, 5);
  std::vector<uint64_t> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) bad += ha[i] != hb[i];
  printf("mismatches: %zu of %zu\n", bad, n);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char *names[2] = {"textbook rounds ", "poseidon::permute"};
  for (int variant = 0; variant < 2; variant++) {
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
      hipEventRecord(e0, 0);
      if (variant == 1) hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, b, 777ull, reps);
      else hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, a, 777ull, reps);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: %.3f ms for %zu permutations -> %.3f G perm/s\n", names[variant], best, n * reps, (double)n * reps / best / 1e6);
  }
  // the timed chains started from the same states: 64 permutations deep, any single wrong output would have propagated
  hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
  size_t bad_long = 0;
  for (size_t i = 0; i < n; i++) bad_long += ha[i] != hb[i];
  printf("mismatches after %d chained permutations of %zu states: %zu\n", reps, n, bad_long);
  // more seeds: 16 x 2^20 chains of 32
  size_t bad_seeds = 0;
  for (uint64_t seed = 1; seed <= 16; seed++) {
    hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, a, seed * 0x9E3779B97F4A7C15ull, 32);
    hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, b, seed * 0x9E3779B97F4A7C15ull, 32);
    hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < n; i++) bad_seeds += ha[i] != hb[i];
  }
  printf("mismatches over 16 more seeds (%zu permutations): %zu\n", (size_t)16 * n * 32, bad_seeds);
  return (bad | bad_long | bad_seeds) != 0;
}
