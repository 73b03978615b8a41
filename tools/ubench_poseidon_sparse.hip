// Prototype + micro-benchmark: Poseidon with the sparse ("fast") factorisation of the 22 partial rounds, written
// for the MI355X integer VALU: every product by a 64-bit constant is (lo32(x)*w + hi32(x)*w') with w' = w*2^32 mod p,
// both constants split into 22/21/21-bit limbs, so a dot product of up to 12 terms accumulates in three u64 chains
// of v_mad_u64_u32 without carries and is reduced once.
// Compares against poseidon::permute (bit-exact) and times both.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_poseidon_sparse tools/ubench_poseidon_sparse.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../city-rollup_amd/csrc/gl.h"
#include "../city-rollup_amd/csrc/poseidon_tables.h"
#include "../city-rollup_amd/csrc/poseidon.h"

namespace sp {
using poseidon::W;
__constant__ uint32_t d_DOT[22][11][6];  // what_j limbs (w0,w1,w2) then (what_j * 2^32) limbs
__constant__ uint64_t d_VS[22][11];
__constant__ uint64_t d_K[22];
__constant__ uint32_t d_DENSE[12][12][6];  // M' = diag(1, INIT) * M
__constant__ uint64_t d_FIRST2[12];        // diag(1, INIT) * FIRST

template <int NT>
__device__ __forceinline__ uint64_t dot_limbs(const uint64_t *x, const uint32_t (*L)[6], gl::u128 extra) {
  uint64_t c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const uint32_t sl = (uint32_t)x[j], sh = (uint32_t)(x[j] >> 32);
    c0 += (uint64_t)sl * L[j][0];
    c1 += (uint64_t)sl * L[j][1];
    c2 += (uint64_t)sl * L[j][2];
    c0 += (uint64_t)sh * L[j][3];
    c1 += (uint64_t)sh * L[j][4];
    c2 += (uint64_t)sh * L[j][5];
  }
  gl::u128 t = extra + c0 + ((gl::u128)c1 << 22) + ((gl::u128)c2 << 43);
  return gl::reduce128_lazy((uint64_t)t, (uint64_t)(t >> 64));
}

__device__ __forceinline__ void permute_sparse(uint64_t (&s)[W]) {
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = poseidon::add_const_lazy(s[i], poseidon::rc(i));
#pragma unroll 1
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = poseidon::sbox_lazy(s[i]);
    poseidon::mds_layer(s, (r + 1) * W);
  }
  // 4th full round: S-box, then the dense layer diag(1, INIT) * (M x + FIRST)
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = poseidon::sbox_lazy(s[i]);
  {
    uint64_t y[W];
#pragma unroll 1
    for (int r = 0; r < W; r++) y[r] = dot_limbs<12>(s, d_DENSE[r], (gl::u128)d_FIRST2[r]);
#pragma unroll
    for (int r = 0; r < W; r++) s[r] = y[r];
  }
#pragma unroll 1
  for (int i = 0; i < 22; i++) {
    const uint64_t s0 = poseidon::add_const_lazy(poseidon::sbox_lazy(s[0]), d_K[i]);
    const uint64_t d = dot_limbs<11>(s + 1, d_DOT[i], (gl::u128)s0 * 25u);  // m00 = 17 + 8
#pragma unroll
    for (int j = 0; j < 11; j++) s[1 + j] = gl::mul_add_lazy(s0, d_VS[i][j], s[1 + j]);
    s[0] = d;
  }
  // last four full rounds: constants of round 26 are not folded anywhere yet
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = poseidon::add_const_lazy(s[i], poseidon::rc(26 * W + i));
#pragma unroll 1
  for (int r = 26; r < 30; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = poseidon::sbox_lazy(s[i]);
    poseidon::mds_layer(s, r + 1 < 30 ? (r + 1) * W : -1);
  }
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(s[i]);
}

template <int VARIANT>  // 0: textbook rounds, 1: sparse partial rounds (this file), 2: poseidon::permute (plane-resident)
__global__ __launch_bounds__(256) void k_chain(uint64_t *out, uint64_t seed, int reps) {
  uint64_t s[W];
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(seed * (i + 1) + t * 0x9E3779B97F4A7C15ull + i);
  for (int r = 0; r < reps; r++) {
    if (VARIANT == 1) permute_sparse(s);
    else if (VARIANT == 2) poseidon::permute(s);
    else poseidon::permute_textbook(s);
  }
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < W; i++) acc ^= s[i] + i;
  out[t] = acc;
}
}  // namespace sp

static uint64_t hmul(uint64_t a, uint64_t b) { return gl::mul(a, b); }

int main() {
  // host tables
  uint64_t M[12][12];
  for (int r = 0; r < 12; r++)
    for (int c = 0; c < 12; c++) M[r][c] = POSEIDON_MDS_CIRC[(c - r + 12) % 12] + (r == 0 && c == 0 ? 8 : 0);
  auto limbs = [](uint64_t w, uint32_t *o) {
    uint64_t w2 = hmul(w, 1ull << 32);
    o[0] = w & 0x3FFFFF; o[1] = (w >> 22) & 0x1FFFFF; o[2] = (uint32_t)(w >> 43);
    o[3] = w2 & 0x3FFFFF; o[4] = (w2 >> 22) & 0x1FFFFF; o[5] = (uint32_t)(w2 >> 43);
  };
  static uint32_t DOT[22][11][6], DENSE[12][12][6];
  static uint64_t VS[22][11], K[22], FIRST2[12];
  for (int i = 0; i < 22; i++) {
    K[i] = POSEIDON_FAST_K[i];
    for (int j = 0; j < 11; j++) { limbs(POSEIDON_FAST_WHATS[i * 11 + j], DOT[i][j]); VS[i][j] = POSEIDON_FAST_VS[i * 11 + j]; }
  }
  for (int r = 0; r < 12; r++) {
    uint64_t f = 0;
    for (int c = 0; c < 12; c++) {
      uint64_t m = 0;  // (diag(1, INIT) * M)[r][c]
      if (r == 0) m = M[0][c];
      else for (int k = 1; k < 12; k++) m = gl::add(m, hmul(POSEIDON_FAST_INIT[(r - 1) * 11 + (k - 1)], M[k][c]));
      limbs(m, DENSE[r][c]);
    }
    if (r == 0) f = POSEIDON_FAST_FIRST[0];
    else for (int k = 1; k < 12; k++) f = gl::add(f, hmul(POSEIDON_FAST_INIT[(r - 1) * 11 + (k - 1)], POSEIDON_FAST_FIRST[k]));
    FIRST2[r] = f;
  }
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RC), POSEIDON_RC, sizeof POSEIDON_RC);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RCD), POSEIDON_RCD, sizeof POSEIDON_RCD);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDK), POSEIDON_DOMD_K, sizeof POSEIDON_DOMD_K);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDLAST), POSEIDON_DOMD_LAST, sizeof POSEIDON_DOMD_LAST);
  hipMemcpyToSymbol(HIP_SYMBOL(sp::d_DOT), DOT, sizeof DOT);
  hipMemcpyToSymbol(HIP_SYMBOL(sp::d_VS), VS, sizeof VS);
  hipMemcpyToSymbol(HIP_SYMBOL(sp::d_K), K, sizeof K);
  hipMemcpyToSymbol(HIP_SYMBOL(sp::d_DENSE), DENSE, sizeof DENSE);
  hipMemcpyToSymbol(HIP_SYMBOL(sp::d_FIRST2), FIRST2, sizeof FIRST2);

  const int blocks = 256 * 16, reps = 64;
  const size_t n = (size_t)blocks * 256;
  uint64_t *a, *b;
  hipMalloc(&a, n * 8);
  hipMalloc(&b, n * 8);
  uint64_t *c;
  hipMalloc(&c, n * 8);
  hipLaunchKernelGGL(sp::k_chain<0>, dim3(blocks), dim3(256), 0, 0, a, 12345ull, 3);
  hipLaunchKernelGGL(sp::k_chain<1>, dim3(blocks), dim3(256), 0, 0, b, 12345ull, 3);
  hipLaunchKernelGGL(sp::k_chain<2>, dim3(blocks), dim3(256), 0, 0, c, 12345ull, 3);
  std::vector<uint64_t> ha(n), hb(n), hc(n);
  hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) bad += (ha[i] != hb[i]) + (ha[i] != hc[i]);
  printf("mismatches: %zu of %zu\n", bad, n);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char *names[3] = {"textbook rounds            ", "sparse partial rounds      ", "plane-resident partial rnds"};
  for (int variant = 0; variant < 3; variant++) {
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
      hipEventRecord(e0, 0);
      if (variant == 1) hipLaunchKernelGGL(sp::k_chain<1>, dim3(blocks), dim3(256), 0, 0, b, 777ull, reps);
      else if (variant == 2) hipLaunchKernelGGL(sp::k_chain<2>, dim3(blocks), dim3(256), 0, 0, c, 777ull, reps);
      else hipLaunchKernelGGL(sp::k_chain<0>, dim3(blocks), dim3(256), 0, 0, a, 777ull, reps);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: %.3f ms for %zu permutations -> %.3f G perm/s\n", names[variant], best,
           n * reps, (double)n * reps / best / 1e6);
  }
  return bad != 0;
}
