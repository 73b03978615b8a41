// Experiment (BASELINE.json north_star: "MFMA only for the dense matrix step inside Poseidon's MDS if it measures as a
// contraction"): the MDS layer of Poseidon-Goldilocks on the matrix cores, against the shift-add VALU version the
// kernels use (poseidon::mds_layer).
//
// Formulation that needs NO cross-lane data movement: V_MFMA_I32_4X4X4_16B_I8 computes 16 independent 4x4x4 products,
// one per group of four lanes; lane l supplies column l%4 of B (4 signed bytes) and receives column l%4 of D (4 x i32),
// while row m of A comes from lane m of the group. With every group supplying the same constant 4x4 block of the MDS
// matrix, each lane gets (block) x (four bytes of ITS OWN state): a lane-per-state mat-vec.
//   state element = 8 bytes  ->  8 byte planes; per plane the 12x12 matrix = 3x3 blocks  ->  72 MFMAs per MDS
//   bytes are unsigned, the MFMA is signed: x ^ 0x80 = x - 128, and 128 * rowsum(M) is preloaded into the accumulator
//   the 96 accumulators (< 2^17 each) are recombined into 12 lazy u64 (the VALU cost that remains)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_mds_mfma tools/ubench_mds_mfma.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../city-rollup_amd/csrc/gl.h"
#include "../city-rollup_amd/csrc/poseidon_tables.h"
#include "../city-rollup_amd/csrc/poseidon.h"

namespace mm {
using poseidon::W;
typedef int v4i __attribute__((ext_vector_type(4)));
__constant__ uint64_t d_ZERO[12];  // the last round adds no constants

struct Consts {
  uint32_t a[3];  // row (lane % 4) of the block at circulant distance d = (cg - rg) mod 3, bytes k = 0..3
  uint32_t a00;   // the (0, 0) block: + 8 on the diagonal entry of row 0
};
__device__ __forceinline__ Consts make_consts() {
  const int C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const int m = threadIdx.x & 3;
  Consts K;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t e = 0;
#pragma unroll
      for (int mm_ = 0; mm_ < 4; mm_++)
        if (m == mm_) e = C[(4 * d + k - mm_ + 12) % 12];
      v |= e << (8 * k);
    }
    K.a[d] = v;
  }
  K.a00 = K.a[0] + (m == 0 ? 8u : 0u);
  return K;
}

// 4x4 byte transpose: o_j = (a.byte j, b.byte j, c.byte j, d.byte j)
__device__ __forceinline__ void transpose4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t &o0, uint32_t &o1, uint32_t &o2,
                                           uint32_t &o3) {
  const uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
  const uint32_t t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
  const uint32_t t2 = __builtin_amdgcn_perm(d, c, 0x05010400u);
  const uint32_t t3 = __builtin_amdgcn_perm(d, c, 0x07030602u);
  o0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
  o1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  o2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
  o3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// s <- MDS * s + c   (lazy u64 in and out)
__device__ __forceinline__ void mds_mfma(uint64_t (&s)[W], const Consts &K, const uint64_t *c) {
  uint32_t xt[8][3];
#pragma unroll
  for (int g = 0; g < 3; g++) {
    transpose4((uint32_t)s[4 * g], (uint32_t)s[4 * g + 1], (uint32_t)s[4 * g + 2], (uint32_t)s[4 * g + 3], xt[0][g], xt[1][g], xt[2][g],
               xt[3][g]);
    transpose4((uint32_t)(s[4 * g] >> 32), (uint32_t)(s[4 * g + 1] >> 32), (uint32_t)(s[4 * g + 2] >> 32), (uint32_t)(s[4 * g + 3] >> 32),
               xt[4][g], xt[5][g], xt[6][g], xt[7][g]);
  }
#pragma unroll
  for (int j = 0; j < 8; j++)
#pragma unroll
    for (int g = 0; g < 3; g++) xt[j][g] ^= 0x80808080u;
  const v4i bias0 = {128 * 264, 128 * 256, 128 * 256, 128 * 256}, bias = {128 * 256, 128 * 256, 128 * 256, 128 * 256};
  v4i p[4][3];  // p[k][rg] = plane 2k + 256 * plane 2k+1
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v4i acc[2][3];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int j = 2 * k + h;
#pragma unroll
      for (int rg = 0; rg < 3; rg++) {
        v4i a = rg == 0 ? bias0 : bias;
#pragma unroll
        for (int cg = 0; cg < 3; cg++) {
          const uint32_t A = (rg == 0 && cg == 0) ? K.a00 : K.a[(cg - rg + 3) % 3];
          a = __builtin_amdgcn_mfma_i32_4x4x4i8((int)A, (int)xt[j][cg], a, 0, 0, 0);
        }
        acc[h][rg] = a;
      }
    }
#pragma unroll
    for (int rg = 0; rg < 3; rg++) p[k][rg] = acc[0][rg] + (acc[1][rg] << 8);
  }
#pragma unroll
  for (int rg = 0; rg < 3; rg++)
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const uint32_t p0 = (uint32_t)p[0][rg][m], p1 = (uint32_t)p[1][rg][m], p2 = (uint32_t)p[2][rg][m], p3 = (uint32_t)p[3][rg][m];
      const uint64_t q0 = (uint64_t)p0 + ((uint64_t)p1 << 16), q1 = (uint64_t)p2 + ((uint64_t)p3 << 16);  // < 2^42
      // value = q0 + q1 * 2^32 + c
      const uint64_t w = q1 << 32;
      uint32_t top = (uint32_t)(q1 >> 32);
      const uint64_t t = q0 + w;
      top += (t < w);
      const uint64_t t2 = t + c[4 * rg + m];
      top += (t2 < t);
      const uint64_t u = ((uint64_t)top << 32) - top;
      uint64_t r = t2 + u;
      if (r < u) r += gl::EPS;
      s[4 * rg + m] = r;
    }
}

// a light nonlinearity between MDS layers (keeps values generic, costs the same in both variants)
__device__ __forceinline__ void stir(uint64_t (&s)[W], int r) {
#pragma unroll
  for (int i = 0; i < W; i++) s[i] ^= (s[i] >> 17) + (uint64_t)(r * 12 + i);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_mds(uint64_t *out, uint64_t seed, int reps) {
  uint64_t s[W];
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = seed * (i + 1) + t * 0x9E3779B97F4A7C15ull + ((uint64_t)i << 60);  // any u64 (lazy)
  const Consts K = make_consts();
  for (int r = 0; r < reps; r++) {
    if (VARIANT == 1) mds_mfma(s, K, d_ZERO);
    else poseidon::mds_layer(s, -1);
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::canon(s[i]);
    stir(s, r);
  }
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < W; i++) acc ^= s[i] * (2 * i + 1);
  out[t] = acc;
}

// whole permutation with the MFMA MDS in every round (full and partial): the S-box stays on the VALU
__device__ __forceinline__ void permute_mfma(uint64_t (&s)[W], const Consts &K) {
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = poseidon::add_const_lazy(s[i], poseidon::rc(i));
#pragma unroll 1
  for (int r = 0; r < poseidon::HALF_FULL; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = poseidon::sbox_lazy(s[i]);
    mds_mfma(s, K, poseidon::d_RC + (r + 1) * W);
  }
#pragma unroll 1
  for (int r = poseidon::HALF_FULL; r < poseidon::HALF_FULL + poseidon::PARTIAL; r++) {
    s[0] = poseidon::sbox_lazy(s[0]);
    mds_mfma(s, K, poseidon::d_RC + (r + 1) * W);
  }
#pragma unroll 1
  for (int r = poseidon::HALF_FULL + poseidon::PARTIAL; r < poseidon::ROUNDS; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = poseidon::sbox_lazy(s[i]);
    mds_mfma(s, K, r + 1 < poseidon::ROUNDS ? poseidon::d_RC + (r + 1) * W : d_ZERO);
  }
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(s[i]);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_perm(uint64_t *out, uint64_t seed, int reps) {
  uint64_t s[W];
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(seed * (i + 1) + t * 0x9E3779B97F4A7C15ull + i);
  const Consts K = make_consts();
  for (int r = 0; r < reps; r++) {
    if (VARIANT == 1) permute_mfma(s, K);
    else poseidon::permute(s);
  }
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < W; i++) acc ^= s[i] + i;
  out[t] = acc;
}
}  // namespace mm

template <class Fn>
static float best_ms(Fn launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0, 0);
    launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RC), POSEIDON_RC, sizeof POSEIDON_RC);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RCD), POSEIDON_RCD, sizeof POSEIDON_RCD);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDK), POSEIDON_DOMD_K, sizeof POSEIDON_DOMD_K);
  hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDLAST), POSEIDON_DOMD_LAST, sizeof POSEIDON_DOMD_LAST);
  const int blocks = 256 * 16;
  const size_t n = (size_t)blocks * 256;
  uint64_t *a, *b;
  hipMalloc(&a, n * 8);
  hipMalloc(&b, n * 8);
  std::vector<uint64_t> ha(n), hb(n);
  size_t bad = 0;
  // 1. MDS only
  hipLaunchKernelGGL(mm::k_mds<0>, dim3(blocks), dim3(256), 0, 0, a, 12345ull, 5);
  hipLaunchKernelGGL(mm::k_mds<1>, dim3(blocks), dim3(256), 0, 0, b, 12345ull, 5);
  hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < n; i++) bad += ha[i] != hb[i];
  printf("MDS layer: mismatches %zu of %zu\n", bad, n);
  const int reps = 512;
  const float v0 = best_ms([&] { hipLaunchKernelGGL(mm::k_mds<0>, dim3(blocks), dim3(256), 0, 0, a, 777ull, reps); });
  const float v1 = best_ms([&] { hipLaunchKernelGGL(mm::k_mds<1>, dim3(blocks), dim3(256), 0, 0, b, 777ull, reps); });
  printf("MDS + canon + stir, shift-add VALU : %.3f ms -> %.2f G layers/s\n", v0, (double)n * reps / v0 / 1e6);
  printf("MDS + canon + stir, MFMA 4x4x4 i8  : %.3f ms -> %.2f G layers/s\n", v1, (double)n * reps / v1 / 1e6);
  // 2. whole permutation
  size_t bad2 = 0;
  hipLaunchKernelGGL(mm::k_perm<0>, dim3(blocks), dim3(256), 0, 0, a, 999ull, 3);
  hipLaunchKernelGGL(mm::k_perm<1>, dim3(blocks), dim3(256), 0, 0, b, 999ull, 3);
  hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < n; i++) bad2 += ha[i] != hb[i];
  printf("permutation: mismatches %zu of %zu\n", bad2, n);
  const int preps = 64;
  const float w0 = best_ms([&] { hipLaunchKernelGGL(mm::k_perm<0>, dim3(blocks), dim3(256), 0, 0, a, 777ull, preps); });
  const float w1 = best_ms([&] { hipLaunchKernelGGL(mm::k_perm<1>, dim3(blocks), dim3(256), 0, 0, b, 777ull, preps); });
  printf("permutation, plane-resident VALU MDS: %.3f ms -> %.3f G perm/s\n", w0, (double)n * preps / w0 / 1e6);
  printf("permutation, MFMA MDS in every round: %.3f ms -> %.3f G perm/s\n", w1, (double)n * preps / w1 / 1e6);
  return (bad || bad2) ? 1 : 0;
}
