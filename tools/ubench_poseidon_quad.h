// EXPERIMENT (not part of the library): four lanes per Poseidon state for the leaf hashes of ONE proof (2^15 leaves: 512 waves
// on 1 024 SIMDs). Measured by tools/ubench_leaf_latency.hip, profiles/r03_ubench_leaf_latency.txt: bit-exact, 0.56 ms against
// 0.65 ms for 2^15 x 135 (-14 %), 2.2 x slower at 2^20 — the launch stops being bound by one wave's latency and becomes bound by
// issue slots at twice the work. Not worth a second leaf-hash path: 0.09 ms of a 3.4 ms proof.
//
// With one lane per leaf (merkle::k_leaf_hash_cols) such a launch is not bound by issue slots but by the latency of one wave's
// instruction stream: a lone wave executes the 12.2 K instructions of a permutation at 7.4 cycles each (4.0 when five waves share
// the SIMD), 38 us per permutation, and a leaf of 135 columns is a chain of seventeen. Twelve lanes per state (poseidon_coop.h)
// cut the chain to 13 us per permutation at fourteen times the instructions: right for a tree level of a few thousand nodes, not
// for 557 K leaf permutations. FOUR lanes per state, three elements each (lane q of a quad holds elements q, q + 4, q + 8), are
// the middle: the twelve S-boxes of a full round are three per lane, the state is exchanged inside the quad by DPP quad_perm
// broadcasts (24 v_mov_dpp per round, no LDS), every lane computes three MDS rows as integer multiply-adds over 32-bit halves —
// ~6.5 K instructions per lane and permutation, 2.2 x the total work of the lane-per-leaf form, on four times the lanes.
// Textbook round structure (constants, S-box, MDS), bit-exact with poseidon::permute.
#pragma once
#include "poseidon.h"

namespace pquad {

// quad_perm broadcast of lane K's value to the four lanes of its quad
template <int K>
__device__ __forceinline__ uint32_t bcast32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, false);
}
template <int K>
__device__ __forceinline__ uint64_t bcast64(uint64_t v) {
  return gl::pack(bcast32<K>(gl::lo32(v)), bcast32<K>(gl::hi32(v)));
}

// D[t] = C[(t - q) mod 12]: row e = q + 4 m of the circulant MDS has coefficient D[(j - 4 m) mod 12] on element j
__device__ __forceinline__ uint32_t circ(int t, int q) {
  constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};  // == POSEIDON_MDS_CIRC (tests/test_tables.py)
  int d = t - q;
  if (d < 0) d += 12;
  return C[d];
}

// One permutation per quad. x[m]: element q + 4 m of the state (any u64 representatives on entry, lazy on exit: the caller
// canonicalises what it stores). All 64 lanes of the wave call this together.
__device__ __forceinline__ void permute(uint64_t (&x)[3], int q, const uint32_t (&D)[12], uint32_t diag) {
  uint64_t rcn[3];
#pragma unroll
  for (int m = 0; m < 3; m++) rcn[m] = poseidon::d_RC[q + 4 * m];
#pragma unroll 1
  for (int r = 0; r < poseidon::ROUNDS; r++) {
    // the constants of this lane's elements are per-lane loads: fetched one round ahead
    uint64_t rc[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
      rc[m] = rcn[m];
      rcn[m] = poseidon::d_RC[(r + 1 < poseidon::ROUNDS ? r + 1 : r) * poseidon::W + q + 4 * m];
      x[m] = poseidon::add_const_lazy(x[m], rc[m]);
    }
    const bool full = r < poseidon::HALF_FULL || r >= poseidon::HALF_FULL + poseidon::PARTIAL;
    if (full) {
#pragma unroll
      for (int m = 0; m < 3; m++) x[m] = poseidon::sbox_lazy(x[m]);
    } else if (q == 0) {
      x[0] = poseidon::sbox_lazy(x[0]);
    }
    // the whole state in every lane: element 4 m + l sits in lane l, register m
    uint64_t all[12];
#pragma unroll
    for (int m = 0; m < 3; m++) {
      all[4 * m + 0] = bcast64<0>(x[m]);
      all[4 * m + 1] = bcast64<1>(x[m]);
      all[4 * m + 2] = bcast64<2>(x[m]);
      all[4 * m + 3] = bcast64<3>(x[m]);
    }
    // three rows of the MDS in two 64-bit accumulators each (low / high words of the lazy elements): 12 x 49 x 2^32 < 2^42
#pragma unroll
    for (int m = 0; m < 3; m++) {
      uint64_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < 12; j++) {
        const uint32_t c = D[(j - 4 * m + 12) % 12];
        lo += (uint64_t)gl::lo32(all[j]) * c;
        hi += (uint64_t)gl::hi32(all[j]) * c;
      }
      if (m == 0) {  // the 8 on the diagonal of row 0: diag = 8 in lane 0, else 0
        lo += (uint64_t)gl::lo32(all[0]) * diag;
        hi += (uint64_t)gl::hi32(all[0]) * diag;
      }
      const uint64_t l = lo + (hi << 32);
      const uint64_t h = (hi >> 32) + (l < lo);
      x[m] = gl::reduce128_lazy(l, h);
    }
  }
}

// leaf digests as merkle::k_leaf_hash_cols<false> computes them (leaf_len > 4: always hashed), four lanes per leaf.
// grid = (ceil(4 n_leaves / 256), n_trees), block = 256
__global__ __launch_bounds__(256) void k_leaf_hash_cols_quad(const uint64_t *__restrict__ cols, size_t n_leaves, int leaf_len, size_t col_stride,
                                                             uint64_t *__restrict__ digests, size_t tree_cols_stride, size_t tree_dig_stride) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int q = (int)(t & 3);
  size_t i = t >> 2;
  const bool live = i < n_leaves;  // the last wave may carry quads without a leaf: they hash leaf 0 and store nothing
  if (!live) i = 0;
  cols += (size_t)blockIdx.y * tree_cols_stride;
  digests += (size_t)blockIdx.y * tree_dig_stride;
  uint32_t D[12];
#pragma unroll
  for (int k = 0; k < 12; k++) D[k] = circ(k, q);
  const uint32_t diag = q == 0 ? 8u : 0u;
  uint64_t x[3] = {0, 0, 0};
  for (int j = 0; j < leaf_len; j += poseidon::RATE) {  // overwrite-mode sponge: a partial last chunk overwrites only its own elements
    if (j + q < leaf_len) x[0] = cols[(size_t)(j + q) * col_stride + i];
    if (j + q + 4 < leaf_len) x[1] = cols[(size_t)(j + q + 4) * col_stride + i];
    permute(x, q, D, diag);
  }
  if (live) digests[4 * i + q] = gl::canon(x[0]);
}

}  // namespace pquad
