// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4 + 22 + 4 rounds) for gfx950.
//
// Replaces plonky2's `PoseidonHash` / `PoseidonPermutation` (un-vendored dependency of the
// reference; call sites: city_crypto/src/hash/traits/hasher.rs:77-159 and every Merkle tree /
// challenger use inside `CircuitData::prove`).
//
// One lane owns one 12-element state (24 VGPRs): no cross-lane traffic, the work is pure
// 32-bit integer VALU. The MDS layer has entries <= 41, so it is evaluated on the 32-bit halves
// of each element with 64-bit accumulators and a single 96-bit reduction per output instead of
// 144 modular multiplications.
#pragma once
#include "gl.h"
#include "poseidon_tables.h"

namespace poseidon {

constexpr int W = 12;
constexpr int RATE = 8;
constexpr int HALF_FULL = 4;
constexpr int PARTIAL = 22;
constexpr int ROUNDS = 2 * HALF_FULL + PARTIAL;

// device copies of the tables (uniform indices -> scalar loads)
__constant__ uint64_t d_RC[ROUNDS * W];
__constant__ uint64_t d_FAST_FIRST[W];
__constant__ uint64_t d_FAST_K[PARTIAL];
__constant__ uint64_t d_FAST_VS[PARTIAL * 11];
__constant__ uint64_t d_FAST_WHATS[PARTIAL * 11];
__constant__ uint64_t d_FAST_INIT[11 * 11];

// value = lo + 2^32 * hi_acc  (lo_acc, hi_acc < 2^44)  -> canonical
__device__ __forceinline__ uint64_t reduce_split(uint64_t lo_acc, uint64_t hi_acc) {
  uint64_t lo = lo_acc + (hi_acc << 32);
  uint64_t hi = (hi_acc >> 32) + (lo < lo_acc ? 1 : 0);  // < 2^13
  // hi * 2^64 == hi * EPS
  uint64_t t = (hi << 32) - hi;
  uint64_t r = lo + t;
  if (r < t) r += gl::EPS;
  return gl::canon(r);
}

__device__ __forceinline__ void mds_layer(uint64_t (&s)[W]) {
  uint32_t lo[W], hi[W];
#pragma unroll
  for (int i = 0; i < W; i++) {
    lo[i] = (uint32_t)s[i];
    hi[i] = (uint32_t)(s[i] >> 32);
  }
  constexpr uint32_t C[W] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll
  for (int r = 0; r < W; r++) {
    uint64_t al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
      al += (uint64_t)lo[(i + r) % W] * C[i];
      ah += (uint64_t)hi[(i + r) % W] * C[i];
    }
    if (r == 0) {
      al += (uint64_t)lo[0] << 3;
      ah += (uint64_t)hi[0] << 3;
    }
    s[r] = reduce_split(al, ah);
  }
}

template <bool FULL>
__device__ __forceinline__ void round_naive(uint64_t (&s)[W], int rnd) {
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::add(s[i], d_RC[rnd * W + i]);
  if (FULL) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::pow7(s[i]);
  } else {
    s[0] = gl::pow7(s[0]);
  }
  mds_layer(s);
}

// Textbook form: 30 x (constants, S-box, dense small-coefficient MDS).
__device__ __forceinline__ void permute_naive(uint64_t (&s)[W]) {
#pragma unroll 1
  for (int r = 0; r < HALF_FULL; r++) round_naive<true>(s, r);
#pragma unroll 1
  for (int r = HALF_FULL; r < HALF_FULL + PARTIAL; r++) round_naive<false>(s, r);
#pragma unroll 1
  for (int r = HALF_FULL + PARTIAL; r < ROUNDS; r++) round_naive<true>(s, r);
}

// Sparse-factorised partial rounds (tables from gen_tables.py::fast_partial).
__device__ __forceinline__ void permute_fast(uint64_t (&s)[W]) {
#pragma unroll 1
  for (int r = 0; r < HALF_FULL; r++) round_naive<true>(s, r);
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::add(s[i], d_FAST_FIRST[i]);
  {
    uint64_t t[11];
#pragma unroll 1
    for (int rr = 0; rr < 11; rr++) {
      // 11-term dot product with a 128+ bit accumulator, one reduction
      uint64_t alo = 0, ahi = 0, acar = 0;
#pragma unroll
      for (int cc = 0; cc < 11; cc++) {
        uint64_t lo, hi;
        gl::mul_wide(d_FAST_INIT[rr * 11 + cc], s[1 + cc], lo, hi);
        alo += lo;
        uint64_t c = alo < lo;
        ahi += hi;
        acar += (ahi < hi);
        ahi += c;
        acar += (ahi < c);
      }
      // value = alo + 2^64 ahi + 2^128 acar ; 2^128 == 2^32 * 2^96 == -2^32
      uint64_t r = gl::reduce128(alo, ahi);
      r = gl::sub(r, gl::canon(acar << 32));
      t[rr] = r;
    }
#pragma unroll
    for (int i = 0; i < 11; i++) s[1 + i] = t[i];
  }
#pragma unroll 1
  for (int i = 0; i < PARTIAL; i++) {
    uint64_t s0 = gl::add(gl::pow7(s[0]), d_FAST_K[i]);
    uint64_t alo, ahi, acar = 0;
    gl::mul_wide(s0, 25, alo, ahi);  // m00 = 17 + 8
#pragma unroll
    for (int j = 0; j < 11; j++) {
      uint64_t lo, hi;
      gl::mul_wide(d_FAST_WHATS[i * 11 + j], s[1 + j], lo, hi);
      alo += lo;
      uint64_t c = alo < lo;
      ahi += hi;
      acar += (ahi < hi);
      ahi += c;
      acar += (ahi < c);
    }
    uint64_t d = gl::sub(gl::reduce128(alo, ahi), gl::canon(acar << 32));
#pragma unroll
    for (int j = 0; j < 11; j++) s[1 + j] = gl::add(s[1 + j], gl::mul(s0, d_FAST_VS[i * 11 + j]));
    s[0] = d;
  }
#pragma unroll 1
  for (int r = HALF_FULL + PARTIAL; r < ROUNDS; r++) round_naive<true>(s, r);
}

#ifndef POSEIDON_VARIANT
#define POSEIDON_VARIANT 0
#endif

__device__ __forceinline__ void permute(uint64_t (&s)[W]) {
#if POSEIDON_VARIANT == 1
  permute_fast(s);
#else
  permute_naive(s);
#endif
}

}  // namespace poseidon
