// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4 + 22 + 4 rounds) for gfx950.
//
// Replaces plonky2's `PoseidonHash` / `PoseidonPermutation` (un-vendored dependency of the
// reference; call sites: city_crypto/src/hash/traits/hasher.rs:77-159 and every Merkle tree /
// challenger use inside `CircuitData::prove`).
//
// One lane owns one 12-element state (24 VGPRs): no cross-lane traffic, pure 32-bit integer VALU.
// What shapes the code (measured on MI355X, profiles/r01_*): `v_mad_u64_u32` is quarter rate
// (8 cycles per wave64) and everything else is full rate (2 cycles), so
//   * the S-box (x^7 = 4 modular multiplications, 14 mads) keeps 64-bit multiplies, but
//   * the MDS layer (circulant, entries <= 41, + diag 8) uses NO multiplies at all: each element
//     is split into three 22-bit limbs and the length-12 cyclic convolution is evaluated per limb
//     in wrap-around 32-bit arithmetic through the CRT split
//         x^12-1 = (x^6-1)(x^6+1),  x^6-1 = (x^3-1)(x^3+1)
//     whose transformed kernels are all +-powers of two ([16,16,32], [-1,-8,2], [2,-4,16,1,-1,-1])
//     -> ~96 shift-adds per limb instead of 288 quarter-rate mads per state.
//   * state is carried lazily (any u64 congruent to the value); the next round's constant is
//     folded into the 96-bit recombination, and only the final output is canonicalised.
#pragma once
#include "gl.h"
#include "poseidon_tables.h"

namespace poseidon {

constexpr int W = 12;
constexpr int RATE = 8;
constexpr int HALF_FULL = 4;
constexpr int PARTIAL = 22;
constexpr int ROUNDS = 2 * HALF_FULL + PARTIAL;

// device copy of the round constants (uniform index -> scalar loads)
__constant__ uint64_t d_RC[ROUNDS * W];

GL_HD uint64_t rc(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_RC[i];
#else
  return POSEIDON_RC[i];
#endif
}
// partial-round constants pushed forward through the MDS (see `permute`)
__constant__ uint64_t d_PK[22];
__constant__ uint64_t d_PLAST[12];
GL_HD uint64_t plane_k(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_PK[i];
#else
  return POSEIDON_PLANE_K[i];
#endif
}
GL_HD uint64_t plane_last(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_PLAST[i];
#else
  return POSEIDON_PLANE_LAST[i];
#endif
}

// ---- lazy field helpers: inputs/outputs are arbitrary u64 congruent to the value ----------
GL_HD uint64_t mul_lazy(uint64_t a, uint64_t b) {
  uint64_t lo, hi;
  gl::mul_wide(a, b, lo, hi);
  return gl::reduce128_lazy(lo, hi);
}
GL_HD uint64_t sbox_lazy(uint64_t x) {
  const uint64_t x2 = mul_lazy(x, x), x4 = mul_lazy(x2, x2), x3 = mul_lazy(x, x2);
  return mul_lazy(x3, x4);
}
GL_HD uint64_t add_const_lazy(uint64_t a, uint64_t c) {  // c canonical
  uint64_t s = a + c;
  if (s < a) {  // wrapped: 2^64 == EPS
    s += gl::EPS;
    if (s < gl::EPS) s += gl::EPS;  // (cannot happen for c < p, kept for safety)
  }
  return s;
}

// ---- MDS layer on one 22-bit limb plane, arithmetic mod 2^32 (exact: true results < 2^31) ----
// y[r] = sum_i C[i] * s[(i+r) % 12] + 8*s[0]*[r==0],  C = {17,15,41,16,2,28,13,13,39,18,34,20}
template <typename T>
GL_HD void mds_limb(const T (&s)[W], T (&y)[W]) {
  T a[6], b[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    a[i] = s[i] + s[i + 6];
    b[i] = s[i] - s[i + 6];
  }
  // cyclic-6 part: a (*) [15,24,18,17,40,14]  via  (x^3-1)(x^3+1)
  T aa0 = a[0] + a[3], aa1 = a[1] + a[4], aa2 = a[2] + a[5];
  T ab0 = a[0] - a[3], ab1 = a[1] - a[4], ab2 = a[2] - a[5];
  T T16 = (aa0 + aa1 + aa2) << 4;
  T E0 = T16 + (aa2 << 4), E1 = T16 + (aa0 << 4), E2 = T16 + (aa1 << 4);
  T F0 = (ab2 << 3) - ab0 - (ab1 << 1);
  T F1 = (T)0 - (ab0 << 3) - ab1 - (ab2 << 1);
  T F2 = (ab0 << 1) - (ab1 << 3) - ab2;
  T pc[6] = {E0 + F0, E1 + F1, E2 + F2, E0 - F0, E1 - F1, E2 - F2};
  // negacyclic-6 part: b (*) [2,-4,16,1,-1,-1] mod (x^6+1)
  // V[k] = sum_{i+j=k} b[i]N[j] - sum_{i+j=k+6} b[i]N[j]
  T nb[6];
#pragma unroll
  for (int i = 0; i < 6; i++) nb[i] = (T)0 - b[i];
  T v[6];
  // N0=2 (<<1), N1=-4 (<<2, neg), N2=16 (<<4), N3=1, N4=-1, N5=-1
  // k=0: b0N0 - (b1N5 + b2N4 + b3N3 + b4N2 + b5N1)
  v[0] = (b[0] << 1) + b[1] + b[2] + nb[3] + (nb[4] << 4) + (b[5] << 2);
  // k=1: b0N1 + b1N0 - (b2N5 + b3N4 + b4N3 + b5N2)
  v[1] = (nb[0] << 2) + (b[1] << 1) + b[2] + b[3] + nb[4] + (nb[5] << 4);
  // k=2: b0N2 + b1N1 + b2N0 - (b3N5 + b4N4 + b5N3)
  v[2] = (b[0] << 4) + (nb[1] << 2) + (b[2] << 1) + b[3] + b[4] + nb[5];
  // k=3: b0N3 + b1N2 + b2N1 + b3N0 - (b4N5 + b5N4)
  v[3] = b[0] + (b[1] << 4) + (nb[2] << 2) + (b[3] << 1) + b[4] + b[5];
  // k=4: b0N4 + b1N3 + b2N2 + b3N1 + b4N0 - (b5N5)
  v[4] = nb[0] + b[1] + (b[2] << 4) + (nb[3] << 2) + (b[4] << 1) + b[5];
  // k=5: b0N5 + b1N4 + b2N3 + b3N2 + b4N1 + b5N0
  v[5] = nb[0] + nb[1] + b[2] + (b[3] << 4) + (nb[4] << 2) + (b[5] << 1);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    y[i] = pc[i] + v[i];
    y[i + 6] = pc[i] - v[i];
  }
  y[0] += s[0] << 3;
}

// y = y0 + 2^22 y1 + 2^44 y2 + c  (y* < 2^31, c < 2^64)  ->  lazy u64
GL_HD uint64_t recombine(uint32_t y0, uint32_t y1, uint32_t y2, uint64_t c) {
  uint64_t v = (uint64_t)y0 + ((uint64_t)y1 << 22);  // < 2^54
  uint64_t w_lo = (uint64_t)y2 << 44;
  uint32_t top = y2 >> 20;                            // weight 2^64
  uint64_t t = v + w_lo;
  top += (t < w_lo);
  uint64_t t2 = t + c;
  top += (t2 < c);
  uint64_t u = ((uint64_t)top << 32) - top;           // top * (2^32 - 1)
  uint64_t r = t2 + u;
  if (r < u) r += gl::EPS;
  return r;
}

// s <- MDS * s + next_rc   (next_rc_base < 0: no constant)
GL_HD void mds_layer(uint64_t (&s)[W], int next_rc_base) {
  uint32_t l0[W], l1[W], l2[W];
#pragma unroll
  for (int i = 0; i < W; i++) {
    uint32_t lo = (uint32_t)s[i], hi = (uint32_t)(s[i] >> 32);
    l0[i] = lo & 0x3FFFFFu;
    l1[i] = ((lo >> 22) | (hi << 10)) & 0x3FFFFFu;
    l2[i] = hi >> 12;
  }
  uint32_t y0[W], y1[W], y2[W];
  mds_limb(l0, y0);
  mds_limb(l1, y1);
  mds_limb(l2, y2);
#pragma unroll
  for (int i = 0; i < W; i++)
    s[i] = recombine(y0[i], y1[i], y2[i], next_rc_base >= 0 ? rc(next_rc_base + i) : 0);
}

// Textbook round structure: 30 x (constants, S-box, MDS); state lazy, canonical on exit. Kept as the plain
// statement of the permutation (tests compare `permute` with it); the kernels use `permute` below.
GL_HD void permute_textbook(uint64_t (&s)[W]) {
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = add_const_lazy(s[i], rc(i));
#pragma unroll 1
  for (int r = 0; r < HALF_FULL; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = sbox_lazy(s[i]);
    mds_layer(s, (r + 1) * W);
  }
#pragma unroll 1
  for (int r = HALF_FULL; r < HALF_FULL + PARTIAL; r++) {
    s[0] = sbox_lazy(s[0]);
    mds_layer(s, (r + 1) * W);
  }
#pragma unroll 1
  for (int r = HALF_FULL + PARTIAL; r < ROUNDS; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = sbox_lazy(s[i]);
    mds_layer(s, r + 1 < ROUNDS ? (r + 1) * W : -1);
  }
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(s[i]);
}

// ---- partial rounds with the state resident in limb planes ------------------------------------------------
// Only element 0 meets an S-box in a partial round; the other eleven go from one MDS straight into the next.
// They therefore stay in their three 22-bit limb planes for all 22 rounds: after each MDS a carry
// normalisation (13 integer ops per element) replaces recombine-to-u64 + split-again (24), and they need no
// round constants at all — those are pushed forward through the MDS offline (gen_tables.plane_constants:
// a scalar K[i] on element 0 per round, one vector LAST at the end).
GL_HD void split3(uint64_t v, uint32_t &a, uint32_t &b, uint32_t &c) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  a = lo & 0x3FFFFFu;
  b = ((lo >> 22) | (hi << 10)) & 0x3FFFFFu;
  c = hi >> 12;
}
// MDS outputs of one element (y0, y1 < 2^31.1, y2 < 2^28.1) -> limbs l0, l1 < 2^23, l2 < 2^20 of a congruent value.
// The part above 2^64 (top) is folded with 2^64 == 2^32 - 1: +top*2^10 on plane 1, -top on plane 0, the latter
// paid for by borrowing one unit of plane 1 (only when top != 0, so nothing ever goes negative).
GL_HD void renorm(uint32_t y0, uint32_t y1, uint32_t y2, uint32_t &l0, uint32_t &l1, uint32_t &l2) {
  const uint32_t t1 = y1 + (y0 >> 22);
  const uint32_t t2 = y2 + (t1 >> 22);
  const uint32_t top = t2 >> 20;
  const uint32_t adj = top < 1u ? top : 1u;
  l0 = (y0 & 0x3FFFFFu) + (adj << 22) - top;
  l1 = (t1 & 0x3FFFFFu) + (top << 10) - adj;
  l2 = t2 & 0xFFFFFu;
}

GL_HD void permute(uint64_t (&s)[W]) {
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = add_const_lazy(s[i], rc(i));
#pragma unroll 1
  for (int r = 0; r < HALF_FULL; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = sbox_lazy(s[i]);
    mds_layer(s, (r + 1) * W);  // r = 3: adds the whole constant vector of the first partial round
  }
  {
    uint32_t l0[W], l1[W], l2[W], y0[W], y1[W], y2[W];
#pragma unroll
    for (int k = 1; k < W; k++) split3(s[k], l0[k], l1[k], l2[k]);
    uint64_t x0 = s[0];
#pragma unroll 1
    for (int i = 0; i < PARTIAL; i++) {
      x0 = sbox_lazy(x0);
      split3(x0, l0[0], l1[0], l2[0]);
      mds_limb(l0, y0);
      mds_limb(l1, y1);
      mds_limb(l2, y2);
#pragma unroll
      for (int k = 0; k < W; k++) renorm(y0[k], y1[k], y2[k], l0[k], l1[k], l2[k]);
      if (i + 1 < PARTIAL) x0 = recombine(l0[0], l1[0], l2[0], plane_k(i + 1));
    }
#pragma unroll
    for (int k = 0; k < W; k++) s[k] = recombine(l0[k], l1[k], l2[k], plane_last(k));
  }
#pragma unroll 1
  for (int r = HALF_FULL + PARTIAL; r < ROUNDS; r++) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = sbox_lazy(s[i]);
    mds_layer(s, r + 1 < ROUNDS ? (r + 1) * W : -1);
  }
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(s[i]);
}

}  // namespace poseidon
