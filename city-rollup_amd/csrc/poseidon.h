// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4 + 22 + 4 rounds) for gfx950.
//
// Replaces plonky2's `PoseidonHash` / `PoseidonPermutation` (un-vendored dependency of the
// reference; call sites: city_crypto/src/hash/traits/hasher.rs:77-159 and every Merkle tree /
// challenger use inside `CircuitData::prove`).
//
// One lane owns one 12-element state (24 VGPRs): no cross-lane traffic, pure 32-bit integer VALU, bound by instruction
// issue. What shapes the code (measured on MI355X, profiles/r01_ubench_valu.txt, r02_ubench_poseidon.txt):
//   * a 64-bit modular multiplication is 15 instructions (gl.h: `mul_wide`, `fold_top` — two carry-outs taken in `asm`);
//   * the MDS layer (circulant, entries <= 41, + diag 8) uses NO multiplies at all: each element
//     is split into three 22-bit limbs and the length-12 cyclic convolution is evaluated per limb
//     in wrap-around 32-bit arithmetic through the CRT split
//         x^12-1 = (x^6-1)(x^6+1),  x^6-1 = (x^3-1)(x^3+1)
//     whose transformed kernels are all +-powers of two ([16,16,32], [-1,-8,2], [2,-4,16,1,-1,-1])
//     -> ~90 shift-adds per limb instead of 288 quarter-rate mads per state;
//   * state is carried lazily (any u64 congruent to the value); the next round's constant is
//     folded into the 96-bit recombination, and only the final output is canonicalised;
//   * the 22 partial rounds never leave the transformed domain of that CRT split (see `permute_until`).
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
// partial-round constants pushed forward through the MDS (gen_tables.plane_constants): one scalar per partial round on
// element 0, one vector after the last; both minus the limb bias of the signed recombination (DOM_BIAS)
__constant__ uint64_t d_DK[PARTIAL];
__constant__ uint64_t d_DLAST[W];
GL_HD uint64_t dom_k(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_DK[i];
#else
  return POSEIDON_DOM_K[i];
#endif
}
GL_HD uint64_t dom_last(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_DLAST[i];
#else
  return POSEIDON_DOM_LAST[i];
#endif
}

// ---- lazy field helpers: inputs/outputs are arbitrary u64 congruent to the value ----------
using gl::fold_top;
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
// as T^-1' . K . T: T = two butterfly levels (`dom_enter`), K = three small component-wise products (`dom_mul`),
// T^-1' = the butterflies back (`dom_leave`). Index layout of a transformed plane: [0..2] = aa, [3..5] = ab, [6..11] = b.
GL_HD void dom_enter(const uint32_t (&s)[W], uint32_t (&u)[W]) {
  uint32_t a[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    a[i] = s[i] + s[i + 6];
    u[6 + i] = s[i] - s[i + 6];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    u[i] = a[i] + a[i + 3];
    u[3 + i] = a[i] - a[i + 3];
  }
}
GL_HD void dom_mul(const uint32_t (&u)[W], uint32_t (&o)[W]) {
  typedef uint32_t T;
  // cyclic-3 part: aa (*) [16,16,32]
  const T T16 = (u[0] + u[1] + u[2]) << 4;
  o[0] = T16 + (u[2] << 4);
  o[1] = T16 + (u[0] << 4);
  o[2] = T16 + (u[1] << 4);
  // negacyclic-3 part: ab (*) [-1,-8,2]
  o[3] = (u[5] << 3) - u[3] - (u[4] << 1);
  o[4] = (T)0 - (u[3] << 3) - u[4] - (u[5] << 1);
  o[5] = (u[3] << 1) - (u[4] << 3) - u[5];
  // negacyclic-6 part: b (*) N mod (x^6+1), N = [2,-4,16,1,-1,-1]:  v[k] = sum_{i+j=k} b[i]N[j] - sum_{i+j=k+6} b[i]N[j]
  const T *b = &u[6];
  T nb[6];
#pragma unroll
  for (int i = 0; i < 6; i++) nb[i] = (T)0 - b[i];
  o[6] = (b[0] << 1) + b[1] + b[2] + nb[3] + (nb[4] << 4) + (b[5] << 2);
  o[7] = (nb[0] << 2) + (b[1] << 1) + b[2] + b[3] + nb[4] + (nb[5] << 4);
  o[8] = (b[0] << 4) + (nb[1] << 2) + (b[2] << 1) + b[3] + b[4] + nb[5];
  o[9] = b[0] + (b[1] << 4) + (nb[2] << 2) + (b[3] << 1) + b[4] + b[5];
  o[10] = nb[0] + b[1] + (b[2] << 4) + (nb[3] << 2) + (b[4] << 1) + b[5];
  o[11] = nb[0] + nb[1] + b[2] + (b[3] << 4) + (nb[4] << 2) + (b[5] << 1);
}
GL_HD void dom_leave(const uint32_t (&o)[W], uint32_t (&y)[W]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const uint32_t p = o[i] + o[3 + i], q = o[i] - o[3 + i];
    y[i] = p + o[6 + i];
    y[i + 6] = p - o[6 + i];
    y[i + 3] = q + o[9 + i];
    y[i + 9] = q - o[9 + i];
  }
}
GL_HD void mds_limb(const uint32_t (&s)[W], uint32_t (&y)[W]) {
  uint32_t u[W], o[W];
  dom_enter(s, u);
  dom_mul(u, o);
  dom_leave(o, y);
  y[0] += s[0] << 3;
}

// y = y0 + 2^22 y1 + 2^44 y2 + c  (y* < 2^32, c < 2^64)  ->  lazy u64
GL_HD uint64_t recombine(uint32_t y0, uint32_t y1, uint32_t y2, uint64_t c) {
  const gl::u128 acc = (gl::u128)c + y0 + ((uint64_t)y1 << 22) + ((gl::u128)y2 << 44);  // < 2^77: plain carry chains
  return fold_top((uint64_t)acc, (uint32_t)(acc >> 64));
}
GL_HD void split3(uint64_t v, uint32_t &a, uint32_t &b, uint32_t &c) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  a = lo & 0x3FFFFFu;
  b = ((lo >> 22) | (hi << 10)) & 0x3FFFFFu;
  c = hi >> 12;
}

// s <- MDS * s + next_rc   (next_rc_base < 0: no constant)
GL_HD void mds_layer(uint64_t (&s)[W], int next_rc_base) {
  uint32_t l0[W], l1[W], l2[W];
#pragma unroll
  for (int i = 0; i < W; i++) split3(s[i], l0[i], l1[i], l2[i]);
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

// ---- partial rounds in the transformed domain of the circulant --------------------------------------------
// Only element 0 meets an S-box in a partial round; the other eleven go from one MDS straight into the next, and they
// need no round constants at all — those are pushed forward through the MDS offline (gen_tables.plane_constants: one
// scalar per round on element 0, one vector at the end). Between two MDS layers T . T^-1' is only a scaling: if
// o = (E, F, v) are the products of one layer, the next layer's transformed input is T(y) = (4E, 4F, 2v). So the state
// stays in the domain for all 22 rounds, as three limb planes of field elements mod p with SIGNED limbs:
//   * per round and plane the 36 butterfly additions are gone; the scaling is folded into the carry normalisation
//     (`renorm_scaled`), which also replaces recombine-to-u64 + split-again;
//   * element 0 of the state is read off the products as E0 + F0 + v0 + 8 * (previous S-box output) — exact integer
//     limbs — recombined, S-boxed, split; the difference to what the domain holds for it is added to aa0, ab0, b0 (T of the
//     unit vector) before those three components are normalised;
//   * a negative top limb folds like a positive one (2^64 == 2^32 - 1: + top 2^10 on plane 1, - top on plane 0), so the
//     normalisation needs no borrow handling.
// Magnitudes: normalised limbs lie in (-2^19, 2^22 + 2^19) (top limb [0, 2^20)); |products| <= 64 x that < 2^28.2, times 4
// < 2^30.2; the element-0 components before their normalisation < 2^30.2 + 2^28.3 + 2^25 < 2^31. Entering from the full
// rounds (limbs < 2^22, so aa < 2^24 would give products up to 2^30) the three aa components are normalised first.
// tests/test_hostsim.py checks the bounds with interval arithmetic and the permutation against the oracle.
constexpr uint32_t DOM_BIAS = 1u << 30;  // makes the signed element-0 limbs (|.| < 2^29) non-negative for `recombine`
// signed carry normalisation in place: |y*| <= 2^31 - 2^10, same value mod p
GL_HD void renorm_s(uint32_t &y0, uint32_t &y1, uint32_t &y2) {
  const int32_t t1 = (int32_t)y1 + ((int32_t)y0 >> 22);
  const int32_t t2 = (int32_t)y2 + (t1 >> 22);
  const int32_t top = t2 >> 20;
  y0 = (y0 & 0x3FFFFFu) - (uint32_t)top;
  y1 = ((uint32_t)t1 & 0x3FFFFFu) + ((uint32_t)top << 10);
  y2 = (uint32_t)t2 & 0xFFFFFu;
}
// the same for 2^S * (o0, o1, o2), 2^S |o*| <= 2^31 - 2^10, without forming the scaled limbs
template <int S>
GL_HD void renorm_scaled(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t &l0, uint32_t &l1, uint32_t &l2) {
  const int32_t t1 = (int32_t)(o1 << S) + ((int32_t)o0 >> (22 - S));
  const int32_t t2 = (int32_t)(o2 << S) + (t1 >> 22);
  const int32_t top = t2 >> 20;
  l0 = ((o0 << S) & 0x3FFFFFu) - (uint32_t)top;
  l1 = ((uint32_t)t1 & 0x3FFFFFu) + ((uint32_t)top << 10);
  l2 = (uint32_t)t2 & 0xFFFFFu;
}

// `stop` is polled between rounds (every full round, every fourth partial round); when it answers true the permutation is
// abandoned and false returned. It must answer the same for every lane of a wave. The proof-of-work search uses it to drop
// candidates that can no longer be the smallest witness.
template <typename Stop>
GL_HD bool permute_until(uint64_t (&s)[W], Stop stop) {
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = add_const_lazy(s[i], rc(i));
#pragma unroll 1
  for (int r = 0; r < HALF_FULL - 1; r++) {
    if (stop()) return false;
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = sbox_lazy(s[i]);
    mds_layer(s, (r + 1) * W);
  }
  {
    // the last full round before the partial rounds: S-boxes, then straight into the domain (no constants: all pushed forward)
    uint32_t u0[W], u1[W], u2[W], o0[W], o1[W], o2[W];
    uint32_t n0, n1, n2;  // limbs of the latest S-box output on element 0
    {
      uint32_t l0[W], l1[W], l2[W];
#pragma unroll
      for (int k = 0; k < W; k++) {
        s[k] = sbox_lazy(s[k]);
        split3(s[k], l0[k], l1[k], l2[k]);
      }
      n0 = l0[0], n1 = l1[0], n2 = l2[0];
      dom_enter(l0, u0);
      dom_enter(l1, u1);
      dom_enter(l2, u2);
#pragma unroll
      for (int k = 0; k < 3; k++) renorm_s(u0[k], u1[k], u2[k]);
      dom_mul(u0, o0);
      dom_mul(u1, o1);
      dom_mul(u2, o2);
    }
#pragma unroll 1
    for (int i = 0; i < PARTIAL; i++) {
      if ((i & 3) == 0 && stop()) return false;
      // element 0 of the state: E0 + F0 + v0 + the diagonal 8 of the MDS on the previous S-box output
      const uint32_t z0 = o0[0] + o0[3] + o0[6], z1 = o1[0] + o1[3] + o1[6], z2 = o2[0] + o2[3] + o2[6];
      const uint64_t x = recombine(z0 + (n0 << 3) + DOM_BIAS, z1 + (n1 << 3) + DOM_BIAS, z2 + (n2 << 3) + DOM_BIAS, dom_k(i));
      split3(sbox_lazy(x), n0, n1, n2);
      // the nine components that do not see element 0 directly: scale and normalise
#pragma unroll
      for (int k = 1; k < 3; k++) {
        renorm_scaled<2>(o0[k], o1[k], o2[k], u0[k], u1[k], u2[k]);
        renorm_scaled<2>(o0[3 + k], o1[3 + k], o2[3 + k], u0[3 + k], u1[3 + k], u2[3 + k]);
      }
#pragma unroll
      for (int k = 7; k < W; k++) renorm_scaled<1>(o0[k], o1[k], o2[k], u0[k], u1[k], u2[k]);
      // aa0, ab0, b0: scaled products + diagonal + (new element 0 - old element 0) = scaled products + new - z
      {
        const uint32_t d0 = n0 - z0, d1 = n1 - z1, d2 = n2 - z2;
        u0[0] = (o0[0] << 2) + d0, u0[3] = (o0[3] << 2) + d0, u0[6] = (o0[6] << 1) + d0;
        u1[0] = (o1[0] << 2) + d1, u1[3] = (o1[3] << 2) + d1, u1[6] = (o1[6] << 1) + d1;
        u2[0] = (o2[0] << 2) + d2, u2[3] = (o2[3] << 2) + d2, u2[6] = (o2[6] << 1) + d2;
        renorm_s(u0[0], u1[0], u2[0]);
        renorm_s(u0[3], u1[3], u2[3]);
        renorm_s(u0[6], u1[6], u2[6]);
      }
      dom_mul(u0, o0);
      dom_mul(u1, o1);
      dom_mul(u2, o2);
    }
    // leave the domain: natural limbs (signed, |.| < 2^29), + what is pending of the constants
    uint32_t y0[W], y1[W], y2[W];
    dom_leave(o0, y0);
    dom_leave(o1, y1);
    dom_leave(o2, y2);
    y0[0] += n0 << 3, y1[0] += n1 << 3, y2[0] += n2 << 3;
#pragma unroll
    for (int k = 0; k < W; k++) s[k] = recombine(y0[k] + DOM_BIAS, y1[k] + DOM_BIAS, y2[k] + DOM_BIAS, dom_last(k));
  }
#pragma unroll 1
  for (int r = HALF_FULL + PARTIAL; r < ROUNDS; r++) {
    if (stop()) return false;
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = sbox_lazy(s[i]);
    mds_layer(s, r + 1 < ROUNDS ? (r + 1) * W : -1);
  }
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(s[i]);
  return true;
}

struct NeverStop {
  GL_HD bool operator()() const { return false; }
};
GL_HD void permute(uint64_t (&s)[W]) { permute_until(s, NeverStop()); }

}  // namespace poseidon
