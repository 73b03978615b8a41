// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4 + 22 + 4 rounds) for gfx950.
//
// Replaces plonky2's `PoseidonHash` / `PoseidonPermutation` (un-vendored dependency of the
// reference; call sites: city_crypto/src/hash/traits/hasher.rs:77-159 and every Merkle tree /
// challenger use inside `CircuitData::prove`).
//
// One lane owns one 12-element state (24 VGPRs): no cross-lane traffic, integer and fp64 VALU only, bound by instruction
// issue. What shapes the code (measured on MI355X, profiles/r01_ubench_valu.txt, r02_ubench_poseidon.txt):
//   * a 64-bit modular multiplication is 12 instructions (gl.h: `mul_wide`, `reduce128_lazy`, `fold_top` — three pieces in `asm`);
//   * the MDS layer (circulant, entries <= 41, + diag 8) uses NO integer multiplies at all: the length-12 cyclic
//     convolution is evaluated per limb plane through the CRT split
//         x^12-1 = (x^6-1)(x^6+1),  x^6-1 = (x^3-1)(x^3+1)
//     whose transformed kernels are all +-powers of two ([16,16,32], [-1,-8,2], [2,-4,16,1,-1,-1]): ~90 operations per plane
//     instead of 288 multiply-adds per state — on two 32-bit limb planes in DOUBLE PRECISION (exact: everything stays below
//     2^53; `mds_layer_d`), which beat the three 22-bit integer planes of `mds_limb` by 15 % on the whole permutation;
//   * state is carried lazily (any u64 congruent to the value); the next round's constant is
//     folded into the recombination, and only the final output is canonicalised;
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
// Constants of the double-precision layers (`recombine_d`): each constant c, minus the bit pattern of 1.5 * 2^52 at both limb
// positions, as the two doubles 1.5 * 2^52 + lo32 and 1.5 * 2^52 + hi32 — added to a limb they convert it to an integer AND
// add the constant in one operation. RCD = the round constants (+ one entry for "no constant"); DDK / DDLAST = the partial-round constants pushed
// forward through the MDS (gen_tables.plane_constants: one scalar per partial round on element 0, one vector after the last).
struct Magic { double m0, m1; };
__constant__ uint64_t d_RCD[2 * (ROUNDS * W + 1)];
__constant__ uint64_t d_DDK[2 * PARTIAL];
__constant__ uint64_t d_DDLAST[2 * W];
GL_HD Magic rcd(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return {__builtin_bit_cast(double, d_RCD[2 * i]), __builtin_bit_cast(double, d_RCD[2 * i + 1])};
#else
  return {__builtin_bit_cast(double, POSEIDON_RCD[2 * i]), __builtin_bit_cast(double, POSEIDON_RCD[2 * i + 1])};
#endif
}
GL_HD Magic domd_k(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return {__builtin_bit_cast(double, d_DDK[2 * i]), __builtin_bit_cast(double, d_DDK[2 * i + 1])};
#else
  return {__builtin_bit_cast(double, POSEIDON_DOMD_K[2 * i]), __builtin_bit_cast(double, POSEIDON_DOMD_K[2 * i + 1])};
#endif
}
GL_HD Magic domd_last(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return {__builtin_bit_cast(double, d_DDLAST[2 * i]), __builtin_bit_cast(double, d_DDLAST[2 * i + 1])};
#else
  return {__builtin_bit_cast(double, POSEIDON_DOMD_LAST[2 * i]), __builtin_bit_cast(double, POSEIDON_DOMD_LAST[2 * i + 1])};
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

// ---- the linear layers on TWO 32-bit limb planes in double precision -------------------------------------------------
// v_fma_f64 issues at the rate of the integer VOP3 forms (4.3 cycles per wave64, profiles/r01_ubench_valu.txt) and a double
// holds 53 bits exactly: a 32-bit limb has room for TWO layers of growth (2 x 8 bits) plus two fractional bits, so the MDS
// needs two planes instead of the three 22-bit integer planes of `mds_limb`, the products by +-2^k are single fused
// multiply-adds with the sign in the constant, and a carry normalisation is needed every second layer only. All arithmetic
// is exact (integers and quarter-integers below 2^53; multiplications by powers of two): no rounding ever happens, the
// result is bit-identical to the integer formulation (`permute_textbook` keeps that one; tests compare the two).
//
// Partial rounds in the transformed domain of the circulant. Only element 0 meets an S-box in a partial round; the other
// eleven go from one MDS straight into the next, and they need no round constants at all — those are pushed forward through
// the MDS offline. Between two layers T . T^-1' is only a scaling: if o = (E, F, v) are the products of one layer, the next
// layer's transformed input is T(y) = (4E, 4F, 2v). So the state never leaves the domain for all 22 rounds:
//   * the state is kept as W = (E, F, v), the products of the last layer, i.e. the transformed state divided by (4, 4, 2) —
//     the scaling T . T^-1' = diag(4, 4, 2) goes into the constants of the next layer's products (64, 64, 128 / 4, 32, 8 /
//     4, 8, 32, 2, 2, 2), and replacing element 0 by its S-box output adds (new - z) / 4 to aa0 and ab0 and (new - z) / 2 to b0:
//     exact quarter-integers;
//   * limb -> integer is (limb + 1.5 * 2^52), whose mantissa holds limb + 2^51 (the offsets are taken out of the constant
//     offline); carry extraction is (x + 1.5 * 2^84) - 1.5 * 2^84, the multiple of 2^32 nearest to x (balanced remainders).
// Magnitudes (tests/test_hostsim.py replays them): a normalised limb is within 2^31 + 2^20 of zero; a layer multiplies by at
// most 256 (aa), 44 (ab), 50 (b); element 0 is E0 + F0 + v0 < 350 x the layer's input; two layers after a normalisation every
// limb is below 2^48.6 with two fractional bits — 51 of the 53 bits — and the limb -> integer conversion is good to 2^51.
// Index layout of a plane as before: [0..2] = aa, [3..5] = ab, [6..11] = b. SCALED: the input is W; else the transformed state.
#if defined(__FAST_MATH__) || defined(__FINITE_MATH_ONLY__) && __FINITE_MATH_ONLY__
#error "poseidon.h: the double-precision layers rely on IEEE semantics ((x + M) - M is a rounding, not x): do not build with -ffast-math"
#endif
template <bool SCALED>
GL_HD void dom_mul_d(const double (&u)[W], double (&o)[W]) {
  constexpr double A = SCALED ? 64.0 : 16.0;
  const double t = (u[0] + u[1] + u[2]) * A;
  o[0] = __builtin_fma(u[2], A, t);
  o[1] = __builtin_fma(u[0], A, t);
  o[2] = __builtin_fma(u[1], A, t);
  constexpr double B1 = SCALED ? 4.0 : 1.0, B2 = SCALED ? 8.0 : 2.0, B8 = SCALED ? 32.0 : 8.0;
  o[3] = __builtin_fma(u[5], B8, __builtin_fma(u[3], -B1, u[4] * -B2));
  o[4] = __builtin_fma(u[3], -B8, __builtin_fma(u[4], -B1, u[5] * -B2));
  o[5] = __builtin_fma(u[3], B2, __builtin_fma(u[4], -B8, u[5] * -B1));
  constexpr double S = SCALED ? 2.0 : 1.0;
  constexpr double c[6][6] = {{2, 1, 1, -1, -16, 4},  {-4, 2, 1, 1, -1, -16}, {16, -4, 2, 1, 1, -1},
                              {1, 16, -4, 2, 1, 1},   {-1, 1, 16, -4, 2, 1},  {-1, -1, 1, 16, -4, 2}};
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double acc = u[6] * (S * c[k][0]);
#pragma unroll
    for (int i = 1; i < 6; i++) acc = __builtin_fma(u[6 + i], S * c[k][i], acc);
    o[6 + k] = acc;
  }
}
GL_HD void dom_enter_d(const double (&s)[W], double (&u)[W]) {
  double a[6];
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
GL_HD void dom_leave_d(const double (&o)[W], double (&y)[W]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double p = o[i] + o[3 + i], q = o[i] - o[3 + i];
    y[i] = p + o[6 + i];
    y[i + 6] = p - o[6 + i];
    y[i + 3] = q + o[9 + i];
    y[i + 9] = q - o[9 + i];
  }
}
// value l + 2^32 h: both limbs to within 2^31 (+ the folded top) of zero, same value mod p
GL_HD void renorm_d(double &l, double &h) {
  constexpr double MAGIC = 0x1.8p84, INV = 0x1p-32;  // (x + MAGIC) - MAGIC: x rounded to a multiple of 2^32
  const double c = (l + MAGIC) - MAGIC;
  l -= c;
  h = __builtin_fma(c, INV, h);
  const double t = (h + MAGIC) - MAGIC;       // t / 2^32 units of 2^64 == 2^32 - 1:  h <- h - t + t / 2^32,  l <- l - t / 2^32
  h = __builtin_fma(t, -(1.0 - INV), h);      // one rounding of an exactly representable result (t (1 - 2^-32) is an integer < 2^51)
  l = __builtin_fma(t, -INV, l);
}
// integer limbs |l|, |h| < 2^51 - 2^32 -> lazy u64 congruent to l + 2^32 h + c, for the constant c that `m` encodes. The bit
// pattern of l + m.m0 is B + l + lo32(c') and that of h + m.m1 is B + h + hi32(c'), B = 0x433 * 2^52 + 2^51 being the pattern
// of 1.5 * 2^52 (exponent field included: nothing is masked) and c' = c - B (1 + 2^32): the two patterns are added as
// integers at their limb positions, and what overflows 2^64 (up to 2^31) is folded like any top word.
GL_HD uint64_t recombine_d(double l, double h, Magic m) {
  const uint64_t a0 = __builtin_bit_cast(uint64_t, l + m.m0), a1 = __builtin_bit_cast(uint64_t, h + m.m1);
  uint32_t w1;
  const uint32_t carry = __builtin_add_overflow(gl::hi32(a0), gl::lo32(a1), &w1);
  return fold_top(gl::pack(gl::lo32(a0), w1), gl::hi32(a1) + carry);
}

// s <- MDS * s + next_rc on two double-precision limb planes (next_rc_base < 0: no constant)
GL_HD void mds_layer_d(uint64_t (&s)[W], int next_rc_base) {
  double ll[W], lh[W], u[W], o[W], yl[W], yh[W];
#pragma unroll
  for (int i = 0; i < W; i++) {
    ll[i] = (double)(uint32_t)s[i];
    lh[i] = (double)(uint32_t)(s[i] >> 32);
  }
  dom_enter_d(ll, u);
  dom_mul_d<false>(u, o);
  dom_leave_d(o, yl);
  yl[0] = __builtin_fma(ll[0], 8.0, yl[0]);
  dom_enter_d(lh, u);
  dom_mul_d<false>(u, o);
  dom_leave_d(o, yh);
  yh[0] = __builtin_fma(lh[0], 8.0, yh[0]);
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = recombine_d(yl[i], yh[i], rcd(next_rc_base >= 0 ? next_rc_base + i : ROUNDS * W));
}

// ---- the scaled layer two deep (partial_rounds): constants derived from dom_mul_d<true> at compile time ------------------------
template <int N> struct DomMat { double a[N][N]; };
template <int N> constexpr DomMat<N> dom_mat_sq(const DomMat<N> &m) {
  DomMat<N> r{};
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      double acc = 0;
      for (int k = 0; k < N; k++) acc += m.a[i][k] * m.a[k][j];
      r.a[i][j] = acc;
    }
  return r;
}
// the blocks of the scaled layer K as matrices (rows: products, columns: inputs): aa = 64 (J + P), P: row i takes column (i + 2) % 3
constexpr DomMat<3> DOM_K_AB = {{{-4, -8, 32}, {-32, -4, -8}, {8, -32, -4}}};
constexpr DomMat<6> DOM_K_B = {{{4, 2, 2, -2, -32, 8}, {-8, 4, 2, 2, -2, -32}, {32, -8, 4, 2, 2, -2},
                                {2, 32, -8, 4, 2, 2}, {-2, 2, 32, -8, 4, 2}, {-2, -2, 2, 32, -8, 4}}};
constexpr DomMat<3> DOM_K_AB2 = dom_mat_sq(DOM_K_AB);
constexpr DomMat<6> DOM_K_B2 = dom_mat_sq(DOM_K_B);
// l K: element 0 of the NEXT layer read off this layer's inputs (rows 0, 3, 6 of K)
GL_HD double dom_next_z(const double (&u)[W]) {
  constexpr double CZ[W] = {64, 64, 128, -4, -8, 32, 4, 2, 2, -2, -32, 8};
  double acc = u[0] * CZ[0];
#pragma unroll
  for (int k = 1; k < W; k++) acc = __builtin_fma(u[k], CZ[k], acc);
  return acc;
}
// o = K^2 u + d K t   (K t = columns 0 of the blocks of K over (4, 4, 2): a change d of element 0, one layer on)
GL_HD void dom_mul2_d(const double (&u)[W], double d, double (&o)[W]) {
  constexpr double KT[W] = {16, 32, 16, -1, -8, 2, 2, -4, 16, 1, -1, -1};
  // aa: (64 (J + P))^2 = 4096 (5 J + P^2), P^2: row i takes column (i + 1) % 3
  const double t = (u[0] + u[1] + u[2]) * 20480.0;
  o[0] = __builtin_fma(u[1], 4096.0, __builtin_fma(d, KT[0], t));
  o[1] = __builtin_fma(u[2], 4096.0, __builtin_fma(d, KT[1], t));
  o[2] = __builtin_fma(u[0], 4096.0, __builtin_fma(d, KT[2], t));
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double acc = d * KT[3 + k];
#pragma unroll
    for (int j = 0; j < 3; j++) acc = __builtin_fma(u[3 + j], DOM_K_AB2.a[k][j], acc);
    o[3 + k] = acc;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double acc = d * KT[6 + k];
#pragma unroll
    for (int j = 0; j < 6; j++) acc = __builtin_fma(u[6 + j], DOM_K_B2.a[k][j], acc);
    o[6 + k] = acc;
  }
}

// The 22 partial rounds with the MDS layers on either side of them. s in: the S-box OUTPUTS of the last full round before
// them (round 3); s out: the state entering the S-boxes of the first full round after them (round 26), constants included.
// `input(i, x)` sees the S-box input x (lazy u64) of partial round i on element 0 and returns what goes into the S-box — x
// itself for the permutation; the PoseidonGate constrains x against a wire and continues with the wire. `stop` as below.
template <typename Input, typename Stop>
GL_HD bool partial_rounds(uint64_t (&s)[W], Input input, Stop stop) {
  // straight into the domain (no constants: all pushed forward)
  double wl[W], wh[W], ol[W], oh[W];
  double nl, nh;  // limbs of the latest S-box output on element 0
  {
    double ll[W], lh[W];
#pragma unroll
    for (int k = 0; k < W; k++) {
      ll[k] = (double)(uint32_t)s[k];
      lh[k] = (double)(uint32_t)(s[k] >> 32);
    }
    nl = ll[0], nh = lh[0];
    dom_enter_d(ll, wl);
    dom_enter_d(lh, wh);
    dom_mul_d<false>(wl, ol);
    dom_mul_d<false>(wh, oh);
  }
  // TWO partial rounds per trip, with ONE application of the layer squared. Let W = (E, F, v) be the products at the top of round r,
  // K the scaled layer (dom_mul_d<true>), t = (1/4, 0, 0 | 1/4, 0, 0 | 1/2, 0, ...) what a change of element 0 is in the domain and
  // l = e_0 + e_3 + e_6 the functional that reads element 0 off the products. Round r: z = l W, S-box, W' = W + (new - z) t. Round r + 1
  // needs only ELEMENT 0 of the next layer — z' = l K W' = <CZ, W'>, twelve multiply-adds instead of the 51 operations of the whole
  // layer — and then W'' = K (K W' + (new' - z') t) = K^2 W' + (new' - z') K t: the squared layer costs what one layer costs (its
  // blocks are dense already: 6 + 9 + 36), so a trip is 12 + 51 + 12 operations per plane where two layers were 102. K^2 grows a limb
  // by at most 2^16 (aa: 4096 (5 J + P^2), rows sum to 2^16), which is the two layers of growth a limb was normalised for anyway:
  // one renorm per trip as before, now in front of the squared layer. Magnitudes (tests/test_hostsim.py replays them): W at the top of
  // a trip is below 2^47.3 with no fractional bits, z below 2^48.9.
#if defined(POSEIDON_PARTIAL_V1)  // one layer per round (rounds 2-3 of the build): kept for the A/B in tools/ubench_poseidon_variants.hip
  // one partial round: read element 0 off the products (a), S-box it, put it back (W = (E, F, v) + (new - z) / (4, 4, 2) on
  // aa0, ab0, b0, in place), optionally normalise, next products (a -> b). Two rounds per trip, ping-pong, so nothing is copied.
  auto round = [&](int i, double (&al)[W], double (&ah)[W], double (&bl)[W], double (&bh)[W], bool normalise) {
    const double zl = al[0] + al[3] + al[6], zh = ah[0] + ah[3] + ah[6];
    // element 0 of the state: E0 + F0 + v0 + the diagonal 8 of the MDS on the previous S-box output
    const uint64_t x = sbox_lazy(input(i, recombine_d(__builtin_fma(nl, 8.0, zl), __builtin_fma(nh, 8.0, zh), domd_k(i))));
    nl = (double)(uint32_t)x;
    nh = (double)(uint32_t)(x >> 32);
    const double dl = nl - zl, dh = nh - zh;
    al[0] = __builtin_fma(dl, 0.25, al[0]), ah[0] = __builtin_fma(dh, 0.25, ah[0]);
    al[3] = __builtin_fma(dl, 0.25, al[3]), ah[3] = __builtin_fma(dh, 0.25, ah[3]);
    al[6] = __builtin_fma(dl, 0.5, al[6]), ah[6] = __builtin_fma(dh, 0.5, ah[6]);
    if (normalise) {
#pragma unroll
      for (int k = 0; k < W; k++) renorm_d(al[k], ah[k]);
    }
    dom_mul_d<true>(al, bl);
    dom_mul_d<true>(ah, bh);
  };
  static_assert(PARTIAL % 2 == 0, "two rounds per trip");
#pragma unroll 1
  for (int i = 0; i < PARTIAL; i += 2) {
    if ((i & 3) == 0 && stop()) return false;
    round(i, ol, oh, wl, wh, false);
    round(i + 1, wl, wh, ol, oh, true);  // a limb holds two layers of growth: normalise every second round
  }
  // leave the domain: natural limbs, + what is pending of the constants
  double yl[W], yh[W];
  dom_leave_d(ol, yl);
  dom_leave_d(oh, yh);
  yl[0] = __builtin_fma(nl, 8.0, yl[0]), yh[0] = __builtin_fma(nh, 8.0, yh[0]);
#pragma unroll
  for (int k = 0; k < W; k++) s[k] = recombine_d(yl[k], yh[k], domd_last(k));
#else
  auto trip = [&](int i, double (&al)[W], double (&ah)[W], double (&bl)[W], double (&bh)[W]) {
    // round i: element 0 = E0 + F0 + v0 + the diagonal 8 of the MDS on the previous S-box output
    {
      const double zl = al[0] + al[3] + al[6], zh = ah[0] + ah[3] + ah[6];
      const uint64_t x = sbox_lazy(input(i, recombine_d(__builtin_fma(nl, 8.0, zl), __builtin_fma(nh, 8.0, zh), domd_k(i))));
      nl = (double)(uint32_t)x;
      nh = (double)(uint32_t)(x >> 32);
      const double dl = nl - zl, dh = nh - zh;
      al[0] = __builtin_fma(dl, 0.25, al[0]), ah[0] = __builtin_fma(dh, 0.25, ah[0]);
      al[3] = __builtin_fma(dl, 0.25, al[3]), ah[3] = __builtin_fma(dh, 0.25, ah[3]);
      al[6] = __builtin_fma(dl, 0.5, al[6]), ah[6] = __builtin_fma(dh, 0.5, ah[6]);
    }
#pragma unroll
    for (int k = 0; k < W; k++) renorm_d(al[k], ah[k]);
    // round i + 1: element 0 one layer ahead, then the squared layer with the change of element 0 carried one layer on (a -> b)
    const double zl = dom_next_z(al), zh = dom_next_z(ah);
    const uint64_t x = sbox_lazy(input(i + 1, recombine_d(__builtin_fma(nl, 8.0, zl), __builtin_fma(nh, 8.0, zh), domd_k(i + 1))));
    nl = (double)(uint32_t)x;
    nh = (double)(uint32_t)(x >> 32);
    dom_mul2_d(al, nl - zl, bl);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);  // one plane after the other: interleaved, the four arrays are live at once and the leaf hash (96 registers) spills
#endif
    dom_mul2_d(ah, nh - zh, bh);
  };
  static_assert(PARTIAL % 4 == 2, "five double trips (ping-pong, nothing is copied) and a last single one");
#pragma unroll 1
  for (int i = 0; i + 4 <= PARTIAL; i += 4) {
    if (stop()) return false;
    trip(i, ol, oh, wl, wh);
    trip(i + 2, wl, wh, ol, oh);
  }
  trip(PARTIAL - 2, ol, oh, wl, wh);
  // leave the domain: natural limbs, + what is pending of the constants
  double yl[W], yh[W];
  dom_leave_d(wl, yl);
  dom_leave_d(wh, yh);
  yl[0] = __builtin_fma(nl, 8.0, yl[0]), yh[0] = __builtin_fma(nh, 8.0, yh[0]);
#pragma unroll
  for (int k = 0; k < W; k++) s[k] = recombine_d(yl[k], yh[k], domd_last(k));
#endif
  return true;
}
struct SameInput {
  GL_HD uint64_t operator()(int, uint64_t x) const { return x; }
};

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
    mds_layer_d(s, (r + 1) * W);
  }
#pragma unroll
  for (int k = 0; k < W; k++) s[k] = sbox_lazy(s[k]);  // the last full round before the partial rounds
  if (!partial_rounds(s, SameInput(), stop)) return false;
#pragma unroll 1
  for (int r = HALF_FULL + PARTIAL; r < ROUNDS; r++) {
    if (stop()) return false;
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = sbox_lazy(s[i]);
    mds_layer_d(s, r + 1 < ROUNDS ? (r + 1) * W : -1);
  }
#pragma unroll
  for (int i = 0; i < W; i++) s[i] = gl::canon(s[i]);
  return true;
}

struct NeverStop {
  GL_HD bool operator()() const { return false; }
};
GL_HD void permute(uint64_t (&s)[W]) { permute_until(s, NeverStop()); }
// What HOST code hashes with (the transcript of a small batch, the verifier, public-input hashes): the integer round structure.
// On an x86 core it takes 1.8 us per permutation; `permute`, whose double-precision layers are shaped for the GPU's issue
// rates, takes 5.7 us there. Same function, bit for bit (tests compare the two over 600 M chained permutations).
inline void permute_host(uint64_t (&s)[W]) { permute_textbook(s); }

}  // namespace poseidon
