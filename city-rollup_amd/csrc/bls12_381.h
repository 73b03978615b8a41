// BLS12-381 base field and G1 group law for the Groth16-wrap MSM kernels (reference call site
// of the CPU prover this accelerates: `gnark_plonky2_wrapper::wrap_plonky2_proof`,
// city_rollup_circuit/src/worker/toolbox/root.rs:296-304 — the arithmetic itself lives in gnark-crypto, not in the tree).
// Curve y^2 = x^3 + 4 over F_p; the constants come from gen_bls_tables.py, which derives p from the published
// parameterisation. __host__ __device__ throughout: the host side of the library (final affine conversion) and the
// CPU unit tests (tests/hostsim) run the very same code as the kernels.
//
// Fp: 14 UNSATURATED limbs of 28 bits (392 bits for the 381-bit modulus), Montgomery form with R = 2^392. With 28-bit
// limbs a column of the schoolbook product — up to 28 partial products below 2^56 — fits a 64-bit accumulator, so the
// product-scanning Montgomery multiplication is nothing but v_mad_u64_u32 with the accumulator as its own 64-bit
// addend: no carry propagation and no zero-extension moves between the multiply-adds (a 12 x 32-bit CIOS spends half
// its instructions on those). ~500 instructions per product instead of ~1300. Values are kept fully reduced (< p, limbs
// < 2^28) and additions / subtractions are plain limb loops with a 28-bit carry — except inside the bucket accumulation,
// which runs on bounded, unreduced values ("loose arithmetic" below).
#pragma once
#include <stdint.h>

#include "bls12_381_tables.h"
#include "gl.h"

namespace bls {

constexpr int NL = 14;
constexpr int LB = 28;
constexpr uint32_t LM = (1u << LB) - 1;
constexpr int NW = 12;  // 32-bit words of a canonical coordinate at the API
struct Fp { uint32_t l[NL]; };

GL_HD Fp fp_zero() { Fp r; for (int i = 0; i < NL; i++) r.l[i] = 0; return r; }
GL_HD Fp fp_one() { Fp r; for (int i = 0; i < NL; i++) r.l[i] = BLS_R1[i]; return r; }
GL_HD bool fp_is_zero(const Fp &a) { uint32_t o = 0; for (int i = 0; i < NL; i++) o |= a.l[i]; return o == 0; }
GL_HD bool fp_eq(const Fp &a, const Fp &b) { uint32_t o = 0; for (int i = 0; i < NL; i++) o |= a.l[i] ^ b.l[i]; return o == 0; }

// a -= p if a >= p (a < 2p, limbs < 2^28)
GL_HD void fp_cond_sub_p(Fp &a) {
  uint32_t t[NL];
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t d = a.l[i] - BLS_P[i] - br;
    t[i] = d & LM;
    br = d >> 31;
  }
  if (!br) {
#pragma unroll
    for (int i = 0; i < NL; i++) a.l[i] = t[i];
  }
}
GL_HD Fp fp_add(const Fp &a, const Fp &b) {
  Fp r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t s = a.l[i] + b.l[i] + c;
    r.l[i] = s & LM;
    c = s >> LB;
  }
  fp_cond_sub_p(r);  // a, b < p < 2^381: the sum fits the 392 bits
  return r;
}
GL_HD Fp fp_sub(const Fp &a, const Fp &b) {
  Fp r;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t d = a.l[i] - b.l[i] - br;
    r.l[i] = d & LM;
    br = d >> 31;
  }
  if (br) {  // went below zero: the limbs hold a - b + 2^392; adding p and dropping the carry gives a - b + p
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t s = r.l[i] + BLS_P[i] + c;
      r.l[i] = s & LM;
      c = s >> LB;
    }
  }
  return r;
}
GL_HD Fp fp_dbl(const Fp &a) { return fp_add(a, a); }

// Montgomery product a*b/R mod p, product scanning. Deliberately NOT inlined: a point addition holds 11-16 of these —
// inlined, one kernel outgrows the instruction cache and the library takes minutes to compile.
template <bool FINAL_SUB>
GL_HD Fp fp_mul_body(const Fp &a, const Fp &b) {
  uint32_t m[NL];
  Fp r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * BLS_P[k - i];
    m[k] = ((uint32_t)acc * BLS_N0) & LM;
    acc += (uint64_t)m[k] * BLS_P[0];  // the low 28 bits of the column are now zero
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a.l[i] * b.l[k - i] + (uint64_t)m[i] * BLS_P[k - i];
    if (k - NL < NL) r.l[k - NL] = (uint32_t)acc & LM;
    acc >>= LB;
  }
  // result < 2p whenever a b < R p (R / p = 2^11.3: e.g. both operands < 32 p), limbs < 2^28
  if (FINAL_SUB) fp_cond_sub_p(r);
  return r;
}
__host__ __device__ __attribute__((noinline)) inline Fp fp_mul(Fp a, Fp b) { return fp_mul_body<true>(a, b); }  // by value: operands travel in VGPRs, not through scratch
// Montgomery square: the same column scan with every cross product a_i a_j (i < j) computed once and doubled —
// 105 + 196 multiply-adds instead of 196 + 196. Not inlined, for the same reason.
template <bool FINAL_SUB>
GL_HD Fp fp_sqr_body(const Fp &a) {
  uint32_t m[NL];
  Fp r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
    uint64_t x = 0;  // cross products of the column: at most 7 x 2^56
#pragma unroll
    for (int i = 0; 2 * i < k; i++) x += (uint64_t)a.l[i] * a.l[k - i];
    acc += x << 1;
    if (k % 2 == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * BLS_P[k - i];
    m[k] = ((uint32_t)acc * BLS_N0) & LM;
    acc += (uint64_t)m[k] * BLS_P[0];
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL; k++) {
    uint64_t x = 0;
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; i++) x += (uint64_t)a.l[i] * a.l[k - i];
    acc += x << 1;
    if (k % 2 == 0 && k / 2 < NL) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)m[i] * BLS_P[k - i];
    if (k - NL < NL) r.l[k - NL] = (uint32_t)acc & LM;
    acc >>= LB;
  }
  if (FINAL_SUB) fp_cond_sub_p(r);
  return r;
}
__host__ __device__ __attribute__((noinline)) inline Fp fp_sqr_mont(Fp a) { return fp_sqr_body<true>(a); }
GL_HD Fp fp_sqr(const Fp &a) { return fp_sqr_mont(a); }

// 12 little-endian 32-bit words of a canonical value (< p)  <->  Montgomery limbs
GL_HD Fp fp_from_canonical(const uint32_t *w) {
  Fp t, r2;
  for (int i = 0; i < NL; i++) {
    const int bit = LB * i, word = bit >> 5, sh = bit & 31;
    uint64_t v = word < NW ? w[word] : 0;
    if (word + 1 < NW) v |= (uint64_t)w[word + 1] << 32;
    t.l[i] = (uint32_t)(v >> sh) & LM;
    r2.l[i] = BLS_R2[i];
  }
  return fp_mul(t, r2);
}
GL_HD void fp_to_canonical(const Fp &a, uint32_t *w) {
  Fp one = fp_zero();
  one.l[0] = 1;
  const Fp r = fp_mul(a, one);
  for (int i = 0; i < NW; i++) w[i] = 0;
  for (int i = 0; i < NL; i++) {
    const int bit = LB * i, word = bit >> 5, sh = bit & 31;
    const uint64_t v = (uint64_t)r.l[i] << sh;
    if (word < NW) w[word] |= (uint32_t)v;
    if (word + 1 < NW) w[word + 1] |= (uint32_t)(v >> 32);
  }
}
// a^(p-2) — host side only in practice (one inversion per MSM)
GL_HD Fp fp_inv(const Fp &a) {
  Fp r = fp_one();
  for (int i = 32 * NW - 1; i >= 0; i--) {
    r = fp_sqr(r);
    uint32_t e = BLS_P32[i / 32];
    if (i / 32 == 0) e -= 2;  // p - 2: the low word of p ends in ...aaab, no borrow
    if ((e >> (i % 32)) & 1) r = fp_mul(r, a);
  }
  return r;
}

// ---- F_p^2 = F_p[u]/(u^2 + 1): the coordinate field of G2 ------------------------------------------------------
struct Fp2 { Fp c0, c1; };
GL_HD Fp2 fp2_add(const Fp2 &a, const Fp2 &b) { return {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
GL_HD Fp2 fp2_sub(const Fp2 &a, const Fp2 &b) { return {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
GL_HD Fp2 fp2_mul(const Fp2 &a, const Fp2 &b) {  // Karatsuba: three base-field products
  const Fp v0 = fp_mul(a.c0, b.c0), v1 = fp_mul(a.c1, b.c1);
  const Fp s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
  return {fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
GL_HD Fp2 fp2_sqr(const Fp2 &a) {  // (c0 + c1)(c0 - c1) + 2 c0 c1 u
  const Fp t = fp_mul(a.c0, a.c1);
  return {fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1)), fp_dbl(t)};
}
GL_HD Fp2 fp2_inv(const Fp2 &a) {  // conj(a) / (c0^2 + c1^2)
  const Fp d = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  return {fp_mul(a.c0, d), fp_mul(fp_sub(fp_zero(), a.c1), d)};
}

// ---- one group law for both groups: the coordinate field is a template parameter ----------------------------
// generic field interface (overloads) + per-field constants
GL_HD Fp f_add(const Fp &a, const Fp &b) { return fp_add(a, b); }
GL_HD Fp f_sub(const Fp &a, const Fp &b) { return fp_sub(a, b); }
GL_HD Fp f_dbl(const Fp &a) { return fp_dbl(a); }
GL_HD Fp f_mul(const Fp &a, const Fp &b) { return fp_mul(a, b); }
GL_HD Fp f_sqr(const Fp &a) { return fp_sqr(a); }
GL_HD Fp f_inv(const Fp &a) { return fp_inv(a); }
GL_HD bool f_is_zero(const Fp &a) { return fp_is_zero(a); }
GL_HD bool f_eq(const Fp &a, const Fp &b) { return fp_eq(a, b); }
GL_HD Fp2 f_add(const Fp2 &a, const Fp2 &b) { return fp2_add(a, b); }
GL_HD Fp2 f_sub(const Fp2 &a, const Fp2 &b) { return fp2_sub(a, b); }
GL_HD Fp2 f_dbl(const Fp2 &a) { return fp2_add(a, a); }
GL_HD Fp2 f_mul(const Fp2 &a, const Fp2 &b) { return fp2_mul(a, b); }
GL_HD Fp2 f_sqr(const Fp2 &a) { return fp2_sqr(a); }
GL_HD Fp2 f_inv(const Fp2 &a) { return fp2_inv(a); }
GL_HD bool f_is_zero(const Fp2 &a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
GL_HD bool f_eq(const Fp2 &a, const Fp2 &b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }

template <class F> struct Field;
template <> struct Field<Fp> {
  static constexpr int WORDS = NW;  // 32-bit words of a canonical element at the API
  static GL_HD Fp zero() { return fp_zero(); }
  static GL_HD Fp one() { return fp_one(); }
  static GL_HD Fp curve_b() { Fp b; for (int i = 0; i < NL; i++) b.l[i] = BLS_B_MONT[i]; return b; }  // y^2 = x^3 + 4
  static GL_HD Fp from_canonical(const uint32_t *w) { return fp_from_canonical(w); }
  static GL_HD void to_canonical(const Fp &a, uint32_t *w) { fp_to_canonical(a, w); }
};
template <> struct Field<Fp2> {
  static constexpr int WORDS = 2 * NW;  // c0 then c1
  static GL_HD Fp2 zero() { return {fp_zero(), fp_zero()}; }
  static GL_HD Fp2 one() { return {fp_one(), fp_zero()}; }
  static GL_HD Fp2 curve_b() { const Fp b = Field<Fp>::curve_b(); return {b, b}; }  // twist: y^2 = x^3 + 4 (1 + u)
  static GL_HD Fp2 from_canonical(const uint32_t *w) { return {fp_from_canonical(w), fp_from_canonical(w + NW)}; }
  static GL_HD void to_canonical(const Fp2 &a, uint32_t *w) { fp_to_canonical(a.c0, w); fp_to_canonical(a.c1, w + NW); }
};

template <class F> struct AffineT { F x, y; };  // Montgomery coordinates; infinity is carried separately by the callers
template <class F> struct JacT { F x, y, z; };  // x = X/Z^2, y = Y/Z^3; z = 0: infinity
using Affine = AffineT<Fp>;   // G1
using Jac = JacT<Fp>;
using Affine2 = AffineT<Fp2>; // G2
using Jac2 = JacT<Fp2>;

template <class F> GL_HD JacT<F> jac_inf() { return {Field<F>::one(), Field<F>::one(), Field<F>::zero()}; }
template <class F> GL_HD bool jac_is_inf(const JacT<F> &p) { return f_is_zero(p.z); }

template <class F> GL_HD JacT<F> jac_double(const JacT<F> &p) {  // a = 0 ("dbl-2009-l")
  if (jac_is_inf(p)) return p;
  const F A = f_sqr(p.x), B = f_sqr(p.y), C = f_sqr(B);
  F t = f_sqr(f_add(p.x, B));
  t = f_sub(f_sub(t, A), C);
  const F D = f_dbl(t), E = f_add(f_dbl(A), A), Fq = f_sqr(E);
  JacT<F> r;
  r.x = f_sub(Fq, f_dbl(D));
  const F C8 = f_dbl(f_dbl(f_dbl(C)));
  r.y = f_sub(f_mul(E, f_sub(D, r.x)), C8);
  r.z = f_dbl(f_mul(p.y, p.z));
  return r;
}
// p + q, q affine and not infinity
template <class F> GL_HD JacT<F> jac_add_mixed(const JacT<F> &p, const AffineT<F> &q) {
  if (jac_is_inf(p)) return {q.x, q.y, Field<F>::one()};
  const F z1z1 = f_sqr(p.z);
  const F u2 = f_mul(q.x, z1z1), s2 = f_mul(f_mul(q.y, p.z), z1z1);
  if (f_eq(p.x, u2)) {
    if (f_eq(p.y, s2)) return jac_double(p);
    return jac_inf<F>();
  }
  const F h = f_sub(u2, p.x), rr = f_sub(s2, p.y);
  const F hh = f_sqr(h), hhh = f_mul(h, hh), v = f_mul(p.x, hh);
  JacT<F> r;
  r.x = f_sub(f_sub(f_sqr(rr), hhh), f_dbl(v));
  r.y = f_sub(f_mul(rr, f_sub(v, r.x)), f_mul(p.y, hhh));
  r.z = f_mul(p.z, h);
  return r;
}
template <class F> GL_HD JacT<F> jac_add(const JacT<F> &p, const JacT<F> &q) {
  if (jac_is_inf(p)) return q;
  if (jac_is_inf(q)) return p;
  const F z1z1 = f_sqr(p.z), z2z2 = f_sqr(q.z);
  const F u1 = f_mul(p.x, z2z2), u2 = f_mul(q.x, z1z1);
  const F s1 = f_mul(f_mul(p.y, q.z), z2z2), s2 = f_mul(f_mul(q.y, p.z), z1z1);
  if (f_eq(u1, u2)) {
    if (f_eq(s1, s2)) return jac_double(p);
    return jac_inf<F>();
  }
  const F h = f_sub(u2, u1), rr = f_sub(s2, s1);
  const F hh = f_sqr(h), hhh = f_mul(h, hh), v = f_mul(u1, hh);
  JacT<F> r;
  r.x = f_sub(f_sub(f_sqr(rr), hhh), f_dbl(v));
  r.y = f_sub(f_mul(rr, f_sub(v, r.x)), f_mul(s1, hhh));
  r.z = f_mul(f_mul(p.z, q.z), h);
  return r;
}
// p + q when p != q; returns false and leaves r alone when the operands are the same point (the caller doubles: keeping the
// doubling out of this body halves its code and its live values)
template <class F> GL_HD bool jac_add_distinct(const JacT<F> &p, const JacT<F> &q, JacT<F> &r) {
  if (jac_is_inf(p)) { r = q; return true; }
  if (jac_is_inf(q)) { r = p; return true; }
  const F z1z1 = f_sqr(p.z), z2z2 = f_sqr(q.z);
  const F u1 = f_mul(p.x, z2z2), u2 = f_mul(q.x, z1z1);
  const F s1 = f_mul(f_mul(p.y, q.z), z2z2), s2 = f_mul(f_mul(q.y, p.z), z1z1);
  if (f_eq(u1, u2)) {
    if (f_eq(s1, s2)) return false;
    r = jac_inf<F>();
    return true;
  }
  const F h = f_sub(u2, u1), rr = f_sub(s2, s1);
  const F hh = f_sqr(h), hhh = f_mul(h, hh), v = f_mul(u1, hh);
  r.x = f_sub(f_sub(f_sqr(rr), hhh), f_dbl(v));
  r.y = f_sub(f_mul(rr, f_sub(v, r.x)), f_mul(s1, hhh));
  r.z = f_mul(f_mul(p.z, q.z), h);
  return true;
}
// Bucket accumulators use extended Jacobian ("XYZZ") coordinates: x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2; zz = 0:
// infinity. Adding an affine point costs 8 products + 2 squarings ("madd-2008-s") against 8 + 3 for the Jacobian
// mixed addition, and the result goes back to Jacobian with two products: (X ZZ, Y ZZZ, ZZ).
template <class F> struct XyzzT { F x, y, zz, zzz; };
template <class F> GL_HD XyzzT<F> xyzz_inf() { return {Field<F>::one(), Field<F>::one(), Field<F>::zero(), Field<F>::zero()}; }
template <class F> GL_HD XyzzT<F> xyzz_add_mixed(const XyzzT<F> &p, const AffineT<F> &q) {  // q not infinity
  if (f_is_zero(p.zz)) return {q.x, q.y, Field<F>::one(), Field<F>::one()};
  const F u2 = f_mul(q.x, p.zz), s2 = f_mul(q.y, p.zzz);
  if (f_eq(p.x, u2)) {
    if (!f_eq(p.y, s2)) return xyzz_inf<F>();
    // doubling of the affine point ("mdbl-2008-s-1", a = 0)
    const F u = f_dbl(q.y), v = f_sqr(u), w = f_mul(u, v), s = f_mul(q.x, v);
    const F xx = f_sqr(q.x), m = f_add(f_dbl(xx), xx);
    XyzzT<F> r;
    r.x = f_sub(f_sqr(m), f_dbl(s));
    r.y = f_sub(f_mul(m, f_sub(s, r.x)), f_mul(w, q.y));
    r.zz = v;
    r.zzz = w;
    return r;
  }
  const F pp_ = f_sub(u2, p.x), rr = f_sub(s2, p.y);
  const F pp = f_sqr(pp_), ppp = f_mul(pp_, pp), qq = f_mul(p.x, pp);
  XyzzT<F> r;
  r.x = f_sub(f_sub(f_sqr(rr), ppp), f_dbl(qq));
  r.y = f_sub(f_mul(rr, f_sub(qq, r.x)), f_mul(p.y, ppp));
  r.zz = f_mul(p.zz, pp);
  r.zzz = f_mul(p.zzz, ppp);
  return r;
}
template <class F> GL_HD JacT<F> xyzz_to_jac(const XyzzT<F> &p) {
  if (f_is_zero(p.zz)) return jac_inf<F>();
  return {f_mul(p.x, p.zz), f_mul(p.y, p.zzz), p.zz};
}

// ---- loose arithmetic for the bucket accumulation (k_bucket_sum / k_heavy_sum: all but a few percent of an MSM) ------
// Keeping every value below p costs a conditional subtraction of p after each product and each addition (56 of the ~100
// instructions of an addition, 56 of the ~450 of a product), and an F_p^2 product in Karatsuba form is five additions or
// subtractions around its three products: a third of a G2 mixed addition was spent normalising. Inside the accumulation the
// values are therefore only BOUNDED, not reduced:
//   * the Montgomery product without its final subtraction returns < 2p with normalised limbs for ANY operands with
//     a b < R p (R / p = 2^11.3) and limbs < 2^29 (a column of the product scan then stays below 2^62.3);
//   * an addition is the limb loop alone (`lz_add`), or not even that when the sum only feeds a product (`lz_add_nc`);
//   * a - b is a + K p - b for a compile-time K with K p >= b, one signed-carry limb loop (`lz_sub<K>`);
//   * the bounds of the accumulator coordinates close under the mixed addition (LooseBound below, in units of p and per
//     F_p component; tests/test_hostsim.py replays them) — for G2 with one weak reduction (x < 16 p) per addition;
//   * "is this difference zero mod p" (the doubling / cancellation case of the group law) is decided by a three-instruction
//     necessary condition — x = k p with k < 32 forces (x_0 * p^-1 mod 2^28) < 32 — and only then exactly.
// Everything that leaves the accumulation (xyzz_to_jac_loose) is canonical again.
struct FpC { uint32_t l[NL]; };
constexpr FpC kp_limbs(uint32_t K) {  // K p in normalised limbs; K < 2^11
  FpC r{};
  uint64_t c = 0;
  for (int i = 0; i < NL; i++) {
    c += (uint64_t)K * BLS_P[i];
    r.l[i] = (uint32_t)(c & LM);
    c >>= LB;
  }
  return r;
}
GL_HD Fp lz_add_nc(const Fp &a, const Fp &b) {  // limbs < 2^29 for normalised a, b: a product operand or a subtrahend only
  Fp r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
GL_HD Fp lz_add(const Fp &a, const Fp &b) {  // normalised limbs; value = a + b (must stay below 2^392)
  Fp r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t t = a.l[i] + b.l[i] + c;
    r.l[i] = t & LM;
    c = t >> LB;
  }
  return r;
}
template <int K>
GL_HD Fp lz_sub(const Fp &a, const Fp &b) {  // a + K p - b, normalised limbs; needs b <= K p; limbs of a, b below 2^30
  constexpr FpC kp = kp_limbs(K);
  Fp r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int32_t t = (int32_t)(a.l[i] - b.l[i]) + (int32_t)kp.l[i] + c;
    r.l[i] = (uint32_t)t & LM;
    c = t >> LB;  // arithmetic: -4 .. 2
  }
  return r;
}
// a -= K p if a >= K p
template <int K>
GL_HD Fp lz_weak(const Fp &a) {
  constexpr FpC kp = kp_limbs(K);
  Fp t;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t d = a.l[i] - kp.l[i] - br;
    t.l[i] = d & LM;
    br = d >> 31;
  }
  return br ? a : t;
}
__host__ __device__ __attribute__((noinline)) inline Fp lz_mul(Fp a, Fp b) { return fp_mul_body<false>(a, b); }
__host__ __device__ __attribute__((noinline)) inline Fp lz_sqr(Fp a) { return fp_sqr_body<false>(a); }
GL_HD Fp lz_canon(const Fp &a) { return fp_mul(a, fp_one()); }  // loose (< 32 p) -> canonical: a R / R
GL_HD bool lz_maybe_zero(const Fp &a) {                           // false: a != 0 mod p for sure (a < 32 p)
  constexpr uint32_t PINV = (0u - BLS_N0) & LM;                   // p^-1 mod 2^28
  return ((a.l[0] * PINV) & LM) < 32u;
}
// F_p^2 (components bounded separately)
GL_HD Fp2 lz_mul2(const Fp2 &a, const Fp2 &b) {  // components of a, b normalised, (a0 + a1)(b0 + b1) < R p; out: c0 < 4p, c1 < 6p
  const Fp v0 = lz_mul(a.c0, b.c0), v1 = lz_mul(a.c1, b.c1);
  const Fp s = lz_mul(lz_add_nc(a.c0, a.c1), lz_add_nc(b.c0, b.c1));
  return {lz_sub<2>(v0, v1), lz_sub<4>(s, lz_add_nc(v0, v1))};
}
template <int K>
GL_HD Fp2 lz_sqr2(const Fp2 &a) {  // components < K p; out: c0 < 2p, c1 < 4p
  const Fp t = lz_mul(a.c0, a.c1);
  return {lz_mul(lz_add_nc(a.c0, a.c1), lz_sub<K>(a.c0, a.c1)), lz_add(t, t)};
}
// one vocabulary for both fields
GL_HD Fp lf_mul(const Fp &a, const Fp &b) { return lz_mul(a, b); }
GL_HD Fp2 lf_mul(const Fp2 &a, const Fp2 &b) { return lz_mul2(a, b); }
template <int K> GL_HD Fp lf_sqr(const Fp &a) { return lz_sqr(a); }
template <int K> GL_HD Fp2 lf_sqr(const Fp2 &a) { return lz_sqr2<K>(a); }
template <int K> GL_HD Fp lf_sub(const Fp &a, const Fp &b) { return lz_sub<K>(a, b); }
template <int K> GL_HD Fp2 lf_sub(const Fp2 &a, const Fp2 &b) { return {lz_sub<K>(a.c0, b.c0), lz_sub<K>(a.c1, b.c1)}; }
GL_HD Fp lf_add_nc(const Fp &a, const Fp &b) { return lz_add_nc(a, b); }
GL_HD Fp2 lf_add_nc(const Fp2 &a, const Fp2 &b) { return {lz_add_nc(a.c0, b.c0), lz_add_nc(a.c1, b.c1)}; }
template <int K> GL_HD Fp lf_weak(const Fp &a) { return lz_weak<K>(a); }
template <int K> GL_HD Fp2 lf_weak(const Fp2 &a) { return {lz_weak<K>(a.c0), lz_weak<K>(a.c1)}; }
GL_HD Fp lf_canon(const Fp &a) { return lz_canon(a); }
GL_HD Fp2 lf_canon(const Fp2 &a) { return {lz_canon(a.c0), lz_canon(a.c1)}; }
GL_HD bool lf_maybe_zero(const Fp &a) { return lz_maybe_zero(a); }
GL_HD bool lf_maybe_zero(const Fp2 &a) { return lz_maybe_zero(a.c0) && lz_maybe_zero(a.c1); }
// bounds of the accumulator coordinates and of the intermediates, in units of p per F_p component:
//   M: a product, S: a square, X / Y: the accumulator's x / y after an addition (zz, zzz are products)
template <class F> struct LooseBound;
template <> struct LooseBound<Fp> { static constexpr int M = 2, S = 2, X = 8, Y = 4; static constexpr bool WEAK = false; };
template <> struct LooseBound<Fp2> { static constexpr int M = 6, S = 4, X = 16, Y = 12; static constexpr bool WEAK = true; };
template <class F> GL_HD XyzzT<F> xyzz_canon(const XyzzT<F> &p) { return {lf_canon(p.x), lf_canon(p.y), lf_canon(p.zz), lf_canon(p.zzz)}; }
// p + q with p loose (x < X p, y < Y p, zz, zzz < M p; or infinity: zz = 0 exactly), q affine canonical and not infinity
template <class F> GL_HD XyzzT<F> xyzz_add_mixed_loose(const XyzzT<F> &p, const AffineT<F> &q) {
  using B = LooseBound<F>;
  if (f_is_zero(p.zz)) return {q.x, q.y, Field<F>::one(), Field<F>::one()};
  const F u2 = lf_mul(q.x, p.zz), s2 = lf_mul(q.y, p.zzz);             // < M
  const F pp_ = lf_sub<B::X>(u2, p.x);                                  // < M + X
  if (lf_maybe_zero(pp_) && f_is_zero(lf_canon(pp_)))                   // same x: doubling or cancellation, by the exact formulas
    return xyzz_add_mixed(xyzz_canon(p), q);
  const F rr = lf_sub<B::Y>(s2, p.y);                                   // < M + Y
  const F pp = lf_sqr<B::M + B::X>(pp_), ppp = lf_mul(pp_, pp), qq = lf_mul(p.x, pp);  // < S, M, M
  XyzzT<F> r;
  // x = rr^2 - ppp - 2 qq  < S + 3 M  (G1: 8 = X; G2: 22, brought below 16 = X)
  const F x = lf_sub<3 * B::M>(lf_sqr<B::M + B::Y>(rr), lf_add_nc(ppp, lf_add_nc(qq, qq)));
  r.x = B::WEAK ? lf_weak<B::X>(x) : x;
  // y = rr (qq - x) - y1 ppp  < 2 M = Y
  r.y = lf_sub<B::M>(lf_mul(rr, lf_sub<B::X>(qq, r.x)), lf_mul(p.y, ppp));
  r.zz = lf_mul(p.zz, pp);
  r.zzz = lf_mul(p.zzz, ppp);
  return r;
}
template <class F> GL_HD JacT<F> xyzz_to_jac_loose(const XyzzT<F> &p) {
  if (f_is_zero(p.zz)) return jac_inf<F>();
  return xyzz_to_jac(xyzz_canon(p));  // the canonical functions want canonical operands
}

template <class F> GL_HD JacT<F> jac_neg(const JacT<F> &p) { return {p.x, f_sub(Field<F>::zero(), p.y), p.z}; }
// k * p for a small scalar (window-reduction offsets), double-and-add from the top bit
template <class F> GL_HD JacT<F> jac_mul_small(const JacT<F> &p, uint32_t k) {
  JacT<F> r = jac_inf<F>();
  for (int i = 31; i >= 0; i--) {
    r = jac_double(r);
    if ((k >> i) & 1) r = jac_add(r, p);
  }
  return r;
}
// Jacobian -> affine, Montgomery coordinates (p must not be infinity)
template <class F> GL_HD AffineT<F> jac_to_affine(const JacT<F> &p) {
  const F zi = f_inv(p.z), zi2 = f_sqr(zi);
  return {f_mul(p.x, zi2), f_mul(p.y, f_mul(zi2, zi))};
}
// host side: Jacobian -> affine canonical words (x then y, Field<F>::WORDS each); returns true for infinity
template <class F> GL_HD bool jac_to_affine_canonical(const JacT<F> &p, uint32_t *xy) {
  constexpr int W = Field<F>::WORDS;
  if (jac_is_inf(p)) {
    for (int i = 0; i < 2 * W; i++) xy[i] = 0;
    return true;
  }
  const AffineT<F> a = jac_to_affine(p);
  Field<F>::to_canonical(a.x, xy);
  Field<F>::to_canonical(a.y, xy + W);
  return false;
}
template <class F> GL_HD AffineT<F> affine_from_canonical(const uint32_t *xy) {
  return {Field<F>::from_canonical(xy), Field<F>::from_canonical(xy + Field<F>::WORDS)};
}
template <class F> GL_HD bool affine_on_curve(const AffineT<F> &q) {
  return f_eq(f_sqr(q.y), f_add(f_mul(f_sqr(q.x), q.x), Field<F>::curve_b()));
}

}  // namespace bls
