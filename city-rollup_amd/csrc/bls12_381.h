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
// < 2^28); additions and subtractions are plain limb loops with a 28-bit carry.
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
__host__ __device__ __attribute__((noinline)) inline Fp fp_mul(Fp a, Fp b) {  // by value: operands travel in VGPRs, not through scratch
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
  // result < 2p (both operands < p, R > 4p)
  fp_cond_sub_p(r);
  return r;
}
GL_HD Fp fp_sqr(const Fp &a) { return fp_mul(a, a); }

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

// ---- G1 -----------------------------------------------------------------------------------------------------
struct Affine { Fp x, y; };        // Montgomery coordinates; infinity is carried separately by the callers
struct Jac { Fp x, y, z; };        // x = X/Z^2, y = Y/Z^3; z = 0: infinity

GL_HD Jac jac_inf() { Jac r; r.x = fp_one(); r.y = fp_one(); r.z = fp_zero(); return r; }
GL_HD bool jac_is_inf(const Jac &p) { return fp_is_zero(p.z); }

GL_HD Jac jac_double(const Jac &p) {  // a = 0 ("dbl-2009-l")
  if (jac_is_inf(p)) return p;
  const Fp A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
  Fp t = fp_sqr(fp_add(p.x, B));
  t = fp_sub(fp_sub(t, A), C);
  const Fp D = fp_dbl(t), E = fp_add(fp_dbl(A), A), F = fp_sqr(E);
  Jac r;
  r.x = fp_sub(F, fp_dbl(D));
  const Fp C8 = fp_dbl(fp_dbl(fp_dbl(C)));
  r.y = fp_sub(fp_mul(E, fp_sub(D, r.x)), C8);
  r.z = fp_dbl(fp_mul(p.y, p.z));
  return r;
}
// p + q, q affine and not infinity
GL_HD Jac jac_add_mixed(const Jac &p, const Affine &q) {
  if (jac_is_inf(p)) { Jac r; r.x = q.x; r.y = q.y; r.z = fp_one(); return r; }
  const Fp z1z1 = fp_sqr(p.z);
  const Fp u2 = fp_mul(q.x, z1z1), s2 = fp_mul(fp_mul(q.y, p.z), z1z1);
  if (fp_eq(p.x, u2)) {
    if (fp_eq(p.y, s2)) return jac_double(p);
    return jac_inf();
  }
  const Fp h = fp_sub(u2, p.x), rr = fp_sub(s2, p.y);
  const Fp hh = fp_sqr(h), hhh = fp_mul(h, hh), v = fp_mul(p.x, hh);
  Jac r;
  r.x = fp_sub(fp_sub(fp_sqr(rr), hhh), fp_dbl(v));
  r.y = fp_sub(fp_mul(rr, fp_sub(v, r.x)), fp_mul(p.y, hhh));
  r.z = fp_mul(p.z, h);
  return r;
}
GL_HD Jac jac_add(const Jac &p, const Jac &q) {
  if (jac_is_inf(p)) return q;
  if (jac_is_inf(q)) return p;
  const Fp z1z1 = fp_sqr(p.z), z2z2 = fp_sqr(q.z);
  const Fp u1 = fp_mul(p.x, z2z2), u2 = fp_mul(q.x, z1z1);
  const Fp s1 = fp_mul(fp_mul(p.y, q.z), z2z2), s2 = fp_mul(fp_mul(q.y, p.z), z1z1);
  if (fp_eq(u1, u2)) {
    if (fp_eq(s1, s2)) return jac_double(p);
    return jac_inf();
  }
  const Fp h = fp_sub(u2, u1), rr = fp_sub(s2, s1);
  const Fp hh = fp_sqr(h), hhh = fp_mul(h, hh), v = fp_mul(u1, hh);
  Jac r;
  r.x = fp_sub(fp_sub(fp_sqr(rr), hhh), fp_dbl(v));
  r.y = fp_sub(fp_mul(rr, fp_sub(v, r.x)), fp_mul(s1, hhh));
  r.z = fp_mul(fp_mul(p.z, q.z), h);
  return r;
}
// k * p for a small scalar (window-reduction offsets), double-and-add from the top bit
GL_HD Jac jac_mul_small(const Jac &p, uint32_t k) {
  Jac r = jac_inf();
  for (int i = 31; i >= 0; i--) {
    r = jac_double(r);
    if ((k >> i) & 1) r = jac_add(r, p);
  }
  return r;
}
// host side: Jacobian -> affine canonical words (x: 12 x 32-bit words, then y); returns true for infinity
GL_HD bool jac_to_affine_canonical(const Jac &p, uint32_t *xy) {
  if (jac_is_inf(p)) {
    for (int i = 0; i < 2 * NW; i++) xy[i] = 0;
    return true;
  }
  const Fp zi = fp_inv(p.z), zi2 = fp_sqr(zi);
  fp_to_canonical(fp_mul(p.x, zi2), xy);
  fp_to_canonical(fp_mul(p.y, fp_mul(zi2, zi)), xy + NW);
  return false;
}
GL_HD bool affine_on_curve(const Affine &q) {
  Fp b;
  for (int i = 0; i < NL; i++) b.l[i] = BLS_B_MONT[i];
  return fp_eq(fp_sqr(q.y), fp_add(fp_mul(fp_sqr(q.x), q.x), b));
}

}  // namespace bls
