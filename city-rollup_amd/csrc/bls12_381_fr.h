// BLS12-381 scalar field F_r (255 bits, 2-adicity 32, generator 7) for the NTTs of the Groth16 wrap proof
// ("witness solve, Fr NTTs, G1/G2 MSMs"). Same unsaturated-limb technique as bls12_381.h:
// 10 limbs of 28 bits, Montgomery radix 2^280, product scanning on v_mad_u64_u32 with no carry chains
// (a column holds at most 20 partial products below 2^56). Values are kept fully reduced.
#pragma once
#include <stdint.h>

#include "bls12_381_tables.h"
#include "gl.h"

namespace blsfr {

constexpr int NL = 10;
constexpr int LB = 28;
constexpr uint32_t LM = (1u << LB) - 1;
constexpr int NW = 8;  // 32-bit words of a canonical element at the API
struct Fr { uint32_t l[NL]; };

GL_HD Fr fr_zero() { Fr r; for (int i = 0; i < NL; i++) r.l[i] = 0; return r; }
GL_HD Fr fr_one() { Fr r; for (int i = 0; i < NL; i++) r.l[i] = BLS_FR_R1[i]; return r; }

GL_HD void fr_cond_sub(Fr &a) {
  uint32_t t[NL];
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t d = a.l[i] - BLS_FR_P[i] - br;
    t[i] = d & LM;
    br = d >> 31;
  }
  if (!br) {
#pragma unroll
    for (int i = 0; i < NL; i++) a.l[i] = t[i];
  }
}
GL_HD Fr fr_add(const Fr &a, const Fr &b) {
  Fr r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t s = a.l[i] + b.l[i] + c;
    r.l[i] = s & LM;
    c = s >> LB;
  }
  fr_cond_sub(r);
  return r;
}
GL_HD Fr fr_sub(const Fr &a, const Fr &b) {
  Fr r;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t d = a.l[i] - b.l[i] - br;
    r.l[i] = d & LM;
    br = d >> 31;
  }
  if (br) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t s = r.l[i] + BLS_FR_P[i] + c;
      r.l[i] = s & LM;
      c = s >> LB;
    }
  }
  return r;
}
// Montgomery product a*b/R mod r (product scanning; ~300 instructions: inlined)
GL_HD Fr fr_mul(const Fr &a, const Fr &b) {
  uint32_t m[NL];
  Fr r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * BLS_FR_P[k - i];
    m[k] = ((uint32_t)acc * BLS_FR_N0) & LM;
    acc += (uint64_t)m[k] * BLS_FR_P[0];
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a.l[i] * b.l[k - i] + (uint64_t)m[i] * BLS_FR_P[k - i];
    r.l[k - NL] = (uint32_t)acc & LM;
    acc >>= LB;
  }
  fr_cond_sub(r);
  return r;
}
// 8 little-endian 32-bit words of a canonical value (< r)  <->  Montgomery limbs
GL_HD Fr fr_from_canonical(const uint32_t *w) {
  Fr t, r2;
  for (int i = 0; i < NL; i++) {
    const int bit = LB * i, word = bit >> 5, sh = bit & 31;
    uint64_t v = word < NW ? w[word] : 0;
    if (word + 1 < NW) v |= (uint64_t)w[word + 1] << 32;
    t.l[i] = (uint32_t)(v >> sh) & LM;
    r2.l[i] = BLS_FR_R2[i];
  }
  return fr_mul(t, r2);
}
GL_HD void fr_to_canonical(const Fr &a, uint32_t *w) {
  Fr one = fr_zero();
  one.l[0] = 1;
  const Fr r = fr_mul(a, one);
  for (int i = 0; i < NW; i++) w[i] = 0;
  for (int i = 0; i < NL; i++) {
    const int bit = LB * i, word = bit >> 5, sh = bit & 31;
    const uint64_t v = (uint64_t)r.l[i] << sh;
    if (word < NW) w[word] |= (uint32_t)v;
    if (word + 1 < NW) w[word + 1] |= (uint32_t)(v >> 32);
  }
}
GL_HD Fr fr_pow_u64(const Fr &a, uint64_t e) {
  Fr r = fr_one(), b = a;
  while (e) {
    if (e & 1) r = fr_mul(r, b);
    b = fr_mul(b, b);
    e >>= 1;
  }
  return r;
}
// a^(r-2)
GL_HD Fr fr_inv(const Fr &a) {
  uint32_t e[NW];
  uint32_t br = 2;  // r - 2 with borrow propagation (r ends in ...ffffffff00000001)
  for (int i = 0; i < NW; i++) {
    const uint64_t d = (uint64_t)BLS_FR_P32[i] - br;
    e[i] = (uint32_t)d;
    br = (uint32_t)(d >> 63);
  }
  Fr r = fr_one();
  for (int i = 32 * NW - 1; i >= 0; i--) {
    r = fr_mul(r, r);
    if ((e[i / 32] >> (i % 32)) & 1) r = fr_mul(r, a);
  }
  return r;
}
GL_HD bool fr_is_canonical(const uint32_t *w) {
  for (int k = NW - 1; k >= 0; k--) {
    if (w[k] < BLS_FR_P32[k]) return true;
    if (w[k] > BLS_FR_P32[k]) return false;
  }
  return false;
}

}  // namespace blsfr
