/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see cityoracle.h): never linked, imported or executed by the product path.
 *
 * Poseidon-Goldilocks over EIGHT states at once with AVX-512 (one state per 64-bit lane): what a CPU prover that uses its vector
 * units does — plonky2 ships AVX2 / AVX-512 Poseidon code for x86 (UPSTREAM-MEMORY; the crate is not in the tree), so a scalar
 * port understates the CPU side. Used ONLY by bench.py's cpu_baseline legs (or_set_simd_poseidon(1)), labelled "port-simd";
 * the checker of the tests stays the scalar textbook form, and tests/test_oracle_simd.py holds this file against it bit for bit.
 * Same map as or_poseidon_permute: textbook round structure (constants, S-box, MDS), the MDS through the multiplier-free
 * decomposition of cityoracle.c `mds_plane` on the two 32-bit halves. Values travel between steps as ANY u64 congruent to the
 * element ("lazy"); only the output is canonical.
 * Compiled into the library whatever -march says (target attribute); or_simd_available() asks the CPU at run time.
 */
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "cityoracle.h"
#include "goldilocks.h"

#define W 12
#define ROUNDS 30
#define HALF_FULL 4
#define PARTIAL 22
#define AVX512 __attribute__((target("avx512f,avx512dq,avx512vl,avx512bw")))

static int g_simd_on = 0;
int or_simd_available(void) {
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
         __builtin_cpu_supports("avx512bw");
}
void or_set_simd_poseidon(int on) { g_simd_on = on != 0 && or_simd_available(); }
int or_simd_poseidon_enabled(void) { return g_simd_on; }

typedef __m512i V;

AVX512 static inline V v_set1(uint64_t x) { return _mm512_set1_epi64((long long)x); }

/* 64 x 64 -> 128 per lane from four 32 x 32 -> 64 products */
AVX512 static inline void v_mul_wide(V a, V b, V *lo, V *hi) {
  const V m32 = v_set1(0xFFFFFFFFull);
  const V ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
  const V ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
  const V t = _mm512_add_epi64(lh, _mm512_srli_epi64(ll, 32));            /* < 2^64 */
  const V u = _mm512_add_epi64(hl, _mm512_and_si512(t, m32));              /* < 2^64 */
  *lo = _mm512_or_si512(_mm512_and_si512(ll, m32), _mm512_slli_epi64(u, 32));
  *hi = _mm512_add_epi64(hh, _mm512_add_epi64(_mm512_srli_epi64(t, 32), _mm512_srli_epi64(u, 32)));
}
/* lo + 2^64 hi -> lazy u64: lo - (hi >> 32) + (hi & M32)(2^32 - 1), every wrap of 2^64 repaid with -+ EPS (goldilocks.h gl_reduce128) */
AVX512 static inline V v_reduce128(V lo, V hi) {
  const V m32 = v_set1(0xFFFFFFFFull);
  const V hh = _mm512_srli_epi64(hi, 32), hl = _mm512_and_si512(hi, m32);
  V t0 = _mm512_sub_epi64(lo, hh);
  const __mmask8 borrow = _mm512_cmplt_epu64_mask(lo, hh);
  t0 = _mm512_mask_sub_epi64(t0, borrow, t0, m32);
  const V t1 = _mm512_sub_epi64(_mm512_slli_epi64(hl, 32), hl);
  V r = _mm512_add_epi64(t0, t1);
  const __mmask8 carry = _mm512_cmplt_epu64_mask(r, t1);
  return _mm512_mask_add_epi64(r, carry, r, m32);
}
AVX512 static inline V v_mul(V a, V b) {
  V lo, hi;
  v_mul_wide(a, b, &lo, &hi);
  return v_reduce128(lo, hi);
}
AVX512 static inline V v_sbox(V x) {
  const V x2 = v_mul(x, x), x4 = v_mul(x2, x2), x3 = v_mul(x, x2);
  return v_mul(x3, x4);
}
/* lazy a + canonical c: one wrap at most, and the repaired sum cannot wrap again (it is below c) */
AVX512 static inline V v_add_const(V a, V c) {
  const V r = _mm512_add_epi64(a, c);
  const __mmask8 carry = _mm512_cmplt_epu64_mask(r, a);
  return _mm512_mask_add_epi64(r, carry, r, v_set1(0xFFFFFFFFull));
}
AVX512 static inline V v_canon(V x) {
  const V p = v_set1(GL_P);
  const __mmask8 ge = _mm512_cmpge_epu64_mask(x, p);
  return _mm512_mask_sub_epi64(x, ge, x, p);
}

#define ADD(a, b) _mm512_add_epi64(a, b)
#define SUB(a, b) _mm512_sub_epi64(a, b)
#define SHL(a, k) _mm512_slli_epi64(a, k)
/* one 32-bit plane of eight states: cityoracle.c mds_plane, lane-wise */
AVX512 static inline void v_mds_plane(const V s[W], V y[W]) {
  V a[6], b[6];
  for (int i = 0; i < 6; i++) {
    a[i] = ADD(s[i], s[i + 6]);
    b[i] = SUB(s[i], s[i + 6]);
  }
  const V aa0 = ADD(a[0], a[3]), aa1 = ADD(a[1], a[4]), aa2 = ADD(a[2], a[5]);
  const V ab0 = SUB(a[0], a[3]), ab1 = SUB(a[1], a[4]), ab2 = SUB(a[2], a[5]);
  const V t16 = SHL(ADD(ADD(aa0, aa1), aa2), 4);
  const V e0 = ADD(t16, SHL(aa2, 4)), e1 = ADD(t16, SHL(aa0, 4)), e2 = ADD(t16, SHL(aa1, 4));
  const V zero = _mm512_setzero_si512();
  const V f0 = SUB(SUB(SHL(ab2, 3), ab0), SHL(ab1, 1));
  const V f1 = SUB(SUB(SUB(zero, SHL(ab0, 3)), ab1), SHL(ab2, 1));
  const V f2 = SUB(SUB(SHL(ab0, 1), SHL(ab1, 3)), ab2);
  const V pc[6] = {ADD(e0, f0), ADD(e1, f1), ADD(e2, f2), SUB(e0, f0), SUB(e1, f1), SUB(e2, f2)};
  V v[6];
  v[0] = ADD(SUB(SUB(ADD(ADD(SHL(b[0], 1), b[1]), b[2]), b[3]), SHL(b[4], 4)), SHL(b[5], 2));
  v[1] = SUB(SUB(ADD(ADD(SUB(SHL(b[1], 1), SHL(b[0], 2)), b[2]), b[3]), b[4]), SHL(b[5], 4));
  v[2] = SUB(ADD(ADD(ADD(SUB(SHL(b[0], 4), SHL(b[1], 2)), SHL(b[2], 1)), b[3]), b[4]), b[5]);
  v[3] = ADD(ADD(ADD(SUB(ADD(b[0], SHL(b[1], 4)), SHL(b[2], 2)), SHL(b[3], 1)), b[4]), b[5]);
  v[4] = ADD(ADD(SUB(ADD(SUB(b[1], b[0]), SHL(b[2], 4)), SHL(b[3], 2)), SHL(b[4], 1)), b[5]);
  v[5] = ADD(SUB(ADD(SUB(SUB(b[2], b[0]), b[1]), SHL(b[3], 4)), SHL(b[4], 2)), SHL(b[5], 1));
  for (int i = 0; i < 6; i++) {
    y[i] = ADD(pc[i], v[i]);
    y[i + 6] = SUB(pc[i], v[i]);
  }
  y[0] = ADD(y[0], SHL(s[0], 3));
}
AVX512 static inline void v_mds_layer(V s[W]) {
  const V m32 = v_set1(0xFFFFFFFFull);
  V lo[W], hi[W], yl[W], yh[W];
  for (int i = 0; i < W; i++) {
    lo[i] = _mm512_and_si512(s[i], m32);
    hi[i] = _mm512_srli_epi64(s[i], 32);
  }
  v_mds_plane(lo, yl);
  v_mds_plane(hi, yh);
  for (int i = 0; i < W; i++) {  /* yl + 2^32 yh, both below 2^42: a 74-bit value */
    const V l = ADD(yl[i], SHL(yh[i], 32));
    const __mmask8 carry = _mm512_cmplt_epu64_mask(l, yl[i]);
    V h = _mm512_srli_epi64(yh[i], 32);
    h = _mm512_mask_add_epi64(h, carry, h, v_set1(1));
    s[i] = v_reduce128(l, h);
  }
}

static uint64_t RCS[ROUNDS * W];
static int rcs_ready = 0;

/* eight states, structure of arrays: st[i][k] = element i of state k. Canonical in, canonical out. */
AVX512 void or_poseidon_permute_x8(uint64_t st[W][8]) {
  if (!rcs_ready) {
    or_poseidon_round_constants(RCS);
    rcs_ready = 1;
  }
  V s[W];
  for (int i = 0; i < W; i++) s[i] = _mm512_loadu_si512((const void *)st[i]);
  for (int r = 0; r < ROUNDS; r++) {
    for (int i = 0; i < W; i++) s[i] = v_add_const(s[i], v_set1(RCS[r * W + i]));
    if (r < HALF_FULL || r >= HALF_FULL + PARTIAL) {
      for (int i = 0; i < W; i++) s[i] = v_sbox(s[i]);
    } else {
      s[0] = v_sbox(s[0]);
    }
    v_mds_layer(s);
  }
  for (int i = 0; i < W; i++) _mm512_storeu_si512((void *)st[i], v_canon(s[i]));
}

/* digests of eight consecutive column-major leaves i0 .. i0 + 7 (hash_or_noop with leaf_len > 4: the overwrite-mode sponge) */
AVX512 void or_simd_leaf_hash_cols_x8(const uint64_t *cols, size_t leaf_len, size_t col_stride, size_t i0, uint64_t *digests_out) {
  uint64_t st[W][8];
  memset(st, 0, sizeof st);
  for (size_t off = 0; off < leaf_len; off += 8) {
    const size_t c = leaf_len - off < 8 ? leaf_len - off : 8;
    for (size_t j = 0; j < c; j++) memcpy(st[j], cols + (off + j) * col_stride + i0, 64);
    or_poseidon_permute_x8(st);
  }
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < 4; j++) digests_out[4 * k + j] = st[j][k];
}
/* eight parents from sixteen consecutive child digests */
AVX512 void or_simd_two_to_one_x8(const uint64_t *children, uint64_t *parents_out) {
  uint64_t st[W][8];
  memset(st, 0, sizeof st);
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < 8; j++) st[j][k] = children[8 * k + j];
  or_poseidon_permute_x8(st);
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < 4; j++) parents_out[4 * k + j] = st[j][k];
}
/* count x 12 states, array of structures, in place (the SIMD form of or_poseidon_permute_many's inner loop; count % 8 == 0) */
AVX512 void or_simd_permute_aos_x8(uint64_t *states) {
  uint64_t st[W][8];
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < W; j++) st[j][k] = states[W * k + j];
  or_poseidon_permute_x8(st);
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < W; j++) states[W * k + j] = st[j][k];
}
