/*
 * ORACLE — TEST INFRASTRUCTURE ONLY. Never linked, imported or executed by the
 * product path (city-rollup_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension
 * F_p[X]/(X^2 - 7).
 *
 * The arithmetic lives in the un-vendored dependency `plonky2_field 0.2.2`
 * (git QEDProtocol/plonky2-hwa @ 6a8ca008, /root/reference/Cargo.lock:4174-4223),
 * type `GoldilocksField`, which the reference names at
 * city_rollup_core_worker/src/lib.rs:25-26 (`F = GoldilocksField`, D = 2).
 * This is a restatement of the published definition (canonical u64 in
 * [0, p)), not a copy of that crate.
 */
#ifndef CITY_ORACLE_GOLDILOCKS_H
#define CITY_ORACLE_GOLDILOCKS_H

#include <stdint.h>

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^32 - 1 == 2^64 mod p */
#define GL_GENERATOR 7ULL    /* multiplicative generator; also LDE coset shift */
#define GL_TWO_ADICITY 32
#define GL_EXT_W 7ULL /* extension is X^2 = 7 */

typedef unsigned __int128 gl_u128;

/* (branch-free forms throughout: the conditions depend on the data, a mispredicted branch costs more than the arithmetic) */
static inline uint64_t gl_canon(uint64_t x) { return x - (GL_P & (0 - (uint64_t)(x >= GL_P))); }

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
  /* a, b canonical */
  uint64_t s = a + b;
  return s - (GL_P & (0 - (uint64_t)((s < a) | (s >= GL_P))));
}

static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
  return a - b + (GL_P & (0 - (uint64_t)(a < b)));
}

static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }

/* slow, obviously-correct reduction (used by tests to cross-check the fast one) */
static inline uint64_t gl_reduce128_slow(gl_u128 x) { return (uint64_t)(x % GL_P); }

/* x = lo + 2^64*(hl + 2^32*hh); 2^64 == 2^32-1, 2^96 == -1 (mod p)  =>  x == lo - hh + hl*(2^32-1) */
static inline uint64_t gl_reduce128(gl_u128 x) {
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  uint64_t hh = hi >> 32, hl = hi & GL_EPS;
  uint64_t t0 = lo - hh;
  t0 -= GL_EPS & (0 - (uint64_t)(lo < hh)); /* borrowed 2^64 == EPS (mod p) */
  uint64_t t1 = hl * GL_EPS;
  uint64_t r = t0 + t1;
  r += GL_EPS & (0 - (uint64_t)(r < t1)); /* carried 2^64 == EPS (mod p) */
  return gl_canon(r);
}

static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
  return gl_reduce128((gl_u128)a * b);
}

static inline uint64_t gl_pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_mul(b, b);
    e >>= 1;
  }
  return r;
}

static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

/* primitive 2^k-th root of unity: g^((p-1)/2^k) with g = 7 (k <= 32). */
static inline uint64_t gl_root_of_unity(int log_n) {
  uint64_t base = gl_pow(GL_GENERATOR, (GL_P - 1) >> GL_TWO_ADICITY); /* order 2^32 */
  for (int i = log_n; i < GL_TWO_ADICITY; i++) base = gl_mul(base, base);
  return base;
}

/* ---- quadratic extension: a = a0 + a1*X, X^2 = 7 ---- */
typedef struct { uint64_t c[2]; } gl2_t;

static inline gl2_t gl2_make(uint64_t a, uint64_t b) { gl2_t r = {{a, b}}; return r; }
static inline gl2_t gl2_from_base(uint64_t a) { return gl2_make(a, 0); }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) {
  return gl2_make(gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1]));
}
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) {
  return gl2_make(gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1]));
}
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
  uint64_t c0 = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], b.c[1])));
  uint64_t c1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
  return gl2_make(c0, c1);
}
static inline gl2_t gl2_scale(gl2_t a, uint64_t s) {
  return gl2_make(gl_mul(a.c[0], s), gl_mul(a.c[1], s));
}
static inline gl2_t gl2_inv(gl2_t a) {
  /* 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - 7 a1^2) */
  uint64_t n = gl_sub(gl_mul(a.c[0], a.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], a.c[1])));
  uint64_t ni = gl_inv(n);
  return gl2_make(gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni));
}
static inline gl2_t gl2_pow(gl2_t b, uint64_t e) {
  gl2_t r = gl2_from_base(1);
  while (e) {
    if (e & 1) r = gl2_mul(r, b);
    b = gl2_mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline int gl2_eq(gl2_t a, gl2_t b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }

#endif
