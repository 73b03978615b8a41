/*
 * ORACLE — TEST INFRASTRUCTURE ONLY. See cityoracle.h / goldilocks.h.
 *
 * CPU restatement (plain C) of the plonky2 0.2.2 primitives city-rollup calls
 * through `CircuitData::prove`. plonky2 is an un-vendored git dependency
 * (QEDProtocol/plonky2-hwa @ 6a8ca008, /root/reference/Cargo.lock:4174-4223),
 * so every function below restates the published algorithm and cites the
 * reference call site / known-answer data that pins it.
 */
#include "cityoracle.h"
#include "goldilocks.h"

#include <stdlib.h>
#include <string.h>

static int g_threads = 1;
void or_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int or_get_threads(void) { return g_threads; }

uint64_t or_gl_add(uint64_t a, uint64_t b) { return gl_add(a, b); }
uint64_t or_gl_sub(uint64_t a, uint64_t b) { return gl_sub(a, b); }
uint64_t or_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t or_gl_mul_slow(uint64_t a, uint64_t b) { return gl_reduce128_slow((gl_u128)a * b); }
uint64_t or_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t or_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }
uint64_t or_gl_root_of_unity(int log_n) { return gl_root_of_unity(log_n); }

/* ------------------------------------------------------------------------- */
/* Poseidon round constants.
 *
 * Not present anywhere in /root/reference (SURVEY.md finding #5). plonky2
 * documents them as "generated with ChaCha8 seeded with 0, 12*30 values drawn
 * uniformly below the field order". Restated here from the public definitions
 * of ChaCha (8 rounds, 64-bit block counter, stream 0), rand_core's
 * `seed_from_u64` (PCG32 expansion of the u64 seed into the 32-byte key) and
 * rand's widening-multiply uniform sampler. Pinned by: the first constant
 * 0xb585f766f2144405 quoted in SURVEY.md finding #5, and by all 2x128
 * iterated hashes of city_crypto/src/hash/cached_zero_hashes.rs:11-2065
 * (tests/test_oracle_golden.py). */

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d)                                                          \
  a += b; d ^= a; d = rotl32(d, 16);                                            \
  c += d; b ^= c; b = rotl32(b, 12);                                            \
  a += b; d ^= a; d = rotl32(d, 8);                                             \
  c += d; b ^= c; b = rotl32(b, 7)

static void chacha8_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
  uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                     key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                     (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
  uint32_t w[16];
  memcpy(w, st, sizeof w);
  for (int i = 0; i < 4; i++) { /* 8 rounds = 4 double rounds */
    QR(w[0], w[4], w[8], w[12]);
    QR(w[1], w[5], w[9], w[13]);
    QR(w[2], w[6], w[10], w[14]);
    QR(w[3], w[7], w[11], w[15]);
    QR(w[0], w[5], w[10], w[15]);
    QR(w[1], w[6], w[11], w[12]);
    QR(w[2], w[7], w[8], w[13]);
    QR(w[3], w[4], w[9], w[14]);
  }
  for (int i = 0; i < 16; i++) out[i] = w[i] + st[i];
}

typedef struct {
  uint32_t key[8];
  uint64_t counter;
  uint32_t buf[16];
  int pos;
} chacha8_rng;

static void rng_seed_from_u64(chacha8_rng *r, uint64_t state) {
  for (int i = 0; i < 8; i++) { /* PCG32 expansion */
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    r->key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  r->counter = 0;
  r->pos = 16;
}
static uint32_t rng_u32(chacha8_rng *r) {
  if (r->pos == 16) {
    chacha8_block(r->key, r->counter++, r->buf);
    r->pos = 0;
  }
  return r->buf[r->pos++];
}
static uint64_t rng_u64(chacha8_rng *r) {
  uint64_t lo = rng_u32(r);
  uint64_t hi = rng_u32(r);
  return lo | (hi << 32);
}
/* uniform in [0, range): widening multiply, reject the biased low zone */
static uint64_t rng_below(chacha8_rng *r, uint64_t range) {
  uint64_t zone = (range << __builtin_clzll(range)) - 1;
  for (;;) {
    gl_u128 m = (gl_u128)rng_u64(r) * range;
    if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
  }
}

#define POS_W 12
#define POS_FULL_HALF 4
#define POS_PARTIAL 22
#define POS_ROUNDS (2 * POS_FULL_HALF + POS_PARTIAL)

static uint64_t RC[POS_ROUNDS * POS_W];
static int rc_ready = 0;
/* MDS matrix of Poseidon-Goldilocks width 12: circulant + diagonal.
 * (UPSTREAM parameter; pinned by the same KATs.) */
static const uint64_t MDS_CIRC[POS_W] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t MDS_DIAG[POS_W] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static void rc_init(void) {
  if (rc_ready) return;
  chacha8_rng r;
  rng_seed_from_u64(&r, 0);
  for (int i = 0; i < POS_ROUNDS * POS_W; i++) RC[i] = rng_below(&r, GL_P);
  rc_ready = 1;
}

void or_poseidon_round_constants(uint64_t *out) {
  rc_init();
  memcpy(out, RC, sizeof RC);
}
void or_poseidon_mds(uint64_t *circ, uint64_t *diag) {
  memcpy(circ, MDS_CIRC, sizeof MDS_CIRC);
  memcpy(diag, MDS_DIAG, sizeof MDS_DIAG);
}

static inline uint64_t sbox7(uint64_t x) {
  uint64_t x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x, x2);
  return gl_mul(x3, x4);
}

static inline void mds_layer(uint64_t s[POS_W]) {
  uint64_t o[POS_W];
  for (int r = 0; r < POS_W; r++) {
    gl_u128 acc = 0; /* 12 * 2^64 * 41 + 8*2^64 fits easily */
    for (int i = 0; i < POS_W; i++) acc += (gl_u128)s[(i + r) % POS_W] * MDS_CIRC[i];
    acc += (gl_u128)s[r] * MDS_DIAG[r];
    o[r] = gl_reduce128(acc);
  }
  memcpy(s, o, sizeof o);
}

/* The MDS layer without 128-bit products (what plonky2's portable code reaches with its `mds_multiply_freq`): each element
 * is split into its 32-bit halves, each half-plane goes through a multiplier-free decomposition of the circulant
 * (mds_plane below), and one 96-bit recombination + reduction per output replaces 12 u128 multiply-adds.
 * Same map as mds_layer: or_set_fast_poseidon(1) routes every permutation of the oracle through it — the CPU
 * BASELINE of bench.py uses it (a port that keeps the textbook MDS would be a strawman); tests/test_oracle_golden.py
 * pins both forms on the reference's known-answer data. */
static int g_fast_poseidon = 0;
void or_set_fast_poseidon(int on) { g_fast_poseidon = on != 0; }

/* One 32-bit plane of the state (values < 2^32 held in u64): y[r] = sum_i C[i] s[(i + r) % 12] + 8 s[0] [r == 0] in
 * wrap-around 64-bit arithmetic (true results < 2^41). The length-12 cyclic correlation splits through
 * x^12 - 1 = (x^6 - 1)(x^6 + 1), x^6 - 1 = (x^3 - 1)(x^3 + 1) into a cyclic-3, a negacyclic-3 and a negacyclic-6 part whose
 * transformed kernels [16,16,32], [-1,-8,2], [2,-4,16,1,-1,-1] are powers of two: shifts and adds only (the same
 * decomposition plonky2 reaches with its `mds_multiply_freq`; here over the integers instead of an FFT). */
static inline void mds_plane(const uint64_t s[POS_W], uint64_t y[POS_W]) {
  uint64_t a[6], b[6];
  for (int i = 0; i < 6; i++) {
    a[i] = s[i] + s[i + 6];
    b[i] = s[i] - s[i + 6];
  }
  const uint64_t aa0 = a[0] + a[3], aa1 = a[1] + a[4], aa2 = a[2] + a[5];
  const uint64_t ab0 = a[0] - a[3], ab1 = a[1] - a[4], ab2 = a[2] - a[5];
  const uint64_t t16 = (aa0 + aa1 + aa2) << 4;
  const uint64_t e0 = t16 + (aa2 << 4), e1 = t16 + (aa0 << 4), e2 = t16 + (aa1 << 4);
  const uint64_t f0 = (ab2 << 3) - ab0 - (ab1 << 1);
  const uint64_t f1 = 0 - (ab0 << 3) - ab1 - (ab2 << 1);
  const uint64_t f2 = (ab0 << 1) - (ab1 << 3) - ab2;
  const uint64_t pc[6] = {e0 + f0, e1 + f1, e2 + f2, e0 - f0, e1 - f1, e2 - f2};
  uint64_t v[6];
  v[0] = (b[0] << 1) + b[1] + b[2] - b[3] - (b[4] << 4) + (b[5] << 2);
  v[1] = (b[1] << 1) - (b[0] << 2) + b[2] + b[3] - b[4] - (b[5] << 4);
  v[2] = (b[0] << 4) - (b[1] << 2) + (b[2] << 1) + b[3] + b[4] - b[5];
  v[3] = b[0] + (b[1] << 4) - (b[2] << 2) + (b[3] << 1) + b[4] + b[5];
  v[4] = b[1] - b[0] + (b[2] << 4) - (b[3] << 2) + (b[4] << 1) + b[5];
  v[5] = b[2] - b[0] - b[1] + (b[3] << 4) - (b[4] << 2) + (b[5] << 1);
  for (int i = 0; i < 6; i++) {
    y[i] = pc[i] + v[i];
    y[i + 6] = pc[i] - v[i];
  }
  y[0] += s[0] << 3;
}

static inline void mds_layer_fast(uint64_t s[POS_W]) {
  uint64_t lo[POS_W], hi[POS_W], yl[POS_W], yh[POS_W];
  for (int i = 0; i < POS_W; i++) {
    lo[i] = (uint32_t)s[i];
    hi[i] = s[i] >> 32;
  }
  mds_plane(lo, yl);
  mds_plane(hi, yh);
  for (int i = 0; i < POS_W; i++) s[i] = gl_reduce128((gl_u128)yl[i] + ((gl_u128)yh[i] << 32));
}

/* The textbook ("naive") form: every round = add constants, S-box (all lanes in
 * full rounds, lane 0 in partial rounds), full MDS. plonky2's optimised partial
 * rounds are algebraically the same map. */
void or_poseidon_permute(uint64_t s[POS_W]) {
  rc_init();
  for (int rnd = 0; rnd < POS_ROUNDS; rnd++) {
    for (int i = 0; i < POS_W; i++) s[i] = gl_add(s[i], RC[rnd * POS_W + i]);
    if (rnd < POS_FULL_HALF || rnd >= POS_FULL_HALF + POS_PARTIAL) {
      for (int i = 0; i < POS_W; i++) s[i] = sbox7(s[i]);
    } else {
      s[0] = sbox7(s[0]);
    }
    if (g_fast_poseidon) mds_layer_fast(s);
    else mds_layer(s);
  }
}

void or_poseidon_permute_many(uint64_t *states, size_t count) {
  rc_init();
  size_t done = 0;
  if (or_simd_poseidon_enabled()) {  /* bench.py's cpu_baseline only: eight states per AVX-512 permutation (poseidon_simd.c) */
    const size_t groups = count / 8;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (size_t g = 0; g < groups; g++) or_simd_permute_aos_x8(states + 8 * g * POS_W);
    done = 8 * groups;
  }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t i = done; i < count; i++) or_poseidon_permute(states + i * POS_W);
}

void or_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]) {
  uint64_t st[POS_W] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t c = n - off < 8 ? n - off : 8;
    memcpy(st, in + off, c * sizeof(uint64_t)); /* overwrite mode */
    or_poseidon_permute(st);
  }
  memcpy(out, st, 4 * sizeof(uint64_t));
}

void or_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]) {
  if (n <= 4) {
    memset(out, 0, 4 * sizeof(uint64_t));
    memcpy(out, in, n * sizeof(uint64_t));
  } else {
    or_hash_no_pad(in, n, out);
  }
}

void or_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
  uint64_t st[POS_W] = {0};
  memcpy(st, l, 32);
  memcpy(st + 4, r, 32);
  or_poseidon_permute(st);
  memcpy(out, st, 32);
}

/* ------------------------------------------------------------------------- */
/* Merkle tree with cap. */

static void merkle_from_leaf_digests(uint64_t *level0, size_t n_leaves, int cap_height,
                                     uint64_t *digests_out, uint64_t *cap_out) {
  size_t cap_n = (size_t)1 << cap_height;
  uint64_t *cur = level0;
  size_t n = n_leaves;
  uint64_t *dout = digests_out;
  uint64_t *owned = NULL;
  while (n > cap_n) {
    if (dout) {
      memcpy(dout, cur, n * 32);
      dout += n * 4;
    }
    uint64_t *next = (uint64_t *)malloc((n / 2) * 32);
    size_t first = 0;
    if (or_simd_poseidon_enabled()) {
      const size_t groups = (n / 2) / 8;
#pragma omp parallel for num_threads(g_threads) schedule(static)
      for (size_t g = 0; g < groups; g++) or_simd_two_to_one_x8(cur + 64 * g, next + 32 * g);
      first = 8 * groups;
    }
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (size_t i = first; i < n / 2; i++)
      or_two_to_one(cur + 8 * i, cur + 8 * i + 4, next + 4 * i);
    if (owned) free(owned);
    owned = next;
    cur = next;
    n /= 2;
  }
  memcpy(cap_out, cur, n * 32);
  if (owned) free(owned);
}

void or_merkle_tree(const uint64_t *leaves, size_t n_leaves, size_t leaf_len,
                    int cap_height, uint64_t *digests_out, uint64_t *cap_out) {
  rc_init();
  uint64_t *lvl = (uint64_t *)malloc(n_leaves * 32);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t i = 0; i < n_leaves; i++)
    or_hash_or_noop(leaves + i * leaf_len, leaf_len, lvl + 4 * i);
  merkle_from_leaf_digests(lvl, n_leaves, cap_height, digests_out, cap_out);
  free(lvl);
}

void or_merkle_tree_cols(const uint64_t *cols, size_t n_leaves, size_t leaf_len,
                         size_t col_stride, int cap_height, uint64_t *digests_out,
                         uint64_t *cap_out) {
  rc_init();
  uint64_t *lvl = (uint64_t *)malloc(n_leaves * 32);
  size_t first_scalar = 0;
  if (or_simd_poseidon_enabled() && leaf_len > 4) {  /* eight consecutive leaves per AVX-512 sponge: a column is contiguous */
    const size_t groups = n_leaves / 8;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (size_t g = 0; g < groups; g++) or_simd_leaf_hash_cols_x8(cols, leaf_len, col_stride, 8 * g, lvl + 32 * g);
    first_scalar = 8 * groups;
  }
#pragma omp parallel num_threads(g_threads)
  {
    uint64_t *row = (uint64_t *)malloc(leaf_len * sizeof(uint64_t));
#pragma omp for schedule(static)
    for (size_t i = first_scalar; i < n_leaves; i++) {
      for (size_t j = 0; j < leaf_len; j++) row[j] = cols[j * col_stride + i];
      or_hash_or_noop(row, leaf_len, lvl + 4 * i);
    }
    free(row);
  }
  merkle_from_leaf_digests(lvl, n_leaves, cap_height, digests_out, cap_out);
  free(lvl);
}

int or_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t index,
                     const uint64_t *siblings, size_t n_siblings, const uint64_t *cap,
                     int cap_height) {
  uint64_t cur[4];
  or_hash_or_noop(leaf, leaf_len, cur);
  for (size_t i = 0; i < n_siblings; i++) {
    uint64_t nxt[4];
    if ((index & 1) == 0) or_two_to_one(cur, siblings + 4 * i, nxt);
    else or_two_to_one(siblings + 4 * i, cur, nxt);
    memcpy(cur, nxt, 32);
    index >>= 1;
  }
  if (index >= ((size_t)1 << cap_height)) return 0;
  return memcmp(cur, cap + 4 * index, 32) == 0;
}

/* ------------------------------------------------------------------------- */
/* NTT. */

void or_bit_reverse(uint64_t *a, int log_n) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; i++) {
    size_t j = 0;
    for (int b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
    if (i < j) {
      uint64_t t = a[i];
      a[i] = a[j];
      a[j] = t;
    }
  }
}

static void ntt_core(uint64_t *a, int log_n, uint64_t omega) {
  size_t n = (size_t)1 << log_n;
  or_bit_reverse(a, log_n);
  uint64_t *tw = (uint64_t *)malloc((n / 2 + 1) * sizeof(uint64_t));
  tw[0] = 1;
  for (size_t i = 1; i < n / 2; i++) tw[i] = gl_mul(tw[i - 1], omega);
  for (int s = 1; s <= log_n; s++) {
    size_t m = (size_t)1 << s, half = m >> 1, step = n >> s;
    for (size_t k = 0; k < n; k += m)
      for (size_t j = 0; j < half; j++) {
        uint64_t t = gl_mul(tw[j * step], a[k + j + half]);
        uint64_t u = a[k + j];
        a[k + j] = gl_add(u, t);
        a[k + j + half] = gl_sub(u, t);
      }
  }
  free(tw);
}

void or_ntt(uint64_t *a, int log_n) {
  if (log_n == 0) return;
  ntt_core(a, log_n, gl_root_of_unity(log_n));
}

void or_intt(uint64_t *a, int log_n) {
  if (log_n == 0) return;
  size_t n = (size_t)1 << log_n;
  ntt_core(a, log_n, gl_inv(gl_root_of_unity(log_n)));
  uint64_t ninv = gl_inv((uint64_t)n % GL_P);
  for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
}

void or_dft_naive(const uint64_t *in, uint64_t *out, int log_n) {
  size_t n = (size_t)1 << log_n;
  uint64_t w = gl_root_of_unity(log_n);
  for (size_t i = 0; i < n; i++) {
    uint64_t x = gl_pow(w, i), acc = 0, xp = 1;
    for (size_t j = 0; j < n; j++) {
      acc = gl_add(acc, gl_mul(in[j], xp));
      xp = gl_mul(xp, x);
    }
    out[i] = acc;
  }
}

void or_coset_lde(const uint64_t *coeffs, int log_n, int rate_bits, uint64_t shift,
                  uint64_t *out) {
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
  memset(out, 0, N * sizeof(uint64_t));
  uint64_t sp = 1;
  for (size_t i = 0; i < n; i++) {
    out[i] = gl_mul(coeffs[i], sp);
    sp = gl_mul(sp, shift);
  }
  or_ntt(out, log_n + rate_bits);
}

void or_commit_batch(const uint64_t *values, size_t k, int log_n, int rate_bits,
                     int cap_height, uint64_t *coeffs_out, uint64_t *lde_out,
                     uint64_t *digests_out, uint64_t *cap_out) {
  rc_init();
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
  uint64_t *lde = lde_out ? lde_out : (uint64_t *)malloc(k * N * sizeof(uint64_t));
#pragma omp parallel num_threads(g_threads)
  {
    uint64_t *c = (uint64_t *)malloc(n * sizeof(uint64_t));
#pragma omp for schedule(dynamic)
    for (size_t p = 0; p < k; p++) {
      memcpy(c, values + p * n, n * sizeof(uint64_t));
      or_intt(c, log_n);
      if (coeffs_out) memcpy(coeffs_out + p * n, c, n * sizeof(uint64_t));
      or_coset_lde(c, log_n, rate_bits, GL_GENERATOR, lde + p * N);
      or_bit_reverse(lde + p * N, log_n + rate_bits);
    }
    free(c);
  }
  or_merkle_tree_cols(lde, N, k, N, cap_height, digests_out, cap_out);
  if (!lde_out) free(lde);
}
