/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see goldilocks.h).
 *
 * BLS12-381 G1 arithmetic and a naive multi-scalar multiplication, the checker for the MSM kernels of SURVEY.md
 * §8(a) A12 (Groth16 wrap proof: `gnark_plonky2_wrapper::wrap_plonky2_proof`, called at
 * city_rollup_circuit/src/worker/toolbox/root.rs:296-304; the arithmetic lives in gnark-crypto, a Go dependency that is
 * not in the tree). Restated from the published curve definition: y^2 = x^3 + 4 over F_p,
 *   p = (x-1)^2 (x^4 - x^2 + 1)/3 + x,  r = x^4 - x^2 + 1,  x = -0xd201000000010000,
 * with the standard generator. PARITY: the constants are pinned by those identities, by the curve equation on the
 * generator and by r*G = infinity (tests/test_oracle_bls.py); the reference holds no MSM vectors (SURVEY.md §8(c):
 * Groth16/MSM parity unpinned).
 *
 * Field elements cross the API as 6 little-endian u64 limbs of the canonical value; points as affine (x, y) = 12 limbs
 * plus an infinity flag; scalars as 4 little-endian u64 limbs (any value < 2^256, reduced implicitly by the group order).
 */
#include "cityoracle.h"

#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp_t;        /* Montgomery form, R = 2^384 */
typedef struct { fp_t x, y, z; } g1_t;         /* Jacobian, z = 0: infinity */

static const uint64_t P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                              0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t GX[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                               0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const uint64_t GY[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                               0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
static const uint64_t R_ORDER[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};

static uint64_t N0;      /* -p^-1 mod 2^64 */
static fp_t R1, R2;      /* R mod p, R^2 mod p */
static int ready = 0;

static int ge_p(const uint64_t a[6]) {
  for (int i = 5; i >= 0; i--) { if (a[i] > P[i]) return 1; if (a[i] < P[i]) return 0; }
  return 1;
}
static void sub_p(uint64_t a[6]) {
  u128 b = 0;
  for (int i = 0; i < 6; i++) { u128 d = (u128)a[i] - P[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
}
static fp_t fp_add(fp_t a, fp_t b) {
  fp_t r; u128 c = 0;
  for (int i = 0; i < 6; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  if (c || ge_p(r.l)) sub_p(r.l); /* a, b < p < 2^381: no carry out of 384 bits in practice */
  return r;
}
static fp_t fp_sub(fp_t a, fp_t b) {
  fp_t r; u128 br = 0;
  for (int i = 0; i < 6; i++) { u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)r.l[i] + P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
  return r;
}
/* Montgomery product a*b/R mod p (CIOS) */
static fp_t fp_mul(fp_t a, fp_t b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 6; i++) {
    u128 c = 0;
    for (int j = 0; j < 6; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * N0;
    c = ((u128)m * P[0] + t[0]) >> 64;
    for (int j = 1; j < 6; j++) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
  }
  fp_t r; memcpy(r.l, t, 48);
  if (t[6] || ge_p(r.l)) sub_p(r.l);
  return r;
}
static fp_t fp_sqr(fp_t a) { return fp_mul(a, a); }
static int fp_is_zero(fp_t a) { uint64_t o = 0; for (int i = 0; i < 6; i++) o |= a.l[i]; return o == 0; }
static int fp_eq(fp_t a, fp_t b) { return memcmp(a.l, b.l, 48) == 0; }

static void init(void) {
  if (ready) return;
  uint64_t inv = 1; /* Newton: inv = p^-1 mod 2^64 */
  for (int i = 0; i < 6; i++) inv *= 2 - P[0] * inv;
  N0 = (uint64_t)0 - inv;
  /* R mod p by doubling 1 exactly 384 times; R^2 by 384 more */
  fp_t one = {{1, 0, 0, 0, 0, 0}}, v = one;
  for (int i = 0; i < 768; i++) { v = fp_add(v, v); if (i == 383) R1 = v; }
  R2 = v;
  ready = 1;
}
static fp_t fp_from_canonical(const uint64_t a[6]) { fp_t t; memcpy(t.l, a, 48); return fp_mul(t, R2); }
static void fp_to_canonical(fp_t a, uint64_t out[6]) { fp_t one = {{1, 0, 0, 0, 0, 0}}; fp_t r = fp_mul(a, one); memcpy(out, r.l, 48); }
static fp_t fp_inv(fp_t a) { /* a^(p-2) */
  uint64_t e[6]; memcpy(e, P, 48); e[0] -= 2;
  fp_t r = R1;
  for (int i = 383; i >= 0; i--) { r = fp_sqr(r); if ((e[i / 64] >> (i % 64)) & 1) r = fp_mul(r, a); }
  return r;
}

/* ---- G1, Jacobian coordinates (x = X/Z^2, y = Y/Z^3) ---- */
static g1_t g1_inf(void) { g1_t r; memset(&r, 0, sizeof r); r.x = R1; r.y = R1; return r; }
static g1_t g1_double(g1_t p) { /* a = 0: dbl-2009-l */
  if (fp_is_zero(p.z)) return p;
  fp_t A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
  fp_t t = fp_add(p.x, B); t = fp_sqr(t); t = fp_sub(fp_sub(t, A), C);
  fp_t D = fp_add(t, t), E = fp_add(fp_add(A, A), A), F = fp_sqr(E);
  g1_t r;
  r.x = fp_sub(F, fp_add(D, D));
  fp_t C8 = fp_add(C, C); C8 = fp_add(C8, C8); C8 = fp_add(C8, C8);
  r.y = fp_sub(fp_mul(E, fp_sub(D, r.x)), C8);
  r.z = fp_mul(p.y, p.z); r.z = fp_add(r.z, r.z);
  return r;
}
static g1_t g1_add(g1_t p, g1_t q) {
  if (fp_is_zero(p.z)) return q;
  if (fp_is_zero(q.z)) return p;
  fp_t z1z1 = fp_sqr(p.z), z2z2 = fp_sqr(q.z);
  fp_t u1 = fp_mul(p.x, z2z2), u2 = fp_mul(q.x, z1z1);
  fp_t s1 = fp_mul(fp_mul(p.y, q.z), z2z2), s2 = fp_mul(fp_mul(q.y, p.z), z1z1);
  if (fp_eq(u1, u2)) return fp_eq(s1, s2) ? g1_double(p) : g1_inf();
  fp_t h = fp_sub(u2, u1), rr = fp_sub(s2, s1);
  fp_t hh = fp_sqr(h), hhh = fp_mul(h, hh), v = fp_mul(u1, hh);
  g1_t r;
  r.x = fp_sub(fp_sub(fp_sqr(rr), hhh), fp_add(v, v));
  r.y = fp_sub(fp_mul(rr, fp_sub(v, r.x)), fp_mul(s1, hhh));
  r.z = fp_mul(fp_mul(p.z, q.z), h);
  return r;
}
static g1_t g1_from_affine(const uint64_t xy[12], int inf) {
  if (inf) return g1_inf();
  g1_t r; r.x = fp_from_canonical(xy); r.y = fp_from_canonical(xy + 6); r.z = R1;
  return r;
}
static void g1_to_affine(g1_t p, uint64_t xy[12], int *inf) {
  if (fp_is_zero(p.z)) { memset(xy, 0, 96); *inf = 1; return; }
  fp_t zi = fp_inv(p.z), zi2 = fp_sqr(zi);
  fp_to_canonical(fp_mul(p.x, zi2), xy);
  fp_to_canonical(fp_mul(p.y, fp_mul(zi2, zi)), xy + 6);
  *inf = 0;
}
static g1_t g1_mul(g1_t p, const uint64_t k[4]) {
  g1_t r = g1_inf();
  for (int i = 255; i >= 0; i--) { r = g1_double(r); if ((k[i / 64] >> (i % 64)) & 1) r = g1_add(r, p); }
  return r;
}

/* ---- exported ---- */
void or_bls_constants(uint64_t p[6], uint64_t r[4], uint64_t gen_xy[12]) {
  memcpy(p, P, 48); memcpy(r, R_ORDER, 32); memcpy(gen_xy, GX, 48); memcpy(gen_xy + 6, GY, 48);
}
int or_bls_g1_on_curve(const uint64_t xy[12]) { /* canonical coordinates < p and y^2 == x^3 + 4 */
  init();
  if (ge_p(xy) || ge_p(xy + 6)) return 0;
  fp_t x = fp_from_canonical(xy), y = fp_from_canonical(xy + 6);
  const uint64_t four[6] = {4, 0, 0, 0, 0, 0};
  return fp_eq(fp_sqr(y), fp_add(fp_mul(fp_sqr(x), x), fp_from_canonical(four)));
}
void or_bls_g1_add(const uint64_t a_xy[12], int a_inf, const uint64_t b_xy[12], int b_inf, uint64_t out_xy[12], int *out_inf) {
  init();
  g1_to_affine(g1_add(g1_from_affine(a_xy, a_inf), g1_from_affine(b_xy, b_inf)), out_xy, out_inf);
}
void or_bls_g1_mul(const uint64_t xy[12], int inf, const uint64_t k[4], uint64_t out_xy[12], int *out_inf) {
  init();
  g1_to_affine(g1_mul(g1_from_affine(xy, inf), k), out_xy, out_inf);
}
/* sum_i scalars[i] * points[i], by definition (double-and-add per term); or_set_threads workers */
void or_bls_g1_msm(const uint64_t *scalars, const uint64_t *points_xy, const uint8_t *points_inf, size_t n,
                   uint64_t out_xy[12], int *out_inf) {
  init();
  int nt = or_get_threads();
  if (nt < 1) nt = 1;
  g1_t *part = (g1_t *)__builtin_alloca(sizeof(g1_t) * (size_t)nt);
  for (int t = 0; t < nt; t++) part[t] = g1_inf();
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int t = 0; t < nt; t++) {
    g1_t acc = g1_inf();
    for (size_t i = n * (size_t)t / (size_t)nt; i < n * (size_t)(t + 1) / (size_t)nt; i++)
      acc = g1_add(acc, g1_mul(g1_from_affine(points_xy + 12 * i, points_inf ? points_inf[i] : 0), scalars + 4 * i));
    part[t] = acc;
  }
  g1_t acc = g1_inf();
  for (int t = 0; t < nt; t++) acc = g1_add(acc, part[t]);
  g1_to_affine(acc, out_xy, out_inf);
}

/* ================================================================================================================
 * G2: the twist y^2 = x^3 + 4(1 + u) over F_p^2 = F_p[u]/(u^2 + 1). Points cross the API as 24 limbs:
 * x.c0, x.c1, y.c0, y.c1 (value c0 + c1*u), 6 little-endian u64 each.
 * ================================================================================================================ */
typedef struct { fp_t c0, c1; } fp2_t;
typedef struct { fp2_t x, y, z; } g2_t;

static const uint64_t G2X0[6] = {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL};
static const uint64_t G2X1[6] = {0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL};
static const uint64_t G2Y0[6] = {0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL};
static const uint64_t G2Y1[6] = {0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};

static fp2_t fp2_add(fp2_t a, fp2_t b) { fp2_t r = {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; return r; }
static fp2_t fp2_sub(fp2_t a, fp2_t b) { fp2_t r = {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; return r; }
static fp2_t fp2_mul(fp2_t a, fp2_t b) { /* schoolbook, u^2 = -1 */
  fp2_t r = {fp_sub(fp_mul(a.c0, b.c0), fp_mul(a.c1, b.c1)), fp_add(fp_mul(a.c0, b.c1), fp_mul(a.c1, b.c0))};
  return r;
}
static fp2_t fp2_sqr(fp2_t a) { return fp2_mul(a, a); }
static int fp2_is_zero(fp2_t a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
static int fp2_eq(fp2_t a, fp2_t b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
static fp2_t fp2_inv(fp2_t a) { /* conj(a) / (c0^2 + c1^2) */
  fp_t d = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  fp_t z = {{0, 0, 0, 0, 0, 0}};
  fp2_t r = {fp_mul(a.c0, d), fp_mul(fp_sub(z, a.c1), d)};
  return r;
}
static fp2_t fp2_one(void) { fp_t z = {{0, 0, 0, 0, 0, 0}}; fp2_t r = {R1, z}; return r; }

static g2_t g2_inf(void) { g2_t r; memset(&r, 0, sizeof r); r.x = fp2_one(); r.y = fp2_one(); return r; }
static g2_t g2_double(g2_t p) {
  if (fp2_is_zero(p.z)) return p;
  fp2_t A = fp2_sqr(p.x), B = fp2_sqr(p.y), C = fp2_sqr(B);
  fp2_t t = fp2_sqr(fp2_add(p.x, B)); t = fp2_sub(fp2_sub(t, A), C);
  fp2_t D = fp2_add(t, t), E = fp2_add(fp2_add(A, A), A), F = fp2_sqr(E);
  g2_t r;
  r.x = fp2_sub(F, fp2_add(D, D));
  fp2_t C8 = fp2_add(C, C); C8 = fp2_add(C8, C8); C8 = fp2_add(C8, C8);
  r.y = fp2_sub(fp2_mul(E, fp2_sub(D, r.x)), C8);
  r.z = fp2_mul(p.y, p.z); r.z = fp2_add(r.z, r.z);
  return r;
}
static g2_t g2_add(g2_t p, g2_t q) {
  if (fp2_is_zero(p.z)) return q;
  if (fp2_is_zero(q.z)) return p;
  fp2_t z1z1 = fp2_sqr(p.z), z2z2 = fp2_sqr(q.z);
  fp2_t u1 = fp2_mul(p.x, z2z2), u2 = fp2_mul(q.x, z1z1);
  fp2_t s1 = fp2_mul(fp2_mul(p.y, q.z), z2z2), s2 = fp2_mul(fp2_mul(q.y, p.z), z1z1);
  if (fp2_eq(u1, u2)) return fp2_eq(s1, s2) ? g2_double(p) : g2_inf();
  fp2_t h = fp2_sub(u2, u1), rr = fp2_sub(s2, s1);
  fp2_t hh = fp2_sqr(h), hhh = fp2_mul(h, hh), v = fp2_mul(u1, hh);
  g2_t r;
  r.x = fp2_sub(fp2_sub(fp2_sqr(rr), hhh), fp2_add(v, v));
  r.y = fp2_sub(fp2_mul(rr, fp2_sub(v, r.x)), fp2_mul(s1, hhh));
  r.z = fp2_mul(fp2_mul(p.z, q.z), h);
  return r;
}
static g2_t g2_from_affine(const uint64_t xy[24], int inf) {
  if (inf) return g2_inf();
  g2_t r;
  r.x.c0 = fp_from_canonical(xy); r.x.c1 = fp_from_canonical(xy + 6);
  r.y.c0 = fp_from_canonical(xy + 12); r.y.c1 = fp_from_canonical(xy + 18);
  r.z = fp2_one();
  return r;
}
static void g2_to_affine(g2_t p, uint64_t xy[24], int *inf) {
  if (fp2_is_zero(p.z)) { memset(xy, 0, 192); *inf = 1; return; }
  fp2_t zi = fp2_inv(p.z), zi2 = fp2_sqr(zi);
  fp2_t x = fp2_mul(p.x, zi2), y = fp2_mul(p.y, fp2_mul(zi2, zi));
  fp_to_canonical(x.c0, xy); fp_to_canonical(x.c1, xy + 6);
  fp_to_canonical(y.c0, xy + 12); fp_to_canonical(y.c1, xy + 18);
  *inf = 0;
}
static g2_t g2_mul(g2_t p, const uint64_t k[4]) {
  g2_t r = g2_inf();
  for (int i = 255; i >= 0; i--) { r = g2_double(r); if ((k[i / 64] >> (i % 64)) & 1) r = g2_add(r, p); }
  return r;
}

void or_bls_g2_generator(uint64_t xy[24]) { memcpy(xy, G2X0, 48); memcpy(xy + 6, G2X1, 48); memcpy(xy + 12, G2Y0, 48); memcpy(xy + 18, G2Y1, 48); }
int or_bls_g2_on_curve(const uint64_t xy[24]) {
  init();
  for (int k = 0; k < 4; k++) if (ge_p(xy + 6 * k)) return 0;
  g2_t p = g2_from_affine(xy, 0);
  const uint64_t four[6] = {4, 0, 0, 0, 0, 0};
  fp2_t b = {fp_from_canonical(four), fp_from_canonical(four)}; /* 4 (1 + u) */
  return fp2_eq(fp2_sqr(p.y), fp2_add(fp2_mul(fp2_sqr(p.x), p.x), b));
}
void or_bls_g2_add(const uint64_t a_xy[24], int a_inf, const uint64_t b_xy[24], int b_inf, uint64_t out_xy[24], int *out_inf) {
  init();
  g2_to_affine(g2_add(g2_from_affine(a_xy, a_inf), g2_from_affine(b_xy, b_inf)), out_xy, out_inf);
}
void or_bls_g2_mul(const uint64_t xy[24], int inf, const uint64_t k[4], uint64_t out_xy[24], int *out_inf) {
  init();
  g2_to_affine(g2_mul(g2_from_affine(xy, inf), k), out_xy, out_inf);
}
void or_bls_g2_msm(const uint64_t *scalars, const uint64_t *points_xy, const uint8_t *points_inf, size_t n,
                   uint64_t out_xy[24], int *out_inf) {
  init();
  int nt = or_get_threads();
  if (nt < 1) nt = 1;
  g2_t *part = (g2_t *)__builtin_alloca(sizeof(g2_t) * (size_t)nt);
  for (int t = 0; t < nt; t++) part[t] = g2_inf();
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int t = 0; t < nt; t++) {
    g2_t acc = g2_inf();
    for (size_t i = n * (size_t)t / (size_t)nt; i < n * (size_t)(t + 1) / (size_t)nt; i++)
      acc = g2_add(acc, g2_mul(g2_from_affine(points_xy + 24 * i, points_inf ? points_inf[i] : 0), scalars + 4 * i));
    part[t] = acc;
  }
  g2_t acc = g2_inf();
  for (int t = 0; t < nt; t++) acc = g2_add(acc, part[t]);
  g2_to_affine(acc, out_xy, out_inf);
}

/* ================================================================================================================
 * Scalar field F_r (r = the group order above, 255 bits, 2-adicity 32, multiplicative generator 7) and its NTT: the
 * transforms of Groth16's quotient computation. Elements cross the API as 4 little-endian u64 of the canonical value.
 * omega_n = 7^((r-1)/n).
 * ================================================================================================================ */
typedef struct { uint64_t l[4]; } fr_t; /* Montgomery, R = 2^256 */
static uint64_t FR_N0;
static fr_t FR_R1, FR_R2;
static int fr_ready = 0;

static int fr_ge(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > R_ORDER[i]) return 1; if (a[i] < R_ORDER[i]) return 0; }
  return 1;
}
static void fr_subm(uint64_t a[4]) {
  u128 b = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - R_ORDER[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
}
static fr_t fr_add(fr_t a, fr_t b) {
  fr_t r; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  if (c || fr_ge(r.l)) fr_subm(r.l);
  return r;
}
static fr_t fr_sub(fr_t a, fr_t b) {
  fr_t r; u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + R_ORDER[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
  return r;
}
static fr_t fr_mul(fr_t a, fr_t b) {
  uint64_t t[6] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * FR_N0;
    c = ((u128)m * R_ORDER[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * R_ORDER[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fr_t r; memcpy(r.l, t, 32);
  if (t[4] || fr_ge(r.l)) fr_subm(r.l);
  return r;
}
static void fr_init(void) {
  if (fr_ready) return;
  uint64_t inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - R_ORDER[0] * inv;
  FR_N0 = (uint64_t)0 - inv;
  fr_t one = {{1, 0, 0, 0}}, v = one;
  for (int i = 0; i < 512; i++) { v = fr_add(v, v); if (i == 255) FR_R1 = v; }
  FR_R2 = v;
  fr_ready = 1;
}
static fr_t fr_from(const uint64_t a[4]) { fr_t t; memcpy(t.l, a, 32); return fr_mul(t, FR_R2); }
static void fr_to(fr_t a, uint64_t out[4]) { fr_t one = {{1, 0, 0, 0}}; fr_t r = fr_mul(a, one); memcpy(out, r.l, 32); }
static fr_t fr_pow(fr_t a, const uint64_t e[4]) {
  fr_t r = FR_R1;
  for (int i = 255; i >= 0; i--) { r = fr_mul(r, r); if ((e[i / 64] >> (i % 64)) & 1) r = fr_mul(r, a); }
  return r;
}
static fr_t fr_inv(fr_t a) { uint64_t e[4]; memcpy(e, R_ORDER, 32); e[0] -= 2; return fr_pow(a, e); }
static fr_t fr_root(int log_n) { /* 7^((r-1) / 2^log_n) */
  uint64_t e[4]; memcpy(e, R_ORDER, 32); e[0] -= 1;
  for (int s = 0; s < log_n; s++) { for (int i = 0; i < 3; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 63); e[3] >>= 1; }
  const uint64_t seven[4] = {7, 0, 0, 0};
  return fr_pow(fr_from(seven), e);
}

int or_fr_is_canonical(const uint64_t a[4]) { return !fr_ge(a); }
void or_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { fr_init(); fr_to(fr_mul(fr_from(a), fr_from(b)), out); }
void or_fr_root_of_unity(int log_n, uint64_t out[4]) { fr_init(); fr_to(fr_root(log_n), out); }
/* O(n^2) definition: out[k] = sum_j in[j] * omega_n^(jk) */
void or_fr_dft_naive(const uint64_t *in, uint64_t *out, int log_n) {
  fr_init();
  size_t n = (size_t)1 << log_n;
  fr_t w = fr_root(log_n), wk = FR_R1;
  for (size_t k = 0; k < n; k++) {
    fr_t acc = {{0, 0, 0, 0}}, x = FR_R1;
    for (size_t j = 0; j < n; j++) { acc = fr_add(acc, fr_mul(fr_from(in + 4 * j), x)); x = fr_mul(x, wk); }
    fr_to(acc, out + 4 * k);
    wk = fr_mul(wk, w);
  }
}
/* in-place NTT, natural order in and out. inverse = 0: evaluations on shift*<omega_n> (shift NULL: the subgroup itself);
 * inverse = 1: the exact inverse of that (including the 1/n and the shift^-i). */
void or_fr_ntt(uint64_t *data, int log_n, int inverse, const uint64_t *shift) {
  fr_init();
  size_t n = (size_t)1 << log_n;
  fr_t *a = (fr_t *)__builtin_malloc(n * sizeof(fr_t));
  for (size_t i = 0; i < n; i++) a[i] = fr_from(data + 4 * i);
  if (!inverse && shift) { fr_t s = fr_from(shift), p = FR_R1; for (size_t i = 0; i < n; i++) { a[i] = fr_mul(a[i], p); p = fr_mul(p, s); } }
  /* bit reversal, then decimation-in-time butterflies */
  for (size_t i = 0; i < n; i++) {
    size_t j = 0;
    for (int b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
    if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (int s = 1; s <= log_n; s++) {
    fr_t wm = fr_root(s);
    if (inverse) wm = fr_inv(wm);
    size_t m = (size_t)1 << s;
    for (size_t k = 0; k < n; k += m) {
      fr_t w = FR_R1;
      for (size_t j = 0; j < m / 2; j++) {
        fr_t t = fr_mul(w, a[k + j + m / 2]), u = a[k + j];
        a[k + j] = fr_add(u, t);
        a[k + j + m / 2] = fr_sub(u, t);
        w = fr_mul(w, wm);
      }
    }
  }
  if (inverse) {
    uint64_t nn[4] = {(uint64_t)n, 0, 0, 0};
    fr_t ninv = fr_inv(fr_from(nn));
    for (size_t i = 0; i < n; i++) a[i] = fr_mul(a[i], ninv);
    if (shift) { fr_t si = fr_inv(fr_from(shift)), p = FR_R1; for (size_t i = 0; i < n; i++) { a[i] = fr_mul(a[i], p); p = fr_mul(p, si); } }
  }
  for (size_t i = 0; i < n; i++) fr_to(a[i], data + 4 * i);
  __builtin_free(a);
}

/* Groth16 quotient, as gnark's computeH does it (UPSTREAM-MEMORY: gnark backend/groth16/bls12-381/prove.go is not in
 * /root/reference; call site city_rollup_circuit/src/worker/toolbox/root.rs:296-304): a, b, c = evaluations of
 * (A w), (B w), (C w) on <omega_n>; h = coset_iNTT((coset_NTT(iNTT a) o coset_NTT(iNTT b) - coset_NTT(iNTT c)) / (g^n - 1)),
 * coset generator g = 7. h overwrites a; b and c are left transformed. Parity unpinned by the reference. */
void or_groth16_quotient(uint64_t *a, uint64_t *b, uint64_t *c, int log_n) {
  fr_init();
  const size_t n = (size_t)1 << log_n;
  const uint64_t g[4] = {7, 0, 0, 0};
  uint64_t *v[3] = {a, b, c};
  for (int k = 0; k < 3; k++) { or_fr_ntt(v[k], log_n, 1, NULL); or_fr_ntt(v[k], log_n, 0, g); }
  uint64_t e[4] = {(uint64_t)n, 0, 0, 0};
  fr_t den = fr_inv(fr_sub(fr_pow(fr_from(g), e), FR_R1));
  for (size_t i = 0; i < n; i++)
    fr_to(fr_mul(fr_sub(fr_mul(fr_from(a + 4 * i), fr_from(b + 4 * i)), fr_from(c + 4 * i)), den), a + 4 * i);
  or_fr_ntt(a, log_n, 1, g);
}
