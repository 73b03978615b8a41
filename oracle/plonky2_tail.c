/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see goldilocks.h).
 *
 * CPU restatement of the part of plonky2 0.2.2 `prove_with_partition_witness` that follows the
 * computation of the polynomials: transcript (Challenger), the three oracle commitments, openings at
 * zeta / g*zeta, the FRI opening proof (batch polynomial, commit phase, proof of work, query rounds),
 * bincode serialisation of ProofWithPublicInputs, and the matching verifier side for the FRI part.
 * plonky2 is an un-vendored dependency (QEDProtocol/plonky2-hwa @ 6a8ca008, Cargo.lock:4174-4223):
 * the algorithm is restated from its published definition (SURVEY.md §3.3 steps 3-4, 6, 8-11,
 * Appendix B). What the reference tree pins, on ALL TEN reference proofs of qbench_data/example.bin
 * (tests/golden/qbench_example.bin): the proof shape and byte layout, every Merkle path, the FRI leaf
 * layout, the FRI fold / final-polynomial relation (tests/test_oracle_fri_reference.py) and
 * fri_combine_initial — i.e. the order of the opened polynomials inside each batch, the alpha-power /
 * shift convention of ReducingFactor and the g*zeta point (tests/test_oracle_fri_combine_reference.py,
 * challenges recovered from the proof bytes by algebra). NOT pinned by reference data: the transcript
 * order (which values are observed when; no circuit digest comes with the fixture proofs) — "parity
 * unpinned" for that fact.
 *
 * The FRI part is generic (or_fri_prove / or_fri_verify over any oracles and opening batches, plonky2
 * `PolynomialBatch::prove_openings` / `verify_fri_proof`): the whole-proof functions below are one
 * client, the STARK-shaped tests (tests/test_gpu_fri_generic.py) another.
 */
#define _POSIX_C_SOURCE 200809L
#include "cityoracle.h"
#include "goldilocks.h"

#include <stdio.h>
#include <time.h>

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Challenger: Poseidon duplex, overwrite mode, rate 8; challenges are popped from the END of the
 * squeezed rate block. */

void or_ch_init(or_challenger *c) { memset(c, 0, sizeof *c); }

/* plonky2: `response.to_canonical_u64().leading_zeros() >= proof_of_work_bits` (0 bits: always true) */
static int pow_ok(uint64_t response, int pow_bits) { return pow_bits <= 0 || (response >> (64 - pow_bits)) == 0; }

static void ch_duplex(or_challenger *c) {
  for (int i = 0; i < c->n_in; i++) c->state[i] = c->in[i];
  c->n_in = 0;
  or_poseidon_permute(c->state);
  memcpy(c->out, c->state, 8 * sizeof(uint64_t));
  c->n_out = 8;
}
void or_ch_observe(or_challenger *c, const uint64_t *e, size_t n) {
  for (size_t i = 0; i < n; i++) {
    c->n_out = 0;
    c->in[c->n_in++] = e[i];
    if (c->n_in == 8) ch_duplex(c);
  }
}
uint64_t or_ch_challenge(or_challenger *c) {
  if (c->n_in > 0 || c->n_out == 0) ch_duplex(c);
  return c->out[--c->n_out];
}
static gl2_t ch_ext(or_challenger *c) {
  uint64_t a = or_ch_challenge(c), b = or_ch_challenge(c);
  return gl2_make(a, b);
}

/* ------------------------------------------------------------------------------------------ */
/* byte buffer (bincode 1.3 default: LE, u64 length prefixes) */

typedef struct { uint8_t *p; size_t len, cap; } buf_t;
static void buf_put(buf_t *b, const void *src, size_t n) {
  if (b->len + n > b->cap) { b->cap = (b->len + n) * 2 + 1024; b->p = (uint8_t *)realloc(b->p, b->cap); }
  memcpy(b->p + b->len, src, n); b->len += n;
}
static void buf_u64(buf_t *b, uint64_t v) { buf_put(b, &v, 8); }
static void buf_felts(buf_t *b, const uint64_t *v, size_t n) { buf_put(b, v, 8 * n); }

/* ------------------------------------------------------------------------------------------ */
/* polynomial batches */

typedef struct or_batch {
  size_t k;      /* polynomials */
  size_t n_salt; /* extra leaf elements after the k polynomial values (zero-knowledge: OR_SALT_SIZE, else 0) */
  int log_n, rate_bits, cap_height;
  uint64_t *coeffs;  /* k x n */
  uint64_t *lde;     /* (k + n_salt) x N, bit-reversed index order; the salt columns come last */
  uint64_t *digests; /* levels below the cap */
  uint64_t *cap;     /* 2^cap_height x 4 */
} batch_t;

/* FriParams without the circuit around it */
typedef struct { int db, rb, ch, pow_bits, nq, n_arity; int arity_bits[8]; } fri_cfg_t;
static int fri_prove_core(const batch_t *const *B, size_t n_oracles, const or_fri_batch *batches, size_t n_batches, const fri_cfg_t *cfg,
                          or_challenger *c, int use_pow_override, uint64_t pow_override, buf_t *out, or_tail_debug *dbg);

static size_t digest_nodes(size_t n_leaves, int cap_height) {
  size_t cap_n = (size_t)1 << cap_height;
  return n_leaves > cap_n ? 2 * n_leaves - 2 * cap_n : 0;
}

static void batch_alloc(batch_t *b, size_t k, int log_n, int rate_bits, int cap_height, const uint64_t *salt) {
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
  b->k = k; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
  b->n_salt = salt ? OR_SALT_SIZE : 0;
  b->coeffs = (uint64_t *)malloc(k * n * 8);
  b->lde = (uint64_t *)malloc((k + b->n_salt) * N * 8);
  if (salt) memcpy(b->lde + k * N, salt, b->n_salt * N * 8);
  b->digests = (uint64_t *)malloc((digest_nodes(N, cap_height) + 1) * 32);
  b->cap = (uint64_t *)malloc(((size_t)1 << cap_height) * 32);
}
static void batch_free(batch_t *b) { free(b->coeffs); free(b->lde); free(b->digests); free(b->cap); }

static void batch_from_values(batch_t *b, const uint64_t *values, size_t k, int log_n, int rate_bits, int cap_height,
                              const uint64_t *salt) {
  batch_alloc(b, k, log_n, rate_bits, cap_height, salt);
  or_commit_batch(values, k, log_n, rate_bits, cap_height, b->coeffs, b->lde, b->digests, b->cap);
  if (salt) /* the leaves are longer than what or_commit_batch hashed: redo the tree over k + n_salt columns */
    or_merkle_tree_cols(b->lde, (size_t)1 << (log_n + rate_bits), k + b->n_salt, (size_t)1 << (log_n + rate_bits), cap_height,
                        b->digests, b->cap);
}
static void batch_from_coeffs(batch_t *b, const uint64_t *coeffs, size_t k, int log_n, int rate_bits, int cap_height,
                              const uint64_t *salt) {
  batch_alloc(b, k, log_n, rate_bits, cap_height, salt);
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
  memcpy(b->coeffs, coeffs, k * n * 8);
  for (size_t p = 0; p < k; p++) {
    or_coset_lde(coeffs + p * n, log_n, rate_bits, GL_GENERATOR, b->lde + p * N);
    or_bit_reverse(b->lde + p * N, log_n + rate_bits);
  }
  or_merkle_tree_cols(b->lde, N, k + b->n_salt, N, cap_height, b->digests, b->cap);
}

/* Merkle path of leaf `idx` out of the level-by-level digest array */
static void merkle_path(const uint64_t *digests, size_t n_leaves, int cap_height, size_t idx, uint64_t *siblings /* depth x 4 */) {
  size_t cap_n = (size_t)1 << cap_height, n = n_leaves, off = 0;
  int lvl = 0;
  while (n > cap_n) {
    memcpy(siblings + 4 * lvl, digests + 4 * (off + (idx ^ 1)), 32);
    off += n; n >>= 1; idx >>= 1; lvl++;
  }
}
static int log2z(size_t x) { int l = 0; while (((size_t)1 << l) < x) l++; return l; }

/* ------------------------------------------------------------------------------------------ */
/* polynomial helpers */

static gl2_t eval_base_poly_ext(const uint64_t *c, size_t n, gl2_t z) {
  gl2_t acc = gl2_from_base(0);
  for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_from_base(c[i]));
  return acc;
}
static gl2_t eval_ext_poly(const gl2_t *c, size_t n, gl2_t z) {
  gl2_t acc = gl2_from_base(0);
  for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), c[i]);
  return acc;
}
/* coset NTT of an extension polynomial: component-wise (the transform is F_p-linear) */
static void ext_coset_ntt(const gl2_t *coeffs, int log_n, uint64_t shift, gl2_t *out) {
  size_t n = (size_t)1 << log_n;
  uint64_t *a = (uint64_t *)malloc(n * 8), *b = (uint64_t *)malloc(n * 8), *oa = (uint64_t *)malloc(n * 8), *ob = (uint64_t *)malloc(n * 8);
  for (size_t i = 0; i < n; i++) { a[i] = coeffs[i].c[0]; b[i] = coeffs[i].c[1]; }
  or_coset_lde(a, log_n, 0, shift, oa);
  or_coset_lde(b, log_n, 0, shift, ob);
  for (size_t i = 0; i < n; i++) out[i] = gl2_make(oa[i], ob[i]);
  free(a); free(b); free(oa); free(ob);
}
static size_t bitrev_sz(size_t x, int bits) { size_t r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

/* ------------------------------------------------------------------------------------------ */
/* FRI prover */

typedef struct {
  int n_layers;
  uint64_t *leaves[8];  /* n_leaves x (2*arity), row-major */
  uint64_t *digests[8];
  uint64_t *cap[8];
  size_t n_leaves[8];
} fri_trees_t;

static int prove_impl(const or_shape *sh, const or_gates *G, const uint64_t circuit_digest[4], const uint64_t *public_inputs, size_t n_pi,
                      const uint64_t *cs_values, const uint64_t *wires_values, const uint64_t *zs_pp_values,
                      const uint64_t *quotient_coeffs, const uint64_t *salts, int use_pow_override, uint64_t pow_override,
                      uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg);

int or_prove_tail(const or_shape *sh, const uint64_t circuit_digest[4], const uint64_t *public_inputs, size_t n_pi,
                  const uint64_t *cs_values, const uint64_t *wires_values, const uint64_t *zs_pp_values,
                  const uint64_t *quotient_coeffs, int use_pow_override, uint64_t pow_override,
                  uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg) {
  return prove_impl(sh, NULL, circuit_digest, public_inputs, n_pi, cs_values, wires_values, zs_pp_values, quotient_coeffs,
                    NULL, use_pow_override, pow_override, proof_out, proof_len, dbg);
}

/* The whole of CircuitData::prove after witness generation: wires -> proof. Z / partial products (A7)
 * and the quotient chunks (A8) are computed here from the transcript challenges and the gate set G. */
int or_prove_full(const or_shape *sh, const or_gates *G, const uint64_t circuit_digest[4], const uint64_t *public_inputs,
                  size_t n_pi, const uint64_t *cs_values, const uint64_t *wires_values, int use_pow_override,
                  uint64_t pow_override, uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg) {
  if (!G) return -100;
  return prove_impl(sh, G, circuit_digest, public_inputs, n_pi, cs_values, wires_values, NULL, NULL, NULL, use_pow_override,
                    pow_override, proof_out, proof_len, dbg);
}

int or_prove_full_zk(const or_shape *sh, const or_gates *G, const uint64_t circuit_digest[4], const uint64_t *public_inputs,
                     size_t n_pi, const uint64_t *cs_values, const uint64_t *wires_values, const uint64_t *salts,
                     int use_pow_override, uint64_t pow_override, uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg) {
  if (!G || !salts) return -100;
  return prove_impl(sh, G, circuit_digest, public_inputs, n_pi, cs_values, wires_values, NULL, NULL, salts, use_pow_override,
                    pow_override, proof_out, proof_len, dbg);
}

/* OR_TIMING=1 in the environment prints the wall time of each phase to stderr (bench diagnostics only) */
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define PHASE(name) do { if (timing) { double t_ = now_s(); fprintf(stderr, "[oracle] %-18s %.3f s\n", phase_name, t_ - phase_t0); phase_t0 = t_; } phase_name = name; } while (0)

static int prove_impl(const or_shape *sh, const or_gates *G, const uint64_t circuit_digest[4], const uint64_t *public_inputs, size_t n_pi,
                      const uint64_t *cs_values, const uint64_t *wires_values, const uint64_t *zs_pp_values,
                      const uint64_t *quotient_coeffs, const uint64_t *salts, int use_pow_override, uint64_t pow_override,
                      uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg) {
  if ((sh->zero_knowledge != 0) != (salts != NULL)) return -3; /* salts exactly when the circuit is zero-knowledge */
  const int timing = getenv("OR_TIMING") != NULL;
  double phase_t0 = now_s();
  const char *phase_name = "commit cs+wires";
  const int db = sh->degree_bits, rb = sh->rate_bits, ch = sh->cap_height;
  const size_t n = (size_t)1 << db, N = n << rb;
  const size_t k_cs = sh->num_constants + sh->num_routed_wires, k_w = sh->num_wires;
  const size_t k_z = (size_t)sh->num_challenges * (1 + sh->num_partial_products);
  const size_t k_q = (size_t)sh->num_challenges * sh->quotient_degree_factor;
  const size_t cap_n = (size_t)1 << ch;

  batch_t B[4];
  const uint64_t *salt1 = salts, *salt2 = salts ? salts + (size_t)OR_SALT_SIZE * N : NULL, *salt3 = salts ? salts + (size_t)2 * OR_SALT_SIZE * N : NULL;
  batch_from_values(&B[0], cs_values, k_cs, db, rb, ch, NULL);
  uint64_t pi_hash[4];
  or_hash_no_pad(public_inputs, n_pi, pi_hash);
  batch_from_values(&B[1], wires_values, k_w, db, rb, ch, salt1);

  or_challenger c;
  or_ch_init(&c);
  or_ch_observe(&c, circuit_digest, 4);
  or_ch_observe(&c, pi_hash, 4);
  or_ch_observe(&c, B[1].cap, cap_n * 4);
  uint64_t betas[8], gammas[8], alphas[8];
  for (int i = 0; i < sh->num_challenges; i++) betas[i] = or_ch_challenge(&c);
  for (int i = 0; i < sh->num_challenges; i++) gammas[i] = or_ch_challenge(&c);
  uint64_t *own_zs = NULL, *own_q = NULL;
  PHASE("zs");
  if (G) { /* A7 */
    own_zs = (uint64_t *)malloc(k_z * n * 8);
    or_zs_partial_products(sh, wires_values, cs_values + (size_t)sh->num_constants * n, G->k_is, betas, gammas, own_zs);
    zs_pp_values = own_zs;
  }
  PHASE("commit zs");
  batch_from_values(&B[2], zs_pp_values, k_z, db, rb, ch, salt2);
  or_ch_observe(&c, B[2].cap, cap_n * 4);
  for (int i = 0; i < sh->num_challenges; i++) alphas[i] = or_ch_challenge(&c);
  PHASE("quotient");
  if (G) { /* A8 */
    own_q = (uint64_t *)malloc(k_q * n * 8);
    int qrc = or_quotient_polys(sh, G, pi_hash, B[0].lde, B[1].lde, B[2].lde, betas, gammas, alphas, own_q);
    if (qrc) return -200 + qrc;
    quotient_coeffs = own_q;
  }
  PHASE("commit quotient");
  batch_from_coeffs(&B[3], quotient_coeffs, k_q, db, rb, ch, salt3);
  free(own_zs);
  free(own_q);
  or_ch_observe(&c, B[3].cap, cap_n * 4);
  gl2_t zeta = ch_ext(&c);
  uint64_t g = gl_root_of_unity(db);
  gl2_t zeta_next = gl2_scale(zeta, g);
  if (dbg) {
    memcpy(dbg->betas, betas, sizeof betas); memcpy(dbg->gammas, gammas, sizeof gammas); memcpy(dbg->alphas, alphas, sizeof alphas);
    dbg->zeta[0] = zeta.c[0]; dbg->zeta[1] = zeta.c[1];
  }

  PHASE("openings");
  /* openings: every polynomial at zeta, the Z polynomials also at g*zeta */
  size_t k_all = k_cs + k_w + k_z + k_q;
  gl2_t *open = (gl2_t *)malloc(k_all * sizeof(gl2_t));
  gl2_t *open_next = (gl2_t *)malloc(sh->num_challenges * sizeof(gl2_t));
  {
    size_t o = 0;
    for (int b = 0; b < 4; b++)
      for (size_t p = 0; p < B[b].k; p++) open[o++] = eval_base_poly_ext(B[b].coeffs + p * n, n, zeta);
    for (int i = 0; i < sh->num_challenges; i++) open_next[i] = eval_base_poly_ext(B[2].coeffs + (size_t)i * n, n, zeta_next);
  }
  /* observe_openings: batch zeta = [constants, sigmas, wires, zs, partial products, quotient], batch g*zeta = [zs_next] */
  for (size_t i = 0; i < k_all; i++) or_ch_observe(&c, open[i].c, 2);
  for (int i = 0; i < sh->num_challenges; i++) or_ch_observe(&c, open_next[i].c, 2);

  PHASE("fri");
  /* ---- the FRI opening proof: plonky2 `fri_instance` = everything at zeta, the Z polynomials at g*zeta ---- */
  buf_t fri = {0};
  {
    const batch_t *oracles[4] = {&B[0], &B[1], &B[2], &B[3]};
    or_fri_range r0[4] = {{0, 0, (uint32_t)k_cs}, {1, 0, (uint32_t)k_w}, {2, 0, (uint32_t)k_z}, {3, 0, (uint32_t)k_q}};
    or_fri_range r1[1] = {{2, 0, (uint32_t)sh->num_challenges}};
    or_fri_batch fb[2] = {{{zeta.c[0], zeta.c[1]}, r0, 4}, {{zeta_next.c[0], zeta_next.c[1]}, r1, 1}};
    fri_cfg_t cfg = {db, rb, ch, sh->pow_bits, sh->num_query_rounds, sh->n_arity, {0}};
    for (int l = 0; l < 8; l++) cfg.arity_bits[l] = sh->arity_bits[l];
    int frc = fri_prove_core(oracles, 4, fb, 2, &cfg, &c, use_pow_override, pow_override, &fri, dbg);
    if (frc) return frc;
  }

  PHASE("serialise");
  /* ---- serialise ---- */
  buf_t out = {0};
  for (int b = 1; b <= 3; b++) { buf_u64(&out, cap_n); buf_felts(&out, B[b].cap, cap_n * 4); }
  {
    size_t o = 0;
    /* constants, plonk_sigmas */
    buf_u64(&out, sh->num_constants); buf_put(&out, open + o, (size_t)sh->num_constants * 16); o += sh->num_constants;
    buf_u64(&out, sh->num_routed_wires); buf_put(&out, open + o, (size_t)sh->num_routed_wires * 16); o += sh->num_routed_wires;
    buf_u64(&out, k_w); buf_put(&out, open + o, k_w * 16); o += k_w;
    buf_u64(&out, sh->num_challenges); buf_put(&out, open + o, (size_t)sh->num_challenges * 16);
    buf_u64(&out, sh->num_challenges); buf_put(&out, open_next, (size_t)sh->num_challenges * 16);
    size_t npp = k_z - sh->num_challenges;
    buf_u64(&out, npp); buf_put(&out, open + o + sh->num_challenges, npp * 16); o += k_z;
    buf_u64(&out, k_q); buf_put(&out, open + o, k_q * 16);
    buf_u64(&out, 0); buf_u64(&out, 0); /* lookup_zs, lookup_zs_next */
  }
  buf_put(&out, fri.p, fri.len); /* FriProof */
  free(fri.p);
  buf_u64(&out, n_pi); buf_felts(&out, public_inputs, n_pi);

  *proof_out = out.p; *proof_len = out.len;
  free(open); free(open_next);
  for (int b = 0; b < 4; b++) batch_free(&B[b]);
  return 0;
}

/* plonky2 `PolynomialBatch::prove_openings` + `fri_proof`: draws alpha, builds the batch polynomial
 *   final = sum_b alpha^(counts of the later batches) * (F_b - F_b(z_b)) / (X - z_b),  F_b = sum_j alpha^j f_bj
 * commits the folded layers, observes the final polynomial, grinds the proof of work (smallest witness), draws the query
 * indices and writes bincode FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }. */
static int fri_prove_core(const batch_t *const *B, size_t n_oracles, const or_fri_batch *batches, size_t n_batches, const fri_cfg_t *cfg,
                          or_challenger *cp, int use_pow_override, uint64_t pow_override, buf_t *outp, or_tail_debug *dbg) {
  const int db = cfg->db, rb = cfg->rb, ch = cfg->ch;
  const size_t n = (size_t)1 << db, N = n << rb, cap_n = (size_t)1 << ch;
  or_challenger c = *cp;
  gl2_t fri_alpha = ch_ext(&c);
  gl2_t *fin = (gl2_t *)calloc(N, sizeof(gl2_t)); /* LDE-padded coefficient vector */
  for (size_t batch = 0; batch < n_batches; batch++) {
    gl2_t *comp = (gl2_t *)calloc(n, sizeof(gl2_t));
    gl2_t ap = gl2_from_base(1);
    size_t cnt = 0;
    for (size_t r = 0; r < batches[batch].n_ranges; r++) {
      const or_fri_range *R = &batches[batch].ranges[r];
      for (size_t p = R->first; p < (size_t)R->first + R->count; p++) {
        const uint64_t *f = B[R->oracle]->coeffs + p * n;
        for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(ap, f[i]));
        ap = gl2_mul(ap, fri_alpha);
        cnt++;
      }
    }
    /* divide_by_linear(point): q[i-1] = comp[i] + z*q[i] */
    gl2_t z = gl2_make(batches[batch].point[0], batches[batch].point[1]);
    gl2_t *q = (gl2_t *)calloc(n, sizeof(gl2_t));
    gl2_t acc = gl2_from_base(0);
    for (size_t i = n; i-- > 1;) { acc = gl2_add(gl2_mul(acc, z), comp[i]); q[i - 1] = acc; }
    /* final = final * alpha^cnt + q   (ReducingFactor::shift_poly with the count of THIS batch) */
    gl2_t sh_f = gl2_pow(fri_alpha, cnt);
    for (size_t i = 0; i < n; i++) fin[i] = gl2_add(gl2_mul(fin[i], sh_f), q[i]);
    free(comp); free(q);
  }
  /* values on the LDE coset (natural order) */
  gl2_t *vals = (gl2_t *)malloc(N * sizeof(gl2_t));
  ext_coset_ntt(fin, db + rb, GL_GENERATOR, vals);

  /* ---- commit phase ---- */
  fri_trees_t T; memset(&T, 0, sizeof T);
  T.n_layers = cfg->n_arity;
  gl2_t *coeffs = fin; size_t clen = N; int clog = db + rb;
  uint64_t shift = GL_GENERATOR;
  for (int l = 0; l < cfg->n_arity; l++) {
    int ab = cfg->arity_bits[l]; size_t arity = (size_t)1 << ab;
    size_t nl = clen >> ab;
    uint64_t *leaves = (uint64_t *)malloc(clen * 16);
    for (size_t i = 0; i < clen; i++) { /* position i of the bit-reversed value vector */
      gl2_t v = vals[bitrev_sz(i, clog)];
      leaves[2 * i] = v.c[0]; leaves[2 * i + 1] = v.c[1];
    }
    T.leaves[l] = leaves; T.n_leaves[l] = nl;
    T.digests[l] = (uint64_t *)malloc((digest_nodes(nl, ch) + 1) * 32);
    T.cap[l] = (uint64_t *)malloc(cap_n * 32);
    or_merkle_tree(leaves, nl, 2 * arity, ch, T.digests[l], T.cap[l]);
    or_ch_observe(&c, T.cap[l], cap_n * 4);
    gl2_t beta = ch_ext(&c);
    if (dbg && l < 8) { dbg->fri_betas[l][0] = beta.c[0]; dbg->fri_betas[l][1] = beta.c[1]; }
    gl2_t *nc = (gl2_t *)malloc(nl * sizeof(gl2_t));
    for (size_t j = 0; j < nl; j++) nc[j] = eval_ext_poly(coeffs + j * arity, arity, beta); /* sum beta^i c[16j+i] */
    if (coeffs != fin) free(coeffs);
    coeffs = nc; clen = nl; clog -= ab;
    shift = gl_pow(shift, arity);
    free(vals);
    vals = (gl2_t *)malloc(clen * sizeof(gl2_t));
    ext_coset_ntt(coeffs, clog, shift, vals);
  }
  size_t final_len = clen >> rb;
  for (size_t i = 0; i < final_len; i++) or_ch_observe(&c, coeffs[i].c, 2);

  /* ---- proof of work: smallest witness whose response has >= pow_bits leading zeros ---- */
  uint64_t pow_witness = 0;
  {
    uint64_t st[12]; memcpy(st, c.state, sizeof st);
    for (int i = 0; i < c.n_in; i++) st[i] = c.in[i];
    int pos = c.n_in;
    if (use_pow_override) pow_witness = pow_override;
    else for (uint64_t base = 0;; base += 4096) { /* smallest witness; blocks of candidates tried in parallel */
      uint64_t found = UINT64_MAX;
#pragma omp parallel for num_threads(or_get_threads()) schedule(static) reduction(min : found)
      for (long long k = 0; k < 4096; k++) {
        uint64_t t[12]; memcpy(t, st, sizeof t);
        t[pos] = base + (uint64_t)k;
        or_poseidon_permute(t);
        if (pow_ok(t[7], cfg->pow_bits) && base + (uint64_t)k < found) found = base + (uint64_t)k;
      }
      if (found != UINT64_MAX) { pow_witness = found; break; }
    }
    or_ch_observe(&c, &pow_witness, 1);
    uint64_t resp = or_ch_challenge(&c);
    if (dbg) dbg->pow_response = resp;
    if (!use_pow_override && !pow_ok(resp, cfg->pow_bits)) return -1;
  }

  /* ---- FriProof ---- */
  buf_t out = *outp;
  buf_u64(&out, cfg->n_arity);
  for (int l = 0; l < cfg->n_arity; l++) { buf_u64(&out, cap_n); buf_felts(&out, T.cap[l], cap_n * 4); }
  buf_u64(&out, cfg->nq);
  int depth0 = db + rb - ch;
  uint64_t sib[64 * 4];
  for (int qi = 0; qi < cfg->nq; qi++) {
    size_t x = (size_t)(or_ch_challenge(&c) % N);
    if (dbg && qi < 64) dbg->query_indices[qi] = x;
    buf_u64(&out, n_oracles);
    for (size_t b = 0; b < n_oracles; b++) {
      buf_u64(&out, B[b]->k + B[b]->n_salt); /* the whole leaf, salt included */
      for (size_t p = 0; p < B[b]->k + B[b]->n_salt; p++) buf_u64(&out, B[b]->lde[p * N + x]);
      merkle_path(B[b]->digests, N, ch, x, sib);
      buf_u64(&out, depth0); buf_felts(&out, sib, (size_t)depth0 * 4);
    }
    buf_u64(&out, cfg->n_arity);
    size_t xi = x;
    for (int l = 0; l < cfg->n_arity; l++) {
      int ab = cfg->arity_bits[l]; size_t arity = (size_t)1 << ab;
      xi >>= ab;
      buf_u64(&out, arity); buf_felts(&out, T.leaves[l] + xi * 2 * arity, 2 * arity);
      int depth = log2z(T.n_leaves[l]) - ch; if (depth < 0) depth = 0;
      merkle_path(T.digests[l], T.n_leaves[l], ch, xi, sib);
      buf_u64(&out, depth); buf_felts(&out, sib, (size_t)depth * 4);
    }
  }
  buf_u64(&out, final_len); buf_put(&out, coeffs, final_len * 16);
  buf_u64(&out, pow_witness);
  *outp = out;
  *cp = c;
  for (int l = 0; l < cfg->n_arity; l++) { free(T.leaves[l]); free(T.digests[l]); free(T.cap[l]); }
  if (coeffs != fin) free(coeffs);
  free(fin); free(vals);
  return 0;
}

/* ---- the generic seams as entry points (plonky2 PolynomialBatch / prove_openings) ---- */
or_batch *or_batch_commit(const uint64_t *polys, size_t k, int log_n, int rate_bits, int cap_height, int from_coeffs, const uint64_t *salt) {
  batch_t *b = (batch_t *)calloc(1, sizeof(batch_t));
  if (from_coeffs) batch_from_coeffs(b, polys, k, log_n, rate_bits, cap_height, salt);
  else batch_from_values(b, polys, k, log_n, rate_bits, cap_height, salt);
  return b;
}
void or_batch_free(or_batch *b) { if (b) { batch_free(b); free(b); } }
const uint64_t *or_batch_cap(const or_batch *b) { return b->cap; }
const uint64_t *or_batch_coeffs(const or_batch *b) { return b->coeffs; }
const uint64_t *or_batch_lde(const or_batch *b) { return b->lde; }
void or_batch_shape(const or_batch *b, size_t *k, int *log_n, int *rate_bits, int *cap_height) {
  *k = b->k; *log_n = b->log_n; *rate_bits = b->rate_bits; *cap_height = b->cap_height;
}
void or_batch_eval_ext(const or_batch *b, size_t first, size_t count, const uint64_t point[2], uint64_t *out) {
  size_t n = (size_t)1 << b->log_n;
  gl2_t z = gl2_make(point[0], point[1]);
  for (size_t j = 0; j < count; j++) {
    gl2_t v = eval_base_poly_ext(b->coeffs + (first + j) * n, n, z);
    out[2 * j] = v.c[0]; out[2 * j + 1] = v.c[1];
  }
}
/* PolynomialBatch::get_lde_values(index * step): the leaf at bit-reversed position = the values at NATURAL position */
void or_batch_lde_rows(const or_batch *b, size_t first_index, size_t count, size_t step, uint64_t *out) {
  int lb = b->log_n + b->rate_bits;
  size_t N = (size_t)1 << lb;
  for (size_t r = 0; r < count; r++) {
    size_t pos = bitrev_sz(((first_index + r) * step) & (N - 1), lb);
    for (size_t j = 0; j < b->k; j++) out[r * b->k + j] = b->lde[j * N + pos];
  }
}
int or_fri_prove(const or_batch *const *oracles, size_t n_oracles, const or_fri_batch *batches, size_t n_batches,
                 const or_fri_params *params, or_challenger *c, int use_pow_override, uint64_t pow_override,
                 uint8_t **out, size_t *len, or_tail_debug *dbg) {
  fri_cfg_t cfg = {params->degree_bits, params->rate_bits, params->cap_height, params->pow_bits, params->num_query_rounds, params->n_arity, {0}};
  for (int l = 0; l < 8; l++) cfg.arity_bits[l] = params->arity_bits[l];
  for (size_t o = 0; o < n_oracles; o++)
    if (oracles[o]->log_n != cfg.db || oracles[o]->rate_bits != cfg.rb || oracles[o]->cap_height != cfg.ch) return -2;
  buf_t b = {0};
  int rc = fri_prove_core(oracles, n_oracles, batches, n_batches, &cfg, c, use_pow_override, pow_override, &b, dbg);
  if (rc) { free(b.p); return rc; }
  *out = b.p; *len = b.len;
  return 0;
}

void or_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------ */
/* FRI verifier side */

/* Interpolate {(x*g^i, evals'[i])} (evals' = bit-reversed evals) and evaluate at beta:
 * plonky2 `compute_evaluation`. x is the point of THIS query in the current layer's domain. */
void or_fri_compute_evaluation(uint64_t x, size_t x_index_within_coset, int arity_bits, const uint64_t *evals /* arity x 2 */,
                               const uint64_t beta[2], uint64_t out[2]) {
  size_t arity = (size_t)1 << arity_bits;
  uint64_t g = gl_root_of_unity(arity_bits);
  gl2_t ev[64]; uint64_t pts[64];
  for (size_t i = 0; i < arity; i++) { size_t r = bitrev_sz(i, arity_bits); ev[i] = gl2_make(evals[2 * r], evals[2 * r + 1]); }
  size_t rev = bitrev_sz(x_index_within_coset, arity_bits);
  uint64_t start = gl_mul(x, gl_pow(g, arity - rev));
  uint64_t gp = 1;
  for (size_t i = 0; i < arity; i++) { pts[i] = gl_mul(start, gp); gp = gl_mul(gp, g); }
  /* Lagrange interpolation at beta (barycentric; exact arithmetic, any correct method agrees) */
  gl2_t b = gl2_make(beta[0], beta[1]);
  gl2_t acc = gl2_from_base(0);
  for (size_t i = 0; i < arity; i++) {
    gl2_t num = gl2_from_base(1); uint64_t den = 1;
    for (size_t j = 0; j < arity; j++) if (j != i) {
      num = gl2_mul(num, gl2_sub(b, gl2_from_base(pts[j])));
      den = gl_mul(den, gl_sub(pts[i], pts[j]));
    }
    acc = gl2_add(acc, gl2_mul(ev[i], gl2_scale(num, gl_inv(den))));
  }
  out[0] = acc.c[0]; out[1] = acc.c[1];
}

/* x = g_coset * omega_N^rev(x_index) */
uint64_t or_fri_query_point(size_t x_index, int log_n) {
  return gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(log_n), bitrev_sz(x_index, log_n)));
}

/* byte reader */
typedef struct { const uint8_t *p; size_t len, o; int bad; } rd_t;
static uint64_t rd_u64(rd_t *r) { if (r->o + 8 > r->len) { r->bad = 1; return 0; } uint64_t v; memcpy(&v, r->p + r->o, 8); r->o += 8; return v; }
static const uint64_t *rd_felts(rd_t *r, size_t n) { if (r->o + 8 * n > r->len) { r->bad = 1; return NULL; } const uint64_t *v = (const uint64_t *)(r->p + r->o); r->o += 8 * n; return v; }

/* ---- generic FRI verifier: plonky2 `Challenger::fri_challenges` + `verify_fri_proof` ----
 * kk[o] / leaf_len[o]: polynomials of oracle o / its leaf length (+ OR_SALT_SIZE when blinded); caps[o]: its cap;
 * batches: the opening batches; opened[b]: the claimed values of batch b in list order (2 u64 each).
 * Reads FriProof from r (leaves r->o behind the pow witness). 0 = accepted. */
static int fri_verify_core(const fri_cfg_t *cfg, size_t n_oracles, const size_t *kk, const size_t *leaf_len, const uint64_t *const *caps,
                           const or_fri_batch *batches, size_t n_batches, const uint64_t *const *opened, or_challenger *cp, rd_t *r,
                           or_tail_debug *dbg) {
  const int db = cfg->db, rb = cfg->rb, ch = cfg->ch;
  const size_t n = (size_t)1 << db, N = n << rb, cap_n = (size_t)1 << ch;
  (void)kk;
  if (rd_u64(r) != (uint64_t)cfg->n_arity) return -4;
  const uint64_t *fcap[8];
  for (int l = 0; l < cfg->n_arity; l++) { if (rd_u64(r) != cap_n) return -4; fcap[l] = rd_felts(r, cap_n * 4); }
  if (rd_u64(r) != (uint64_t)cfg->nq) return -5;
  size_t q_off = r->o;
  /* skip the queries to reach final poly / pow */
  int depth0 = db + rb - ch;
  for (int qi = 0; qi < cfg->nq; qi++) {
    if (rd_u64(r) != n_oracles) return -6;
    for (size_t b = 0; b < n_oracles; b++) { if (rd_u64(r) != leaf_len[b]) return -6; rd_felts(r, leaf_len[b]); if (rd_u64(r) != (uint64_t)depth0) return -6; rd_felts(r, (size_t)depth0 * 4); }
    if (rd_u64(r) != (uint64_t)cfg->n_arity) return -6;
    size_t nl = N;
    for (int l = 0; l < cfg->n_arity; l++) {
      size_t arity = (size_t)1 << cfg->arity_bits[l]; nl >>= cfg->arity_bits[l];
      if (rd_u64(r) != arity) return -6;
      rd_felts(r, 2 * arity);
      int depth = log2z(nl) - ch; if (depth < 0) depth = 0;
      if (rd_u64(r) != (uint64_t)depth) return -6;
      rd_felts(r, (size_t)depth * 4);
    }
    if (r->bad) return -7;
  }
  size_t final_len = rd_u64(r);
  int total_arity = 0;
  for (int l = 0; l < cfg->n_arity; l++) total_arity += cfg->arity_bits[l];
  if (final_len != n >> total_arity) return -7;
  const uint64_t *final_poly = rd_felts(r, final_len * 2);
  uint64_t pow_witness = rd_u64(r);
  if (r->bad) return -7;

  or_challenger c = *cp;
  gl2_t alpha = ch_ext(&c);
  gl2_t betas[8];
  for (int l = 0; l < cfg->n_arity; l++) { or_ch_observe(&c, fcap[l], cap_n * 4); betas[l] = ch_ext(&c); }
  or_ch_observe(&c, final_poly, final_len * 2);
  or_ch_observe(&c, &pow_witness, 1);
  uint64_t resp = or_ch_challenge(&c);
  if (dbg) { dbg->pow_response = resp;
             for (int l = 0; l < cfg->n_arity; l++) { dbg->fri_betas[l][0] = betas[l].c[0]; dbg->fri_betas[l][1] = betas[l].c[1]; } }
  if (!pow_ok(resp, cfg->pow_bits)) return -8;

  /* reduced openings: sum alpha^j opening_j per batch; shift = alpha^(count of the batch) */
  gl2_t *red = (gl2_t *)malloc(n_batches * sizeof(gl2_t)), *shf = (gl2_t *)malloc(n_batches * sizeof(gl2_t));
  for (size_t b = 0; b < n_batches; b++) {
    size_t cnt = 0;
    for (size_t i = 0; i < batches[b].n_ranges; i++) cnt += batches[b].ranges[i].count;
    red[b] = gl2_from_base(0);
    for (size_t j = cnt; j-- > 0;) red[b] = gl2_add(gl2_mul(red[b], alpha), gl2_make(opened[b][2 * j], opened[b][2 * j + 1]));
    shf[b] = gl2_pow(alpha, cnt);
  }
  int rc = 0;
  rd_t q = {r->p, r->len, q_off, 0};
  for (int qi = 0; qi < cfg->nq && !rc; qi++) {
    size_t x_index = (size_t)(or_ch_challenge(&c) % N);
    if (dbg && qi < 64) dbg->query_indices[qi] = x_index;
    rd_u64(&q);
    const uint64_t *ev[16];
    for (size_t b = 0; b < n_oracles; b++) {
      rd_u64(&q); ev[b] = rd_felts(&q, leaf_len[b]); rd_u64(&q);
      const uint64_t *sibs = rd_felts(&q, (size_t)depth0 * 4);
      if (!or_merkle_verify(ev[b], leaf_len[b], x_index, sibs, depth0, caps[b], ch)) { rc = -10 - (int)b; break; }
    }
    if (rc) break;
    uint64_t x = or_fri_query_point(x_index, db + rb);
    /* fri_combine_initial */
    gl2_t sum = gl2_from_base(0);
    for (size_t b = 0; b < n_batches; b++) {
      gl2_t re = gl2_from_base(0);
      for (size_t i = batches[b].n_ranges; i-- > 0;) {
        const or_fri_range *R = &batches[b].ranges[i];
        for (size_t j = R->count; j-- > 0;) re = gl2_add(gl2_mul(re, alpha), gl2_from_base(ev[R->oracle][R->first + j]));
      }
      gl2_t num = gl2_sub(re, red[b]), den = gl2_sub(gl2_from_base(x), gl2_make(batches[b].point[0], batches[b].point[1]));
      sum = gl2_add(gl2_mul(sum, shf[b]), gl2_mul(num, gl2_inv(den))); /* alpha.shift(sum) uses the count of THIS batch */
    }
    rd_u64(&q);
    gl2_t old = sum; size_t xi = x_index; size_t nl = N;
    for (int l = 0; l < cfg->n_arity; l++) {
      int ab = cfg->arity_bits[l]; size_t arity = (size_t)1 << ab; nl >>= ab;
      rd_u64(&q); const uint64_t *fe = rd_felts(&q, 2 * arity);
      int depth = log2z(nl) - ch; if (depth < 0) depth = 0;
      rd_u64(&q); const uint64_t *sibs = rd_felts(&q, (size_t)depth * 4);
      size_t within = xi & (arity - 1), coset = xi >> ab;
      if (fe[2 * within] != old.c[0] || fe[2 * within + 1] != old.c[1]) { rc = -20 - l; break; }
      uint64_t o2[2];
      or_fri_compute_evaluation(x, within, ab, fe, betas[l].c, o2);
      old = gl2_make(o2[0], o2[1]);
      if (!or_merkle_verify(fe, 2 * arity, coset, sibs, depth, fcap[l], ch)) { rc = -30 - l; break; }
      for (int s2 = 0; s2 < ab; s2++) x = gl_mul(x, x);
      xi = coset;
    }
    if (rc) break;
    gl2_t fp = eval_ext_poly((const gl2_t *)final_poly, final_len, gl2_from_base(x));
    if (!gl2_eq(fp, old)) rc = -40;
  }
  free(red); free(shf);
  if (!rc) *cp = c;
  return rc;
}

int or_fri_verify(const or_fri_params *params, const uint32_t *num_polys, const uint32_t *blinding, size_t n_oracles,
                  const uint64_t *const *caps, const or_fri_batch *batches, size_t n_batches, const uint64_t *const *opened,
                  or_challenger *c, const uint8_t *proof, size_t len, or_tail_debug *dbg) {
  if (n_oracles > 16) return -2;
  fri_cfg_t cfg = {params->degree_bits, params->rate_bits, params->cap_height, params->pow_bits, params->num_query_rounds, params->n_arity, {0}};
  for (int l = 0; l < 8; l++) cfg.arity_bits[l] = params->arity_bits[l];
  size_t kk[16], leaf_len[16];
  for (size_t o = 0; o < n_oracles; o++) { kk[o] = num_polys[o]; leaf_len[o] = kk[o] + (blinding[o] ? OR_SALT_SIZE : 0); }
  rd_t r = {proof, len, 0, 0};
  int rc = fri_verify_core(&cfg, n_oracles, kk, leaf_len, caps, batches, n_batches, opened, c, &r, dbg);
  if (rc) return rc;
  if (r.bad || r.o != len) return -7;
  return 0;
}

/* Verify everything of a proof that does not need the gate constraints: transcript, proof of work,
 * Merkle paths of all queries, fri_combine_initial against the openings, fold consistency, final poly.
 * Returns 0 if ok, otherwise a negative code naming the first failing check. */
int or_verify_tail(const or_shape *sh, const uint64_t circuit_digest[4], const uint64_t *cs_cap,
                   const uint8_t *proof, size_t len, or_tail_debug *dbg) {
  const int db = sh->degree_bits, rb = sh->rate_bits, ch = sh->cap_height;
  const size_t cap_n = (size_t)1 << ch;
  const size_t k_cs = sh->num_constants + sh->num_routed_wires, k_w = sh->num_wires;
  const size_t k_z = (size_t)sh->num_challenges * (1 + sh->num_partial_products);
  const size_t k_q = (size_t)sh->num_challenges * sh->quotient_degree_factor;
  rd_t r = {proof, len, 0, 0};
  const uint64_t *caps[4]; caps[0] = cs_cap;
  for (int b = 1; b <= 3; b++) { if (rd_u64(&r) != cap_n) return -2; caps[b] = rd_felts(&r, cap_n * 4); }
  size_t cnt[9] = {(size_t)sh->num_constants, (size_t)sh->num_routed_wires, k_w, (size_t)sh->num_challenges, (size_t)sh->num_challenges,
                   k_z - sh->num_challenges, k_q, 0, 0};
  const uint64_t *op[9];
  for (int i = 0; i < 9; i++) { if (rd_u64(&r) != cnt[i]) return -3; op[i] = rd_felts(&r, cnt[i] * 2); }
  if (r.bad) return -7;
  size_t fri_off = r.o;
  /* the public inputs sit behind the FriProof: parse it once to find them */
  size_t kk[4] = {k_cs, k_w, k_z, k_q};
  /* FriParams::hiding: the blinded oracles' leaves carry OR_SALT_SIZE salt elements after the polynomial values; the
   * Merkle check covers the whole leaf, fri_combine_initial only the unsalted part (FriInitialTreeProof::unsalted_eval) */
  size_t leaf_len[4];
  for (int b = 0; b < 4; b++) leaf_len[b] = kk[b] + ((sh->zero_knowledge && b >= 1) ? OR_SALT_SIZE : 0);
  fri_cfg_t cfg = {db, rb, ch, sh->pow_bits, sh->num_query_rounds, sh->n_arity, {0}};
  for (int l = 0; l < 8; l++) cfg.arity_bits[l] = sh->arity_bits[l];
  size_t pi_off;
  {
    /* length of the FriProof from the shape alone */
    int depth0 = db + rb - ch;
    size_t words = 1 + (size_t)sh->n_arity * (1 + cap_n * 4) + 1, per_q = 1;
    for (int b = 0; b < 4; b++) per_q += 1 + leaf_len[b] + 1 + (size_t)depth0 * 4;
    per_q += 1;
    size_t nl = (size_t)1 << (db + rb); int total = 0;
    for (int l = 0; l < sh->n_arity; l++) {
      nl >>= sh->arity_bits[l]; total += sh->arity_bits[l];
      int depth = log2z(nl) - ch; if (depth < 0) depth = 0;
      per_q += 1 + ((size_t)2 << sh->arity_bits[l]) + 1 + (size_t)depth * 4;
    }
    words += per_q * (size_t)sh->num_query_rounds + 1 + 2 * (((size_t)1 << db) >> total) + 1;
    pi_off = fri_off + 8 * words;
  }
  if (pi_off + 8 > len) return -7;
  rd_t rp = {proof, len, pi_off, 0};
  size_t n_pi = rd_u64(&rp);
  const uint64_t *pi = rd_felts(&rp, n_pi);
  if (rp.bad || rp.o != len) return -7;

  /* transcript */
  uint64_t pi_hash[4]; or_hash_no_pad(pi, n_pi, pi_hash);
  or_challenger c; or_ch_init(&c);
  or_ch_observe(&c, circuit_digest, 4); or_ch_observe(&c, pi_hash, 4); or_ch_observe(&c, caps[1], cap_n * 4);
  uint64_t vb[8], vg[8], va[8];
  for (int i = 0; i < sh->num_challenges; i++) vb[i] = or_ch_challenge(&c);
  for (int i = 0; i < sh->num_challenges; i++) vg[i] = or_ch_challenge(&c);
  or_ch_observe(&c, caps[2], cap_n * 4);
  for (int i = 0; i < sh->num_challenges; i++) va[i] = or_ch_challenge(&c);
  if (dbg) { memcpy(dbg->betas, vb, sizeof vb); memcpy(dbg->gammas, vg, sizeof vg); memcpy(dbg->alphas, va, sizeof va); }
  or_ch_observe(&c, caps[3], cap_n * 4);
  gl2_t zeta = ch_ext(&c);
  gl2_t zeta_next = gl2_scale(zeta, gl_root_of_unity(db));
  if (dbg) { dbg->zeta[0] = zeta.c[0]; dbg->zeta[1] = zeta.c[1]; }
  /* zeta batch order: constants, sigmas, wires, zs, partial products, quotient ; next batch: zs_next */
  int zorder[6] = {0, 1, 2, 3, 5, 6};
  size_t k_all = k_cs + k_w + k_z + k_q;
  uint64_t *open0 = (uint64_t *)malloc(k_all * 16);
  {
    size_t o = 0;
    for (int i = 0; i < 6; i++) { or_ch_observe(&c, op[zorder[i]], cnt[zorder[i]] * 2); memcpy(open0 + o, op[zorder[i]], cnt[zorder[i]] * 16); o += cnt[zorder[i]] * 2; }
  }
  or_ch_observe(&c, op[4], cnt[4] * 2);

  or_fri_range r0[4] = {{0, 0, (uint32_t)k_cs}, {1, 0, (uint32_t)k_w}, {2, 0, (uint32_t)k_z}, {3, 0, (uint32_t)k_q}};
  or_fri_range r1[1] = {{2, 0, (uint32_t)sh->num_challenges}};
  or_fri_batch fb[2] = {{{zeta.c[0], zeta.c[1]}, r0, 4}, {{zeta_next.c[0], zeta_next.c[1]}, r1, 1}};
  const uint64_t *opened[2] = {open0, op[4]};
  rd_t rf = {proof, len, fri_off, 0};
  int rc = fri_verify_core(&cfg, 4, kk, leaf_len, caps, fb, 2, opened, &c, &rf, dbg);
  free(open0);
  if (rc) return rc;
  if (rf.o != pi_off) return -7;
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A7: permutation argument — Z and partial products (plonky2 `wires_permutation_partial_products_and_zs`,
 * SURVEY.md §3.3 step 5 / §8(a) A7). For every row i (x = omega^i) and routed wire j:
 *   num_j = w_ij + beta * k_j * x + gamma ,  den_j = w_ij + beta * sigma_j(x) + gamma
 * the 80 quotients are multiplied in chunks of `quotient_degree_factor` (10 chunk products), the running
 * product over chunks gives the 9 partial products and Z(g x); Z(1) = 1.
 * Output order = the committed batch: [Z_0 .. Z_{c-1}, pp(challenge 0) 0..npp-1, pp(challenge 1) ...]. */
void or_zs_partial_products(const or_shape *sh, const uint64_t *wires_values, const uint64_t *sigma_values,
                            const uint64_t *k_is, const uint64_t *betas, const uint64_t *gammas,
                            uint64_t *out) {
  const size_t n = (size_t)1 << sh->degree_bits;
  const int R = sh->num_routed_wires, chunk = sh->quotient_degree_factor, npp = sh->num_partial_products;
  const int nchunks = (R + chunk - 1) / chunk; /* == npp + 1 */
  const int nc = sh->num_challenges;
  uint64_t omega = gl_root_of_unity(sh->degree_bits);
  /* phase 1 (rows are independent: or_set_threads workers): the quotient of every chunk of every row */
  uint64_t *q = (uint64_t *)malloc((size_t)nc * nchunks * n * 8);
#pragma omp parallel for num_threads(or_get_threads()) schedule(static)
  for (long long ii = 0; ii < (long long)n; ii++) {
    const size_t i = (size_t)ii;
    const uint64_t x = gl_pow(omega, i);
    for (int c = 0; c < nc; c++)
      for (int t = 0; t < nchunks; t++) {
        uint64_t prod = 1;
        for (int j = t * chunk; j < R && j < (t + 1) * chunk; j++) {
          uint64_t w = wires_values[(size_t)j * n + i];
          uint64_t num = gl_add(gl_add(w, gl_mul(betas[c], gl_mul(k_is[j], x))), gammas[c]);
          uint64_t den = gl_add(gl_add(w, gl_mul(betas[c], sigma_values[(size_t)j * n + i])), gammas[c]);
          prod = gl_mul(prod, gl_mul(num, gl_inv(den)));
        }
        q[((size_t)c * nchunks + t) * n + i] = prod;
      }
  }
  /* phase 2: the running product down the rows */
  for (int c = 0; c < nc; c++) {
    uint64_t *Z = out + (size_t)c * n;
    uint64_t *PP = out + ((size_t)nc + (size_t)c * npp) * n;
    uint64_t z = 1;
    for (size_t i = 0; i < n; i++) {
      uint64_t acc = z;
      Z[i] = z;
      for (int t = 0; t < nchunks; t++) {
        acc = gl_mul(acc, q[((size_t)c * nchunks + t) * n + i]);
        if (t < npp) PP[(size_t)t * n + i] = acc;
      }
      z = acc; /* Z(g x) */
    }
  }
  free(q);
}
