/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see goldilocks.h / cityoracle.h). Checker of the generic AIR machinery of the product
 * (include/cityprover.h "the STARK's own two steps as GENERIC device machinery": cp_air_program, cp_air_quotient_commit,
 * cp_air_map_dev, cp_cubic_batch_inverse_dev, cp_column_prefix_sum_dev, cp_stark_prove / cp_stark_verify).
 *
 * What it follows. The reference proves a SHA-256 STARK inside witness generation
 * (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524; 418 free + 912 extended columns, :55-79; row
 * count :310-312; call site city_rollup_circuit/src/sighash_circuits/sighash.rs:132-146) with starkyx 0.1.0
 * (git QEDProtocol/starkyx @ a53ea106, /root/reference/Cargo.toml:112 and :131-132 — ABSENT from the tree). Its AIR cannot be
 * restated; this file restates the protocol AROUND an AIR as plonky2's starky defines it (UPSTREAM-MEMORY; the shape starkyx
 * is built on, smartgadget.rs:48-49 `plonky2::{stark::config::GenericCombinedConfig, Plonky2Air}`), with the AIR supplied
 * as data: a straight-line program over F_p. PARITY UNPINNED: the reference holds no STARK proof, challenge or constraint
 * vector (smartgadget.rs:505-513 asserts digests only). The interpreter here is the direct one — every op evaluated into an
 * array of values, no dead-value elimination, no segments, no slots — so that the product's compiled form is checked against
 * an independent evaluation.
 */
#include <stdlib.h>
#include <string.h>

#include "cityoracle.h"
#include "goldilocks.h"

enum { AIR_LOCAL = 0, AIR_NEXT, AIR_PUBLIC, AIR_GLOBAL, AIR_CHALLENGE, AIR_CONST, AIR_ADD, AIR_SUB, AIR_MUL, AIR_NEG, AIR_INV,
       AIR_ASSERT_ZERO, AIR_ASSERT_ZERO_TRANSITION, AIR_ASSERT_ZERO_FIRST_ROW, AIR_ASSERT_ZERO_LAST_ROW, AIR_STORE };

static int defines_value(uint32_t op) { return op <= AIR_INV; }

/* 0 = well-formed; else 1 + the index of the first offending op */
size_t or_air_check(const or_air_program *p) {
  for (size_t i = 0; i < p->n_ops; i++) /* an output column is stored at most once */
    if (p->ops[i].op == AIR_STORE)
      for (size_t j = 0; j < i; j++)
        if (p->ops[j].op == AIR_STORE && p->ops[j].a == p->ops[i].a) return i + 1;
  for (size_t i = 0; i < p->n_ops; i++) {
    const or_air_op *o = &p->ops[i];
    int bad = 0;
    switch (o->op) {
      case AIR_LOCAL: case AIR_NEXT: bad = o->a >= p->n_columns; break;
      case AIR_PUBLIC: bad = o->a >= p->n_public; break;
      case AIR_GLOBAL: bad = o->a >= p->n_global; break;
      case AIR_CHALLENGE: bad = o->a >= p->n_challenge; break;
      case AIR_CONST: bad = o->a >= p->n_consts || p->consts[o->a] >= GL_P; break;
      case AIR_ADD: case AIR_SUB: case AIR_MUL:
        bad = o->a >= i || o->b >= i || !defines_value(p->ops[o->a].op) || !defines_value(p->ops[o->b].op);
        break;
      case AIR_NEG: bad = o->a >= i || !defines_value(p->ops[o->a].op); break;
      case AIR_INV: bad = !p->map || o->a >= i || !defines_value(p->ops[o->a].op); break;
      case AIR_ASSERT_ZERO: case AIR_ASSERT_ZERO_TRANSITION: case AIR_ASSERT_ZERO_FIRST_ROW: case AIR_ASSERT_ZERO_LAST_ROW:
        bad = p->map || o->a >= i || !defines_value(p->ops[o->a].op);
        break;
      case AIR_STORE: bad = !p->map || o->a >= p->n_out_columns || o->b >= i || !defines_value(p->ops[o->b].op); break;
      default: bad = 1;
    }
    if (bad || o->c != 0) return i + 1;
  }
  return 0;
}

size_t or_air_num_constraints(const or_air_program *p) {
  size_t k = 0;
  for (size_t i = 0; i < p->n_ops; i++) k += p->ops[i].op >= AIR_ASSERT_ZERO && p->ops[i].op <= AIR_ASSERT_ZERO_LAST_ROW;
  return k;
}

/* every op on one row over F_p: vals[i] = the value op i defines (0 for sinks / stores) */
void or_air_eval_row(const or_air_program *p, const uint64_t *local, const uint64_t *next, const uint64_t *publics,
                     const uint64_t *globals, const uint64_t *challenges, uint64_t *vals) {
  for (size_t i = 0; i < p->n_ops; i++) {
    const or_air_op *o = &p->ops[i];
    uint64_t v = 0;
    switch (o->op) {
      case AIR_LOCAL: v = local[o->a]; break;
      case AIR_NEXT: v = next[o->a]; break;
      case AIR_PUBLIC: v = publics[o->a]; break;
      case AIR_GLOBAL: v = globals[o->a]; break;
      case AIR_CHALLENGE: v = challenges[o->a]; break;
      case AIR_CONST: v = p->consts[o->a]; break;
      case AIR_ADD: v = gl_add(vals[o->a], vals[o->b]); break;
      case AIR_SUB: v = gl_sub(vals[o->a], vals[o->b]); break;
      case AIR_MUL: v = gl_mul(vals[o->a], vals[o->b]); break;
      case AIR_NEG: v = gl_neg(vals[o->a]); break;
      case AIR_INV: v = vals[o->a] ? gl_inv(vals[o->a]) : 0; break;
      default: break;
    }
    vals[i] = v;
  }
}

/* the same over F_p^2 (what a verifier does at zeta); every input an extension element */
static void air_eval_row_ext(const or_air_program *p, const gl2_t *local, const gl2_t *next, const gl2_t *publics, const gl2_t *globals,
                             const gl2_t *challenges, gl2_t *vals) {
  for (size_t i = 0; i < p->n_ops; i++) {
    const or_air_op *o = &p->ops[i];
    gl2_t v = gl2_make(0, 0);
    switch (o->op) {
      case AIR_LOCAL: v = local[o->a]; break;
      case AIR_NEXT: v = next[o->a]; break;
      case AIR_PUBLIC: v = publics[o->a]; break;
      case AIR_GLOBAL: v = globals[o->a]; break;
      case AIR_CHALLENGE: v = challenges[o->a]; break;
      case AIR_CONST: v = gl2_from_base(p->consts[o->a]); break;
      case AIR_ADD: v = gl2_add(vals[o->a], vals[o->b]); break;
      case AIR_SUB: v = gl2_sub(vals[o->a], vals[o->b]); break;
      case AIR_MUL: v = gl2_mul(vals[o->a], vals[o->b]); break;
      case AIR_NEG: v = gl2_sub(gl2_make(0, 0), vals[o->a]); break;
      default: break;
    }
    vals[i] = v;
  }
}

/* constraint values (unfiltered) on one row over F_p^2, program order; kinds_out optional */
void or_air_eval_ext(const or_air_program *p, const uint64_t *local, const uint64_t *next, const uint64_t *publics,
                     const uint64_t *globals, const uint64_t *challenges, uint64_t *out, uint32_t *kinds_out) {
  gl2_t *vals = malloc((p->n_ops ? p->n_ops : 1) * sizeof(gl2_t));
  air_eval_row_ext(p, (const gl2_t *)local, (const gl2_t *)next, (const gl2_t *)publics, (const gl2_t *)globals, (const gl2_t *)challenges, vals);
  size_t k = 0;
  for (size_t i = 0; i < p->n_ops; i++) {
    const uint32_t op = p->ops[i].op;
    if (op < AIR_ASSERT_ZERO || op > AIR_ASSERT_ZERO_LAST_ROW) continue;
    out[2 * k] = vals[p->ops[i].a].c[0];
    out[2 * k + 1] = vals[p->ops[i].a].c[1];
    if (kinds_out) kinds_out[k] = op;
    k++;
  }
  free(vals);
}

/* a map program over n rows of value columns (column-major, natural order; next of the last row = row 0) */
void or_air_map(const or_air_program *p, const uint64_t *in_cols, uint64_t *out_cols, size_t n, const uint64_t *publics,
                const uint64_t *globals, const uint64_t *challenges) {
  uint64_t *vals = malloc((p->n_ops ? p->n_ops : 1) * 8), *local = malloc((p->n_columns ? p->n_columns : 1) * 8),
           *next = malloc((p->n_columns ? p->n_columns : 1) * 8);
  for (size_t r = 0; r < n; r++) {
    for (size_t c = 0; c < p->n_columns; c++) {
      local[c] = in_cols[c * n + r];
      next[c] = in_cols[c * n + (r + 1) % n];
    }
    or_air_eval_row(p, local, next, publics, globals, challenges, vals);
    for (size_t i = 0; i < p->n_ops; i++)
      if (p->ops[i].op == AIR_STORE) out_cols[(size_t)p->ops[i].a * n + r] = vals[p->ops[i].b];
  }
  free(vals); free(local); free(next);
}

/* ---- cubic extension F_p[X]/(X^3 - m1 X - m0): inversion by Cramer's rule on the multiplication matrix ---- */
static void cubic_mulx(const uint64_t m[2], const uint64_t v[3], uint64_t out[3]) { /* X * v */
  out[0] = gl_mul(m[0], v[2]);
  out[1] = gl_add(v[0], gl_mul(m[1], v[2]));
  out[2] = v[1];
}
void or_cubic_mul(const uint64_t m[2], const uint64_t a[3], const uint64_t b[3], uint64_t out[3]) {
  uint64_t ax[3], axx[3], r[3];
  cubic_mulx(m, a, ax);
  cubic_mulx(m, ax, axx);
  for (int i = 0; i < 3; i++) r[i] = gl_add(gl_add(gl_mul(a[i], b[0]), gl_mul(ax[i], b[1])), gl_mul(axx[i], b[2]));
  memcpy(out, r, sizeof r);
}
static uint64_t det2(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return gl_sub(gl_mul(a, d), gl_mul(b, c)); }
void or_cubic_inverse(const uint64_t m[2], const uint64_t a[3], uint64_t out[3]) {
  /* columns of M: a, X a, X^2 a; solve M y = (1, 0, 0): y_j = cofactor(0, j) / det M */
  uint64_t c0[3], c1[3], c2[3];
  memcpy(c0, a, sizeof c0);
  cubic_mulx(m, c0, c1);
  cubic_mulx(m, c1, c2);
  const uint64_t k0 = det2(c1[1], c2[1], c1[2], c2[2]);         /* minor of M[0][0] */
  const uint64_t k1 = gl_neg(det2(c0[1], c2[1], c0[2], c2[2])); /* - minor of M[0][1] */
  const uint64_t k2 = det2(c0[1], c1[1], c0[2], c1[2]);
  const uint64_t det = gl_add(gl_add(gl_mul(c0[0], k0), gl_mul(c1[0], k1)), gl_mul(c2[0], k2));
  const uint64_t di = det ? gl_inv(det) : 0;
  out[0] = gl_mul(k0, di); out[1] = gl_mul(k1, di); out[2] = gl_mul(k2, di);
}
/* count x n elements, element e of row i = (cols[3e][i], cols[3e+1][i], cols[3e+2][i]); in place */
void or_cubic_batch_inverse(const uint64_t m[2], uint64_t *cols, size_t count, size_t n) {
  for (size_t e = 0; e < count; e++)
    for (size_t i = 0; i < n; i++) {
      uint64_t a[3] = {cols[(3 * e) * n + i], cols[(3 * e + 1) * n + i], cols[(3 * e + 2) * n + i]}, r[3];
      or_cubic_inverse(m, a, r);
      for (int j = 0; j < 3; j++) cols[(3 * e + j) * n + i] = r[j];
    }
}
void or_column_prefix_sum(uint64_t *cols, size_t k, size_t n, int exclusive) {
  for (size_t c = 0; c < k; c++) {
    uint64_t run = 0;
    for (size_t i = 0; i < n; i++) {
      const uint64_t v = cols[c * n + i];
      if (exclusive) { cols[c * n + i] = run; run = gl_add(run, v); }
      else { run = gl_add(run, v); cols[c * n + i] = run; }
    }
  }
}

/* ---- starky `compute_quotient_polys`: out = n_alphas * 2^q coefficient vectors of length n (challenge-major) ---- */
static size_t bitrev(size_t x, int bits) {
  size_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}
int or_air_quotient(const or_air_program *p, const or_batch *const *oracles, size_t n_oracles, int qdb, const uint64_t *publics,
                    const uint64_t *globals, const uint64_t *challenges, const uint64_t *alphas, size_t n_alphas, uint64_t *out) {
  if (!n_oracles) return -1;
  size_t kk[16], width = 0;
  int db, rb, ch;
  or_batch_shape(oracles[0], &kk[0], &db, &rb, &ch);
  for (size_t o = 0; o < n_oracles; o++) {
    int d2, r2, c2;
    if (o >= 16) return -1;
    or_batch_shape(oracles[o], &kk[o], &d2, &r2, &c2);
    if (d2 != db || r2 != rb) return -1;
    width += kk[o];
  }
  if (width != p->n_columns || qdb > rb || qdb < 0) return -1;
  const size_t n = (size_t)1 << db, M = n << qdb, N = n << rb, step = (size_t)1 << (rb - qdb);
  const int log_N = db + rb, log_M = db + qdb;
  const uint64_t g = gl_root_of_unity(db), g_last = gl_pow(g, n - 1), wM = gl_root_of_unity(log_M), n_inv = gl_inv((uint64_t)n % GL_P);
  uint64_t *vals = malloc((p->n_ops ? p->n_ops : 1) * 8), *local = malloc((width ? width : 1) * 8), *next = malloc((width ? width : 1) * 8);
  uint64_t *q = malloc(n_alphas * M * 8);
  uint64_t x = GL_GENERATOR;
  for (size_t i = 0; i < M; i++) {
    const size_t s_loc = bitrev(i * step, log_N), s_nxt = bitrev(((i + ((size_t)1 << qdb)) % M) * step, log_N);
    size_t c = 0;
    for (size_t o = 0; o < n_oracles; o++) {
      const uint64_t *lde = or_batch_lde(oracles[o]);
      for (size_t j = 0; j < kk[o]; j++, c++) {
        local[c] = lde[j * N + s_loc];
        next[c] = lde[j * N + s_nxt];
      }
    }
    or_air_eval_row(p, local, next, publics, globals, challenges, vals);
    const uint64_t zh = gl_sub(gl_pow(x, n), 1), z_last = gl_sub(x, g_last);
    const uint64_t l_first = gl_mul(gl_mul(zh, n_inv), gl_inv(gl_sub(x, 1)));
    const uint64_t l_last = gl_mul(gl_mul(gl_mul(zh, n_inv), g_last), gl_inv(z_last));
    const uint64_t zh_inv = gl_inv(zh);
    for (size_t a = 0; a < n_alphas; a++) {
      uint64_t acc = 0;
      for (size_t k = 0; k < p->n_ops; k++) {
        const uint32_t op = p->ops[k].op;
        if (op < AIR_ASSERT_ZERO || op > AIR_ASSERT_ZERO_LAST_ROW) continue;
        uint64_t cv = vals[p->ops[k].a];
        if (op == AIR_ASSERT_ZERO_TRANSITION) cv = gl_mul(cv, z_last);
        else if (op == AIR_ASSERT_ZERO_FIRST_ROW) cv = gl_mul(cv, l_first);
        else if (op == AIR_ASSERT_ZERO_LAST_ROW) cv = gl_mul(cv, l_last);
        acc = gl_add(gl_mul(acc, alphas[a]), cv);
      }
      q[a * M + i] = gl_mul(acc, zh_inv);
    }
    x = gl_mul(x, wM);
  }
  /* coset iNTT on 7<omega_M>: iNTT, then coefficient j times 7^-j; the chunks of n are consecutive */
  const uint64_t s_inv = gl_inv(GL_GENERATOR);
  for (size_t a = 0; a < n_alphas; a++) {
    or_intt(q + a * M, log_M);
    uint64_t sp = 1;
    for (size_t j = 0; j < M; j++) {
      out[a * M + j] = gl_mul(q[a * M + j], sp);
      sp = gl_mul(sp, s_inv);
    }
  }
  free(vals); free(local); free(next); free(q);
  return 0;
}

/* ---- the whole prover / verifier (same order as cp_stark_prove; include/cityprover.h) ---- */
typedef struct { uint8_t *p; size_t len, cap; } bytes_t;
static void put(bytes_t *b, const void *src, size_t n) {
  if (b->len + n > b->cap) {
    b->cap = (b->len + n) * 2 + 64;
    b->p = realloc(b->p, b->cap);
  }
  memcpy(b->p + b->len, src, n);
  b->len += n;
}
static void put_u64(bytes_t *b, uint64_t v) { put(b, &v, 8); }
static void put_cap(bytes_t *b, const uint64_t *cap, int ch) {
  put_u64(b, (uint64_t)1 << ch);
  put(b, cap, ((size_t)32) << ch);
}

int or_stark_prove(const or_stark_desc *d, const uint64_t *trace_values, const uint64_t *publics, const uint64_t *globals,
                   or_challenger *c, int use_pow_override, uint64_t pow_override, uint8_t **proof_out, size_t *proof_len) {
  const int db = d->degree_bits, rb = d->fri.rate_bits, ch = d->fri.cap_height, qdb = d->quotient_degree_bits;
  const size_t n = (size_t)1 << db, k0 = d->n_trace_columns, k1 = d->n_extended_columns, kq = (size_t)d->num_challenges << qdb;
  or_batch *T0 = or_batch_commit(trace_values, k0, db, rb, ch, 0, NULL), *T1 = NULL, *Q = NULL;
  or_ch_observe(c, or_batch_cap(T0), (size_t)4 << ch);
  uint64_t *rch = malloc((d->n_round_challenges ? d->n_round_challenges : 1) * 8);
  int rc = 0;
  if (k1) {
    for (uint32_t i = 0; i < d->n_round_challenges; i++) rch[i] = or_ch_challenge(c);
    /* a row of a map step = the execution trace followed by the extended columns as filled so far */
    uint64_t *all = calloc((k0 + k1) * n, 8);
    memcpy(all, trace_values, k0 * n * 8);
    uint64_t *ext = all + k0 * n;
    for (size_t s = 0; s < d->n_steps; s++) {
      const or_stark_step *st = &d->steps[s];
      if (st->kind == 0) or_air_map(st->program, all, ext, n, publics, globals, rch);
      else if (st->kind == 1) or_cubic_batch_inverse(st->modulus, ext + (size_t)st->first * n, st->count, n);
      else or_column_prefix_sum(ext + (size_t)st->first * n, st->count, n, st->flags & 1);
    }
    T1 = or_batch_commit(ext, k1, db, rb, ch, 0, NULL);
    free(all);
    or_ch_observe(c, or_batch_cap(T1), (size_t)4 << ch);
  }
  uint64_t alphas[8];
  for (uint32_t i = 0; i < d->num_challenges; i++) alphas[i] = or_ch_challenge(c);
  const or_batch *tr[2] = {T0, T1};
  const size_t n_tr = k1 ? 2 : 1;
  uint64_t *qc = malloc(kq * n * 8);
  rc = or_air_quotient(d->constraints, tr, n_tr, qdb, publics, globals, rch, alphas, d->num_challenges, qc);
  if (rc == 0) {
    Q = or_batch_commit(qc, kq, db, rb, ch, 1, NULL);
    or_ch_observe(c, or_batch_cap(Q), (size_t)4 << ch);
    uint64_t zeta[2], zeta_next[2];
    zeta[0] = or_ch_challenge(c);
    zeta[1] = or_ch_challenge(c);
    const uint64_t g = gl_root_of_unity(db);
    zeta_next[0] = gl_mul(zeta[0], g);
    zeta_next[1] = gl_mul(zeta[1], g);
    const size_t kt = k0 + k1;
    uint64_t *loc = malloc((kt + kq) * 16), *nxt = malloc(kt * 16);
    or_batch_eval_ext(T0, 0, k0, zeta, loc);
    if (k1) or_batch_eval_ext(T1, 0, k1, zeta, loc + 2 * k0);
    or_batch_eval_ext(Q, 0, kq, zeta, loc + 2 * kt);
    or_batch_eval_ext(T0, 0, k0, zeta_next, nxt);
    if (k1) or_batch_eval_ext(T1, 0, k1, zeta_next, nxt + 2 * k0);
    or_ch_observe(c, loc, 2 * (kt + kq));
    or_ch_observe(c, nxt, 2 * kt);
    const or_batch *oracles[3] = {T0, k1 ? T1 : Q, Q};
    const size_t n_or = k1 ? 3 : 2;
    const uint32_t qi = (uint32_t)(n_or - 1);
    or_fri_range r0[3] = {{0, 0, (uint32_t)k0}, {1, 0, (uint32_t)k1}, {qi, 0, (uint32_t)kq}}, r1[2] = {{0, 0, (uint32_t)k0}, {1, 0, (uint32_t)k1}};
    if (!k1) r0[1] = r0[2];
    or_fri_batch fb[2];
    memcpy(fb[0].point, zeta, 16);
    fb[0].ranges = r0; fb[0].n_ranges = k1 ? 3 : 2;
    memcpy(fb[1].point, zeta_next, 16);
    fb[1].ranges = r1; fb[1].n_ranges = k1 ? 2 : 1;
    uint8_t *fri = NULL;
    size_t fri_len = 0;
    rc = or_fri_prove(oracles, n_or, fb, 2, &d->fri, c, use_pow_override, pow_override, &fri, &fri_len, NULL);
    if (rc == 0) {
      bytes_t b = {0};
      put_u64(&b, n_tr);
      put_cap(&b, or_batch_cap(T0), ch);
      if (k1) put_cap(&b, or_batch_cap(T1), ch);
      put_cap(&b, or_batch_cap(Q), ch);
      put_u64(&b, kt); put(&b, loc, kt * 16);
      put_u64(&b, kt); put(&b, nxt, kt * 16);
      put_u64(&b, kq); put(&b, loc + 2 * kt, kq * 16);
      put(&b, fri, fri_len);
      *proof_out = b.p;
      *proof_len = b.len;
    }
    or_free(fri);
    free(loc); free(nxt);
  }
  free(qc); free(rch);
  or_batch_free(T0);
  if (T1) or_batch_free(T1);
  if (Q) or_batch_free(Q);
  return rc;
}

typedef struct { const uint8_t *p; size_t len, o; int bad; } rd_t;
static uint64_t rd_u64(rd_t *r) {
  uint64_t v = 0;
  if (r->o + 8 > r->len) { r->bad = 1; return 0; }
  memcpy(&v, r->p + r->o, 8);
  r->o += 8;
  return v;
}
static const uint64_t *rd_words(rd_t *r, size_t n) { /* n u64, canonical */
  if (r->bad || n > (r->len - r->o) / 8) { r->bad = 1; return NULL; }
  const uint64_t *w = (const uint64_t *)(r->p + r->o);
  for (size_t i = 0; i < n; i++) {
    uint64_t v;
    memcpy(&v, r->p + r->o + 8 * i, 8);
    if (v >= GL_P) r->bad = 1;
  }
  r->o += 8 * n;
  return w;
}

/* 0 = accepted; 1 malformed; 2 constraints fail at zeta; negative: or_fri_verify's code */
int or_stark_verify(const or_stark_desc *d, const uint64_t *publics, const uint64_t *globals, or_challenger *c, const uint8_t *proof,
                    size_t len) {
  const int db = d->degree_bits, ch = d->fri.cap_height, qdb = d->quotient_degree_bits;
  const size_t n = (size_t)1 << db, k0 = d->n_trace_columns, k1 = d->n_extended_columns, kt = k0 + k1, kq = (size_t)d->num_challenges << qdb;
  const size_t n_tr = k1 ? 2 : 1, cap_w = (size_t)4 << ch;
  rd_t r = {proof, len, 0, 0};
  if (rd_u64(&r) != n_tr) return 1;
  const uint64_t *caps[3];
  for (size_t i = 0; i < n_tr + 1; i++) {
    if (rd_u64(&r) != ((uint64_t)1 << ch)) return 1;
    caps[i] = rd_words(&r, cap_w);
  }
  if (rd_u64(&r) != kt) return 1;
  const uint64_t *loc = rd_words(&r, 2 * kt);
  if (rd_u64(&r) != kt) return 1;
  const uint64_t *nxt = rd_words(&r, 2 * kt);
  if (rd_u64(&r) != kq) return 1;
  const uint64_t *qz = rd_words(&r, 2 * kq);
  if (r.bad) return 1;
  uint64_t *lq = malloc((kt + kq) * 16);
  memcpy(lq, loc, kt * 16);
  memcpy(lq + 2 * kt, qz, kq * 16);
  /* transcript */
  or_ch_observe(c, caps[0], cap_w);
  uint64_t *rch = malloc((d->n_round_challenges ? d->n_round_challenges : 1) * 16);
  if (k1) {
    for (uint32_t i = 0; i < d->n_round_challenges; i++) { rch[2 * i] = or_ch_challenge(c); rch[2 * i + 1] = 0; }
    or_ch_observe(c, caps[1], cap_w);
  }
  uint64_t alphas[8];
  for (uint32_t i = 0; i < d->num_challenges; i++) alphas[i] = or_ch_challenge(c);
  or_ch_observe(c, caps[n_tr], cap_w);
  gl2_t zeta;
  zeta.c[0] = or_ch_challenge(c);
  zeta.c[1] = or_ch_challenge(c);
  /* constraints at zeta */
  const size_t nc = or_air_num_constraints(d->constraints);
  uint64_t *cv = malloc((nc ? nc : 1) * 16), *pe = malloc((d->n_public ? d->n_public : 1) * 16), *ge = malloc((d->n_global ? d->n_global : 1) * 16);
  uint32_t *kinds = malloc((nc ? nc : 1) * 4);
  for (uint32_t i = 0; i < d->n_public; i++) { pe[2 * i] = publics[i]; pe[2 * i + 1] = 0; }
  for (uint32_t i = 0; i < d->n_global; i++) { ge[2 * i] = globals[i]; ge[2 * i + 1] = 0; }
  or_air_eval_ext(d->constraints, loc, nxt, pe, ge, rch, cv, kinds);
  const uint64_t g = gl_root_of_unity(db), g_last = gl_pow(g, n - 1), n_inv = gl_inv((uint64_t)n % GL_P);
  const gl2_t zn = gl2_pow(zeta, n), zh = gl2_sub(zn, gl2_from_base(1)), z_last = gl2_sub(zeta, gl2_from_base(g_last));
  const gl2_t l_first = gl2_mul(gl2_scale(zh, n_inv), gl2_inv(gl2_sub(zeta, gl2_from_base(1))));
  const gl2_t l_last = gl2_mul(gl2_scale(zh, gl_mul(n_inv, g_last)), gl2_inv(z_last));
  int rc = 0;
  for (uint32_t a = 0; a < d->num_challenges && rc == 0; a++) {
    gl2_t acc = gl2_make(0, 0);
    for (size_t k = 0; k < nc; k++) {
      gl2_t v = gl2_make(cv[2 * k], cv[2 * k + 1]);
      if (kinds[k] == AIR_ASSERT_ZERO_TRANSITION) v = gl2_mul(v, z_last);
      else if (kinds[k] == AIR_ASSERT_ZERO_FIRST_ROW) v = gl2_mul(v, l_first);
      else if (kinds[k] == AIR_ASSERT_ZERO_LAST_ROW) v = gl2_mul(v, l_last);
      acc = gl2_add(gl2_scale(acc, alphas[a]), v);
    }
    gl2_t t = gl2_make(0, 0);
    for (size_t k = (size_t)1 << qdb; k-- > 0;) t = gl2_add(gl2_mul(t, zn), gl2_make(qz[2 * ((a << qdb) + k)], qz[2 * ((a << qdb) + k) + 1]));
    if (!gl2_eq(gl2_mul(t, zh), acc)) rc = 2;
  }
  if (rc == 0) {
    or_ch_observe(c, lq, 2 * (kt + kq));
    or_ch_observe(c, nxt, 2 * kt);
    const size_t n_or = n_tr + 1;
    const uint32_t qi = (uint32_t)n_tr;
    uint32_t np[3] = {(uint32_t)k0, (uint32_t)(k1 ? k1 : kq), (uint32_t)kq}, bl[3] = {0, 0, 0};
    or_fri_range r0[3] = {{0, 0, (uint32_t)k0}, {1, 0, (uint32_t)k1}, {qi, 0, (uint32_t)kq}}, r1[2] = {{0, 0, (uint32_t)k0}, {1, 0, (uint32_t)k1}};
    if (!k1) r0[1] = r0[2];
    or_fri_batch fb[2];
    const uint64_t zn2[2] = {gl_mul(zeta.c[0], g), gl_mul(zeta.c[1], g)};
    memcpy(fb[0].point, zeta.c, 16);
    fb[0].ranges = r0; fb[0].n_ranges = k1 ? 3 : 2;
    memcpy(fb[1].point, zn2, 16);
    fb[1].ranges = r1; fb[1].n_ranges = k1 ? 2 : 1;
    const uint64_t *opened[2] = {lq, nxt};
    const int frc = or_fri_verify(&d->fri, np, bl, n_or, caps, fb, 2, opened, c, proof + r.o, len - r.o, NULL);
    if (frc) rc = frc < 0 ? frc : -1000 - frc;
  }
  free(lq); free(rch); free(cv); free(pe); free(ge); free(kinds);
  return rc;
}
