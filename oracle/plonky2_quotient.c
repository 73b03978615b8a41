/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see goldilocks.h).
 *
 * A8: the quotient polynomials and the verifier-side vanishing
 * check, restated from plonky2 0.2.2 `compute_quotient_polys` / `eval_vanishing_poly{,_base_batch}` /
 * `check_partial_products` / `compute_filter` (un-vendored dependency, QEDProtocol/plonky2-hwa @
 * 6a8ca008). PARITY UNPINNED by the reference tree: no fixture under /root/reference carries verifier
 * data for a proof, so nothing there can check a quotient value; this file is validated by its own
 * verifier side (the openings of a proof must satisfy Z_H(zeta) * sum zeta^(n i) t_i(zeta) == vanishing(zeta)),
 * which is an independent formula path, and the GPU path is compared bit for bit with this file.
 *
 * Gate set restated here (constraint formulas of the upstream gates named in
 * city_common_circuit/src/builder/pad_circuit.rs:31-55): Noop, Constant, PublicInput, Arithmetic, Poseidon, and the in-tree
 * Comparison / U32Arithmetic / U32RangeCheck gates here; every other gate of the set in plonky2_gates.c.
 */
#include "cityoracle.h"
#include "gates_internal.h"
#include "goldilocks.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define UNUSED_SELECTOR 0xFFFFFFFFULL /* u32::MAX */

typedef struct {
  const or_shape *sh;
  const or_gates *g;
  gl2_t pi_hash[4];
} vctx_t;

static int gate_num_constraints(const or_gate *g) {
  switch (g->type) {
    case OR_GATE_NOOP: return 0;
    case OR_GATE_CONSTANT: return g->param;
    case OR_GATE_PUBLIC_INPUT: return 4;
    case OR_GATE_ARITHMETIC: return g->param;
    case OR_GATE_POSEIDON: return 123;
    case OR_GATE_COMPARISON: return 6 + 5 * g->param2 + (g->param + g->param2 - 1) / g->param2; /* comparison.rs:310-312 */
    case OR_GATE_U32_ARITHMETIC: return g->param * 36;  /* arithmetic_u32.rs:269-271 */
    case OR_GATE_U32_RANGE_CHECK: return g->param * 17; /* range_check_u32.rs:158-160 */
    default: return or_extra_gate_num_constraints(g); /* plonky2_gates.c */
  }
}

static gl2_t range_prod(gl2_t v, int count) { /* prod_{x<count} (v - x) */
  gl2_t p = v;
  for (int x = 1; x < count; x++) p = gl2_mul(p, gl2_sub(v, gl2_from_base((uint64_t)x)));
  return p;
}

/* ComparisonGate — city_common_circuit/src/u32/gates/comparison.rs:96-200 */
static void comparison_gate_eval(const or_gate *g, const gl2_t *w, gl2_t *out) {
  const int nch = g->param2, cb = (g->param + nch - 1) / nch, csz = 1 << cb;
  const gl2_t one = gl2_from_base(1), base = gl2_from_base((uint64_t)csz);
  int c = 0;
  gl2_t fc = gl2_from_base(0), sc = gl2_from_base(0);
  for (int i = nch - 1; i >= 0; i--) { fc = gl2_add(gl2_mul(fc, base), w[4 + i]); sc = gl2_add(gl2_mul(sc, base), w[4 + nch + i]); }
  out[c++] = gl2_sub(fc, w[0]);
  out[c++] = gl2_sub(sc, w[1]);
  gl2_t msd = gl2_from_base(0);
  for (int i = 0; i < nch; i++) {
    gl2_t f = w[4 + i], s = w[4 + nch + i];
    out[c++] = range_prod(f, csz);
    out[c++] = range_prod(s, csz);
    gl2_t diff = gl2_sub(s, f), dummy = w[4 + 2 * nch + i], eq = w[4 + 3 * nch + i];
    out[c++] = gl2_sub(gl2_mul(diff, dummy), gl2_sub(one, eq));
    out[c++] = gl2_mul(eq, diff);
    gl2_t inter = w[4 + 4 * nch + i];
    out[c++] = gl2_sub(inter, gl2_mul(eq, msd));
    msd = gl2_add(inter, gl2_mul(gl2_sub(one, eq), diff));
  }
  out[c++] = gl2_sub(w[3], msd);
  gl2_t bits = gl2_from_base(0);
  for (int i = cb; i >= 0; i--) bits = gl2_add(gl2_add(bits, bits), w[4 + 5 * nch + i]);
  for (int i = 0; i <= cb; i++) out[c++] = gl2_mul(w[4 + 5 * nch + i], gl2_sub(one, w[4 + 5 * nch + i]));
  out[c++] = gl2_sub(gl2_add(base, w[3]), bits);
  out[c++] = gl2_sub(w[2], w[4 + 5 * nch + cb]);
}

/* U32ArithmeticGate — city_common_circuit/src/u32/gates/arithmetic_u32.rs:90-150 */
static void u32_arithmetic_gate_eval(const or_gate *g, const gl2_t *w, gl2_t *out) {
  const int nops = g->param;
  const gl2_t one = gl2_from_base(1), four = gl2_from_base(4);
  int c = 0;
  for (int i = 0; i < nops; i++) {
    gl2_t m0 = w[6 * i], m1 = w[6 * i + 1], addend = w[6 * i + 2], lo = w[6 * i + 3], hi = w[6 * i + 4], inv = w[6 * i + 5];
    gl2_t computed = gl2_add(gl2_mul(m0, m1), addend);
    gl2_t diff = gl2_sub(gl2_from_base(0xFFFFFFFFULL), hi);
    out[c++] = gl2_mul(gl2_sub(gl2_mul(inv, diff), one), lo);
    out[c++] = gl2_sub(gl2_add(gl2_scale(hi, 1ULL << 32), lo), computed);
    gl2_t cl = gl2_from_base(0), chh = gl2_from_base(0);
    for (int j = 31; j >= 0; j--) {
      gl2_t limb = w[6 * nops + 32 * i + j];
      out[c++] = range_prod(limb, 4);
      if (j < 16) cl = gl2_add(gl2_mul(four, cl), limb); else chh = gl2_add(gl2_mul(four, chh), limb);
    }
    out[c++] = gl2_sub(cl, lo);
    out[c++] = gl2_sub(chh, hi);
  }
}

/* U32RangeCheckGate — city_common_circuit/src/u32/gates/range_check_u32.rs:57-80 */
static void u32_range_check_gate_eval(const or_gate *g, const gl2_t *w, gl2_t *out) {
  const int n = g->param;
  const gl2_t four = gl2_from_base(4);
  int c = 0;
  for (int i = 0; i < n; i++) {
    gl2_t sum = gl2_from_base(0);
    for (int j = 15; j >= 0; j--) sum = gl2_add(gl2_mul(sum, four), w[n + 16 * i + j]);
    out[c++] = gl2_sub(sum, w[i]);
    for (int j = 0; j < 16; j++) out[c++] = range_prod(w[n + 16 * i + j], 4);
  }
}

/* ---- PoseidonGate (plonky2 gates/poseidon.rs; named in city_common_circuit/src/builder/pad_circuit.rs:31-55).
 * Wires: inputs 0..11, outputs 12..23, swap 24, delta 25..28, S-box inputs of full rounds 1..3 at 29..64,
 * of the 22 partial rounds at 65..86, of the last 4 full rounds at 87..134. 123 constraints, in this order:
 * swap boolean; 4 delta definitions; 36 + 22 + 48 S-box-input anchors; 12 outputs.
 * The gate is evaluated in the textbook round structure (full constant vector + full MDS every round);
 * upstream uses its sparse partial-round factorisation, which is the same polynomial map between anchors. */
static uint64_t PG_RC[360], PG_CIRC[12], PG_DIAG[12];
static int pg_ready = 0;
static void pg_init(void) {
  if (pg_ready) return;
  or_poseidon_round_constants(PG_RC);
  or_poseidon_mds(PG_CIRC, PG_DIAG);
  pg_ready = 1;
}
static gl2_t ext_pow7(gl2_t x) { gl2_t x2 = gl2_mul(x, x), x4 = gl2_mul(x2, x2), x3 = gl2_mul(x, x2); return gl2_mul(x3, x4); }
static void ext_mds(gl2_t s[12]) {
  gl2_t o[12];
  for (int r = 0; r < 12; r++) {
    gl2_t acc = gl2_from_base(0);
    for (int i = 0; i < 12; i++) acc = gl2_add(acc, gl2_scale(s[(i + r) % 12], PG_CIRC[i]));
    acc = gl2_add(acc, gl2_scale(s[r], PG_DIAG[r]));
    o[r] = acc;
  }
  memcpy(s, o, sizeof o);
}
static void poseidon_gate_eval(const gl2_t *w, gl2_t *out) {
  pg_init();
  int c = 0;
  gl2_t swap = w[24];
  out[c++] = gl2_mul(swap, gl2_sub(swap, gl2_from_base(1)));
  for (int i = 0; i < 4; i++) out[c++] = gl2_sub(gl2_mul(swap, gl2_sub(w[i + 4], w[i])), w[25 + i]);
  gl2_t st[12];
  for (int i = 0; i < 4; i++) { st[i] = gl2_add(w[i], w[25 + i]); st[i + 4] = gl2_sub(w[i + 4], w[25 + i]); }
  for (int i = 8; i < 12; i++) st[i] = w[i];
  int rnd = 0;
  for (int r = 0; r < 4; r++, rnd++) {
    for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from_base(PG_RC[rnd * 12 + i]));
    if (r != 0)
      for (int i = 0; i < 12; i++) { gl2_t in = w[29 + 12 * (r - 1) + i]; out[c++] = gl2_sub(st[i], in); st[i] = in; }
    for (int i = 0; i < 12; i++) st[i] = ext_pow7(st[i]);
    ext_mds(st);
  }
  for (int r = 0; r < 22; r++, rnd++) {
    for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from_base(PG_RC[rnd * 12 + i]));
    gl2_t in = w[65 + r];
    out[c++] = gl2_sub(st[0], in);
    st[0] = ext_pow7(in);
    ext_mds(st);
  }
  for (int r = 0; r < 4; r++, rnd++) {
    for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from_base(PG_RC[rnd * 12 + i]));
    for (int i = 0; i < 12; i++) { gl2_t in = w[87 + 12 * r + i]; out[c++] = gl2_sub(st[i], in); st[i] = in; }
    for (int i = 0; i < 12; i++) st[i] = ext_pow7(st[i]);
    ext_mds(st);
  }
  for (int i = 0; i < 12; i++) out[c++] = gl2_sub(st[i], w[12 + i]);
}

int or_gates_num_constraints(const or_gates *g) {
  int m = 0;
  for (int i = 0; i < g->n_gates; i++) {
    int c = gate_num_constraints(&g->gates[i]);
    if (c < 0 || c > 512) return -1;
    if (c > m) m = c;
  }
  return m;
}

/* unfiltered constraints of one gate; consts = local constants AFTER the selector prefix */
static void gate_eval(const or_gate *g, const gl2_t *consts, const gl2_t *wires, const gl2_t *pi_hash, gl2_t *out) {
  switch (g->type) {
    case OR_GATE_CONSTANT:
      for (int i = 0; i < g->param; i++) out[i] = gl2_sub(consts[i], wires[i]);
      break;
    case OR_GATE_PUBLIC_INPUT:
      for (int i = 0; i < 4; i++) out[i] = gl2_sub(wires[i], pi_hash[i]);
      break;
    case OR_GATE_ARITHMETIC:
      for (int i = 0; i < g->param; i++) {
        gl2_t m0 = wires[4 * i], m1 = wires[4 * i + 1], addend = wires[4 * i + 2], output = wires[4 * i + 3];
        gl2_t computed = gl2_add(gl2_mul(gl2_mul(m0, m1), consts[0]), gl2_mul(addend, consts[1]));
        out[i] = gl2_sub(output, computed);
      }
      break;
    case OR_GATE_POSEIDON:
      poseidon_gate_eval(wires, out);
      break;
    case OR_GATE_COMPARISON:
      comparison_gate_eval(g, wires, out);
      break;
    case OR_GATE_U32_ARITHMETIC:
      u32_arithmetic_gate_eval(g, wires, out);
      break;
    case OR_GATE_U32_RANGE_CHECK:
      u32_range_check_gate_eval(g, wires, out);
      break;
    default:
      or_extra_gate_eval(g, consts, wires, out); /* plonky2_gates.c */
      break;
  }
}

/* filter of gate `row` inside its selector group: prod_{i in group, i != row} (i - s) [* (UNUSED - s)] */
static gl2_t gate_filter(int row, const or_gate *g, gl2_t s, int many_selectors) {
  gl2_t f = gl2_from_base(1);
  for (int i = g->group_start; i < g->group_end; i++)
    if (i != row) f = gl2_mul(f, gl2_sub(gl2_from_base((uint64_t)i), s));
  if (many_selectors) f = gl2_mul(f, gl2_sub(gl2_from_base(UNUSED_SELECTOR), s));
  return f;
}

/* vanishing(x) for every challenge, at one point. All inputs in F_p^2 (the prover embeds base values).
 * consts: all num_constants "constants" openings (selectors first); l0 = L_0(x); x = the point itself. */
static void eval_vanishing(const vctx_t *v, gl2_t x, gl2_t l0, const gl2_t *consts, const gl2_t *wires,
                           const gl2_t *zs, const gl2_t *zs_next, const gl2_t *pps, const gl2_t *sigmas,
                           const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas, gl2_t *out) {
  const or_shape *sh = v->sh;
  const or_gates *G = v->g;
  const int nc = sh->num_challenges, R = sh->num_routed_wires, chunk = sh->quotient_degree_factor,
            npp = sh->num_partial_products, ngc = or_gates_num_constraints(G);
  int n_terms = nc + nc * (npp + 1) + ngc;
  gl2_t *terms = (gl2_t *)calloc((size_t)n_terms, sizeof(gl2_t));
  int t = 0;
  /* L_0(x) (Z(x) - 1) */
  for (int c = 0; c < nc; c++) terms[t++] = gl2_mul(l0, gl2_sub(zs[c], gl2_from_base(1)));
  /* partial-product checks */
  for (int c = 0; c < nc; c++) {
    for (int k = 0; k <= npp; k++) {
      gl2_t prev = k == 0 ? zs[c] : pps[c * npp + k - 1];
      gl2_t next = k == npp ? zs_next[c] : pps[c * npp + k];
      gl2_t pn = gl2_from_base(1), pd = gl2_from_base(1);
      for (int j = k * chunk; j < R && j < (k + 1) * chunk; j++) {
        gl2_t sid = gl2_scale(x, G->k_is[j]);
        gl2_t num = gl2_add(gl2_add(wires[j], gl2_scale(sid, betas[c])), gl2_from_base(gammas[c]));
        gl2_t den = gl2_add(gl2_add(wires[j], gl2_scale(sigmas[j], betas[c])), gl2_from_base(gammas[c]));
        pn = gl2_mul(pn, num);
        pd = gl2_mul(pd, den);
      }
      terms[t++] = gl2_sub(gl2_mul(prev, pn), gl2_mul(next, pd));
    }
  }
  /* gate constraints: filtered, summed per constraint index */
  gl2_t tmp[512];
  for (int gi = 0; gi < G->n_gates; gi++) {
    const or_gate *g = &G->gates[gi];
    int c = gate_num_constraints(g);
    if (c <= 0) continue;
    gl2_t f = gate_filter(gi, g, consts[g->selector_index], G->num_selectors > 1);
    gate_eval(g, consts + G->num_selectors, wires, v->pi_hash, tmp);
    for (int i = 0; i < c; i++) terms[t + i] = gl2_add(terms[t + i], gl2_mul(tmp[i], f));
  }
  /* reduce_with_powers_multi: sum_i alpha^i term_i */
  for (int c = 0; c < nc; c++) {
    gl2_t acc = gl2_from_base(0);
    for (int i = n_terms - 1; i >= 0; i--) acc = gl2_add(gl2_scale(acc, alphas[c]), terms[i]);
    out[c] = acc;
  }
  free(terms);
}

static size_t bitrev_sz(size_t x, int bits) { size_t r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

/* Quotient chunk polynomials (coefficients), num_challenges*quotient_degree_factor x n.
 * cs_lde / wires_lde / zs_lde: bit-reversed LDEs on 7*<omega_N>, poly-major (as or_commit_batch returns). */
int or_quotient_polys(const or_shape *sh, const or_gates *G, const uint64_t pi_hash[4], const uint64_t *cs_lde,
                      const uint64_t *wires_lde, const uint64_t *zs_lde, const uint64_t *betas,
                      const uint64_t *gammas, const uint64_t *alphas, uint64_t *out_coeffs) {
  const int db = sh->degree_bits, rb = sh->rate_bits, nc = sh->num_challenges;
  if ((1 << rb) != sh->quotient_degree_factor) return -1; /* step = 1 only (the configuration in use) */
  if (or_gates_num_constraints(G) < 0) return -2;
  const size_t n = (size_t)1 << db, N = n << rb;
  const int R = sh->num_routed_wires, ncst = sh->num_constants, W = sh->num_wires, npp = sh->num_partial_products;
  vctx_t v = {sh, G, {{{pi_hash[0], 0}}, {{pi_hash[1], 0}}, {{pi_hash[2], 0}}, {{pi_hash[3], 0}}}};
  /* Z_H on the coset takes 2^rb values: g^n * w^i - 1, w = omega_{2^rb} */
  uint64_t gpn = gl_pow(GL_GENERATOR, n), w8 = gl_root_of_unity(rb), zh_inv[64], zh[64];
  for (int i = 0; i < (1 << rb); i++) { zh[i] = gl_sub(gl_mul(gpn, gl_pow(w8, i)), 1); zh_inv[i] = gl_inv(zh[i]); }
  uint64_t omegaN = gl_root_of_unity(db + rb);
  uint64_t *vals = (uint64_t *)malloc((size_t)nc * N * 8); /* natural order */
  /* points are independent: the same axis rayon splits upstream (or_set_threads workers, contiguous ranges) */
#pragma omp parallel num_threads(or_get_threads())
  {
    gl2_t *consts = (gl2_t *)malloc((size_t)(ncst + W + R + 4 * nc + nc * npp) * sizeof(gl2_t));
    gl2_t *wires = consts + ncst, *sig = wires + W, *zs = sig + R, *zsn = zs + nc, *pps = zsn + nc;
    int nt = 1, tid = 0;
#ifdef _OPENMP
    nt = omp_get_num_threads();
    tid = omp_get_thread_num();
#endif
    const size_t lo = N * (size_t)tid / (size_t)nt, hi = N * (size_t)(tid + 1) / (size_t)nt;
    uint64_t x = gl_mul(GL_GENERATOR, gl_pow(omegaN, lo)); /* coset point of natural index i */
    for (size_t i = lo; i < hi; i++) {
      size_t s = bitrev_sz(i, db + rb), sn = bitrev_sz((i + ((size_t)1 << rb)) % N, db + rb);
      for (int j = 0; j < ncst; j++) consts[j] = gl2_from_base(cs_lde[(size_t)j * N + s]);
      for (int j = 0; j < R; j++) sig[j] = gl2_from_base(cs_lde[(size_t)(ncst + j) * N + s]);
      for (int j = 0; j < W; j++) wires[j] = gl2_from_base(wires_lde[(size_t)j * N + s]);
      for (int c = 0; c < nc; c++) { zs[c] = gl2_from_base(zs_lde[(size_t)c * N + s]); zsn[c] = gl2_from_base(zs_lde[(size_t)c * N + sn]); }
      for (int j = 0; j < nc * npp; j++) pps[j] = gl2_from_base(zs_lde[(size_t)(nc + j) * N + s]);
      /* L_0(x) = Z_H(x) / (n (x - 1)) */
      uint64_t l0 = gl_mul(zh[i & ((1 << rb) - 1)], gl_inv(gl_mul((uint64_t)n, gl_sub(x, 1))));
      gl2_t res[8];
      eval_vanishing(&v, gl2_from_base(x), gl2_from_base(l0), consts, wires, zs, zsn, pps, sig, betas, gammas, alphas, res);
      for (int c = 0; c < nc; c++) vals[(size_t)c * N + i] = gl_mul(res[c].c[0], zh_inv[i & ((1 << rb) - 1)]);
      x = gl_mul(x, omegaN);
    }
    free(consts);
  }
  /* coset iFFT: values on 7*<omega_N> -> coefficients; chunk j of challenge c = coeffs[j*n .. (j+1)*n) */
  uint64_t ginv = gl_inv(GL_GENERATOR);
  for (int c = 0; c < nc; c++) {
    uint64_t *p = vals + (size_t)c * N;
    or_intt(p, db + rb);
    uint64_t sp = 1;
    for (size_t i = 0; i < N; i++) { p[i] = gl_mul(p[i], sp); sp = gl_mul(sp, ginv); }
    memcpy(out_coeffs + (size_t)c * N, p, N * 8);
  }
  free(vals);
  return 0;
}

/* Verifier side: with the openings of a proof and the challenges, check
 *   Z_H(zeta) * sum_i zeta^(n i) t_{c,i}(zeta) == vanishing_c(zeta)   for every challenge c.
 * openings (ext, 2 u64 each): constants[num_constants], sigmas[R], wires[W], zs[nc], zs_next[nc],
 * partial_products[nc*npp], quotient[nc*qdf]. Returns 0 when the identity holds. */
int or_check_vanishing(const or_shape *sh, const or_gates *G, const uint64_t pi_hash[4], const uint64_t zeta[2],
                       const uint64_t *op_constants, const uint64_t *op_sigmas, const uint64_t *op_wires,
                       const uint64_t *op_zs, const uint64_t *op_zs_next, const uint64_t *op_pps,
                       const uint64_t *op_quotient, const uint64_t *betas, const uint64_t *gammas,
                       const uint64_t *alphas) {
  const int nc = sh->num_challenges, qdf = sh->quotient_degree_factor;
  const size_t n = (size_t)1 << sh->degree_bits;
  vctx_t v = {sh, G, {{{pi_hash[0], 0}}, {{pi_hash[1], 0}}, {{pi_hash[2], 0}}, {{pi_hash[3], 0}}}};
  gl2_t z = gl2_make(zeta[0], zeta[1]);
  gl2_t zn = gl2_pow(z, n);
  gl2_t zh = gl2_sub(zn, gl2_from_base(1));
  gl2_t l0 = gl2_mul(zh, gl2_inv(gl2_scale(gl2_sub(z, gl2_from_base(1)), (uint64_t)n)));
  gl2_t res[8];
  eval_vanishing(&v, z, l0, (const gl2_t *)op_constants, (const gl2_t *)op_wires, (const gl2_t *)op_zs,
                 (const gl2_t *)op_zs_next, (const gl2_t *)op_pps, (const gl2_t *)op_sigmas, betas, gammas, alphas, res);
  for (int c = 0; c < nc; c++) {
    gl2_t acc = gl2_from_base(0);
    for (int i = qdf - 1; i >= 0; i--)
      acc = gl2_add(gl2_mul(acc, zn), gl2_make(op_quotient[2 * (c * qdf + i)], op_quotient[2 * (c * qdf + i) + 1]));
    if (!gl2_eq(gl2_mul(acc, zh), res[c])) return -(c + 1);
  }
  return 0;
}
