/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see goldilocks.h).
 *
 * A8, second half of the gate set: constraint polynomials `eval_unfiltered` of
 *   - the remaining in-tree city-rollup gates, following the reference source line by line:
 *       U32AddManyGate        city_common_circuit/src/u32/gates/add_many_u32.rs:93-140
 *       U32SubtractionGate    city_common_circuit/src/u32/gates/subtraction_u32.rs:89-125
 *       U32InterleaveGate     city_common_circuit/src/u32/gates/interleave_u32.rs:90-128
 *       UninterleaveToU32Gate city_common_circuit/src/u32/gates/uninterleave_to_u32.rs:82-130
 *       UninterleaveToB32Gate city_common_circuit/src/u32/gates/uninterleave_to_b32.rs:82-131
 *   - the remaining upstream gates of the city-common gate set (city_common_circuit/src/builder/pad_circuit.rs:31-55),
 *     restated from plonky2 0.2.2 (un-vendored, QEDProtocol/plonky2-hwa @ 6a8ca008): ArithmeticExtensionGate,
 *     MulExtensionGate, BaseSumGate<B>, RandomAccessGate, ReducingGate, ReducingExtensionGate, PoseidonMdsGate, ExponentiationGate,
 *     CosetInterpolationGate. PARITY UNPINNED for these eight (no reference data can check them; see
 *     plonky2_quotient.c header) — validated by prove -> verify round trips on satisfying witnesses and by rejection
 *     of corrupted ones.
 *
 * Written in the reference's own shape (wire-accessor functions + a constraint vector that is pushed to), on
 * purpose different from the product's fused formulation in csrc/gates.h.
 * All values are F_p^2 (the prover side embeds base-field values); a D=2 "extension algebra" element is a pair of
 * such values (a0, a1) standing for a0 + a1*X, X^2 = 7.
 */
#include "gates_internal.h"

#include <stdlib.h>

typedef struct { gl2_t c[2]; } alg_t; /* plonky2 ExtensionAlgebra<F::Extension, 2> */

static alg_t alg_get(const gl2_t *w, int start) { alg_t r = {{w[start], w[start + 1]}}; return r; }
static alg_t alg_add(alg_t x, alg_t y) { alg_t r = {{gl2_add(x.c[0], y.c[0]), gl2_add(x.c[1], y.c[1])}}; return r; }
static alg_t alg_sub(alg_t x, alg_t y) { alg_t r = {{gl2_sub(x.c[0], y.c[0]), gl2_sub(x.c[1], y.c[1])}}; return r; }
static alg_t alg_mul(alg_t x, alg_t y) {
  /* schoolbook product, then reduce X^2 -> W = 7 */
  gl2_t p0 = gl2_mul(x.c[0], y.c[0]), p1 = gl2_add(gl2_mul(x.c[0], y.c[1]), gl2_mul(x.c[1], y.c[0])),
        p2 = gl2_mul(x.c[1], y.c[1]);
  alg_t r = {{gl2_add(p0, gl2_scale(p2, 7)), p1}};
  return r;
}
static alg_t alg_scalar_mul(alg_t x, gl2_t s) { alg_t r = {{gl2_mul(x.c[0], s), gl2_mul(x.c[1], s)}}; return r; }
static alg_t alg_from_scalar(gl2_t s) { alg_t r = {{s, gl2_from_base(0)}}; return r; }

typedef struct { gl2_t *v; int n; } cvec; /* the `constraints` vector */
static void push(cvec *c, gl2_t x) { c->v[c->n++] = x; }
static void push_alg(cvec *c, alg_t x) { push(c, x.c[0]); push(c, x.c[1]); } /* to_basefield_array */

static gl2_t limb_product(gl2_t limb, int max_limb) { /* (0..max_limb).map(|x| limb - x).product() */
  gl2_t p = gl2_from_base(1);
  for (int x = 0; x < max_limb; x++) p = gl2_mul(p, gl2_sub(limb, gl2_from_base((uint64_t)x)));
  return p;
}
/* reduce_with_powers(terms, base) = sum terms[i] * base^i over terms given in iteration order */
static gl2_t reduce_with_powers_idx(const gl2_t *w, const int *idx, int n, uint64_t base) {
  gl2_t acc = gl2_from_base(0);
  for (int i = n - 1; i >= 0; i--) acc = gl2_add(gl2_scale(acc, base), w[idx[i]]);
  return acc;
}

/* ---- U32AddManyGate (add_many_u32.rs) ---- */
static void add_many(const or_gate *g, const gl2_t *w, cvec *out) {
  const int num_ops = g->param, num_addends = g->param2;
  const int num_result_limbs = 16, num_carry_limbs = 2, num_limbs = num_result_limbs + num_carry_limbs; /* :70-78 */
  for (int i = 0; i < num_ops; i++) {
    const int op = (num_addends + 3) * i; /* :47-66 */
    gl2_t computed_output = gl2_from_base(0);
    for (int j = 0; j < num_addends; j++) computed_output = gl2_add(computed_output, w[op + j]);
    computed_output = gl2_add(computed_output, w[op + num_addends]); /* + carry */
    gl2_t output_result = w[op + num_addends + 1], output_carry = w[op + num_addends + 2];
    gl2_t combined_output = gl2_add(gl2_scale(output_carry, 1ull << 32), output_result);
    push(out, gl2_sub(combined_output, computed_output));
    gl2_t combined_result_limbs = gl2_from_base(0), combined_carry_limbs = gl2_from_base(0);
    for (int j = num_limbs - 1; j >= 0; j--) {
      gl2_t this_limb = w[(num_addends + 3) * num_ops + num_limbs * i + j]; /* :80-84 */
      push(out, limb_product(this_limb, 4));
      if (j < num_result_limbs) combined_result_limbs = gl2_add(gl2_scale(combined_result_limbs, 4), this_limb);
      else combined_carry_limbs = gl2_add(gl2_scale(combined_carry_limbs, 4), this_limb);
    }
    push(out, gl2_sub(combined_result_limbs, output_result));
    push(out, gl2_sub(combined_carry_limbs, output_carry));
  }
}

/* ---- U32SubtractionGate (subtraction_u32.rs) ---- */
static void subtraction(const or_gate *g, const gl2_t *w, cvec *out) {
  const int num_ops = g->param, num_limbs = 16;
  for (int i = 0; i < num_ops; i++) {
    gl2_t input_x = w[5 * i], input_y = w[5 * i + 1], input_borrow = w[5 * i + 2];
    gl2_t result_initial = gl2_sub(gl2_sub(input_x, input_y), input_borrow);
    gl2_t output_result = w[5 * i + 3], output_borrow = w[5 * i + 4];
    push(out, gl2_sub(output_result, gl2_add(result_initial, gl2_scale(output_borrow, 1ull << 32))));
    gl2_t combined_limbs = gl2_from_base(0);
    for (int j = num_limbs - 1; j >= 0; j--) {
      gl2_t this_limb = w[5 * num_ops + num_limbs * i + j];
      push(out, limb_product(this_limb, 4));
      combined_limbs = gl2_add(gl2_scale(combined_limbs, 4), this_limb);
    }
    push(out, gl2_sub(combined_limbs, output_result));
    push(out, gl2_mul(output_borrow, gl2_sub(gl2_from_base(1), output_borrow)));
  }
}

/* ---- U32InterleaveGate (interleave_u32.rs): bit wires big-endian, so `bits.iter().rev()` is little-endian ---- */
static void interleave(const or_gate *g, const gl2_t *w, cvec *out) {
  const int num_ops = g->param, NUM_BITS = 32;
  int rev[64];
  for (int i = 0; i < num_ops; i++) {
    const int start = 2 * num_ops + NUM_BITS * i; /* wires_ith_bit_decomposition */
    for (int k = 0; k < NUM_BITS; k++) rev[k] = start + NUM_BITS - 1 - k;
    push(out, gl2_sub(reduce_with_powers_idx(w, rev, NUM_BITS, 2), w[2 * i]));
    push(out, gl2_sub(reduce_with_powers_idx(w, rev, NUM_BITS, 4), w[2 * i + 1]));
    for (int k = 0; k < NUM_BITS; k++) push(out, limb_product(w[start + k], 2));
  }
}

/* ---- UninterleaveToU32Gate / UninterleaveToB32Gate ---- */
static void uninterleave(const or_gate *g, const gl2_t *w, cvec *out, int to_b32) {
  const int num_ops = g->param, NUM_BITS = 64;
  int rev[64];
  for (int i = 0; i < num_ops; i++) {
    const int start = 3 * num_ops + NUM_BITS * i;
    for (int k = 0; k < NUM_BITS; k++) rev[k] = start + NUM_BITS - 1 - k;
    push(out, gl2_sub(reduce_with_powers_idx(w, rev, NUM_BITS, 2), w[3 * i]));
    gl2_t computed_x_evens = gl2_from_base(0), computed_x_odds = gl2_from_base(0);
    for (int j = 0; j < NUM_BITS / 2; j++) {
      const int e = NUM_BITS / 2 - j - 1;
      const uint64_t coeff = to_b32 ? 1ull << (2 * e) : 1ull << e;
      computed_x_evens = gl2_add(computed_x_evens, gl2_scale(w[start + 2 * j], coeff));
      computed_x_odds = gl2_add(computed_x_odds, gl2_scale(w[start + 2 * j + 1], coeff));
    }
    push(out, gl2_sub(computed_x_evens, w[3 * i + 1]));
    push(out, gl2_sub(computed_x_odds, w[3 * i + 2]));
    for (int k = 0; k < NUM_BITS; k++) push(out, limb_product(w[start + k], 2));
  }
}

/* ---- ArithmeticExtensionGate / MulExtensionGate ---- */
static void arithmetic_extension(const or_gate *g, const gl2_t *consts, const gl2_t *w, cvec *out) {
  for (int i = 0; i < g->param; i++) {
    alg_t multiplicand_0 = alg_get(w, 8 * i), multiplicand_1 = alg_get(w, 8 * i + 2), addend = alg_get(w, 8 * i + 4),
          output = alg_get(w, 8 * i + 6);
    alg_t computed_output = alg_add(alg_scalar_mul(alg_mul(multiplicand_0, multiplicand_1), consts[0]),
                                    alg_scalar_mul(addend, consts[1]));
    push_alg(out, alg_sub(output, computed_output));
  }
}
static void mul_extension(const or_gate *g, const gl2_t *consts, const gl2_t *w, cvec *out) {
  for (int i = 0; i < g->param; i++) {
    alg_t multiplicand_0 = alg_get(w, 6 * i), multiplicand_1 = alg_get(w, 6 * i + 2), output = alg_get(w, 6 * i + 4);
    push_alg(out, alg_sub(output, alg_scalar_mul(alg_mul(multiplicand_0, multiplicand_1), consts[0])));
  }
}

/* ---- BaseSumGate<B>: WIRE_SUM = 0, START_LIMBS = 1 ---- */
static void base_sum(const or_gate *g, const gl2_t *w, cvec *out) {
  const int num_limbs = g->param, B = g->param2;
  int *idx = (int *)malloc(sizeof(int) * (size_t)num_limbs);
  for (int i = 0; i < num_limbs; i++) idx[i] = 1 + i;
  push(out, gl2_sub(reduce_with_powers_idx(w, idx, num_limbs, (uint64_t)B), w[0]));
  for (int i = 0; i < num_limbs; i++) push(out, limb_product(w[1 + i], B));
  free(idx);
}

/* ---- RandomAccessGate { bits, num_copies, num_extra_constants } ---- */
static void random_access(const or_gate *g, const gl2_t *consts, const gl2_t *w, cvec *out) {
  const int bits = g->param, num_copies = g->param2, num_extra_constants = g->param3, vec_size = 1 << bits;
  const int start_extra_constants = (2 + vec_size) * num_copies, num_routed_wires = start_extra_constants + num_extra_constants;
  gl2_t *list_items = (gl2_t *)malloc(sizeof(gl2_t) * (size_t)vec_size);
  for (int copy = 0; copy < num_copies; copy++) {
    gl2_t access_index = w[(2 + vec_size) * copy], claimed_element = w[(2 + vec_size) * copy + 1];
    for (int i = 0; i < vec_size; i++) list_items[i] = w[(2 + vec_size) * copy + 2 + i];
    const gl2_t *bit = w + num_routed_wires + copy * bits; /* wire_bit(i, copy) */
    for (int i = 0; i < bits; i++) push(out, gl2_mul(bit[i], gl2_sub(bit[i], gl2_from_base(1))));
    gl2_t reconstructed_index = gl2_from_base(0);
    for (int i = bits - 1; i >= 0; i--) reconstructed_index = gl2_add(gl2_add(reconstructed_index, reconstructed_index), bit[i]);
    push(out, gl2_sub(reconstructed_index, access_index));
    int len = vec_size;
    for (int i = 0; i < bits; i++) { /* fold pairs: x + b*(y - x) */
      for (int k = 0; k < len / 2; k++)
        list_items[k] = gl2_add(list_items[2 * k], gl2_mul(bit[i], gl2_sub(list_items[2 * k + 1], list_items[2 * k])));
      len /= 2;
    }
    push(out, gl2_sub(list_items[0], claimed_element));
  }
  for (int i = 0; i < num_extra_constants; i++) push(out, gl2_sub(consts[i], w[start_extra_constants + i]));
  free(list_items);
}

/* ---- ReducingGate / ReducingExtensionGate (D = 2): output 0..2, alpha 2..4, old_acc 4..6, coeffs from 6 ---- */
static void reducing(const or_gate *g, const gl2_t *w, cvec *out, int ext_coeffs) {
  const int num_coeffs = g->param, D = 2, START_COEFFS = 3 * D;
  const int start_accs = START_COEFFS + (ext_coeffs ? D * num_coeffs : num_coeffs);
  alg_t alpha = alg_get(w, D), acc = alg_get(w, 2 * D);
  for (int i = 0; i < num_coeffs; i++) {
    alg_t coeff = ext_coeffs ? alg_get(w, START_COEFFS + D * i) : alg_from_scalar(w[START_COEFFS + i]);
    alg_t accs_i = i == num_coeffs - 1 ? alg_get(w, 0) : alg_get(w, start_accs + D * i); /* wires_accs(i) */
    push_alg(out, alg_sub(alg_add(alg_mul(acc, alpha), coeff), accs_i));
    acc = accs_i;
  }
}

/* ---- PoseidonMdsGate: inputs i at wires 2i.., outputs at 24 + 2i.. ---- */
static void poseidon_mds(const gl2_t *w, cvec *out) {
  uint64_t circ[12], diag[12];
  or_poseidon_mds(circ, diag);
  for (int r = 0; r < 12; r++) {
    alg_t res = alg_from_scalar(gl2_from_base(0));
    for (int i = 0; i < 12; i++) res = alg_add(res, alg_scalar_mul(alg_get(w, 2 * ((i + r) % 12)), gl2_from_base(circ[i])));
    res = alg_add(res, alg_scalar_mul(alg_get(w, 2 * r), gl2_from_base(diag[r])));
    push_alg(out, alg_sub(alg_get(w, 24 + 2 * r), res));
  }
}

/* ---- CosetInterpolationGate { subgroup_bits, degree } ---- */
typedef struct { alg_t eval, prod; } interp_t;
static interp_t partial_interpolate(const uint64_t *domain, const alg_t *values, const uint64_t *weights, int from, int to,
                                    alg_t point, interp_t st) {
  for (int i = from; i < to; i++) {
    alg_t term = alg_sub(point, alg_from_scalar(gl2_from_base(domain[i])));
    alg_t next_eval = alg_add(alg_mul(st.eval, term), alg_mul(values[i], alg_scalar_mul(st.prod, gl2_from_base(weights[i]))));
    st.prod = alg_mul(st.prod, term);
    st.eval = next_eval;
  }
  return st;
}
static void coset_interpolation(const or_gate *g, const gl2_t *w, cvec *out) {
  const int subgroup_bits = g->param, degree = g->param2, D = 2, num_points = 1 << subgroup_bits;
  const int num_intermediates = (num_points - 2) / (degree - 1);
  const int start_values = 1, start_evaluation_point = start_values + num_points * D,
            start_evaluation_value = start_evaluation_point + D, start_intermediates = start_evaluation_value + D,
            end_intermediates = start_intermediates + D * 2 * num_intermediates;
  uint64_t domain[64], weights[64];
  alg_t values[64];
  /* two_adic_subgroup + barycentric_weights by their definition: w_i = 1 / prod_{j != i} (x_i - x_j) */
  uint64_t gen = gl_root_of_unity(subgroup_bits);
  domain[0] = 1;
  for (int i = 1; i < num_points; i++) domain[i] = gl_mul(domain[i - 1], gen);
  for (int i = 0; i < num_points; i++) {
    uint64_t p = 1;
    for (int j = 0; j < num_points; j++)
      if (j != i) p = gl_mul(p, gl_sub(domain[i], domain[j]));
    weights[i] = gl_inv(p);
    values[i] = alg_get(w, start_values + i * D);
  }
  gl2_t shift = w[0];
  alg_t evaluation_point = alg_get(w, start_evaluation_point), shifted_evaluation_point = alg_get(w, end_intermediates);
  push_alg(out, alg_sub(evaluation_point, alg_scalar_mul(shifted_evaluation_point, shift)));
  interp_t st = {alg_from_scalar(gl2_from_base(0)), alg_from_scalar(gl2_from_base(1))};
  st = partial_interpolate(domain, values, weights, 0, degree, shifted_evaluation_point, st);
  for (int i = 0; i < num_intermediates; i++) {
    alg_t intermediate_eval = alg_get(w, start_intermediates + D * i),
          intermediate_prod = alg_get(w, start_intermediates + D * (num_intermediates + i));
    push_alg(out, alg_sub(intermediate_eval, st.eval));
    push_alg(out, alg_sub(intermediate_prod, st.prod));
    int start_index = 1 + (degree - 1) * (i + 1);
    int end_index = start_index + degree - 1 < num_points ? start_index + degree - 1 : num_points;
    interp_t from_wires = {intermediate_eval, intermediate_prod};
    st = partial_interpolate(domain, values, weights, start_index, end_index, shifted_evaluation_point, from_wires);
  }
  push_alg(out, alg_sub(alg_get(w, start_evaluation_value), st.eval));
}

/* ---- ExponentiationGate { num_power_bits } (plonky2 0.2.2 gates/exponentiation.rs, UPSTREAM-MEMORY):
 * wire_base = 0, wire_power_bit(i) = 1 + i, wire_output = 1 + n, wire_intermediate_value(i) = 2 + n + i ---- */
static void exponentiation(const or_gate *g, const gl2_t *w, cvec *out) {
  const int num_power_bits = g->param;
  const gl2_t base = w[0], one = gl2_from_base(1);
  for (int i = 0; i < num_power_bits; i++) {
    gl2_t prev_intermediate_value = i == 0 ? one : gl2_mul(w[2 + num_power_bits + i - 1], w[2 + num_power_bits + i - 1]);
    /* power_bits is in LE order, but we accumulate in BE order */
    gl2_t cur_bit = w[1 + (num_power_bits - i - 1)];
    gl2_t not_cur_bit = gl2_sub(one, cur_bit);
    gl2_t computed_intermediate_value = gl2_mul(prev_intermediate_value, gl2_add(gl2_mul(cur_bit, base), not_cur_bit));
    push(out, gl2_sub(computed_intermediate_value, w[2 + num_power_bits + i]));
  }
  push(out, gl2_sub(w[1 + num_power_bits], w[2 + num_power_bits + num_power_bits - 1]));
}

int or_extra_gate_num_constraints(const or_gate *g) {
  switch (g->type) {
    case OR_GATE_U32_ADD_MANY: return g->param * (3 + 18);   /* add_many_u32.rs:266-268 */
    case OR_GATE_U32_SUBTRACTION: return g->param * (3 + 16); /* subtraction_u32.rs:215-217 */
    case OR_GATE_U32_INTERLEAVE: return g->param * (32 + 1 + 1); /* interleave_u32.rs:214-216 */
    case OR_GATE_UNINTERLEAVE_TO_U32:
    case OR_GATE_UNINTERLEAVE_TO_B32: return g->param * (64 + 1 + 2); /* uninterleave_to_u32.rs:246-248 */
    case OR_GATE_ARITHMETIC_EXT:
    case OR_GATE_MUL_EXT: return g->param * 2;
    case OR_GATE_BASE_SUM: return 1 + g->param;
    case OR_GATE_RANDOM_ACCESS: return g->param2 * (g->param + 2) + g->param3;
    case OR_GATE_REDUCING:
    case OR_GATE_REDUCING_EXT: return 2 * g->param;
    case OR_GATE_POSEIDON_MDS: return 24;
    case OR_GATE_EXPONENTIATION: return g->param >= 1 ? g->param + 1 : -1;
    case OR_GATE_COSET_INTERPOLATION:
      if (g->param < 1 || g->param > 5 || g->param2 < 2) return -1;
      return (2 + 2 * (((1 << g->param) - 2) / (g->param2 - 1))) * 2;
    default: return -1;
  }
}

int or_extra_gate_eval(const or_gate *g, const gl2_t *consts, const gl2_t *w, gl2_t *out) {
  cvec c = {out, 0};
  switch (g->type) {
    case OR_GATE_U32_ADD_MANY: add_many(g, w, &c); break;
    case OR_GATE_U32_SUBTRACTION: subtraction(g, w, &c); break;
    case OR_GATE_U32_INTERLEAVE: interleave(g, w, &c); break;
    case OR_GATE_UNINTERLEAVE_TO_U32: uninterleave(g, w, &c, 0); break;
    case OR_GATE_UNINTERLEAVE_TO_B32: uninterleave(g, w, &c, 1); break;
    case OR_GATE_ARITHMETIC_EXT: arithmetic_extension(g, consts, w, &c); break;
    case OR_GATE_MUL_EXT: mul_extension(g, consts, w, &c); break;
    case OR_GATE_BASE_SUM: base_sum(g, w, &c); break;
    case OR_GATE_RANDOM_ACCESS: random_access(g, consts, w, &c); break;
    case OR_GATE_REDUCING: reducing(g, w, &c, 0); break;
    case OR_GATE_REDUCING_EXT: reducing(g, w, &c, 1); break;
    case OR_GATE_POSEIDON_MDS: poseidon_mds(w, &c); break;
    case OR_GATE_COSET_INTERPOLATION: coset_interpolation(g, w, &c); break;
    case OR_GATE_EXPONENTIATION: exponentiation(g, w, &c); break;
    default: return -1;
  }
  return c.n;
}
