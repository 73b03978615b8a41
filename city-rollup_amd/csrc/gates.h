// Gate constraint polynomials (A8), written once over a generic field type so that the quotient kernel
// (F = canonical u64, device) and the host verifier (F = quadratic extension) evaluate the same code.
//
// Upstream plonky2 gates (named at city_common_circuit/src/builder/pad_circuit.rs:31-55; formulas restated
// from plonky2 0.2.2, un-vendored): Noop, Constant, PublicInput, Arithmetic, Poseidon.
// In-tree city-rollup gates (formulas follow the reference source line by line):
//   Comparison        city_common_circuit/src/u32/gates/comparison.rs:96-200      (88 constraints @ (32,16), deg 4)
//   U32Arithmetic     city_common_circuit/src/u32/gates/arithmetic_u32.rs:90-150  (36 per op, deg 4)
//   U32RangeCheck     city_common_circuit/src/u32/gates/range_check_u32.rs:57-80  (17 per limb, deg 4)
//   U32AddMany        city_common_circuit/src/u32/gates/add_many_u32.rs:93-140    (21 per op, deg 4)
//   U32Subtraction    city_common_circuit/src/u32/gates/subtraction_u32.rs:89-125 (19 per op, deg 4)
//   U32Interleave     city_common_circuit/src/u32/gates/interleave_u32.rs:90-128  (34 per op, deg 2)
//   UninterleaveToU32 city_common_circuit/src/u32/gates/uninterleave_to_u32.rs:82-130 (67 per op, deg 2)
//   UninterleaveToB32 city_common_circuit/src/u32/gates/uninterleave_to_b32.rs:82-131 (67 per op, deg 2)
// Remaining upstream gates of the city-common gate set (pad_circuit.rs:31-55), restated from plonky2 0.2.2:
// ArithmeticExtension, MulExtension, BaseSum<B>, RandomAccess, Reducing, ReducingExtension, PoseidonMds, Exponentiation,
// CosetInterpolation. Extension-valued wires are pairs of wires forming an element of the extension ALGEBRA
// (F[X]/(X^2-7) over F = base field for the prover, F = F_p^2 for the verifier): `Alg<F>` below.
// Each `eval` calls emit(k, value) for constraint k in the order the reference pushes them.
#pragma once
#include "gl.h"
#include "poseidon_tables.h"

namespace gates {

enum {
  NOOP = 0, CONSTANT = 1, PUBLIC_INPUT = 2, ARITHMETIC = 3, POSEIDON = 4,
  COMPARISON = 5,       // param = num_bits, param2 = num_chunks
  U32_ARITHMETIC = 6,   // param = num_ops
  U32_RANGE_CHECK = 7,  // param = num_input_limbs
  U32_ADD_MANY = 8,     // param = num_ops, param2 = num_addends
  U32_SUBTRACTION = 9,  // param = num_ops
  U32_INTERLEAVE = 10,  // param = num_ops
  UNINTERLEAVE_TO_U32 = 11,  // param = num_ops
  UNINTERLEAVE_TO_B32 = 12,  // param = num_ops
  ARITHMETIC_EXT = 13,  // param = num_ops
  MUL_EXT = 14,         // param = num_ops
  BASE_SUM = 15,        // param = num_limbs, param2 = base B
  RANDOM_ACCESS = 16,   // param = bits, param2 = num_copies, param3 = num_extra_constants
  REDUCING = 17,        // param = num_coeffs
  REDUCING_EXT = 18,    // param = num_coeffs
  POSEIDON_MDS = 19,
  COSET_INTERPOLATION = 20,  // param = subgroup_bits, param2 = degree
  EXPONENTIATION = 21,  // param = num_power_bits
  N_TYPES = 22
};

constexpr int MAX_RANDOM_ACCESS_BITS = 4;
constexpr int MAX_COSET_BITS = 5;

struct Gate { int type, selector_index, group_start, group_end, param, param2, param3; };

template <class F> struct Ops;
template <> struct Ops<uint64_t> {
  static GL_HD uint64_t add(uint64_t a, uint64_t b) { return gl::add(a, b); }
  static GL_HD uint64_t sub(uint64_t a, uint64_t b) { return gl::sub(a, b); }
  static GL_HD uint64_t mul(uint64_t a, uint64_t b) { return gl::mul(a, b); }
  static GL_HD uint64_t from(uint64_t v) { return v; }  // v < p
};
template <> struct Ops<gl::Ext> {
  static GL_HD gl::Ext add(gl::Ext a, gl::Ext b) { return gl::ext_add(a, b); }
  static GL_HD gl::Ext sub(gl::Ext a, gl::Ext b) { return gl::ext_sub(a, b); }
  static GL_HD gl::Ext mul(gl::Ext a, gl::Ext b) { return gl::ext_mul(a, b); }
  static GL_HD gl::Ext from(uint64_t v) { return gl::Ext{v, 0}; }
};

GL_HD int ceil_div(int a, int b) { return (a + b - 1) / b; }

GL_HD int num_constraints(const Gate &g) {
  switch (g.type) {
    case CONSTANT: return g.param;
    case PUBLIC_INPUT: return 4;
    case ARITHMETIC: return g.param;
    case POSEIDON: return 123;
    case COMPARISON: return 6 + 5 * g.param2 + ceil_div(g.param, g.param2);  // comparison.rs:310-312
    case U32_ARITHMETIC: return g.param * 36;                               // arithmetic_u32.rs:269-271
    case U32_RANGE_CHECK: return g.param * 17;                              // range_check_u32.rs:158-160
    case U32_ADD_MANY: return g.param * 21;                                 // add_many_u32.rs:266-268
    case U32_SUBTRACTION: return g.param * 19;                              // subtraction_u32.rs:215-217
    case U32_INTERLEAVE: return g.param * 34;                               // interleave_u32.rs:214-216
    case UNINTERLEAVE_TO_U32: case UNINTERLEAVE_TO_B32: return g.param * 67;  // uninterleave_to_u32.rs:246-248
    case ARITHMETIC_EXT: case MUL_EXT: return 2 * g.param;
    case BASE_SUM: return 1 + g.param;
    case RANDOM_ACCESS: return g.param2 * (g.param + 2) + g.param3;
    case REDUCING: case REDUCING_EXT: return 2 * g.param;
    case POSEIDON_MDS: return 24;
    case COSET_INTERPOLATION: return 2 * (2 + 2 * (((1 << g.param) - 2) / (g.param2 - 1)));
    case EXPONENTIATION: return g.param + 1;
    default: return 0;
  }
}
// constants columns (after the selectors) a gate reads
GL_HD int num_constants(const Gate &g) {
  switch (g.type) {
    case CONSTANT: return g.param;
    case ARITHMETIC: case ARITHMETIC_EXT: return 2;
    case MUL_EXT: return 1;
    case RANDOM_ACCESS: return g.param3;
    default: return 0;
  }
}
// wires a gate instance touches (for shape validation)
GL_HD int num_wires(const Gate &g) {
  switch (g.type) {
    case CONSTANT: return g.param;
    case PUBLIC_INPUT: return 4;
    case ARITHMETIC: return 4 * g.param;
    case POSEIDON: return 135;
    case COMPARISON: return 4 + 5 * g.param2 + ceil_div(g.param, g.param2) + 1;
    case U32_ARITHMETIC: return g.param * 38;
    case U32_RANGE_CHECK: return g.param * 17;
    case U32_ADD_MANY: return g.param * (g.param2 + 3 + 18);
    case U32_SUBTRACTION: return g.param * 21;
    case U32_INTERLEAVE: return g.param * 34;
    case UNINTERLEAVE_TO_U32: case UNINTERLEAVE_TO_B32: return g.param * 67;
    case ARITHMETIC_EXT: return 8 * g.param;
    case MUL_EXT: return 6 * g.param;
    case BASE_SUM: return 1 + g.param;
    case RANDOM_ACCESS: return g.param2 * (2 + (1 << g.param)) + g.param3 + g.param2 * g.param;
    case REDUCING: return 6 + g.param + 2 * (g.param - 1);
    case REDUCING_EXT: return 6 + 2 * g.param + 2 * (g.param - 1);
    case POSEIDON_MDS: return 48;
    case COSET_INTERPOLATION: return 1 + 2 * (1 << g.param) + 4 + 4 * (((1 << g.param) - 2) / (g.param2 - 1)) + 2;
    case EXPONENTIATION: return 2 + 2 * g.param;
    default: return 0;
  }
}

// plonky2's ExtensionAlgebra<F, 2>: a + b*X with X^2 = 7, components in F (F = base field: this is F_p^2 itself)
template <class F> struct Alg { F a, b; };
template <class F> GL_HD Alg<F> alg_add(Alg<F> x, Alg<F> y) { using O = Ops<F>; return {O::add(x.a, y.a), O::add(x.b, y.b)}; }
template <class F> GL_HD Alg<F> alg_sub(Alg<F> x, Alg<F> y) { using O = Ops<F>; return {O::sub(x.a, y.a), O::sub(x.b, y.b)}; }
template <class F> GL_HD Alg<F> alg_mul(Alg<F> x, Alg<F> y) {
  using O = Ops<F>;
  return {O::add(O::mul(x.a, y.a), O::mul(O::from(7), O::mul(x.b, y.b))), O::add(O::mul(x.a, y.b), O::mul(x.b, y.a))};
}
template <class F> GL_HD Alg<F> alg_scale(Alg<F> x, F s) { using O = Ops<F>; return {O::mul(x.a, s), O::mul(x.b, s)}; }
template <class F, class WF> GL_HD Alg<F> alg_at(WF W, int start) { return {W(start), W(start + 1)}; }

// prod_{x < count} (v - x)
template <class F>
GL_HD F range_product(F v, int count) {
  using O = Ops<F>;
  F p = v;
  for (int x = 1; x < count; x++) p = O::mul(p, O::sub(v, O::from((uint64_t)x)));
  return p;
}

// ---- PoseidonGate over a generic field (textbook rounds); the device uses the lazy-u64 twin in quotient.h ----
template <class F>
GL_HD F pow7(F x) {
  using O = Ops<F>;
  F x2 = O::mul(x, x), x4 = O::mul(x2, x2), x3 = O::mul(x, x2);
  return O::mul(x3, x4);
}
template <class F>
GL_HD void mds(F (&s)[12]) {
  using O = Ops<F>;
  F o[12];
  for (int r = 0; r < 12; r++) {
    F acc = O::from(0);
    for (int i = 0; i < 12; i++) acc = O::add(acc, O::mul(s[(i + r) % 12], O::from((uint64_t)POSEIDON_MDS_CIRC[i])));
    if (r == 0) acc = O::add(acc, O::mul(s[0], O::from(8)));
    o[r] = acc;
  }
  for (int r = 0; r < 12; r++) s[r] = o[r];
}
template <class F, class WF, class EF>
GL_HD void poseidon_gate(WF W, EF emit) {
  using O = Ops<F>;
  int c = 0;
  const F swap = W(24);
  emit(c++, O::mul(swap, O::sub(swap, O::from(1))));
  for (int i = 0; i < 4; i++) emit(c++, O::sub(O::mul(swap, O::sub(W(i + 4), W(i))), W(25 + i)));
  F st[12];
  for (int i = 0; i < 4; i++) { st[i] = O::add(W(i), W(25 + i)); st[i + 4] = O::sub(W(i + 4), W(25 + i)); }
  for (int i = 8; i < 12; i++) st[i] = W(i);
  int rnd = 0;
  for (int r = 0; r < 4; r++, rnd++) {
    for (int i = 0; i < 12; i++) st[i] = O::add(st[i], O::from(POSEIDON_RC[rnd * 12 + i]));
    if (r != 0)
      for (int i = 0; i < 12; i++) { F in = W(29 + 12 * (r - 1) + i); emit(c++, O::sub(st[i], in)); st[i] = in; }
    for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
    mds(st);
  }
  for (int r = 0; r < 22; r++, rnd++) {
    for (int i = 0; i < 12; i++) st[i] = O::add(st[i], O::from(POSEIDON_RC[rnd * 12 + i]));
    F in = W(65 + r);
    emit(c++, O::sub(st[0], in));
    st[0] = pow7(in);
    mds(st);
  }
  for (int r = 0; r < 4; r++, rnd++) {
    for (int i = 0; i < 12; i++) st[i] = O::add(st[i], O::from(POSEIDON_RC[rnd * 12 + i]));
    for (int i = 0; i < 12; i++) { F in = W(87 + 12 * r + i); emit(c++, O::sub(st[i], in)); st[i] = in; }
    for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
    mds(st);
  }
  for (int i = 0; i < 12; i++) emit(c++, O::sub(st[i], W(12 + i)));
}

// Unfiltered constraints of every gate type except Poseidon-on-device.
// W(j): local wire j; C(j): gate constant j (after the selector columns); PI(j): public-inputs hash element j.
// TYPE is a compile-time constant so that a caller instantiating one gate type carries only that gate's code
// (the quotient kernels are one instantiation per gate type); `eval` below dispatches at run time.
template <int TYPE, class F, class WF, class CF, class PF, class EF>
GL_HD void eval_t(const Gate &g, WF W, CF C, PF PI, EF emit) {
  using O = Ops<F>;
  switch (TYPE) {
    case CONSTANT:
      for (int k = 0; k < g.param; k++) emit(k, O::sub(C(k), W(k)));
      break;
    case PUBLIC_INPUT:
      for (int k = 0; k < 4; k++) emit(k, O::sub(W(k), PI(k)));
      break;
    case ARITHMETIC: {
      const F c0 = C(0), c1 = C(1);
      for (int k = 0; k < g.param; k++) {
        F computed = O::add(O::mul(O::mul(W(4 * k), W(4 * k + 1)), c0), O::mul(W(4 * k + 2), c1));
        emit(k, O::sub(W(4 * k + 3), computed));
      }
      break;
    }
    case POSEIDON:
      poseidon_gate<F>(W, emit);
      break;
    case COMPARISON: {  // comparison.rs:96-200
      const int num_chunks = g.param2, chunk_bits = ceil_div(g.param, g.param2), chunk_size = 1 << chunk_bits;
      int c = 0;
      const F base = O::from((uint64_t)chunk_size);
      F fc = O::from(0), sc = O::from(0);  // reduce_with_powers(chunks, 2^chunk_bits)
      for (int i = num_chunks - 1; i >= 0; i--) {
        fc = O::add(O::mul(fc, base), W(4 + i));
        sc = O::add(O::mul(sc, base), W(4 + num_chunks + i));
      }
      emit(c++, O::sub(fc, W(0)));
      emit(c++, O::sub(sc, W(1)));
      F msd = O::from(0);
      for (int i = 0; i < num_chunks; i++) {
        const F f = W(4 + i), s = W(4 + num_chunks + i);
        emit(c++, range_product(f, chunk_size));
        emit(c++, range_product(s, chunk_size));
        const F diff = O::sub(s, f), dummy = W(4 + 2 * num_chunks + i), eq = W(4 + 3 * num_chunks + i);
        emit(c++, O::sub(O::mul(diff, dummy), O::sub(O::from(1), eq)));
        emit(c++, O::mul(eq, diff));
        const F inter = W(4 + 4 * num_chunks + i);
        emit(c++, O::sub(inter, O::mul(eq, msd)));
        msd = O::add(inter, O::mul(O::sub(O::from(1), eq), diff));
      }
      const F msd_w = W(3);
      emit(c++, O::sub(msd_w, msd));
      F bits = O::from(0);
      for (int i = chunk_bits; i >= 0; i--) bits = O::add(O::add(bits, bits), W(4 + 5 * num_chunks + i));
      for (int i = 0; i <= chunk_bits; i++) {
        const F b = W(4 + 5 * num_chunks + i);
        emit(c++, O::mul(b, O::sub(O::from(1), b)));
      }
      emit(c++, O::sub(O::add(base, msd_w), bits));
      emit(c++, O::sub(W(2), W(4 + 5 * num_chunks + chunk_bits)));
      break;
    }
    case U32_ARITHMETIC: {  // arithmetic_u32.rs:90-150
      const int num_ops = g.param;
      int c = 0;
      for (int i = 0; i < num_ops; i++) {
        const F m0 = W(6 * i), m1 = W(6 * i + 1), addend = W(6 * i + 2), lo = W(6 * i + 3), hi = W(6 * i + 4),
                inv = W(6 * i + 5);
        const F computed = O::add(O::mul(m0, m1), addend);
        const F diff = O::sub(O::from(0xFFFFFFFFull), hi);
        const F hi_not_max = O::sub(O::mul(inv, diff), O::from(1));
        emit(c++, O::mul(hi_not_max, lo));
        emit(c++, O::sub(O::add(O::mul(hi, O::from(1ull << 32)), lo), computed));
        F cl = O::from(0), ch = O::from(0);
        const F four = O::from(4);
        for (int j = 31; j >= 0; j--) {
          const F limb = W(6 * num_ops + 32 * i + j);
          emit(c++, range_product(limb, 4));
          if (j < 16) cl = O::add(O::mul(four, cl), limb);
          else ch = O::add(O::mul(four, ch), limb);
        }
        emit(c++, O::sub(cl, lo));
        emit(c++, O::sub(ch, hi));
      }
      break;
    }
    case U32_RANGE_CHECK: {  // range_check_u32.rs:57-80
      const int n = g.param;
      int c = 0;
      const F four = O::from(4);
      for (int i = 0; i < n; i++) {
        F sum = O::from(0);
        for (int j = 15; j >= 0; j--) sum = O::add(O::mul(sum, four), W(n + 16 * i + j));
        emit(c++, O::sub(sum, W(i)));
        for (int j = 0; j < 16; j++) emit(c++, range_product(W(n + 16 * i + j), 4));
      }
      break;
    }
    case U32_ADD_MANY: {  // add_many_u32.rs:93-140
      const int num_ops = g.param, na = g.param2, stride = na + 3;
      int c = 0;
      const F four = O::from(4);
      for (int i = 0; i < num_ops; i++) {
        F computed = W(stride * i + na);  // input carry
        for (int j = 0; j < na; j++) computed = O::add(computed, W(stride * i + j));
        const F out_result = W(stride * i + na + 1), out_carry = W(stride * i + na + 2);
        emit(c++, O::sub(O::add(O::mul(out_carry, O::from(1ull << 32)), out_result), computed));
        F cr = O::from(0), cc = O::from(0);
        for (int j = 17; j >= 0; j--) {
          const F limb = W(stride * num_ops + 18 * i + j);
          emit(c++, range_product(limb, 4));
          if (j < 16) cr = O::add(O::mul(four, cr), limb);
          else cc = O::add(O::mul(four, cc), limb);
        }
        emit(c++, O::sub(cr, out_result));
        emit(c++, O::sub(cc, out_carry));
      }
      break;
    }
    case U32_SUBTRACTION: {  // subtraction_u32.rs:89-125
      const int num_ops = g.param;
      int c = 0;
      const F four = O::from(4);
      for (int i = 0; i < num_ops; i++) {
        const F x = W(5 * i), y = W(5 * i + 1), borrow = W(5 * i + 2), out_result = W(5 * i + 3), out_borrow = W(5 * i + 4);
        const F initial = O::sub(O::sub(x, y), borrow);
        emit(c++, O::sub(out_result, O::add(initial, O::mul(O::from(1ull << 32), out_borrow))));
        F comb = O::from(0);
        for (int j = 15; j >= 0; j--) {
          const F limb = W(5 * num_ops + 16 * i + j);
          emit(c++, range_product(limb, 4));
          comb = O::add(O::mul(four, comb), limb);
        }
        emit(c++, O::sub(comb, out_result));
        emit(c++, O::mul(out_borrow, O::sub(O::from(1), out_borrow)));
      }
      break;
    }
    case U32_INTERLEAVE: {  // interleave_u32.rs:90-128 — bit wires are big-endian
      const int num_ops = g.param;
      int c = 0;
      for (int i = 0; i < num_ops; i++) {
        const int b0 = 2 * num_ops + 32 * i;
        F cx = O::from(0), ci = O::from(0);
        for (int k = 0; k < 32; k++) {
          const F bit = W(b0 + k);
          cx = O::add(O::add(cx, cx), bit);
          ci = O::add(O::mul(ci, O::from(4)), bit);
        }
        emit(c++, O::sub(cx, W(2 * i)));
        emit(c++, O::sub(ci, W(2 * i + 1)));
        for (int k = 0; k < 32; k++) emit(c++, range_product(W(b0 + k), 2));
      }
      break;
    }
    case UNINTERLEAVE_TO_U32:    // uninterleave_to_u32.rs:82-130
    case UNINTERLEAVE_TO_B32: {  // uninterleave_to_b32.rs:82-131 (even/odd bits weighted by 4^k instead of 2^k)
      const int num_ops = g.param;
      const F base = O::from(TYPE == UNINTERLEAVE_TO_U32 ? 2 : 4);
      int c = 0;
      for (int i = 0; i < num_ops; i++) {
        const int b0 = 3 * num_ops + 64 * i;
        F ci = O::from(0), ev = O::from(0), od = O::from(0);
        for (int k = 0; k < 64; k++) {
          const F bit = W(b0 + k);
          ci = O::add(O::add(ci, ci), bit);
          if (k & 1) od = O::add(O::mul(od, base), bit);
          else ev = O::add(O::mul(ev, base), bit);
        }
        emit(c++, O::sub(ci, W(3 * i)));
        emit(c++, O::sub(ev, W(3 * i + 1)));
        emit(c++, O::sub(od, W(3 * i + 2)));
        for (int k = 0; k < 64; k++) emit(c++, range_product(W(b0 + k), 2));
      }
      break;
    }
    case ARITHMETIC_EXT: {  // plonky2 ArithmeticExtensionGate: out = c0*m0*m1 + c1*addend in the extension algebra
      const F c0 = C(0), c1 = C(1);
      for (int i = 0; i < g.param; i++) {
        const Alg<F> m0 = alg_at<F>(W, 8 * i), m1 = alg_at<F>(W, 8 * i + 2), ad = alg_at<F>(W, 8 * i + 4),
                     out = alg_at<F>(W, 8 * i + 6);
        const Alg<F> d = alg_sub(out, alg_add(alg_scale(alg_mul(m0, m1), c0), alg_scale(ad, c1)));
        emit(2 * i, d.a);
        emit(2 * i + 1, d.b);
      }
      break;
    }
    case MUL_EXT: {  // plonky2 MulExtensionGate: out = c0*m0*m1
      const F c0 = C(0);
      for (int i = 0; i < g.param; i++) {
        const Alg<F> m0 = alg_at<F>(W, 6 * i), m1 = alg_at<F>(W, 6 * i + 2), out = alg_at<F>(W, 6 * i + 4);
        const Alg<F> d = alg_sub(out, alg_scale(alg_mul(m0, m1), c0));
        emit(2 * i, d.a);
        emit(2 * i + 1, d.b);
      }
      break;
    }
    case BASE_SUM: {  // plonky2 BaseSumGate<B>: wire 0 = sum, wires 1.. = little-endian limbs
      const int limbs = g.param, B = g.param2;
      F s = O::from(0);
      for (int i = limbs - 1; i >= 0; i--) s = O::add(O::mul(s, O::from((uint64_t)B)), W(1 + i));
      emit(0, O::sub(s, W(0)));
      for (int i = 0; i < limbs; i++) emit(1 + i, range_product(W(1 + i), B));
      break;
    }
    case RANDOM_ACCESS: {  // plonky2 RandomAccessGate
      const int bits = g.param, copies = g.param2, extra = g.param3, vec = 1 << bits;
      const int routed = (2 + vec) * copies + extra;
      int c = 0;
      for (int cp = 0; cp < copies; cp++) {
        const int base = (2 + vec) * cp, wb = routed + cp * bits;
        for (int b = 0; b < bits; b++) { const F bit = W(wb + b); emit(c++, O::mul(bit, O::sub(bit, O::from(1)))); }
        F idx = O::from(0);
        for (int b = bits - 1; b >= 0; b--) idx = O::add(O::add(idx, idx), W(wb + b));
        emit(c++, O::sub(idx, W(base)));
        F list[1 << MAX_RANDOM_ACCESS_BITS];
#pragma unroll
        for (int k = 0; k < (1 << MAX_RANDOM_ACCESS_BITS); k++) list[k] = k < vec ? W(base + 2 + k) : O::from(0);
#pragma unroll
        for (int b = 0; b < MAX_RANDOM_ACCESS_BITS; b++) {
          if (b < bits) {
            const F bit = W(wb + b);
#pragma unroll
            for (int k = 0; k < ((1 << MAX_RANDOM_ACCESS_BITS) >> (b + 1)); k++)
              list[k] = O::add(list[2 * k], O::mul(bit, O::sub(list[2 * k + 1], list[2 * k])));
          }
        }
        emit(c++, O::sub(list[0], W(base + 1)));
      }
      for (int i = 0; i < extra; i++) emit(c++, O::sub(C(i), W((2 + vec) * copies + i)));
      break;
    }
    case REDUCING:        // plonky2 ReducingGate: acc_{i} = acc_{i-1}*alpha + coeff_i (coeffs in the base field)
    case REDUCING_EXT: {  // plonky2 ReducingExtensionGate: same with extension coefficients
      const int n = g.param;
      const bool ext = TYPE == REDUCING_EXT;
      const int start_accs = 6 + (ext ? 2 * n : n);
      const Alg<F> alpha = alg_at<F>(W, 2);
      Alg<F> acc = alg_at<F>(W, 4);
      for (int i = 0; i < n; i++) {
        const Alg<F> coeff = ext ? alg_at<F>(W, 6 + 2 * i) : Alg<F>{W(6 + i), O::from(0)};
        const Alg<F> next = i == n - 1 ? alg_at<F>(W, 0) : alg_at<F>(W, start_accs + 2 * i);
        const Alg<F> d = alg_sub(alg_add(alg_mul(acc, alpha), coeff), next);
        emit(2 * i, d.a);
        emit(2 * i + 1, d.b);
        acc = next;
      }
      break;
    }
    case POSEIDON_MDS: {  // plonky2 PoseidonMdsGate: 12 extension inputs (wires 0..23) -> 12 outputs (wires 24..47)
      for (int r = 0; r < 12; r++)
        for (int h = 0; h < 2; h++) {
          F acc = O::from(0);
          for (int i = 0; i < 12; i++) acc = O::add(acc, O::mul(W(((i + r) % 12) * 2 + h), O::from((uint64_t)POSEIDON_MDS_CIRC[i])));
          if (r == 0) acc = O::add(acc, O::mul(W(h), O::from(8)));
          emit(2 * r + h, O::sub(W(24 + 2 * r + h), acc));
        }
      break;
    }
    case COSET_INTERPOLATION: {  // plonky2 CosetInterpolationGate (barycentric, chunked by `degree`)
      const int bits = g.param, deg = g.param2, n = 1 << bits, n_inter = (n - 2) / (deg - 1);
      const int w_point = 1 + 2 * n, w_value = w_point + 2, w_inter = w_value + 2, w_shifted = w_inter + 4 * n_inter;
      int c = 0;
      const Alg<F> sp = alg_at<F>(W, w_shifted);
      Alg<F> d = alg_sub(alg_at<F>(W, w_point), alg_scale(sp, W(0)));
      emit(c++, d.a);
      emit(c++, d.b);
      // domain = <omega_n> in natural order; the barycentric weight of point x_i of a full subgroup is x_i / n
      // omega_n = 7^((p-1)/n) and 1/n for n = 2^bits <= 32 (literals: a per-point pow/inv would cost more than the gate)
      constexpr uint64_t OMEGA[6] = {1ull, 0xffffffff00000000ull, 0x1000000000000ull, 0xfffffffeff000001ull,
                                     0xefffffff00000001ull, 0x3fffffffc000ull};
      constexpr uint64_t N_INV[6] = {1ull, 0x7fffffff80000001ull, 0xbfffffff40000001ull, 0xdfffffff20000001ull,
                                     0xefffffff10000001ull, 0xf7ffffff08000001ull};
      uint64_t omega = OMEGA[1], n_inv = N_INV[1];
#pragma unroll
      for (int b = 2; b <= MAX_COSET_BITS; b++)
        if (bits == b) { omega = OMEGA[b]; n_inv = N_INV[b]; }
      uint64_t x = 1;
      Alg<F> ev{O::from(0), O::from(0)}, pr{O::from(1), O::from(0)};
      int boundary = deg, k = 0;
      for (int i = 0; i < n; i++) {
        if (i == boundary) {  // the k-th pair of intermediate wires takes over
          const Alg<F> ie = alg_at<F>(W, w_inter + 2 * k), ip = alg_at<F>(W, w_inter + 2 * (n_inter + k));
          d = alg_sub(ie, ev); emit(c++, d.a); emit(c++, d.b);
          d = alg_sub(ip, pr); emit(c++, d.a); emit(c++, d.b);
          ev = ie; pr = ip; k++; boundary += deg - 1;
        }
        const Alg<F> term{O::sub(sp.a, O::from(x)), sp.b};
        const Alg<F> val = alg_at<F>(W, 1 + 2 * i);
        const Alg<F> nev = alg_add(alg_mul(ev, term), alg_mul(val, alg_scale(pr, O::from(gl::mul(x, n_inv)))));
        pr = alg_mul(pr, term);
        ev = nev;
        x = gl::mul(x, omega);
      }
      d = alg_sub(alg_at<F>(W, w_value), ev);
      emit(c++, d.a);
      emit(c++, d.b);
      break;
    }
    case EXPONENTIATION: {  // plonky2 ExponentiationGate (gates/exponentiation.rs, UPSTREAM-MEMORY): wire 0 = base,
      // 1..n = power bits (little-endian), n+1 = output, n+2.. = intermediate values; square-and-multiply from the top bit
      const int n = g.param;
      const F base = W(0), one = O::from(1);
      for (int i = 0; i < n; i++) {
        const F prev = i == 0 ? one : O::mul(W(2 + n + i - 1), W(2 + n + i - 1));
        const F bit = W(1 + (n - 1 - i));
        const F mul_by = O::add(O::mul(bit, base), O::sub(one, bit));
        emit(i, O::sub(O::mul(prev, mul_by), W(2 + n + i)));
      }
      emit(n, O::sub(W(1 + n), W(2 + n + n - 1)));
      break;
    }
    default: break;
  }
}

template <class F, class WF, class CF, class PF, class EF>
GL_HD void eval(const Gate &g, WF W, CF C, PF PI, EF emit) {
  switch (g.type) {
#define CITY_GATE_CASE(T) case T: eval_t<T, F>(g, W, C, PI, emit); break;
    CITY_GATE_CASE(CONSTANT) CITY_GATE_CASE(PUBLIC_INPUT) CITY_GATE_CASE(ARITHMETIC) CITY_GATE_CASE(POSEIDON)
    CITY_GATE_CASE(COMPARISON) CITY_GATE_CASE(U32_ARITHMETIC) CITY_GATE_CASE(U32_RANGE_CHECK) CITY_GATE_CASE(U32_ADD_MANY)
    CITY_GATE_CASE(U32_SUBTRACTION) CITY_GATE_CASE(U32_INTERLEAVE) CITY_GATE_CASE(UNINTERLEAVE_TO_U32)
    CITY_GATE_CASE(UNINTERLEAVE_TO_B32) CITY_GATE_CASE(ARITHMETIC_EXT) CITY_GATE_CASE(MUL_EXT) CITY_GATE_CASE(BASE_SUM)
    CITY_GATE_CASE(RANDOM_ACCESS) CITY_GATE_CASE(REDUCING) CITY_GATE_CASE(REDUCING_EXT) CITY_GATE_CASE(POSEIDON_MDS)
    CITY_GATE_CASE(COSET_INTERPOLATION) CITY_GATE_CASE(EXPONENTIATION)
#undef CITY_GATE_CASE
    default: break;
  }
}

}  // namespace gates
