// Gate constraint polynomials (A8), written once over a generic field type so that the quotient kernel
// (F = canonical u64, device) and the host verifier (F = quadratic extension) evaluate the same code.
//
// Upstream plonky2 gates (named at city_common_circuit/src/builder/pad_circuit.rs:31-55; formulas restated
// from plonky2 0.2.2, un-vendored): Noop, Constant, PublicInput, Arithmetic, Poseidon.
// In-tree city-rollup gates (formulas follow the reference source line by line):
//   Comparison        city_common_circuit/src/u32/gates/comparison.rs:96-200      (88 constraints @ (32,16), deg 4)
//   U32Arithmetic     city_common_circuit/src/u32/gates/arithmetic_u32.rs:90-150  (36 per op, deg 4)
//   U32RangeCheck     city_common_circuit/src/u32/gates/range_check_u32.rs:57-80  (17 per limb, deg 4)
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
  N_TYPES = 8
};

struct Gate { int type, selector_index, group_start, group_end, param, param2; };

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
    default: return 0;
  }
}

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
template <class F, class WF, class CF, class PF, class EF>
GL_HD void eval(const Gate &g, WF W, CF C, PF PI, EF emit) {
  using O = Ops<F>;
  switch (g.type) {
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
    default: break;
  }
}

}  // namespace gates
