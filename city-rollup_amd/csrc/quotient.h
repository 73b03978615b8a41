// A8: quotient / constraint batch evaluation — plonky2
// `compute_quotient_polys` + `eval_vanishing_poly_base_batch` (un-vendored dependency). For every point
// x = 7*omega_N^i of the LDE coset and every challenge c:
//     t_c(x) = ( sum_t alpha_c^t * term_t(x) ) / Z_H(x)
// with the terms in plonky2's order: L_0(x)(Z_c'(x)-1) for every challenge c', then the partial-product
// checks of every challenge, then the gate constraints (filtered by the selector polynomials and summed
// per constraint index over all gates).
//
// One lane per LDE point, in STORAGE order (bit-reversed), so that all column reads of a point are coalesced
// across the wave. The work is split into one launch per piece — the permutation argument (k_quot_perm), one
// launch per gate of the circuit (k_quot_gate<TYPE>: an instantiation carries only that gate's code, so the
// light gates run at 5-8 waves per SIMD instead of the 2 a single fused kernel gets from PoseidonGate's
// register footprint) and k_quot_finish (divide by Z_H, scatter to the natural order the inverse transform
// wants). The pieces add into a [proof][challenge][N] accumulator in storage order; field addition is exact, so
// the split does not change a bit of the result.
//
// Parity: unpinned by reference data (see oracle/plonky2_quotient.c); bit-exact against that oracle,
// whose verifier side checks the vanishing identity at zeta.
#pragma once
#include "gates.h"
#include "gl.h"
#include "poseidon.h"

namespace quot {

constexpr int MAXC = 4;        // num_challenges supported by the kernel
constexpr int MAX_GATES = 32;
constexpr uint64_t UNUSED_SELECTOR = 0xFFFFFFFFull;  // u32::MAX

enum { GATE_NOOP = gates::NOOP, GATE_CONSTANT = gates::CONSTANT, GATE_PUBLIC_INPUT = gates::PUBLIC_INPUT,
       GATE_ARITHMETIC = gates::ARITHMETIC, GATE_POSEIDON = gates::POSEIDON };

using Gate = gates::Gate;

struct Args {
  const uint64_t *const *cs_lde;  // per proof: (num_constants + R) x N, bit-reversed
  const uint64_t *wires_lde;      // [proof][W][N]
  size_t wires_stride;
  const uint64_t *zs_lde;         // [proof][nc*(1+npp)][N]
  size_t zs_stride;
  const uint64_t *k_is;           // [R]
  const uint64_t *chal;           // [proof][2][nc]: betas, gammas
  const uint64_t *apow;           // [proof][nc][n_terms]: alpha_c^t
  const uint64_t *pi_hash;        // public_inputs_hash of proof p at pi_hash + p*pi_stride (4 elements)
  size_t pi_stride;
  const uint64_t *zh;             // [2^rb]  Z_H on the coset, and
  const uint64_t *zh_inv;         // [2^rb]  its inverses
  const uint64_t *omega_tab;      // power table of omega_N
  const uint64_t *l0_tab;         // [N], storage order: L_0(x) = Z_H(x) / (n (x - 1)) on the coset (per shape, cached per context)
  uint64_t *out;                  // [proof][nc][N], NATURAL index order
  size_t out_stride;
  uint64_t *acc;                  // [proof][nc][N], storage order: sum_t alpha^t term_t before the division by Z_H
  uint64_t n_field;               // n as a field element
  size_t N;
  int log_N, rb;
  int ncst, R, W, nc, npp, chunk, n_gates, num_selectors, n_terms;
  int flip;                       // 1: odd gate launches walk the batch backwards (see k_quot_gate)
  // small batches (k_quot_all): every piece writes its own slice, parts[(piece * B + proof) * out_stride + c * N + s], piece =
  // gate index or n_gates for the permutation argument; k_quot_finish adds the n_parts slices instead of reading `acc`
  uint64_t *parts;
  int n_parts;
  int t0_gates;                   // index of the first gate constraint among the terms
  Gate gates[MAX_GATES];
};

__device__ __forceinline__ uint64_t pow_tab(const uint64_t *T, uint64_t e) {
  uint32_t e0 = (uint32_t)e & 2047, e1 = (uint32_t)(e >> 11) & 2047, e2 = (uint32_t)(e >> 22);
  uint64_t r = T[e0];
  if (e1) r = gl::mul(r, T[2048 + e1]);
  if (e2) r = gl::mul(r, T[4096 + e2]);
  return r;
}

GL_HD int gate_num_constraints(const Gate &g) { return gates::num_constraints(g); }

// L_0 on the LDE coset, storage order: depends on the shape only, so the one field inversion per point is paid once per
// (degree, rate) and context instead of once per proof and point (it was a Fermat inversion per lane of k_quot_perm).
__global__ __launch_bounds__(256) void k_fill_l0(uint64_t *out, size_t N, int log_N, int rb, const uint64_t *omega_tab,
                                                 const uint64_t *zh, uint64_t n_field) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= N) return;
  const uint32_t i = __brev((uint32_t)s) >> (32 - log_N);
  const uint64_t x = gl::mul(7, pow_tab(omega_tab, i));
  out[s] = gl::mul(zh[i & ((1u << rb) - 1)], gl::inv(gl::mul(n_field, gl::sub(x, 1))));
}

// L_0(x)(Z(x)-1) for every challenge, then the partial-product checks: terms 0 .. nc*(npp+2), for the point at storage
// (bit-reversed) position s of `proof`; out[c * N] = the sum for challenge c
template <class WR>  // WR(j): wire j of this point (from HBM, or from a tile a workgroup has staged in LDS: k_quot_tile)
__device__ __forceinline__ void perm_sum(const Args &a, size_t s, size_t proof, WR W, uint64_t (&res)[MAXC]) {
  const uint32_t i = __brev((uint32_t)s) >> (32 - a.log_N);  // natural index
  const size_t N = a.N;
  const uint64_t *cs = a.cs_lde[proof] + s;
  const uint64_t *zsb = a.zs_lde + proof * a.zs_stride;
  const uint32_t i_next = (i + (1u << a.rb)) & (uint32_t)(N - 1);
  const size_t s_next = __brev(i_next) >> (32 - a.log_N);
  const uint64_t *betas = a.chal + proof * 2 * a.nc, *gammas = betas + a.nc;
  const uint64_t *apow = a.apow + proof * (size_t)a.nc * a.n_terms;
  const int nc = a.nc;

  const uint64_t x = gl::mul(7, pow_tab(a.omega_tab, i));
  const uint64_t l0 = a.l0_tab[s];

  uint64_t acc[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; c++) acc[c] = 0;
  int t = 0;
  auto add_term = [&](uint64_t term, int idx) {
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < nc) acc[c] = gl::mul_add_lazy(term, apow[(size_t)c * a.n_terms + idx], acc[c]);  // lazy u64
  };
  for (int c = 0; c < nc; c++) add_term(gl::mul(l0, gl::sub(zsb[(size_t)c * N + s], 1)), t++);
  // partial-product checks: prev * prod(num chunk) - next * prod(den chunk)
  for (int c = 0; c < nc; c++) {
    const uint64_t beta = betas[c], gamma = gammas[c];
    const uint64_t bx = gl::mul(beta, x);
    uint64_t prev = zsb[(size_t)c * N + s];
    for (int k = 0; k <= a.npp; k++) {
      uint64_t next = k == a.npp ? zsb[(size_t)c * N + s_next] : zsb[((size_t)nc + (size_t)c * a.npp + k) * N + s];
      uint64_t pn = 1, pd = 1;
      for (int j = k * a.chunk; j < a.R && j < (k + 1) * a.chunk; j++) {
        // lazy values inside the running products (any u64 congruent to the value; the term below canonicalises):
        // w + gamma + beta k x and w + gamma + beta sigma as one multiply-add each
        const uint64_t base = gl::add(W(j), gamma);
        pn = poseidon::mul_lazy(pn, gl::mul_add_lazy(bx, a.k_is[j], base));
        pd = poseidon::mul_lazy(pd, gl::mul_add_lazy(beta, cs[(size_t)(a.ncst + j) * N], base));
      }
      add_term(gl::sub(gl::mul(prev, pn), gl::mul(next, pd)), t++);
      prev = next;
    }
  }
#pragma unroll
  for (int c = 0; c < MAXC; c++) res[c] = c < nc ? gl::canon(acc[c]) : 0;
}
__device__ __forceinline__ void perm_eval(const Args &a, size_t s, size_t proof, uint64_t *out) {
  const uint64_t *w = a.wires_lde + proof * a.wires_stride + s;
  const size_t N = a.N;
  uint64_t res[MAXC];
  perm_sum(a, s, proof, [&](int j) { return w[(size_t)j * N]; }, res);
#pragma unroll
  for (int c = 0; c < MAXC; c++)
    if (c < a.nc) out[(size_t)c * N] = res[c];
}
// grid = (N/256, B)
__global__ __launch_bounds__(256) void k_quot_perm(Args a) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.N) return;
  perm_eval(a, s, blockIdx.y, a.acc + (size_t)blockIdx.y * a.out_stride + s);
}

template <int TYPE, class WR>  // WR(j): wire j of this point; res[c] = filter * sum_k alpha_c^(t0+k) * constraint_k (canonical)
__device__ __forceinline__ void gate_sum(const Args &a, int gi, int t0, size_t s, size_t proof, WR W, uint64_t (&res)[MAXC]) {
  const size_t N = a.N;
  const uint64_t *cs = a.cs_lde[proof] + s;
  const uint64_t *apow = a.apow + proof * (size_t)a.nc * a.n_terms + t0;
  const int nc = a.nc;
  const Gate g = a.gates[gi];

  uint64_t acc[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; c++) acc[c] = 0;
  auto add_term = [&](uint64_t term, int idx) {
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < nc) acc[c] = gl::mul_add_lazy(term, apow[(size_t)c * a.n_terms + idx], acc[c]);  // lazy u64
  };
  if constexpr (TYPE == gates::POSEIDON) {
    // plonky2 PoseidonGate: wires 0..11 in, 12..23 out, 24 swap, 25..28 delta, S-box inputs of full rounds
    // 1..3 at 29.., of the 22 partial rounds at 65.., of the last 4 full rounds at 87.. (123 constraints).
    // Textbook round structure with the lazy permutation primitives; anchors are canonicalised.
    int c = 0;
    const uint64_t swap = W(24);
    add_term(gl::mul(swap, gl::sub(swap, 1)), c++);
    uint64_t st[poseidon::W];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint64_t l = W(k), r = W(k + 4), d = W(25 + k);
      add_term(gl::sub(gl::mul(swap, gl::sub(r, l)), d), c++);
      st[k] = gl::add(l, d);
      st[k + 4] = gl::sub(r, d);
    }
#pragma unroll
    for (int k = 8; k < 12; k++) st[k] = W(k);
#pragma unroll
    for (int k = 0; k < 12; k++) st[k] = poseidon::add_const_lazy(st[k], poseidon::rc(k));
    int rnd = 0;
#pragma unroll 1
    for (int r = 0; r < 4; r++, rnd++) {
      if (r != 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
          uint64_t in = W(29 + 12 * (r - 1) + k);
          add_term(gl::sub(gl::canon(st[k]), in), c++);
          st[k] = in;
        }
      }
#pragma unroll
      for (int k = 0; k < 12; k++) st[k] = poseidon::sbox_lazy(st[k]);
      if (r < 3) poseidon::mds_layer_d(st, (rnd + 1) * 12);
    }
    // the 22 partial rounds in the transformed domain (poseidon.h): the S-box input of each is constrained against its wire
    // and the wire goes on into the S-box
    poseidon::partial_rounds(
        st,
        [&](int r, uint64_t x) {
          const uint64_t in = W(65 + r);
          add_term(gl::sub(gl::canon(x), in), c + r);
          return in;
        },
        poseidon::NeverStop());
    c += 22;
    rnd += 22;
#pragma unroll 1
    for (int r = 0; r < 4; r++, rnd++) {
#pragma unroll
      for (int k = 0; k < 12; k++) {
        uint64_t in = W(87 + 12 * r + k);
        add_term(gl::sub(gl::canon(st[k]), in), c++);
        st[k] = in;
      }
#pragma unroll
      for (int k = 0; k < 12; k++) st[k] = poseidon::sbox_lazy(st[k]);
      poseidon::mds_layer_d(st, rnd + 1 < poseidon::ROUNDS ? (rnd + 1) * 12 : -1);
    }
#pragma unroll
    for (int k = 0; k < 12; k++) add_term(gl::sub(gl::canon(st[k]), W(12 + k)), c++);
  } else if constexpr (TYPE == gates::POSEIDON_MDS) {
    // PoseidonMdsGate: the MDS on 12 extension elements = the base-field MDS on their a-components and on their
    // b-components: two double-precision layers (poseidon.h `mds_layer_d`) — same values as gates.h's generic form
    uint64_t sa[12], sb[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      sa[k] = W(2 * k);
      sb[k] = W(2 * k + 1);
    }
    poseidon::mds_layer_d(sa, -1);
    poseidon::mds_layer_d(sb, -1);
#pragma unroll
    for (int r = 0; r < 12; r++) {
      add_term(gl::sub(W(24 + 2 * r), gl::canon(sa[r])), 2 * r);
      add_term(gl::sub(W(24 + 2 * r + 1), gl::canon(sb[r])), 2 * r + 1);
    }
  } else {  // the generic constraint code shared with the host verifier (gates.h)
    const uint64_t *consts = cs + (size_t)a.num_selectors * N;
    const uint64_t *pih = a.pi_hash + proof * a.pi_stride;
    gates::eval_t<TYPE, uint64_t>(
        g, W, [&](int j) { return consts[(size_t)j * N]; },
        [&](int j) { return pih[j]; }, [&](int k, uint64_t v) { add_term(v, k); });
  }
  // filter of this gate inside its selector group
  const uint64_t sv = cs[(size_t)g.selector_index * N];
  uint64_t f = 1;
  for (int r = g.group_start; r < g.group_end; r++)
    if (r != gi) f = gl::mul(f, gl::sub((uint64_t)r, sv));
  if (a.num_selectors > 1) f = gl::mul(f, gl::sub(UNUSED_SELECTOR, sv));
#pragma unroll
  for (int c = 0; c < MAXC; c++) res[c] = c < nc ? gl::mul(f, acc[c]) : 0;
}
// Gate `gi` (of type TYPE) at storage position s of `proof`, wires from HBM: added to out[c * N] (ACCUMULATE) or written there
template <int TYPE, bool ACCUMULATE>
__device__ __forceinline__ void gate_eval(const Args &a, int gi, int t0, size_t s, size_t proof, uint64_t *out) {
  const size_t N = a.N;
  const uint64_t *w = a.wires_lde + proof * a.wires_stride + s;
  uint64_t res[MAXC];
  gate_sum<TYPE>(a, gi, t0, s, proof, [&](int j) { return w[(size_t)j * N]; }, res);
#pragma unroll
  for (int c = 0; c < MAXC; c++)
    if (c < a.nc) out[(size_t)c * N] = ACCUMULATE ? gl::add(out[(size_t)c * N], res[c]) : res[c];
}


// grid = (N/256, B): acc[c] += gate `gi`
template <int TYPE>
__global__ __launch_bounds__(256) void k_quot_gate(Args a, int gi, int t0) {
  // Fourteen of these kernels stream the same wire columns one after the other. Every other launch (odd gi) walks the
  // batch backwards — last proof first, last rows first — so that it starts on what the previous launch touched last and
  // finds it in the memory-side cache instead of HBM (CITYPROVER_QUOT_FLIP=0 turns it off for measurements).
  const bool flip = (gi & 1) && a.flip;
  const size_t s = (size_t)(flip ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 256 + threadIdx.x;
  if (s >= a.N) return;
  const size_t proof = flip ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  gate_eval<TYPE, true>(a, gi, t0, s, proof, a.acc + proof * a.out_stride + s);
}

// THE ARITHMETIC GROUP (round 4, VERDICT r3 "next" #3). ConstantGate, PublicInputGate, ArithmeticGate, ArithmeticExtensionGate and
// MulExtensionGate all read the FIRST wires of a row - 4, 8 and 6 wires per operation, ~80 wires each at the recursion
// configuration - and little else: as five launches they stream the same 80 wire columns from HBM three times over (and the
// accumulator five times). Here one lane walks the row once in windows of 24 wires (the least common multiple of 4, 8 and 6: six
// Arithmetic operations, three ArithmeticExtension, four MulExtension per window), holds a window in registers and evaluates
// every operation of every member gate on it. Each member keeps its own alpha-weighted sum (its selector filter multiplies the
// sum once, as in gate_eval); the constraint indices, and therefore the alpha powers, are the gates' own - the same field
// elements in another order of exact additions: the same bits. NC: accumulators compiled in (2 for the product's two challenges).
struct ArithGroup { int constant, public_input, arithmetic, arithmetic_ext, mul_ext; };  // gate index of each member, -1 = not in the circuit
GL_HD bool in_arith_group(int type) {
  return type == gates::CONSTANT || type == gates::PUBLIC_INPUT || type == gates::ARITHMETIC || type == gates::ARITHMETIC_EXT || type == gates::MUL_EXT;
}
template <int NC, class WR>
__device__ __forceinline__ void arith_group_sum(const Args &a, const ArithGroup &G, int t0, size_t s, size_t proof, WR W, uint64_t (&res)[MAXC]) {
  const size_t N = a.N;
  const uint64_t *cs = a.cs_lde[proof] + s;
  const uint64_t *apow = a.apow + proof * (size_t)a.nc * a.n_terms + t0;
  const uint64_t *consts = cs + (size_t)a.num_selectors * N;
  const int nc = a.nc;
  uint64_t accA[NC], accE[NC], accM[NC], accS[NC];  // Arithmetic, ArithmeticExtension, MulExtension, the two small gates (pre-filtered)
#pragma unroll
  for (int c = 0; c < NC; c++) accA[c] = accE[c] = accM[c] = accS[c] = 0;
  auto term = [&](uint64_t (&acc)[NC], uint64_t v, int idx) {
#pragma unroll
    for (int c = 0; c < NC; c++)
      if (c < nc) acc[c] = gl::mul_add_lazy(v, apow[(size_t)c * a.n_terms + idx], acc[c]);
  };
  auto filter_of = [&](int gi) -> uint64_t {
    const Gate g = a.gates[gi];
    const uint64_t sv = cs[(size_t)g.selector_index * N];
    uint64_t f = 1;
    for (int r = g.group_start; r < g.group_end; r++)
      if (r != gi) f = gl::mul(f, gl::sub((uint64_t)r, sv));
    if (a.num_selectors > 1) f = gl::mul(f, gl::sub(UNUSED_SELECTOR, sv));
    return f;
  };
  const int nA = G.arithmetic >= 0 ? a.gates[G.arithmetic].param : 0, nE = G.arithmetic_ext >= 0 ? a.gates[G.arithmetic_ext].param : 0,
            nM = G.mul_ext >= 0 ? a.gates[G.mul_ext].param : 0, nK = G.constant >= 0 ? a.gates[G.constant].param : 0;
  const uint64_t c0 = consts[0], c1 = (G.arithmetic >= 0 || G.arithmetic_ext >= 0) ? consts[N] : 0;
  int last = 0;  // wires the members read
  if (4 * nA > last) last = 4 * nA;
  if (8 * nE > last) last = 8 * nE;
  if (6 * nM > last) last = 6 * nM;
  if (nK > last) last = nK;
  if (G.public_input >= 0 && last < 4) last = 4;
  // the two small gates read the first wires only: their terms are multiplied by their filters at once and share one sum
  if (G.constant >= 0) {
    const uint64_t f = filter_of(G.constant);
    for (int k = 0; k < nK; k++) term(accS, gl::mul(f, gl::sub(consts[(size_t)k * N], W(k))), k);
  }
  if (G.public_input >= 0) {
    const uint64_t f = filter_of(G.public_input);
    const uint64_t *pih = a.pi_hash + proof * a.pi_stride;
    for (int k = 0; k < 4; k++) term(accS, gl::mul(f, gl::sub(W(k), pih[k])), k);
  }
  using gates::Alg;
  for (int base = 0; base < last; base += 24) {
    uint64_t x[24];
#pragma unroll
    for (int j = 0; j < 24; j++) x[j] = base + j < last ? W(base + j) : 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {  // ArithmeticGate: w3 - (w0 w1 c0 + w2 c1)
      const int k = base / 4 + j;
      if (k < nA) term(accA, gl::sub(x[4 * j + 3], gl::add(gl::mul(gl::mul(x[4 * j], x[4 * j + 1]), c0), gl::mul(x[4 * j + 2], c1))), k);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {  // ArithmeticExtensionGate: out - (c0 m0 m1 + c1 addend) in F[X]/(X^2 - 7)
      const int i = base / 8 + j;
      if (i < nE) {
        const Alg<uint64_t> m0{x[8 * j], x[8 * j + 1]}, m1{x[8 * j + 2], x[8 * j + 3]}, ad{x[8 * j + 4], x[8 * j + 5]}, o{x[8 * j + 6], x[8 * j + 7]};
        const Alg<uint64_t> d = gates::alg_sub(o, gates::alg_add(gates::alg_scale(gates::alg_mul(m0, m1), c0), gates::alg_scale(ad, c1)));
        term(accE, d.a, 2 * i);
        term(accE, d.b, 2 * i + 1);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {  // MulExtensionGate: out - c0 m0 m1
      const int i = base / 6 + j;
      if (i < nM) {
        const Alg<uint64_t> m0{x[6 * j], x[6 * j + 1]}, m1{x[6 * j + 2], x[6 * j + 3]}, o{x[6 * j + 4], x[6 * j + 5]};
        const Alg<uint64_t> d = gates::alg_sub(o, gates::alg_scale(gates::alg_mul(m0, m1), c0));
        term(accM, d.a, 2 * i);
        term(accM, d.b, 2 * i + 1);
      }
    }
  }
  const uint64_t fA = G.arithmetic >= 0 ? filter_of(G.arithmetic) : 0, fE = G.arithmetic_ext >= 0 ? filter_of(G.arithmetic_ext) : 0,
                 fM = G.mul_ext >= 0 ? filter_of(G.mul_ext) : 0;
#pragma unroll
  for (int c = 0; c < MAXC; c++) res[c] = 0;
#pragma unroll
  for (int c = 0; c < NC; c++)
    if (c < nc) res[c] = gl::add(gl::add(gl::mul(fA, accA[c]), gl::mul(fE, accE[c])), gl::add(gl::mul(fM, accM[c]), gl::canon(accS[c])));
}
// grid = (N/256, B): acc[c] += every member of the arithmetic group
template <int NC>
__global__ __launch_bounds__(256) void k_quot_arith_group(Args a, ArithGroup G, int t0) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.N) return;
  const size_t N = a.N, proof = blockIdx.y;
  const uint64_t *w = a.wires_lde + proof * a.wires_stride + s;
  uint64_t res[MAXC];
  arith_group_sum<NC>(a, G, t0, s, proof, [&](int j) { return w[(size_t)j * N]; }, res);
  uint64_t *out = a.acc + proof * a.out_stride + s;
#pragma unroll
  for (int c = 0; c < NC; c++)
    if (c < a.nc) out[(size_t)c * N] = gl::add(out[(size_t)c * N], res[c]);
}

// THE WHOLE QUOTIENT OF A TILE OF 64 POINTS IN ONE WORKGROUP (round 4, VERDICT r3 "next" #3). With a launch per gate every gate
// streams its ~100 of the 135 wire columns from HBM again: 5.9x the bytes of "every column read once" (5.2x with the arithmetic
// group). Here a workgroup stages the 135 wires of its 64 points in LDS ONCE (69 KB, [wire][lane]: conflict-free), and every
// piece of the quotient - the permutation argument, each gate, the arithmetic group - is evaluated by a WAVE of its own on those
// points, all at the same time, reading wires from LDS; the partial sums meet in LDS (the tile's memory, reused), are divided by
// Z_H and written to the point's natural position: the accumulator array, its read-modify-write per launch and k_quot_finish are
// gone as well. The pieces are ordered so that the waves a SIMD receives (wave w of a workgroup runs on SIMD w mod 4) carry about
// the same work (prover_tail.inc: longest piece first onto the lightest SIMD). One workgroup per CU (the heaviest gate's registers
// for every wave, at most 128: four waves per SIMD). Field addition is exact: the same bits whatever the split.
struct TilePieces {
  int n;            // waves with a piece
  int kind[16];     // 0: permutation argument, 1: one gate (gi), 2: the arithmetic group
  int gi[16];
  ArithGroup G;
};
constexpr int TILE = 64;
__global__ __launch_bounds__(1024) void k_quot_tile(Args a, TilePieces P) {
  extern __shared__ uint64_t tile[];  // [W][TILE] wires; then [pieces][MAXC][TILE] partial sums
  const int lane = threadIdx.x & (TILE - 1), wave = threadIdx.x / TILE, n_waves = blockDim.x / TILE;
  const size_t N = a.N, proof = blockIdx.y;
  const size_t s = (size_t)blockIdx.x * TILE + lane;  // N is a multiple of TILE (the host checks)
  const uint64_t *w = a.wires_lde + proof * a.wires_stride + s;
  for (int j = wave; j < a.W; j += n_waves) tile[j * TILE + lane] = w[(size_t)j * N];
  __syncthreads();
  uint64_t res[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; c++) res[c] = 0;
  auto W = [&](int j) { return tile[j * TILE + lane]; };
  if (wave < P.n && (P.kind[wave] != 1 || P.gi[wave] >= 0)) {
    const int kind = P.kind[wave], gi = P.gi[wave];
    if (kind == 0) perm_sum(a, s, proof, W, res);
    else if (kind == 2) {
      if (a.nc <= 2) arith_group_sum<2>(a, P.G, a.t0_gates, s, proof, W, res);
      else arith_group_sum<MAXC>(a, P.G, a.t0_gates, s, proof, W, res);
    } else {
      switch (a.gates[gi].type) {
#define CITY_QUOT_CASE(T) case gates::T: gate_sum<gates::T>(a, gi, a.t0_gates, s, proof, W, res); break;
        CITY_QUOT_CASE(CONSTANT) CITY_QUOT_CASE(PUBLIC_INPUT) CITY_QUOT_CASE(ARITHMETIC) CITY_QUOT_CASE(POSEIDON) CITY_QUOT_CASE(COMPARISON)
        CITY_QUOT_CASE(U32_ARITHMETIC) CITY_QUOT_CASE(U32_RANGE_CHECK) CITY_QUOT_CASE(U32_ADD_MANY) CITY_QUOT_CASE(U32_SUBTRACTION)
        CITY_QUOT_CASE(U32_INTERLEAVE) CITY_QUOT_CASE(UNINTERLEAVE_TO_U32) CITY_QUOT_CASE(UNINTERLEAVE_TO_B32) CITY_QUOT_CASE(ARITHMETIC_EXT)
        CITY_QUOT_CASE(MUL_EXT) CITY_QUOT_CASE(BASE_SUM) CITY_QUOT_CASE(RANDOM_ACCESS) CITY_QUOT_CASE(REDUCING) CITY_QUOT_CASE(REDUCING_EXT)
        CITY_QUOT_CASE(POSEIDON_MDS) CITY_QUOT_CASE(COSET_INTERPOLATION) CITY_QUOT_CASE(EXPONENTIATION)
#undef CITY_QUOT_CASE
        default: break;  // Noop: no constraints
      }
    }
  }
  __syncthreads();  // every wave is done with the wires: their memory takes the partial sums
  if (wave < P.n) {
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < a.nc) tile[(wave * MAXC + c) * TILE + lane] = res[c];
  }
  __syncthreads();
  if (wave == 0) {
    const uint32_t i = __brev((uint32_t)s) >> (32 - a.log_N);
    const uint64_t zh_inv = a.zh_inv[i & ((1u << a.rb) - 1)];
    uint64_t *out = a.out + proof * a.out_stride + i;
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < a.nc) {
        uint64_t v = 0;
        for (int p = 0; p < P.n; p++) v = gl::add(v, tile[(p * MAXC + c) * TILE + lane]);
        out[(size_t)c * N] = gl::mul(v, zh_inv);
      }
  }
}

// SMALL BATCHES (one or two proofs): a gate kernel of 2^15 points is 512 waves on 1 024 SIMDs, and fifteen of them in a row are
// fifteen half-empty launches (0.67 ms of a lone proof's 3.2 ms of kernels). Here every piece of the quotient — each gate and the
// permutation argument — is a slice of ONE grid (blockIdx.z) and writes its own slice of `parts`; k_quot_finish adds the slices.
// Field addition is exact and commutative: the same bits as the launch-per-gate form. Register allocation is that of the
// heaviest gate for everyone, which is why batches that fill the chip anyway keep one launch per gate.
// grid = (N/256, B, n_gates + 1)
__global__ __launch_bounds__(256) void k_quot_all(Args a) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.N) return;
  const size_t proof = blockIdx.y;
  const int gi = blockIdx.z;
  uint64_t *out = a.parts + ((size_t)gi * gridDim.y + proof) * a.out_stride + s;
  if (gi == a.n_gates) {
    perm_eval(a, s, proof, out);
    return;
  }
  switch (a.gates[gi].type) {
#define CITY_QUOT_CASE(T) case gates::T: gate_eval<gates::T, false>(a, gi, a.t0_gates, s, proof, out); break;
    CITY_QUOT_CASE(CONSTANT) CITY_QUOT_CASE(PUBLIC_INPUT) CITY_QUOT_CASE(ARITHMETIC) CITY_QUOT_CASE(POSEIDON) CITY_QUOT_CASE(COMPARISON)
    CITY_QUOT_CASE(U32_ARITHMETIC) CITY_QUOT_CASE(U32_RANGE_CHECK) CITY_QUOT_CASE(U32_ADD_MANY) CITY_QUOT_CASE(U32_SUBTRACTION)
    CITY_QUOT_CASE(U32_INTERLEAVE) CITY_QUOT_CASE(UNINTERLEAVE_TO_U32) CITY_QUOT_CASE(UNINTERLEAVE_TO_B32) CITY_QUOT_CASE(ARITHMETIC_EXT)
    CITY_QUOT_CASE(MUL_EXT) CITY_QUOT_CASE(BASE_SUM) CITY_QUOT_CASE(RANDOM_ACCESS) CITY_QUOT_CASE(REDUCING) CITY_QUOT_CASE(REDUCING_EXT)
    CITY_QUOT_CASE(POSEIDON_MDS) CITY_QUOT_CASE(COSET_INTERPOLATION) CITY_QUOT_CASE(EXPONENTIATION)
#undef CITY_QUOT_CASE
    default:  // Noop: no constraints
      for (int c = 0; c < a.nc; c++) out[(size_t)c * a.N] = 0;
  }
}

// grid = (N/256, B): t_c(x) = acc / Z_H(x), written to the point's natural position
__global__ __launch_bounds__(256) void k_quot_finish(Args a) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.N) return;
  const size_t proof = blockIdx.y;
  const uint32_t i = __brev((uint32_t)s) >> (32 - a.log_N);
  const uint64_t zh_inv = a.zh_inv[i & ((1u << a.rb) - 1)];
  uint64_t *out = a.out + proof * a.out_stride + i;
  if (a.n_parts > 0) {  // the slices of k_quot_all
    for (int c = 0; c < a.nc; c++) {
      uint64_t v = 0;
      for (int g = 0; g < a.n_parts; g++) v = gl::add(v, a.parts[((size_t)g * gridDim.y + proof) * a.out_stride + (size_t)c * a.N + s]);
      out[(size_t)c * a.N] = gl::mul(v, zh_inv);
    }
    return;
  }
  const uint64_t *in = a.acc + proof * a.out_stride + s;
  for (int c = 0; c < a.nc; c++) out[(size_t)c * a.N] = gl::mul(in[(size_t)c * a.N], zh_inv);
}

}  // namespace quot
