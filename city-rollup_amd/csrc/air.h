// Generic AIR machinery (include/cityprover.h "the STARK's own two steps as GENERIC device machinery"; SURVEY.md §8(a) A13 /
// §8(f) N3). The SHA-256 STARK of the sighash circuit (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524,
// 418 + 912 columns :55-79) evaluates its constraints through starkyx's generic parser trait; a fork records them as a flat
// straight-line program (rust/starkyx-patch/recording_parser.rs) and this file runs such a program:
//   * host: validation, the constraint degree, and a small COMPILER from the recorded SSA ops to a compact bytecode — dead
//     values dropped, loads and uniform values (constants, publics, globals, challenges) folded into operands, the constraints
//     cut into independent segments (a segment = a run of consecutive sinks + the backward slice of ops they need; its
//     contribution to the alpha-fold is its own Horner sum times alpha^(sinks after it), so segments are summed afterwards —
//     field addition is exact, the bits do not depend on the cut), temporaries assigned by liveness to per-lane slots, a value
//     consumed only by the next instruction kept in a register (the accumulator) and never stored;
//   * device: an interpreter — one lane per point of the quotient coset (storage order: all column loads of a wave are
//     coalesced), one wave per workgroup, grid = (point blocks, segments); the bytecode, the uniform table and the column
//     pointer table are read through the scalar unit (uniform addresses), slots live in LDS ([slot][lane]: conflict-free
//     ds_read_b64 / ds_write_b64), slots beyond the LDS budget in a global scratch array.
// A trace of 2^10 rows is 32 waves of points; cut into ~128 segments it is 4 096 waves. Nothing of any particular AIR is here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <queue>
#include <string>
#include <type_traits>
#include <vector>

#include "gl.h"

namespace air {

// raw op codes = CP_AIR_* (include/cityprover.h)
enum : uint32_t { R_LOCAL = 0, R_NEXT, R_PUBLIC, R_GLOBAL, R_CHALLENGE, R_CONST, R_ADD, R_SUB, R_MUL, R_NEG, R_INV,
                  R_ASSERT, R_ASSERT_TRANSITION, R_ASSERT_FIRST, R_ASSERT_LAST, R_STORE };
struct RawOp { uint32_t op, a, b, c; };
inline bool defines_value(uint32_t op) { return op <= R_INV; }
inline bool is_sink(uint32_t op) { return op >= R_ASSERT && op <= R_ASSERT_LAST; }
inline bool is_uniform(uint32_t op) { return op >= R_PUBLIC && op <= R_CONST; }
inline bool is_load(uint32_t op) { return op == R_LOCAL || op == R_NEXT; }
inline bool is_arith(uint32_t op) { return op >= R_ADD && op <= R_INV; }

// bytecode: one 64-bit word per instruction
//   bits 0-3 op | 4-6 kind of a | 7-9 kind of b | 10-23 dst slot (DST_NONE: result only in the accumulator; SINK: the
//   selector 0..3; STORE: unused) | 24-43 index of a | 44-63 index of b (STORE: the output column)
enum : uint32_t { I_ADD = 0, I_SUB = 1, I_MUL = 2, I_INV = 3, I_SINK = 4, I_STORE = 5, I_LOAD = 6 };  // I_LOAD: slot dst <- column a (kind LOCAL / NEXT)
enum : uint32_t { K_ACC = 0, K_SLOT = 1, K_UNI = 2, K_LOCAL = 3, K_NEXT = 4 };
constexpr uint32_t DST_NONE = 0x3FFF;
constexpr uint32_t MAX_INDEX = 1u << 20;
GL_HD uint64_t encode(uint32_t op, uint32_t ka, uint32_t kb, uint32_t dst, uint32_t ia, uint32_t ib) {
  return (uint64_t)op | ((uint64_t)ka << 4) | ((uint64_t)kb << 7) | ((uint64_t)dst << 10) | ((uint64_t)ia << 24) | ((uint64_t)ib << 44);
}

struct Program {
  int kind = 0;  // CP_AIR_CONSTRAINTS / CP_AIR_MAP
  std::vector<RawOp> ops;
  std::vector<uint64_t> consts;
  uint32_t n_columns = 0, n_public = 0, n_global = 0, n_challenge = 0, n_out_columns = 0;
  // derived
  std::vector<uint32_t> roots;  // sinks (constraint programs) or stores (map programs), program order
  std::vector<uint8_t> live;    // reaches a root
  size_t n_live = 0;
  uint32_t max_degree = 0;
  // uniform table layout: consts | publics | globals | challenges | 0
  uint32_t uni_public() const { return (uint32_t)consts.size(); }
  uint32_t uni_global() const { return uni_public() + n_public; }
  uint32_t uni_challenge() const { return uni_global() + n_global; }
  uint32_t uni_zero() const { return uni_challenge() + n_challenge; }
  uint32_t uni_size() const { return uni_zero() + 1; }
};

// "" = well-formed (and p.roots / live / max_degree filled); else what is wrong
inline std::string analyse(Program &p) {
  const size_t n = p.ops.size();
  auto bad = [&](size_t i, const char *why) { return "op " + std::to_string(i) + ": " + why; };
  if (n > ((size_t)1 << 24)) return "more than 2^24 ops";
  if (p.kind != 0 && p.kind != 1) return "kind must be CP_AIR_CONSTRAINTS or CP_AIR_MAP";
  if (p.n_columns >= MAX_INDEX || p.n_out_columns >= MAX_INDEX) return "more than 2^20 - 1 columns";
  if ((uint64_t)p.consts.size() + p.n_public + p.n_global + p.n_challenge + 1 >= MAX_INDEX) return "more than 2^20 - 2 uniform values (constants + publics + globals + challenges)";
  if (p.kind == 0 && p.n_out_columns) return "n_out_columns of a constraint program must be 0";
  for (uint64_t c : p.consts)
    if (c >= gl::P) return "a constant is not canonical";
  std::vector<uint32_t> deg(n, 0);
  std::vector<uint8_t> stored(p.n_out_columns, 0);
  p.roots.clear();
  p.max_degree = 0;
  for (size_t i = 0; i < n; i++) {
    const RawOp &o = p.ops[i];
    if (o.c != 0) return bad(i, "reserved field c is not 0");
    auto val = [&](uint32_t x) { return x < i && defines_value(p.ops[x].op); };
    auto sat = [](uint64_t d) { return (uint32_t)std::min<uint64_t>(d, 1u << 30); };
    switch (o.op) {
      case R_LOCAL: case R_NEXT:
        if (o.a >= p.n_columns) return bad(i, "column out of range");
        deg[i] = 1;
        break;
      case R_PUBLIC: if (o.a >= p.n_public) return bad(i, "public input out of range"); break;
      case R_GLOBAL: if (o.a >= p.n_global) return bad(i, "global value out of range"); break;
      case R_CHALLENGE: if (o.a >= p.n_challenge) return bad(i, "challenge out of range"); break;
      case R_CONST: if (o.a >= p.consts.size()) return bad(i, "constant out of range"); break;
      case R_ADD: case R_SUB:
        if (!val(o.a) || !val(o.b)) return bad(i, "operand is not an earlier value");
        deg[i] = std::max(deg[o.a], deg[o.b]);
        break;
      case R_MUL:
        if (!val(o.a) || !val(o.b)) return bad(i, "operand is not an earlier value");
        deg[i] = sat((uint64_t)deg[o.a] + deg[o.b]);
        break;
      case R_NEG:
        if (!val(o.a)) return bad(i, "operand is not an earlier value");
        deg[i] = deg[o.a];
        break;
      case R_INV:
        if (p.kind != 1) return bad(i, "INV is for map programs only");
        if (!val(o.a)) return bad(i, "operand is not an earlier value");
        break;
      case R_ASSERT: case R_ASSERT_TRANSITION: case R_ASSERT_FIRST: case R_ASSERT_LAST:
        if (p.kind != 0) return bad(i, "a map program has no constraints");
        if (!val(o.a)) return bad(i, "operand is not an earlier value");
        // degree in units of n: a first / last row constraint is multiplied by a Lagrange basis polynomial (degree n - 1: one
        // more unit); the transition factor (x - g^(n-1)) is one more in DEGREE, which the bound d <= 2^q + 1 already leaves room for
        p.max_degree = std::max(p.max_degree, sat((uint64_t)deg[o.a] + (o.op == R_ASSERT_FIRST || o.op == R_ASSERT_LAST ? 1 : 0)));
        p.roots.push_back((uint32_t)i);
        break;
      case R_STORE:
        if (p.kind != 1) return bad(i, "STORE is for map programs only");
        if (o.a >= p.n_out_columns) return bad(i, "output column out of range");
        if (!val(o.b)) return bad(i, "operand is not an earlier value");
        if (stored[o.a]) return bad(i, "output column stored twice");
        stored[o.a] = 1;
        p.roots.push_back((uint32_t)i);
        break;
      default: return bad(i, "unknown op code");
    }
  }
  // liveness from the roots
  p.live.assign(n, 0);
  for (size_t i = n; i-- > 0;) {
    const RawOp &o = p.ops[i];
    if (is_sink(o.op)) { p.live[i] = 1; p.live[o.a] = 1; }
    else if (o.op == R_STORE) { p.live[i] = 1; p.live[o.b] = 1; }
    else if (p.live[i]) {
      if (o.op == R_ADD || o.op == R_SUB || o.op == R_MUL) p.live[o.a] = p.live[o.b] = 1;
      else if (o.op == R_NEG || o.op == R_INV) p.live[o.a] = 1;
    }
  }
  p.n_live = 0;
  for (uint8_t l : p.live) p.n_live += l;
  return "";
}

struct Compiled {
  std::vector<uint64_t> code;
  std::vector<uint32_t> seg_off;      // [S + 1] into code
  std::vector<uint32_t> sinks_after;  // [S]: constraints in later segments (the exponent of this segment's alpha weight)
  uint32_t n_slots = 0;               // per-lane temporaries (max over segments)
  uint32_t n_segments() const { return (uint32_t)sinks_after.size(); }
};

// Cut the roots into at most `want_segments` runs of about equal work and emit each run's backward slice.
// prefetch: column operands are loaded into slots in batches of up to PREFETCH_BATCH, ahead of the instructions that use them (one
// memory latency per batch instead of one per operand); false (the default of the library: stark.inc air_prefetch): every column
// operand is a load at its use, hidden by the other waves of the CU.
constexpr size_t PREFETCH_BATCH = 8, PREFETCH_WINDOW = 32;
inline Compiled compile(const Program &p, uint32_t want_segments, bool prefetch = false) {
  Compiled C;
  const size_t n = p.ops.size();
  const size_t n_roots = p.roots.size();
  if (want_segments < 1) want_segments = 1;
  // work by position: live arithmetic ops up to each root
  std::vector<uint32_t> arith_before(n + 1, 0);
  for (size_t i = 0; i < n; i++) arith_before[i + 1] = arith_before[i] + (p.live[i] && (is_arith(p.ops[i].op) || is_sink(p.ops[i].op) || p.ops[i].op == R_STORE));
  const uint32_t total = arith_before[n];
  const uint32_t target = std::max<uint32_t>(1, (total + want_segments - 1) / want_segments);
  std::vector<std::pair<size_t, size_t>> runs;  // [first root, last root] index ranges
  {
    size_t first = 0;
    uint32_t start_work = 0;
    for (size_t r = 0; r < n_roots; r++) {
      const uint32_t w = arith_before[p.roots[r] + 1];
      if (w - start_work >= target || r + 1 == n_roots) {
        runs.push_back({first, r});
        first = r + 1;
        start_work = w;
      }
    }
    if (runs.empty()) runs.push_back({0, 0});  // a program without roots: one empty segment
  }
  std::vector<uint32_t> stamp(n, 0xFFFFFFFFu), uses(n, 0), slot(n, DST_NONE);
  std::vector<uint32_t> need;
  // one abstract instruction per emitted op: operands are uniform-table entries, columns (to be prefetched) or earlier values
  enum : uint8_t { O_UNI = 0, O_COL = 1, O_VAL = 2 };
  struct Operand { uint8_t kind; uint32_t ix; };  // O_COL: (next << 20) | column; O_VAL: the raw op index of the value
  struct AInstr { uint32_t op; Operand a, b; uint32_t aux; uint32_t def; };  // aux: sink selector / store column; def: raw op index or ~0
  std::vector<AInstr> ai;
  C.seg_off.push_back(0);
  for (size_t s = 0; s < runs.size(); s++) {
    need.clear();
    // backward slice of this run's roots
    std::vector<uint32_t> stack;
    if (n_roots)
      for (size_t r = runs[s].first; r <= runs[s].second; r++) stack.push_back(p.roots[r]);
    while (!stack.empty()) {
      const uint32_t i = stack.back();
      stack.pop_back();
      if (stamp[i] == (uint32_t)s) continue;
      stamp[i] = (uint32_t)s;
      uses[i] = 0;
      slot[i] = DST_NONE;
      const RawOp &o = p.ops[i];
      if (is_load(o.op) || is_uniform(o.op)) continue;  // folded into operands: nothing to emit, nothing to visit
      need.push_back(i);
      if (o.op == R_STORE) stack.push_back(o.b);
      else {
        stack.push_back(o.a);
        if (o.op == R_ADD || o.op == R_SUB || o.op == R_MUL) stack.push_back(o.b);
      }
    }
    std::sort(need.begin(), need.end());
    // ---- abstract instructions ----
    auto operand = [&](uint32_t x) -> Operand {
      const RawOp &src = p.ops[x];
      if (src.op == R_LOCAL) return {O_COL, src.a};
      if (src.op == R_NEXT) return {O_COL, (1u << 20) | src.a};
      if (src.op == R_CONST) return {O_UNI, src.a};
      if (src.op == R_PUBLIC) return {O_UNI, p.uni_public() + src.a};
      if (src.op == R_GLOBAL) return {O_UNI, p.uni_global() + src.a};
      if (src.op == R_CHALLENGE) return {O_UNI, p.uni_challenge() + src.a};
      return {O_VAL, x};
    };
    ai.clear();
    const Operand zero{O_UNI, p.uni_zero()};
    for (uint32_t i : need) {
      const RawOp &o = p.ops[i];
      switch (o.op) {
        case R_ADD: ai.push_back({I_ADD, operand(o.a), operand(o.b), 0, i}); break;
        case R_SUB: ai.push_back({I_SUB, operand(o.a), operand(o.b), 0, i}); break;
        case R_MUL: ai.push_back({I_MUL, operand(o.a), operand(o.b), 0, i}); break;
        case R_NEG: ai.push_back({I_SUB, zero, operand(o.a), 0, i}); break;  // 0 - a
        case R_INV: ai.push_back({I_INV, operand(o.a), zero, 0, i}); break;
        case R_STORE: ai.push_back({I_STORE, operand(o.b), zero, o.a, 0xFFFFFFFFu}); break;
        default: ai.push_back({I_SINK, operand(o.a), zero, o.op - R_ASSERT, 0xFFFFFFFFu}); break;
      }
    }
    for (const AInstr &x : ai) {
      if (x.a.kind == O_VAL) uses[x.a.ix]++;
      if (x.b.kind == O_VAL) uses[x.b.ix]++;
    }
    // ---- emission: slots by liveness, column operands prefetched in batches ----
    std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> free_slots;
    uint32_t next_slot = 0;
    auto take_slot = [&]() {
      if (free_slots.empty()) return next_slot++;
      const uint32_t v = free_slots.top();
      free_slots.pop();
      return v;
    };
    struct Loaded { uint32_t slot, uses_left; };
    std::map<uint32_t, Loaded> loaded;  // column key -> where it sits and how many operand occurrences it still serves
    for (size_t t = 0; t < ai.size(); t++) {
      const AInstr &x = ai[t];
      if (prefetch) {
        // a column operand that is not in a slot: load it together with the next columns the instructions ahead will need
        const bool miss = (x.a.kind == O_COL && !loaded.count(x.a.ix)) || (x.b.kind == O_COL && !loaded.count(x.b.ix));
        if (miss) {
          std::vector<uint32_t> keys;
          size_t u = t;
          for (; u < ai.size() && u < t + PREFETCH_WINDOW && keys.size() < PREFETCH_BATCH; u++)
            for (const Operand *o : {&ai[u].a, &ai[u].b})
              if (o->kind == O_COL && !loaded.count(o->ix) && std::find(keys.begin(), keys.end(), o->ix) == keys.end() && keys.size() < PREFETCH_BATCH)
                keys.push_back(o->ix);
          // the batch serves the occurrences of its keys in [t, u): count them (the last scanned instruction may be cut short when
          // the batch filled up on its first operand: count only what was taken)
          for (uint32_t k : keys) loaded[k] = Loaded{take_slot(), 0};
          for (size_t v = t; v < u; v++)
            for (const Operand *o : {&ai[v].a, &ai[v].b})
              if (o->kind == O_COL) {
                auto it = loaded.find(o->ix);
                if (it != loaded.end() && std::find(keys.begin(), keys.end(), o->ix) != keys.end()) it->second.uses_left++;
              }
          for (uint32_t k : keys) C.code.push_back(encode(I_LOAD, (k >> 20) ? K_NEXT : K_LOCAL, K_UNI, loaded[k].slot, k & 0xFFFFF, 0));
        }
      }
      const uint32_t prev = t ? ai[t - 1].def : 0xFFFFFFFFu;  // the value the accumulator holds (when the last instruction defined one)
      auto place = [&](const Operand &o, uint32_t &k, uint32_t &ix) {
        if (o.kind == O_UNI) { k = K_UNI; ix = o.ix; }
        else if (o.kind == O_COL) {
          if (prefetch) { k = K_SLOT; ix = loaded[o.ix].slot; }
          else { k = (o.ix >> 20) ? K_NEXT : K_LOCAL; ix = o.ix & 0xFFFFF; }
        } else if (o.ix == prev) { k = K_ACC; ix = 0; }
        else { k = K_SLOT; ix = slot[o.ix]; }
      };
      uint32_t ka, ia, kb, ib, dst = DST_NONE;
      place(x.a, ka, ia);
      place(x.b, kb, ib);
      if (x.op == I_SINK) dst = x.aux;
      if (x.op == I_STORE) ib = x.aux;
      // operands consumed: slots whose last use this was go back to the pool BEFORE the result takes one (the interpreter reads both
      // operands before it writes)
      for (const Operand *o : {&x.a, &x.b}) {
        if (o->kind == O_VAL) {
          if (--uses[o->ix] == 0 && slot[o->ix] != DST_NONE) { free_slots.push(slot[o->ix]); slot[o->ix] = DST_NONE; }
        } else if (o->kind == O_COL && prefetch) {
          auto it = loaded.find(o->ix);
          if (--it->second.uses_left == 0) { free_slots.push(it->second.slot); loaded.erase(it); }
        }
      }
      if (x.def != 0xFFFFFFFFu) {
        // a slot unless every use is an operand of the very next instruction
        uint32_t next_uses = 0;
        if (t + 1 < ai.size()) next_uses = (ai[t + 1].a.kind == O_VAL && ai[t + 1].a.ix == x.def) + (ai[t + 1].b.kind == O_VAL && ai[t + 1].b.ix == x.def);
        if (uses[x.def] > next_uses) dst = slot[x.def] = take_slot();
      }
      C.code.push_back(encode(x.op, ka, kb, dst, ia, ib));
    }
    C.n_slots = std::max(C.n_slots, next_slot);
    C.seg_off.push_back((uint32_t)C.code.size());
    C.sinks_after.push_back(p.kind == 0 && n_roots ? (uint32_t)(n_roots - 1 - runs[s].second) : 0);
  }
  return C;
}

// compile-time loop over the points of a lane: indices are constants in the frontend already, so the small per-point arrays
// (positions, flags) are scalarised into registers instead of living in scratch behind "dynamic" indices
template <int N, class F>
GL_HD void for_k(F f) {
  if constexpr (N > 0) {
    for_k<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// K points per lane: a value of the interpreter is K field elements (one per point), so that one decoded instruction - the scalar
// work that bounds this kernel - serves K x 64 points
template <int K>
struct Vec { uint64_t v[K]; };
template <int K> GL_HD Vec<K> vsplat(uint64_t x) { Vec<K> r; for (int k = 0; k < K; k++) r.v[k] = x; return r; }
template <int K> GL_HD Vec<K> vadd(const Vec<K> &a, const Vec<K> &b) { Vec<K> r; for (int k = 0; k < K; k++) r.v[k] = gl::add(a.v[k], b.v[k]); return r; }
template <int K> GL_HD Vec<K> vsub(const Vec<K> &a, const Vec<K> &b) { Vec<K> r; for (int k = 0; k < K; k++) r.v[k] = gl::sub(a.v[k], b.v[k]); return r; }
template <int K> GL_HD Vec<K> vmul(const Vec<K> &a, const Vec<K> &b) { Vec<K> r; for (int k = 0; k < K; k++) r.v[k] = gl::mul(a.v[k], b.v[k]); return r; }
template <int K> GL_HD Vec<K> vscale(const Vec<K> &a, uint64_t s) { Vec<K> r; for (int k = 0; k < K; k++) r.v[k] = gl::mul(a.v[k], s); return r; }
template <int K> GL_HD Vec<K> vinv(const Vec<K> &a) { Vec<K> r; for (int k = 0; k < K; k++) r.v[k] = a.v[k] ? gl::inv(a.v[k]) : 0; return r; }

// per-point selector values of the quotient (starky's ConstraintConsumer): z_last = x - g^(n-1), L_0(x), L_(n-1)(x)
template <int K>
struct Selectors { Vec<K> z_last, l_first, l_last; };

constexpr int MAX_ALPHAS = 4;

// ---- the instruction semantics, shared by the host executor (tests/hostsim; cp_air_program_eval_ext has its own F_p^2 form) and the
// device interpreter ----
template <class Mem>  // Mem: K, code(pc), alpha(c), slot_read(i), slot_write(i, v), uni(i), local(i), next(i), store(col, v)
GL_HD void run_segment(uint32_t first, uint32_t last, Mem &m, int n_alphas, const Selectors<Mem::K> &sel,
                       Vec<Mem::K> *acc_out /* [MAX_ALPHAS], Horner sums of this segment */) {
  constexpr int K = Mem::K;
  using V = Vec<K>;
  V acc = vsplat<K>(0);
#pragma unroll
  for (int c = 0; c < MAX_ALPHAS; c++) acc_out[c] = vsplat<K>(0);
  for (uint32_t pc = first; pc < last; pc++) {
    const uint64_t w = m.code(pc);
    const uint32_t op = (uint32_t)w & 15, ka = (uint32_t)(w >> 4) & 7, kb = (uint32_t)(w >> 7) & 7, dst = (uint32_t)(w >> 10) & 0x3FFF,
                   ia = (uint32_t)(w >> 24) & 0xFFFFF, ib = (uint32_t)(w >> 44);
    auto fetch = [&](uint32_t k, uint32_t i) -> V {
      switch (k) {
        case K_ACC: return acc;
        case K_SLOT: return m.slot_read(i);
        case K_UNI: return vsplat<K>(m.uni(i));
        case K_LOCAL: return m.local(i);
        default: return m.next(i);
      }
    };
    if (op == I_LOAD) {
      if constexpr (K > 1) {  // batched prefetch is a one-point-per-lane form (stark.inc keeps K = 1 when it is on): a plain load here
        m.slot_write(dst, ka == K_NEXT ? m.next(ia) : m.local(ia));
        continue;
      }
      // a run of up to PREFETCH_BATCH loads: every load is issued before the first value is needed, then the slots are written
      V v[PREFETCH_BATCH];
      uint32_t d[PREFETCH_BATCH];
      uint32_t r = 0;
#pragma unroll
      for (uint32_t j = 0; j < PREFETCH_BATCH; j++) {
        if (j == r && pc + j < last) {
          const uint64_t wj = m.code(pc + j);
          if (((uint32_t)wj & 15) == I_LOAD) {
            const uint32_t col = (uint32_t)(wj >> 24) & 0xFFFFF;
            v[j] = (((uint32_t)(wj >> 4) & 7) == K_NEXT) ? m.next(col) : m.local(col);
            d[j] = (uint32_t)(wj >> 10) & 0x3FFF;
            r = j + 1;
          }
        }
      }
#pragma unroll
      for (uint32_t j = 0; j < PREFETCH_BATCH; j++)
        if (j < r) m.slot_write(d[j], v[j]);
      pc += r - 1;
      continue;
    }
    const V a = fetch(ka, ia);
    if (op == I_SINK) {
      V v = a;
      if (dst != 0) {  // element-wise selects keep the three selector vectors in registers (a select between whole structs goes through scratch)
        V sv;
        for (int k = 0; k < K; k++) sv.v[k] = dst == 1 ? sel.z_last.v[k] : dst == 2 ? sel.l_first.v[k] : sel.l_last.v[k];
        v = vmul<K>(a, sv);
      }
#pragma unroll
      for (int c = 0; c < MAX_ALPHAS; c++)
        if (c < n_alphas) acc_out[c] = vadd<K>(vscale<K>(acc_out[c], m.alpha(c)), v);
      continue;
    }
    if (op == I_STORE) { m.store(ib, a); continue; }
    V r;
    if (op == I_INV) r = vinv<K>(a);
    else {
      const V b = fetch(kb, ib);
      r = op == I_ADD ? vadd<K>(a, b) : op == I_SUB ? vsub<K>(a, b) : vmul<K>(a, b);
    }
    if (dst != DST_NONE) m.slot_write(dst, r);
    acc = r;
  }
}

// ---- device side ----
constexpr int WAVE = 64;
constexpr uint32_t MAX_LDS_SLOTS = 10;  // x K points per lane: 512 B per slot, point and wave (10 KB per wave at K = 2: sixteen waves per
                                        // CU); slots beyond live in global scratch (the allocator hands out the lowest free number first, so
                                        // the high numbers are the long-lived, rarely touched ones). Measured: profiles/r04_air_ab3.jsonl -
                                        // the launch is bound by the latency of its dependent scalar-load -> vector-load -> LDS chains, and
                                        // resident waves are what hides it: 4 / 6 / 8 / 10 / 12 slots at two points per lane = 2.23 / 2.04 /
                                        // 1.93 / 1.88 / 2.32 ms at 2^16 rows (24 slots, one point: 3.09)

struct KArgs {
  const uint64_t *code;
  const uint32_t *seg_off;
  const uint64_t *uni;
  const uint64_t *const *cols;   // n_columns column base pointers
  const uint64_t *sel;           // quotient: [3][M] z_last, l_first, l_last in storage order
  const uint64_t *alphas;        // [n_alphas]
  const uint64_t *weights;       // [S][n_alphas]: alpha^(sinks after the segment)
  uint64_t *parts;               // quotient: [S][n_alphas][M]
  uint64_t *spill;               // [n_slots - n_lds][S * lanes of the grid]
  uint64_t *const *out_cols;     // map: n_out_columns column base pointers
  size_t M;                      // points (quotient: n << q; map: n rows)
  size_t spill_stride;           // S * (point lanes of the grid, whole waves)
  int degree_bits, n_alphas, n_lds;
};

// The bytecode, the uniform table, the column pointer table, the alphas and the weights are read at wave-uniform addresses. Read
// through ordinary global pointers they are VECTOR loads (the kernel also stores to global memory, so the compiler may not treat
// them as invariant): one memory round trip per interpreted instruction for the instruction word alone, another for a pointer
// before the column load that needs it. Through the constant address space they are scalar loads (s_load: the scalar cache,
// invariant by definition) - the kernel never writes what it reads this way.
#if defined(__HIP_DEVICE_COMPILE__)
#define AIR_CONSTANT_AS __attribute__((address_space(4)))
#else
#define AIR_CONSTANT_AS
#endif
template <class T>
__device__ __forceinline__ const AIR_CONSTANT_AS T *constant_as(const T *p) {
  return (const AIR_CONSTANT_AS T *)p;
}

template <int KP>
struct DevMem {
  static constexpr int K = KP;
  uint64_t *lds;  // this wave's slots, [slot][point of the lane][lane]
  const KArgs &a;
  size_t pos[KP], npos[KP], gid[KP];
  bool active[KP];
  int lane;
  __device__ __forceinline__ uint64_t code(uint32_t pc) const { return constant_as(a.code)[pc]; }
  __device__ __forceinline__ uint64_t alpha(int c) const { return constant_as(a.alphas)[c]; }
  __device__ __forceinline__ Vec<KP> slot_read(uint32_t i) const {
    Vec<KP> r;
    if ((int)i < a.n_lds) for_k<KP>([&](auto k) { r.v[k] = lds[(i * KP + k) * WAVE + lane]; });
    else for_k<KP>([&](auto k) { r.v[k] = a.spill[(size_t)(i - a.n_lds) * a.spill_stride + gid[k]]; });
    return r;
  }
  __device__ __forceinline__ void slot_write(uint32_t i, const Vec<KP> &v) const {
    if ((int)i < a.n_lds) for_k<KP>([&](auto k) { lds[(i * KP + k) * WAVE + lane] = v.v[k]; });
    else for_k<KP>([&](auto k) { a.spill[(size_t)(i - a.n_lds) * a.spill_stride + gid[k]] = v.v[k]; });
  }
  __device__ __forceinline__ uint64_t uni(uint32_t i) const { return constant_as(a.uni)[i]; }
  __device__ __forceinline__ Vec<KP> local(uint32_t i) const {
    const uint64_t *c = (const uint64_t *)constant_as(a.cols)[i];
    Vec<KP> r;
    for_k<KP>([&](auto k) { r.v[k] = c[pos[k]]; });
    return r;
  }
  __device__ __forceinline__ Vec<KP> next(uint32_t i) const {
    const uint64_t *c = (const uint64_t *)constant_as(a.cols)[i];
    Vec<KP> r;
    for_k<KP>([&](auto k) { r.v[k] = c[npos[k]]; });
    return r;
  }
  __device__ __forceinline__ void store(uint32_t col, const Vec<KP> &v) const {
    uint64_t *c = (uint64_t *)constant_as(a.out_cols)[col];
    for_k<KP>([&](auto k) {
      if (active[k]) c[pos[k]] = v.v[k];
    });
  }
};

// MODE 0: quotient (points in storage order: pos = [coset block][bit-reversed row]); MODE 1: map (pos = row, natural order).
// A lane works on KP points, 64 apart: the wave covers KP * 64 consecutive points.
template <int MODE, int KP>
__global__ __launch_bounds__(WAVE) void k_run(KArgs a) {
  extern __shared__ uint64_t lds[];
  const int lane = threadIdx.x;
  const uint32_t seg = blockIdx.y;
  DevMem<KP> m{lds, a, {}, {}, {}, {}, lane};
  Selectors<KP> sel;
  for_k<KP>([&](auto k) {
    const size_t p0 = ((size_t)blockIdx.x * KP + k) * WAVE + lane;
    m.active[k] = p0 < a.M;
    const size_t pos = m.active[k] ? p0 : a.M - 1;  // idle lanes of a short grid shadow the last point (loads stay in bounds)
    m.pos[k] = pos;
    m.gid[k] = ((size_t)seg * gridDim.x * KP + (size_t)blockIdx.x * KP + k) * WAVE + lane;
    if (MODE == 0) {
      const size_t n = (size_t)1 << a.degree_bits;
      const uint32_t q = (uint32_t)(pos & (n - 1));
      const uint32_t r = a.degree_bits ? __brev(q) >> (32 - a.degree_bits) : 0;
      const uint32_t r1 = (r + 1) & (uint32_t)(n - 1);
      m.npos[k] = (pos & ~(n - 1)) | (a.degree_bits ? __brev(r1) >> (32 - a.degree_bits) : 0);
      sel.z_last.v[k] = a.sel[pos];
      sel.l_first.v[k] = a.sel[a.M + pos];
      sel.l_last.v[k] = a.sel[2 * a.M + pos];
    } else {
      m.npos[k] = pos + 1 == a.M ? 0 : pos + 1;
      sel.z_last.v[k] = sel.l_first.v[k] = sel.l_last.v[k] = 0;
    }
  });
  Vec<KP> acc[MAX_ALPHAS];
  run_segment(constant_as(a.seg_off)[seg], constant_as(a.seg_off)[seg + 1], m, MODE == 0 ? a.n_alphas : 0, sel, acc);
  if (MODE == 0) {
#pragma unroll
    for (int c = 0; c < MAX_ALPHAS; c++)
      if (c < a.n_alphas) {
        const uint64_t wgt = constant_as(a.weights)[seg * a.n_alphas + c];
        for_k<KP>([&](auto k) {
          if (m.active[k]) a.parts[((size_t)seg * a.n_alphas + c) * a.M + m.pos[k]] = gl::mul(acc[c].v[k], wgt);
        });
      }
  }
}

// z_last, l_first, l_last at every point of the quotient coset 7<omega_M>, storage order. omega_tab: power table of omega_M
// (3 x 2048, see get_pow_table). One-off per (degree_bits, q) and context.
__global__ __launch_bounds__(256) void k_selectors(uint64_t *out, size_t M, int log_M, int degree_bits, const uint64_t *omega_tab, uint64_t g_last,
                                                   uint64_t n_inv) {
  const size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pos >= M) return;
  // storage position -> natural index of the coset: [coset block c'][bit-reversed row q] <-> nat = r * 2^q_bits + c
  const uint32_t nat = log_M ? __brev((uint32_t)pos) >> (32 - log_M) : 0;
  const uint32_t e0 = nat & 2047, e1 = (nat >> 11) & 2047, e2 = nat >> 22;
  uint64_t w = omega_tab[e0];
  if (e1) w = gl::mul(w, omega_tab[2048 + e1]);
  if (e2) w = gl::mul(w, omega_tab[4096 + e2]);
  const uint64_t x = gl::mul(7, w);
  uint64_t xn = x;
  for (int i = 0; i < degree_bits; i++) xn = gl::sqr(xn);
  const uint64_t zh = gl::sub(xn, 1), zl = gl::sub(x, g_last), zhn = gl::mul(zh, n_inv);
  out[pos] = zl;
  out[M + pos] = gl::mul(zhn, gl::inv(gl::sub(x, 1)));
  out[2 * M + pos] = gl::mul(gl::mul(zhn, g_last), gl::inv(zl));
}

// sum of the segments' parts, / Z_H (2^q distinct values: zh_inv[nat mod 2^q]), scattered to the NATURAL order the coset iNTT reads.
// A workgroup = 64 points x 4 groups of segments (a small trace is cut into hundreds of segments: one thread per point would walk
// them all in sequence); the four partial sums meet in LDS.
__global__ __launch_bounds__(256) void k_finish(const uint64_t *parts, uint32_t n_segments, int n_alphas, size_t M, int log_M, int q_bits,
                                                const uint64_t *zh_inv, uint64_t *out) {
  __shared__ uint64_t part[MAX_ALPHAS][4][WAVE];
  const int lane = threadIdx.x & (WAVE - 1), grp = threadIdx.x / WAVE;
  const size_t pos = (size_t)blockIdx.x * WAVE + lane;
  const bool active = pos < M;
#pragma unroll
  for (int c = 0; c < MAX_ALPHAS; c++)
    if (c < n_alphas) {
      uint64_t s = 0;
      if (active)
        for (uint32_t g = grp; g < n_segments; g += 4) s = gl::add(s, parts[((size_t)g * n_alphas + c) * M + pos]);
      part[c][grp][lane] = s;
    }
  __syncthreads();
  if (grp != 0 || !active) return;
  const uint32_t nat = log_M ? __brev((uint32_t)pos) >> (32 - log_M) : 0;
  const uint64_t zi = zh_inv[nat & ((1u << q_bits) - 1)];
#pragma unroll
  for (int c = 0; c < MAX_ALPHAS; c++)
    if (c < n_alphas) {
      const uint64_t s = gl::add(gl::add(part[c][0][lane], part[c][1][lane]), gl::add(part[c][2][lane], part[c][3][lane]));
      out[(size_t)c * M + nat] = gl::mul(s, zi);
    }
}

}  // namespace air
