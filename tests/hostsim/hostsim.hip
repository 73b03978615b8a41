// TEST-ONLY: host-side instantiation of the __host__ __device__ arithmetic in city-rollup_amd/csrc
// so that the exact device formulas (lazy reductions, limb MDS, shift twiddles) can be unit-tested and
// run under sanitizers on the CPU. Never loaded by the product path.
#include "../../city-rollup_amd/csrc/gl.h"
#include "../../city-rollup_amd/csrc/poseidon.h"

extern "C" {
void hs_poseidon_permute(uint64_t *states, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint64_t s[12];
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    poseidon::permute(s);
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
  }
}
void hs_poseidon_permute_textbook(uint64_t *states, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint64_t s[12];
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    poseidon::permute_textbook(s);
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
  }
}
// double-precision layers (poseidon.h): carry normalisation of a two-limb value; one plane through T / K / T^-1'
void hs_renorm_d(const double *in, double *out) {
  out[0] = in[0], out[1] = in[1];
  poseidon::renorm_d(out[0], out[1]);
}
void hs_dom_d(int op, const double *s, double *y) {  // 0: dom_enter_d, 1: dom_mul_d<false>, 2: dom_mul_d<true>, 3: dom_leave_d
  double a[12], b[12];
  for (int i = 0; i < 12; i++) a[i] = s[i];
  if (op == 0) poseidon::dom_enter_d(a, b);
  else if (op == 1) poseidon::dom_mul_d<false>(a, b);
  else if (op == 2) poseidon::dom_mul_d<true>(a, b);
  else poseidon::dom_leave_d(a, b);
  for (int i = 0; i < 12; i++) y[i] = b[i];
}
uint64_t hs_recombine_d(double l, double h, uint64_t c) {  // c: the constant already minus B (1 + 2^32), as the tables hold it
  return poseidon::recombine_d(l, h, poseidon::Magic{0x1.8p52 + (double)(uint32_t)c, 0x1.8p52 + (double)(uint32_t)(c >> 32)});
}
void hs_mds_layer(uint64_t *s, int which) {  // 0: integer planes (mds_layer), 1: double-precision planes (mds_layer_d); no constant
  uint64_t t[12];
  for (int i = 0; i < 12; i++) t[i] = s[i];
  if (which) poseidon::mds_layer_d(t, -1); else poseidon::mds_layer(t, -1);
  for (int i = 0; i < 12; i++) s[i] = gl::canon(t[i]);
}
uint64_t hs_mul(uint64_t a, uint64_t b) { return gl::mul(a, b); }
uint64_t hs_mul_lazy(uint64_t a, uint64_t b) { return poseidon::mul_lazy(a, b); }
uint64_t hs_add(uint64_t a, uint64_t b) { return gl::add(a, b); }
uint64_t hs_sub(uint64_t a, uint64_t b) { return gl::sub(a, b); }
void hs_mds_limb(const uint32_t *s, uint32_t *y) {
  uint32_t a[12], b[12];
  for (int i = 0; i < 12; i++) a[i] = s[i];
  poseidon::mds_limb(a, b);
  for (int i = 0; i < 12; i++) y[i] = b[i];
}
}
#include "../../city-rollup_amd/csrc/ntt16.h"
extern "C" {
#define MP(K) case K: return ntt16::mul_pow2<K>(x);
uint64_t hs_mul_pow2(uint64_t x, int k) {
  switch (k) {
    MP(0) MP(1) MP(5) MP(12) MP(24) MP(31) MP(32) MP(33) MP(36) MP(48) MP(60) MP(63) MP(64) MP(65) MP(72) MP(84) MP(95)
    default: return ~0ull;
  }
}
void hs_round16(uint64_t *x, int inverse) {
  uint64_t r[16];
  for (int i = 0; i < 16; i++) r[i] = x[i];
  if (inverse) ntt16::round16<true, 4, true>(r, 1); else ntt16::round16<false, 4, true>(r, 1);
  for (int i = 0; i < 16; i++) x[i] = r[i];
}
}

// ---- BLS12-381 (csrc/bls12_381.h) ----
#include "../../city-rollup_amd/csrc/bls12_381.h"
extern "C" {
// canonical 12-word operands -> canonical (a*b mod p, a+b, a-b by op 0/1/2; 3: a^-1; 4: a^2 by the dedicated square)
void hs_bls_fp_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
  bls::Fp x = bls::fp_from_canonical(a), y = bls::fp_from_canonical(b), r;
  switch (op) {
    case 0: r = bls::fp_mul(x, y); break;
    case 1: r = bls::fp_add(x, y); break;
    case 2: r = bls::fp_sub(x, y); break;
    case 4: r = bls::fp_sqr_mont(x); break;
    default: r = bls::fp_inv(x); break;
  }
  bls::fp_to_canonical(r, out);
}
// F_p^2: 24-word operands (c0, c1); op 0 mul, 1 sqr, 2 inv
void hs_bls_fp2_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
  using F2 = bls::Field<bls::Fp2>;
  bls::Fp2 x = F2::from_canonical(a), y = F2::from_canonical(b), r;
  switch (op) {
    case 0: r = bls::fp2_mul(x, y); break;
    case 1: r = bls::fp2_sqr(x); break;
    default: r = bls::fp2_inv(x); break;
  }
  F2::to_canonical(r, out);
}
}
// op 0: general add (both lifted to Jacobian, p scaled to a non-trivial z first), 1: mixed add, 2: double p, 3: p * k
template <class F>
static int hs_group_op(int op, const uint32_t *p_xy, int p_inf, const uint32_t *q_xy, int q_inf, uint32_t k, uint32_t *out_xy) {
  using Fd = bls::Field<F>;
  bls::JacT<F> p = bls::jac_inf<F>(), q = bls::jac_inf<F>(), r;
  bls::AffineT<F> qa = bls::affine_from_canonical<F>(q_xy);
  if (!p_inf) {  // give p a z != 1 so the projective formulas are exercised: (x z^2, y z^3, z), z = 3
    const bls::AffineT<F> pa = bls::affine_from_canonical<F>(p_xy);
    const F z = bls::f_add(Fd::one(), bls::f_add(Fd::one(), Fd::one())), z2 = bls::f_sqr(z);
    p = {bls::f_mul(pa.x, z2), bls::f_mul(pa.y, bls::f_mul(z2, z)), z};
  }
  if (!q_inf) q = {qa.x, qa.y, Fd::one()};
  switch (op) {
    case 0: r = bls::jac_add(p, q); break;
    case 1: r = q_inf ? p : bls::jac_add_mixed(p, qa); break;
    case 2: r = bls::jac_double(p); break;
    default: r = bls::jac_mul_small(p, k); break;
  }
  return bls::jac_to_affine_canonical(r, out_xy) ? 1 : 0;
}
// the bucket accumulation on loose (bounded, unreduced) values: sum of n affine points (xy: n x 2 x WORDS canonical words,
// none at infinity) through xyzz_add_mixed_loose, result canonical affine; returns 1 for infinity. max_bound_p (optional):
// the largest coordinate component seen, in units of p rounded up — what LooseBound promises is checked by the caller.
template <class F>
static int hs_bucket_chain(const uint32_t *xy, size_t n, uint32_t *out_xy) {
  constexpr int W = bls::Field<F>::WORDS;
  bls::XyzzT<F> acc = bls::xyzz_inf<F>();
  for (size_t i = 0; i < n; i++) acc = bls::xyzz_add_mixed_loose(acc, bls::affine_from_canonical<F>(xy + 2 * W * i));
  return bls::jac_to_affine_canonical(bls::xyzz_to_jac_loose(acc), out_xy) ? 1 : 0;
}
extern "C" {
int hs_bls_g1_chain(const uint32_t *xy, size_t n, uint32_t *out_xy) { return hs_bucket_chain<bls::Fp>(xy, n, out_xy); }
int hs_bls_g2_chain(const uint32_t *xy, size_t n, uint32_t *out_xy) { return hs_bucket_chain<bls::Fp2>(xy, n, out_xy); }
// loose field primitives against Python integers: op 0 lz_mul, 1 lz_add, 2 lz_sub<8>, 3 lz_weak<16>, 4 lz_canon; a, b: 14 raw limbs
void hs_bls_lz_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
  bls::Fp x, y, r;
  for (int i = 0; i < bls::NL; i++) x.l[i] = a[i], y.l[i] = b[i];
  switch (op) {
    case 0: r = bls::lz_mul(x, y); break;
    case 1: r = bls::lz_add(x, y); break;
    case 2: r = bls::lz_sub<8>(x, y); break;
    case 3: r = bls::lz_weak<16>(x); break;
    default: r = bls::lz_canon(x); break;
  }
  for (int i = 0; i < bls::NL; i++) out[i] = r.l[i];
}
int hs_bls_lz_maybe_zero(const uint32_t *a) {
  bls::Fp x;
  for (int i = 0; i < bls::NL; i++) x.l[i] = a[i];
  return bls::lz_maybe_zero(x) ? 1 : 0;
}
}
extern "C" {
int hs_bls_g1_op(int op, const uint32_t *p_xy, int p_inf, const uint32_t *q_xy, int q_inf, uint32_t k, uint32_t *out_xy) {
  return hs_group_op<bls::Fp>(op, p_xy, p_inf, q_xy, q_inf, k, out_xy);
}
int hs_bls_g2_op(int op, const uint32_t *p_xy, int p_inf, const uint32_t *q_xy, int q_inf, uint32_t k, uint32_t *out_xy) {
  return hs_group_op<bls::Fp2>(op, p_xy, p_inf, q_xy, q_inf, k, out_xy);
}
}

// ---- BLS12-381 scalar field (csrc/bls12_381_fr.h) ----
#include "../../city-rollup_amd/csrc/bls12_381_fr.h"
extern "C" {
// 8-word canonical operands; op 0 mul, 1 add, 2 sub, 3 inverse of a, 4 a^(b[0])
void hs_bls_fr_op(int op, const uint32_t *a, const uint32_t *b, uint32_t *out) {
  blsfr::Fr x = blsfr::fr_from_canonical(a), y = blsfr::fr_from_canonical(b), r;
  switch (op) {
    case 0: r = blsfr::fr_mul(x, y); break;
    case 1: r = blsfr::fr_add(x, y); break;
    case 2: r = blsfr::fr_sub(x, y); break;
    case 3: r = blsfr::fr_inv(x); break;
    default: r = blsfr::fr_pow_u64(x, ((uint64_t)b[1] << 32) | b[0]); break;
  }
  blsfr::fr_to_canonical(r, out);
}
}

// ---- host_util.h: worker threads that cannot take the process down ----
#include "../../city-rollup_amd/csrc/host_util.h"
extern "C" {
// runs parallel_for over n indices on max_threads threads with the `fail_after`-th thread creation failing
// (-1: none); out[p] += p + 1 for every index done; returns the number of indices whose body ran
long hs_parallel_for(size_t n, size_t max_threads, long fail_after, uint64_t *out) {
  hostu::fault_counter(0).store(fail_after);
  std::atomic<long> ran{0};
  hostu::parallel_for(n, max_threads, [&](size_t p) { out[p] += p + 1; ran++; });
  hostu::fault_counter(0).store(-1);
  return ran.load();
}
// body that throws std::bad_alloc at index `bad`: returns 1 when the exception reached the caller after all helpers
// were joined, 0 when nothing was thrown
int hs_parallel_for_throw(size_t n, size_t max_threads, size_t bad) {
  try {
    hostu::parallel_for(n, max_threads, [&](size_t p) { if (p == bad) throw std::bad_alloc(); });
  } catch (const std::bad_alloc &) {
    return 1;
  }
  return 0;
}

// ---- dev_pool.h (the batch-handle buffer pool) over a counting allocator with a capacity: a script of operations in, the
// allocator's and the pool's counters out. op: 0 alloc(device, bytes) -> handle index, 1 release(handle, reusable),
// 2 trim(device), 3 malloc(device, bytes) -> handle index (the path every other allocation of the library takes)
}  // extern "C"
#include "../../city-rollup_amd/csrc/dev_pool.h"
#include <set>
namespace poolsim {
struct Fake {
  static constexpr int OOM = 2;
  static inline size_t capacity = 0, in_use = 0, n_malloc = 0, n_free = 0, n_oom = 0;
  static inline std::map<void *, size_t> live;
  static inline uintptr_t next = 0x1000;
  static int malloc(void **p, size_t bytes) {
    if (in_use + bytes > capacity) { n_oom++; return OOM; }
    *p = (void *)next;
    next += 0x1000;
    live[*p] = bytes;
    in_use += bytes;
    n_malloc++;
    return 0;
  }
  static void free(void *p) {
    auto it = live.find(p);
    if (it == live.end()) { n_free = (size_t)-1; return; }  // double free / foreign pointer: poison the counter
    in_use -= it->second;
    live.erase(it);
    n_free++;
  }
};
}  // namespace poolsim
extern "C" {
// ops: n x {op, device, bytes_or_handle, reusable}; results: n x int64 (handle index, -OOM, or bytes trimmed);
// counters_out: {in_use, n_malloc, n_free, n_oom, live buffers, pooled bytes dev0, pooled bytes dev1, hits0, misses0, trims0}
int hs_pool_script(size_t n_devices, size_t pool_cap, size_t capacity, const int64_t *ops, size_t n, int64_t *results, uint64_t *counters_out) {
  using poolsim::Fake;
  Fake::capacity = capacity; Fake::in_use = 0; Fake::n_malloc = Fake::n_free = Fake::n_oom = 0; Fake::live.clear();
  DevPoolT<Fake> pool(n_devices, pool_cap);
  struct H { void *p; size_t bytes; int device; };
  std::vector<H> handles;
  std::set<void *> out;  // buffers currently handed out: a pool must never hand one out twice
  for (size_t i = 0; i < n; i++) {
    const int64_t op = ops[4 * i], dev = ops[4 * i + 1], arg = ops[4 * i + 2], reusable = ops[4 * i + 3];
    if (op == 0 || op == 3) {
      void *p = nullptr;
      const int e = op == 0 ? pool.alloc((int)dev, &p, (size_t)arg) : pool.malloc((int)dev, &p, (size_t)arg);
      if (e) { results[i] = -e; continue; }
      if (!out.insert(p).second) return -1;
      handles.push_back({p, (size_t)arg, (int)dev});
      results[i] = (int64_t)handles.size() - 1;
    } else if (op == 1) {
      H &h = handles[(size_t)arg];
      out.erase(h.p);
      pool.release(h.device, h.p, h.bytes, reusable != 0);
      results[i] = 0;
    } else {
      results[i] = (int64_t)pool.trim((int)dev);
    }
  }
  counters_out[0] = Fake::in_use; counters_out[1] = Fake::n_malloc; counters_out[2] = Fake::n_free; counters_out[3] = Fake::n_oom;
  counters_out[4] = Fake::live.size();
  counters_out[5] = pool.stats(0).bytes; counters_out[6] = pool.stats(1).bytes;
  counters_out[7] = pool.stats(0).hits; counters_out[8] = pool.stats(0).misses; counters_out[9] = pool.stats(0).trims;
  return 0;
}
}

// ---- air.h: the host half (analyse + compile) and the instruction semantics the device interpreter shares (run_segment),
// executed on one row / point with plain arrays behind the memory interface ----
#include <cstdio>
#include <cstring>
#include "../../city-rollup_amd/csrc/air.h"
#include "../../city-rollup_amd/csrc/ext3.h"
namespace airsim {
struct HostMem {
  static constexpr int K = 1;
  std::vector<uint64_t> slots;
  const uint64_t *u, *loc, *nxt;
  uint64_t *out;
  const uint64_t *prog, *al;
  uint64_t code(uint32_t pc) const { return prog[pc]; }
  uint64_t alpha(int c) const { return al[c]; }
  air::Vec<1> slot_read(uint32_t i) const { return {{slots[i]}}; }
  void slot_write(uint32_t i, const air::Vec<1> &v) { slots[i] = v.v[0]; }
  uint64_t uni(uint32_t i) const { return u[i]; }
  air::Vec<1> local(uint32_t i) const { return {{loc[i]}}; }
  air::Vec<1> next(uint32_t i) const { return {{nxt[i]}}; }
  void store(uint32_t col, const air::Vec<1> &v) { out[col] = v.v[0]; }
};
}  // namespace airsim
extern "C" {
// dims: n_columns, n_public, n_global, n_challenge, n_out_columns. Returns 0, or -1 with the analyser's message in err (256 bytes).
// acc_out[c] = sum over segments of (the segment's Horner sum) * alpha_c^(sinks after it) - what k_run + k_finish add up before the
// division by Z_H; stores_out: the stored columns of a map program; info_out: segments, slots, instructions, live ops, max degree
int hs_air_point(int kind, const uint32_t *ops, size_t n_ops, const uint64_t *consts, size_t n_consts, const uint32_t *dims, uint32_t want_segments, int prefetch,
                 const uint64_t *local, const uint64_t *next, const uint64_t *publics, const uint64_t *globals, const uint64_t *challenges,
                 const uint64_t *alphas, int n_alphas, const uint64_t *sel, uint64_t *acc_out, uint64_t *stores_out, uint32_t *info_out, char *err) {
  air::Program P;
  P.kind = kind;
  P.ops.resize(n_ops);
  if (n_ops) memcpy(P.ops.data(), ops, n_ops * 16);
  P.consts.assign(consts, consts + n_consts);
  P.n_columns = dims[0]; P.n_public = dims[1]; P.n_global = dims[2]; P.n_challenge = dims[3]; P.n_out_columns = dims[4];
  const std::string why = air::analyse(P);
  if (!why.empty()) { snprintf(err, 256, "%s", why.c_str()); return -1; }
  const air::Compiled C = air::compile(P, want_segments, prefetch != 0);
  std::vector<uint64_t> uni(P.consts);
  uni.insert(uni.end(), publics, publics + P.n_public);
  uni.insert(uni.end(), globals, globals + P.n_global);
  uni.insert(uni.end(), challenges, challenges + P.n_challenge);
  uni.push_back(0);
  airsim::HostMem m{std::vector<uint64_t>(C.n_slots ? C.n_slots : 1, 0xDEADBEEFull), uni.data(), local, next, stores_out, C.code.data(), alphas};
  const air::Selectors<1> S{{{sel[0]}}, {{sel[1]}}, {{sel[2]}}};
  for (int c = 0; c < n_alphas; c++) acc_out[c] = 0;
  for (uint32_t s = 0; s < C.n_segments(); s++) {
    air::Vec<1> acc[air::MAX_ALPHAS];
    std::fill(m.slots.begin(), m.slots.end(), 0xDEADBEEFull);  // a segment must not read what another one left
    air::run_segment(C.seg_off[s], C.seg_off[s + 1], m, kind == 0 ? n_alphas : 0, S, acc);
    for (int c = 0; c < n_alphas; c++) acc_out[c] = gl::add(acc_out[c], gl::mul(acc[c].v[0], gl::pow(alphas[c], C.sinks_after[s])));
  }
  info_out[0] = C.n_segments(); info_out[1] = C.n_slots; info_out[2] = (uint32_t)C.code.size(); info_out[3] = (uint32_t)P.n_live; info_out[4] = P.max_degree;
  return 0;
}
// ext3.h on the host: op 0 = mul, 1 = inverse through adjugate_row / norm
void hs_cubic(int op, const uint64_t *m, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  ext3::E r;
  if (op == 0) r = ext3::mul(m[0], m[1], ext3::E{{a[0], a[1], a[2]}}, ext3::E{{b[0], b[1], b[2]}});
  else {
    uint64_t norm;
    r = ext3::adjugate_row(m[0], m[1], ext3::E{{a[0], a[1], a[2]}}, norm);
    const uint64_t ni = norm ? gl::inv(norm) : 0;
    for (int i = 0; i < 3; i++) r.c[i] = gl::mul(r.c[i], ni);
  }
  for (int i = 0; i < 3; i++) out[i] = r.c[i];
}
}

