// Kernels for the tail of CircuitData::prove: openings at zeta / g*zeta, the FRI batch polynomial,
// FRI folding, layer-tree leaves, proof-of-work grinding and query gathering.
// Semantics follow plonky2 0.2.2 `PolynomialBatch::prove_openings` / `fri_proof` (un-vendored
// dependency); the fold / final-polynomial / fri_combine_initial relations are pinned on the reference
// proofs by tests/test_oracle_fri_reference.py and tests/test_oracle_fri_combine_reference.py.
//
// Every kernel carries a PROOF dimension (blockIdx.y, or .z): a call proves B independent proofs of the
// same shape at once, so that at the product shape (n = 2^12, a few hundred KB per proof and step)
// launches are amortised over the batch and the grids are large enough to be throughput- rather than
// latency-bound. Per-proof buffers are laid out [proof][...] with a fixed stride; only the circuit's
// constants/sigmas oracle is reached through a per-proof pointer table (proofs of different circuits
// can share a batch).
#pragma once
#include "gl.h"
#include "poseidon.h"

namespace fri {

using gl::Ext;

__device__ __forceinline__ Ext ext_pow(Ext b, uint64_t e) {
  Ext r{1, 0};
  while (e) {
    if (e & 1) r = gl::ext_mul(r, b);
    b = gl::ext_mul(b, b);
    e >>= 1;
  }
  return r;
}

// pts: P points; out[p][j] = pts[p]^j, j < n
__global__ void k_ext_powers(const Ext *__restrict__ pts, size_t n, Ext *__restrict__ out) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  out[(size_t)blockIdx.y * n + j] = ext_pow(pts[blockIdx.y], j);
}

// block reduce of an Ext over 256 threads (result valid in thread 0)
__device__ __forceinline__ Ext block_sum(Ext v, Ext *sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = gl::ext_add(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  return sh[0];
}

// out[proof][out_off + p] = sum_j coeffs_proof[p][j] * zpow[proof*zstride + j]
// coeffs_proof = table ? table[proof] : base + proof*proof_stride.   grid = (k, B)
__global__ __launch_bounds__(256) void k_eval_at_point(const uint64_t *__restrict__ base, size_t proof_stride,
                                                       const uint64_t *const *__restrict__ table, size_t n,
                                                       const Ext *__restrict__ zpow, size_t zstride,
                                                       Ext *__restrict__ out, size_t out_stride, size_t out_off) {
  __shared__ Ext sh[256];
  const size_t proof = blockIdx.y;
  const uint64_t *c = (table ? table[proof] : base + proof * proof_stride) + (size_t)blockIdx.x * n;
  const Ext *zp = zpow + proof * zstride;
  Ext acc{0, 0};
  for (size_t j = threadIdx.x; j < n; j += 256) {
    uint64_t v = c[j];
    Ext z = zp[j];
    acc.a = gl::add(acc.a, gl::mul(v, z.a));
    acc.b = gl::add(acc.b, gl::mul(v, z.b));
  }
  Ext r = block_sum(acc, sh);
  if (threadIdx.x == 0) out[proof * out_stride + out_off + blockIdx.x] = r;
}

// Oracles of a FRI instance as the kernels see them: oracle o of proof p is at base[o] + p*stride[o], or, when
// table[o] is set, at table[o][p] (the constants/sigmas oracle of a plonky2 proof belongs to the circuit, so proofs of
// different circuits reach it through a per-proof pointer table).
constexpr int MAX_ORACLES = 8;   // plonky2: 4 (constants+sigmas, wires, Z/partial products, quotient); a STARK: trace rounds + quotient
constexpr int MAX_RANGES = 16;   // polynomial ranges of one opening batch per k_combine launch (longer lists: several launches)

struct BatchRefs {
  const uint64_t *base[MAX_ORACLES];           // coefficient arrays, polynomial j of oracle o at + j*n
  size_t proof_stride[MAX_ORACLES];
  const uint64_t *const *table[MAX_ORACLES];   // per-proof pointer table (device) or null
  int n_ranges;
  int r_oracle[MAX_RANGES], r_first[MAX_RANGES], r_count[MAX_RANGES];  // FriBatchInfo::polynomials as (oracle, first, count) runs
  size_t n;
};

// comp[proof][c] (+)= sum_{listed polynomials p, in order} apow[proof][apow_off + idx(p)] * f_p[c].   grid = (n/256, B)
// (plonky2 `ReducingFactor::reduce_polys_base` over one FriBatchInfo)
__global__ __launch_bounds__(256) void k_combine(BatchRefs refs, const Ext *__restrict__ apow, size_t apow_stride, size_t apow_off,
                                                 int accumulate, Ext *__restrict__ comp) {
  size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= refs.n) return;
  const size_t proof = blockIdx.y;
  const Ext *ap = apow + proof * apow_stride + apow_off;
  Ext acc{0, 0};
  if (accumulate) acc = comp[proof * refs.n + c];
  int idx = 0;
  for (int r = 0; r < refs.n_ranges; r++) {
    const int o = refs.r_oracle[r];
    const uint64_t *f = (refs.table[o] ? refs.table[o][proof] : refs.base[o] + proof * refs.proof_stride[o]) + (size_t)refs.r_first[r] * refs.n + c;
    // eight polynomials per trip: their loads are issued together (a lone proof's launch is 64 waves, each walking ~130
    // polynomials: one load per step of the running sum was one trip to memory per step)
    const int cnt = refs.r_count[r];
    int p = 0;
    for (; p + 8 <= cnt; p += 8, idx += 8) {
      uint64_t v[8];
      Ext a[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        v[u] = f[(size_t)(p + u) * refs.n];
        a[u] = ap[idx + u];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        acc.a = gl::add(acc.a, gl::mul(v[u], a[u].a));
        acc.b = gl::add(acc.b, gl::mul(v[u], a[u].b));
      }
    }
    for (; p < cnt; p++, idx++) {
      uint64_t v = f[(size_t)p * refs.n];
      Ext a = ap[idx];
      acc.a = gl::add(acc.a, gl::mul(v, a.a));
      acc.b = gl::add(acc.b, gl::mul(v, a.b));
    }
  }
  comp[proof * refs.n + c] = acc;
}

// The same sum cut into `gridDim.z` parts of the polynomial LIST (a few short polynomials of a wide oracle - the 1 338 columns of a
// STARK at 2^10 rows - are sixteen waves each walking the whole list: 0.49 ms of latency for 11 MB): part z sums the listed
// polynomials number [z * per, (z + 1) * per) into parts[(z * B + proof) * n + c]; k_combine_reduce adds the parts. Exact additions:
// the same bits. grid = (n/256, B, parts)
__global__ __launch_bounds__(256) void k_combine_split(BatchRefs refs, const Ext *__restrict__ apow, size_t apow_stride, size_t apow_off, int per,
                                                       Ext *__restrict__ parts) {
  size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= refs.n) return;
  const size_t proof = blockIdx.y;
  const Ext *ap = apow + proof * apow_stride + apow_off;
  const int lo = (int)blockIdx.z * per, hi = lo + per;
  Ext acc{0, 0};
  int idx = 0;
  for (int r = 0; r < refs.n_ranges; r++) {
    const int cnt = refs.r_count[r];
    const int p0 = lo > idx ? lo - idx : 0, p1 = hi - idx < cnt ? hi - idx : cnt;  // this part's share of the run
    if (p0 < p1) {
      const int o = refs.r_oracle[r];
      const uint64_t *f = (refs.table[o] ? refs.table[o][proof] : refs.base[o] + proof * refs.proof_stride[o]) + (size_t)refs.r_first[r] * refs.n + c;
      int p = p0;
      for (; p + 8 <= p1; p += 8) {
        uint64_t v[8];
        Ext a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          v[u] = f[(size_t)(p + u) * refs.n];
          a[u] = ap[idx + p + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          acc.a = gl::add(acc.a, gl::mul(v[u], a[u].a));
          acc.b = gl::add(acc.b, gl::mul(v[u], a[u].b));
        }
      }
      for (; p < p1; p++) {
        const uint64_t v = f[(size_t)p * refs.n];
        const Ext a = ap[idx + p];
        acc.a = gl::add(acc.a, gl::mul(v, a.a));
        acc.b = gl::add(acc.b, gl::mul(v, a.b));
      }
    }
    idx += cnt;
  }
  parts[((size_t)blockIdx.z * gridDim.y + proof) * refs.n + c] = acc;
}
// comp[proof][c] (+)= sum_z parts[z][proof][c].   grid = (n/256, B)
__global__ __launch_bounds__(256) void k_combine_reduce(const Ext *__restrict__ parts, int n_parts, size_t n, int accumulate, Ext *__restrict__ comp) {
  size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const size_t proof = blockIdx.y;
  Ext acc{0, 0};
  if (accumulate) acc = comp[proof * n + c];
  for (int z = 0; z < n_parts; z++) acc = gl::ext_add(acc, parts[((size_t)z * gridDim.y + proof) * n + c]);
  comp[proof * n + c] = acc;
}

// q = (comp - comp(z)) / (X - z):  q[i] = z^-(i+1) * sum_{j>i} comp[j] z^j ;  q[n-1] = 0.
// fin = fin * shift + q  (first == 1: fin = q). One workgroup per proof (grid = (1, B)).
// zpow / zinvpow: tables of the proof's point (stride zstride per proof); shifts[proof*shift_stride]; fin: [proof][re | im][n].
__global__ __launch_bounds__(256) void k_divide_linear_accumulate(const Ext *__restrict__ comp, const Ext *__restrict__ zpow,
                                                                  const Ext *__restrict__ zinvpow, size_t zstride, size_t n,
                                                                  const Ext *__restrict__ shifts, size_t shift_stride, int first,
                                                                  uint64_t *__restrict__ fin) {
  __shared__ Ext tot[256];
  const int t = threadIdx.x;
  const size_t proof = blockIdx.y;
  comp += proof * n;
  zpow += proof * zstride;
  zinvpow += proof * zstride;
  uint64_t *fin_re = fin + proof * 2 * n, *fin_im = fin_re + n;
  const Ext shift = shifts[proof * shift_stride];
  const size_t per = (n + 255) / 256;
  const size_t lo = (size_t)t * per < n ? (size_t)t * per : n, hi = lo + per < n ? lo + per : n;
  Ext s{0, 0};
  for (size_t j = hi; j-- > lo;) s = gl::ext_add(s, gl::ext_mul(comp[j], zpow[j]));
  tot[t] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {  // inclusive suffix scan over chunk totals
    Ext v = tot[t];
    if (t + d < 256) v = gl::ext_add(v, tot[t + d]);
    __syncthreads();
    tot[t] = v;
    __syncthreads();
  }
  Ext run = (t + 1 < 256) ? tot[t + 1] : Ext{0, 0};  // sum of comp[j] z^j for j >= hi
  for (size_t i = hi; i-- > lo;) {
    // q[n-1] = 0 (run == 0 there), so the missing table entry z^-n is never needed
    Ext zi = (i + 1 < n) ? zinvpow[i + 1] : Ext{0, 0};
    Ext q = gl::ext_mul(run, zi);
    Ext f{0, 0};
    if (!first) f = gl::ext_mul(Ext{fin_re[i], fin_im[i]}, shift);
    f = gl::ext_add(f, q);
    fin_re[i] = f.a;
    fin_im[i] = f.b;
    run = gl::ext_add(run, gl::ext_mul(comp[i], zpow[i]));
  }
}

// FRI fold in coefficient space: out[j] = sum_{i<arity} beta^i c[arity*j + i].
// in: [proof][re | im][n_in], out: [proof][re | im][n_out] at out + proof*out_stride.   grid = (blocks, B)
// (n_in = physical length of the re / im arrays; coefficients at index >= n_in are zero: the first
// layer folds the degree-<n batch polynomial as a length-N vector without materialising the padding.)
__global__ void k_fold(const uint64_t *__restrict__ in, size_t in_stride, size_t n_in, size_t n_out, int arity,
                       const Ext *__restrict__ betas, uint64_t *__restrict__ out, size_t out_stride) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_out) return;
  const size_t proof = blockIdx.y;
  const uint64_t *re = in + proof * in_stride, *im = re + n_in;
  const Ext beta = betas[proof];
  Ext acc{0, 0};
  for (int i = arity - 1; i >= 0; i--) {
    size_t idx = (size_t)arity * j + i;
    Ext c = idx < n_in ? Ext{re[idx], im[idx]} : Ext{0, 0};
    acc = gl::ext_add(gl::ext_mul(acc, beta), c);
  }
  uint64_t *o = out + proof * out_stride;
  o[j] = acc.a;
  o[n_out + j] = acc.b;
}

// FRI layer leaves: leaf j = the `arity` extension values at bit-reversed positions [arity*j, arity*(j+1)),
// flattened (a0,b0,a1,b1,...). vals: [proof][re | im][n_vals].   grid = (blocks, B)
__global__ __launch_bounds__(256) void k_leaf_hash_fri(const uint64_t *__restrict__ vals, size_t vals_stride, size_t n_vals,
                                                       size_t n_leaves, int arity, uint64_t *__restrict__ digests,
                                                       size_t dig_stride) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_leaves) return;
  const uint64_t *re = vals + (size_t)blockIdx.y * vals_stride, *im = re + n_vals;
  uint64_t s[poseidon::W];
#pragma unroll
  for (int k = 0; k < poseidon::W; k++) s[k] = 0;
  const int len = 2 * arity;
  if (len <= 4) {
    for (int e = 0; e < len; e++) s[e] = (e & 1) ? im[arity * j + (e >> 1)] : re[arity * j + (e >> 1)];
  } else {
    for (int e0 = 0; e0 < len; e0 += poseidon::RATE) {
#pragma unroll
      for (int k = 0; k < poseidon::RATE; k++) {
        int e = e0 + k;
        if (e < len) s[k] = (e & 1) ? im[arity * j + (e >> 1)] : re[arity * j + (e >> 1)];
      }
      poseidon::permute(s);
    }
  }
  uint64_t *d = digests + (size_t)blockIdx.y * dig_stride + 4 * j;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}

// Proof of work (plonky2 `fri_proof_of_work`: the witness is hashed after the transcript's pending inputs; this
// build, like the oracle, returns the SMALLEST witness whose response has `pow_bits` leading zero bits, which makes the
// proof bytes deterministic). st[proof] = sponge state with the pending inputs already written, pos[proof] = slot the
// witness goes to.
// Persistent workgroups over a work queue: the candidates of a proof are handed out in ranges of 256, in increasing
// order (next_range[proof], atomicAdd); a workgroup keeps taking ranges of "its" proof until a witness below the next
// range is known (best[proof], atomicMin), then moves on to a proof that is still searching, and exits when none is.
// So the chip stays full until the last proof is done, and the work beyond each proof's first witness is only what
// was in flight when it was found — and that is dropped at the next round boundary of its permutation (`permute_until`
// polls best[proof]): with ~400 K candidates resident, finishing them all cost as much again as the search itself.
// Ranges complete out of order, but a range is handed out only after all smaller ones, and a range is abandoned only when a
// witness BELOW it is known: the minimum over all found witnesses is the smallest one.
// Exit: next_range only grows and is bounded by count / 256, so every workgroup runs out of work.
struct PowState { uint64_t s[12]; int pos; int pad; };
__global__ __launch_bounds__(256) void k_pow_grind(const PowState *__restrict__ st, uint64_t start, uint64_t count,
                                                   int pow_bits, unsigned n_proofs, const unsigned long long *__restrict__ done,
                                                   unsigned long long *__restrict__ best, unsigned *__restrict__ next_range) {
  __shared__ int s_proof;
  __shared__ unsigned s_range;
  const unsigned ranges = (unsigned)(count >> 8);
  unsigned home = blockIdx.x % n_proofs;
  for (;;) {
    if (threadIdx.x == 0) {
      int found = -1;
      unsigned r = 0;
      for (unsigned k = 0; k < n_proofs && found < 0; k++) {
        const unsigned q = home + k < n_proofs ? home + k : home + k - n_proofs;
        if (done[q] != ~0ull) continue;  // found by an earlier launch, or supplied by the caller
        const unsigned peek = __hip_atomic_load(next_range + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (peek >= ranges) continue;
        if (__hip_atomic_load(best + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < start + ((uint64_t)peek << 8)) continue;
        r = atomicAdd(next_range + q, 1u);
        if (r >= ranges) continue;
        found = (int)q;  // taken: it is computed even if a smaller witness has appeared meanwhile (harmless)
      }
      s_proof = found;
      s_range = r;
    }
    __syncthreads();
    const int q = s_proof;
    const unsigned r = s_range;
    __syncthreads();
    if (q < 0) return;
    home = (unsigned)q;
    const uint64_t cand = start + ((uint64_t)r << 8) + threadIdx.x;
    uint64_t s[poseidon::W];
    const int pos = st[q].pos;
#pragma unroll
    for (int k = 0; k < poseidon::W; k++) s[k] = st[q].s[k];
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (k == pos) s[k] = cand;
    // a witness below this range has appeared: nothing in the range can be the smallest any more
    const uint64_t range_start = start + ((uint64_t)r << 8);
    const unsigned long long *bq = best + q;
    if (!poseidon::permute_until(s, [=]() { return __hip_atomic_load(bq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < range_start; })) continue;
    if (pow_bits <= 0 || (s[7] >> (64 - pow_bits)) == 0) atomicMin(best + q, (unsigned long long)cand);
  }
}

// Query gathering: one workgroup per (query round, proof); writes the bincode words of a FriQueryRound.
struct QueryRefs {
  int n_oracles;
  const uint64_t *lde[MAX_ORACLES];           // oracle o of proof p: lde[o] + p*lde_stride[o], or lde_table[o][p]
  const uint64_t *digests[MAX_ORACLES];       //                      digests[o] + p*dig_stride[o], or dig_table[o][p]
  const uint64_t *const *lde_table[MAX_ORACLES];
  const uint64_t *const *dig_table[MAX_ORACLES];
  size_t lde_stride[MAX_ORACLES], dig_stride[MAX_ORACLES];
  int k[MAX_ORACLES];
  const uint64_t *salt[MAX_ORACLES];          // blinded oracle: n_salt[o] extra leaf elements ([proof][n_salt][N]), else null
  size_t salt_stride[MAX_ORACLES];
  int n_salt[MAX_ORACLES];
  size_t N;
  int depth0;                       // log2(N) - cap_height
  int n_layers;
  const uint64_t *fvals[8];         // FRI layer values [proof][re | im][f_nvals]
  size_t fvals_stride[8], f_nvals[8];
  const uint64_t *fdig[8];
  size_t fdig_stride[8];
  size_t f_leaves[8];
  int f_arity_bits[8];
  int f_depth[8];
  size_t words_per_query;
  int n_queries;
};

__device__ __forceinline__ void copy_path(const uint64_t *dig, size_t n_leaves, int depth, size_t idx,
                                          uint64_t *out) {
  for (int w = threadIdx.x; w < depth * 4; w += blockDim.x) {
    int l = w >> 2;
    size_t off = 0, n = n_leaves;
    for (int i = 0; i < l; i++) { off += n; n >>= 1; }
    out[w] = dig[4 * (off + ((idx >> l) ^ 1)) + (w & 3)];
  }
}

__global__ __launch_bounds__(256) void k_gather_queries(QueryRefs r, const uint64_t *__restrict__ indices,
                                                        uint64_t *__restrict__ out) {
  const size_t proof = blockIdx.y;
  uint64_t *o = out + (proof * r.n_queries + blockIdx.x) * r.words_per_query;
  const size_t x = (size_t)indices[proof * r.n_queries + blockIdx.x];
  size_t w = 0;
  if (threadIdx.x == 0) o[w] = (uint64_t)r.n_oracles;
  w += 1;
  for (int b = 0; b < r.n_oracles; b++) {
    const uint64_t *lde = r.lde_table[b] ? r.lde_table[b][proof] : r.lde[b] + proof * r.lde_stride[b];
    const uint64_t *dig = r.dig_table[b] ? r.dig_table[b][proof] : r.digests[b] + proof * r.dig_stride[b];
    const int ns = r.salt[b] ? r.n_salt[b] : 0;
    if (threadIdx.x == 0) o[w] = (uint64_t)(r.k[b] + ns);   // the whole leaf: values, then the salt
    w += 1;
    for (int p = threadIdx.x; p < r.k[b]; p += blockDim.x) o[w + p] = lde[(size_t)p * r.N + x];
    w += r.k[b];
    for (int p = threadIdx.x; p < ns; p += blockDim.x) o[w + p] = r.salt[b][proof * r.salt_stride[b] + (size_t)p * r.N + x];
    w += ns;
    if (threadIdx.x == 0) o[w] = (uint64_t)r.depth0;
    w += 1;
    copy_path(dig, r.N, r.depth0, x, o + w);
    w += (size_t)r.depth0 * 4;
  }
  if (threadIdx.x == 0) o[w] = (uint64_t)r.n_layers;
  w += 1;
  size_t xi = x;
  for (int l = 0; l < r.n_layers; l++) {
    const int ab = r.f_arity_bits[l], arity = 1 << ab;
    const uint64_t *re = r.fvals[l] + proof * r.fvals_stride[l], *im = re + r.f_nvals[l];
    xi >>= ab;
    if (threadIdx.x == 0) o[w] = (uint64_t)arity;
    w += 1;
    for (int e = threadIdx.x; e < 2 * arity; e += blockDim.x)
      o[w + e] = (e & 1) ? im[(size_t)arity * xi + (e >> 1)] : re[(size_t)arity * xi + (e >> 1)];
    w += 2 * arity;
    if (threadIdx.x == 0) o[w] = (uint64_t)r.f_depth[l];
    w += 1;
    copy_path(r.fdig[l] + proof * r.fdig_stride[l], r.f_leaves[l], r.f_depth[l], xi, o + w);
    w += (size_t)r.f_depth[l] * 4;
  }
}

}  // namespace fri
