// Kernels for the tail of CircuitData::prove: openings at zeta / g*zeta, the FRI batch polynomial,
// FRI folding, layer-tree leaves, proof-of-work grinding and query gathering.
// Semantics follow plonky2 0.2.2 `PolynomialBatch::prove_openings` / `fri_proof` (un-vendored
// dependency); the fold / final-polynomial / fri_combine_initial relations are pinned on the reference
// proofs by tests/test_oracle_fri_reference.py and tests/test_oracle_fri_combine_reference.py.
//
// All of this is small at the product shape (n = 2^12): a few hundred KB per kernel, L2-resident and
// latency-bound; the kernels are written for few launches and no host round-trips beyond the ones the
// Fiat-Shamir transcript forces.
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

// out[j] = z^j (and optionally inv_out[j] = z^-j), j < n
__global__ void k_ext_powers(Ext z, Ext zinv, size_t n, Ext *__restrict__ out, Ext *__restrict__ inv_out) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  out[j] = ext_pow(z, j);
  if (inv_out) inv_out[j] = ext_pow(zinv, j);
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

// out[p] = sum_j coeffs[p][j] * zpow[j]   (one workgroup per polynomial)
__global__ __launch_bounds__(256) void k_eval_at_point(const uint64_t *__restrict__ coeffs, size_t stride, size_t n,
                                                       const Ext *__restrict__ zpow, Ext *__restrict__ out) {
  __shared__ Ext sh[256];
  const uint64_t *c = coeffs + (size_t)blockIdx.x * stride;
  Ext acc{0, 0};
  for (size_t j = threadIdx.x; j < n; j += 256) {
    uint64_t v = c[j];
    Ext z = zpow[j];
    acc.a = gl::add(acc.a, gl::mul(v, z.a));
    acc.b = gl::add(acc.b, gl::mul(v, z.b));
  }
  Ext r = block_sum(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

struct BatchRefs {
  const uint64_t *base[4];  // coefficient form, polynomial p of batch b at base[b] + p*stride
  int k[4];
  size_t stride;
};

// comp[c] = sum_{p over the listed polynomials, in order} apow[p] * f_p[c]
__global__ __launch_bounds__(256) void k_combine(BatchRefs refs, const Ext *__restrict__ apow, size_t n,
                                                 Ext *__restrict__ comp) {
  size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  Ext acc{0, 0};
  int idx = 0;
  for (int b = 0; b < 4; b++) {
    const uint64_t *f = refs.base[b] + c;
    for (int p = 0; p < refs.k[b]; p++, idx++) {
      uint64_t v = f[(size_t)p * refs.stride];
      Ext a = apow[idx];
      acc.a = gl::add(acc.a, gl::mul(v, a.a));
      acc.b = gl::add(acc.b, gl::mul(v, a.b));
    }
  }
  comp[c] = acc;
}

// q = (comp - comp(z)) / (X - z):  q[i] = z^-(i+1) * sum_{j>i} comp[j] z^j ;  q[n-1] = 0.
// fin = fin * shift + q  (first == 1: fin = q). One workgroup, n <= 2^16.
__global__ __launch_bounds__(256) void k_divide_linear_accumulate(const Ext *__restrict__ comp, const Ext *__restrict__ zpow,
                                                                  const Ext *__restrict__ zinvpow, Ext zinv_n, size_t n,
                                                                  Ext shift, int first, uint64_t *__restrict__ fin_re,
                                                                  uint64_t *__restrict__ fin_im) {
  __shared__ Ext tot[256];
  const int t = threadIdx.x;
  const size_t per = (n + 255) / 256;
  const size_t lo = (size_t)t * per < n ? (size_t)t * per : n, hi = lo + per < n ? lo + per : n;
  // local suffix sum of G[j] = comp[j] z^j over the thread's chunk
  Ext s{0, 0};
  for (size_t j = hi; j-- > lo;) s = gl::ext_add(s, gl::ext_mul(comp[j], zpow[j]));
  tot[t] = s;
  __syncthreads();
  // exclusive suffix scan over chunk totals (Hillis-Steele on the reversed order)
  for (int d = 1; d < 256; d <<= 1) {
    Ext v = tot[t];
    if (t + d < 256) v = gl::ext_add(v, tot[t + d]);
    __syncthreads();
    tot[t] = v;
    __syncthreads();
  }
  Ext above = (t + 1 < 256) ? tot[t + 1] : Ext{0, 0};  // sum of G[j] for j >= hi
  // walk the chunk downwards: S[i] = sum_{j>i} G[j]
  Ext run = above;
  for (size_t i = hi; i-- > lo;) {
    // z^-(i+1): the table holds z^-j for j < n; index n uses zinv_n
    Ext zi = (i + 1 < n) ? zinvpow[i + 1] : zinv_n;
    Ext q = gl::ext_mul(run, zi);
    Ext f{0, 0};
    if (!first) f = gl::ext_mul(Ext{fin_re[i], fin_im[i]}, shift);
    f = gl::ext_add(f, q);
    fin_re[i] = f.a;
    fin_im[i] = f.b;
    run = gl::ext_add(run, gl::ext_mul(comp[i], zpow[i]));
  }
}

// FRI fold in coefficient space: out[j] = sum_{i<arity} beta^i c[arity*j + i]
__global__ void k_fold(const uint64_t *__restrict__ re, const uint64_t *__restrict__ im, size_t n_out, int arity,
                       Ext beta, uint64_t *__restrict__ out_re, uint64_t *__restrict__ out_im) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_out) return;
  Ext acc{0, 0};
  for (int i = arity - 1; i >= 0; i--) {
    size_t idx = (size_t)arity * j + i;
    acc = gl::ext_add(gl::ext_mul(acc, beta), Ext{re[idx], im[idx]});
  }
  out_re[j] = acc.a;
  out_im[j] = acc.b;
}

// FRI layer leaves: leaf j = the `arity` extension values at bit-reversed positions [arity*j, arity*(j+1)),
// flattened (a0,b0,a1,b1,...). Writes the leaf digests and (for the query phase) nothing else: the values
// themselves stay in re/im.
__global__ __launch_bounds__(256) void k_leaf_hash_fri(const uint64_t *__restrict__ re, const uint64_t *__restrict__ im,
                                                       size_t n_leaves, int arity, uint64_t *__restrict__ digests) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_leaves) return;
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
  uint64_t *d = digests + 4 * j;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}

// Proof of work: candidates start .. start+count-1; records the smallest one whose response has
// `pow_bits` leading zero bits. state = sponge state with the pending inputs already written.
struct PowState { uint64_t s[12]; };
__global__ __launch_bounds__(256) void k_pow_grind(PowState st, int pos, uint64_t start, uint64_t count, int pow_bits,
                                                   unsigned long long *__restrict__ best) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  uint64_t cand = start + i;
  uint64_t s[poseidon::W];
#pragma unroll
  for (int k = 0; k < poseidon::W; k++) s[k] = st.s[k];
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (k == pos) s[k] = cand;
  poseidon::permute(s);
  if ((s[7] >> (64 - pow_bits)) == 0) atomicMin(best, (unsigned long long)cand);
}

// Query gathering. One workgroup per query round; writes the bincode words of one FriQueryRound.
struct QueryRefs {
  const uint64_t *lde[4];      // bit-reversed LDE, column-major, stride N
  const uint64_t *digests[4];  // levels below the cap
  int k[4];
  size_t N;
  int depth0;                  // log2(N) - cap_height
  int n_layers;
  const uint64_t *fre[8], *fim[8];  // FRI layer values (bit-reversed order)
  const uint64_t *fdig[8];
  size_t f_leaves[8];
  int f_arity_bits[8];
  int f_depth[8];
  size_t words_per_query;
};

__device__ __forceinline__ void copy_path(const uint64_t *dig, size_t n_leaves, int depth, size_t idx,
                                          uint64_t *out) {
  // level l sibling = digests[off_l + ((idx >> l) ^ 1)]
  for (int w = threadIdx.x; w < depth * 4; w += blockDim.x) {
    int l = w >> 2;
    size_t off = 0, n = n_leaves;
    for (int i = 0; i < l; i++) { off += n; n >>= 1; }
    out[w] = dig[4 * (off + ((idx >> l) ^ 1)) + (w & 3)];
  }
}

__global__ __launch_bounds__(256) void k_gather_queries(QueryRefs r, const uint64_t *__restrict__ indices,
                                                        uint64_t *__restrict__ out) {
  uint64_t *o = out + (size_t)blockIdx.x * r.words_per_query;
  const size_t x = (size_t)indices[blockIdx.x];
  size_t w = 0;
  if (threadIdx.x == 0) o[w] = 4;
  w += 1;
  for (int b = 0; b < 4; b++) {
    if (threadIdx.x == 0) o[w] = (uint64_t)r.k[b];
    w += 1;
    for (int p = threadIdx.x; p < r.k[b]; p += blockDim.x) o[w + p] = r.lde[b][(size_t)p * r.N + x];
    w += r.k[b];
    if (threadIdx.x == 0) o[w] = (uint64_t)r.depth0;
    w += 1;
    copy_path(r.digests[b], r.N, r.depth0, x, o + w);
    w += (size_t)r.depth0 * 4;
  }
  if (threadIdx.x == 0) o[w] = (uint64_t)r.n_layers;
  w += 1;
  size_t xi = x;
  for (int l = 0; l < r.n_layers; l++) {
    const int ab = r.f_arity_bits[l], arity = 1 << ab;
    xi >>= ab;
    if (threadIdx.x == 0) o[w] = (uint64_t)arity;
    w += 1;
    for (int e = threadIdx.x; e < 2 * arity; e += blockDim.x)
      o[w + e] = (e & 1) ? r.fim[l][(size_t)arity * xi + (e >> 1)] : r.fre[l][(size_t)arity * xi + (e >> 1)];
    w += 2 * arity;
    if (threadIdx.x == 0) o[w] = (uint64_t)r.f_depth[l];
    w += 1;
    copy_path(r.fdig[l], r.f_leaves[l], r.f_depth[l], xi, o + w);
    w += (size_t)r.f_depth[l] * 4;
  }
}

}  // namespace fri
