// Multi-scalar multiplication over BLS12-381 G1 and G2 (the MSMs of the Groth16 wrap proof; the
// kernels are templates over the coordinate field F = Fp (G1) or Fp2 (G2)),
// bucket method with SIGNED digits: k = sum_w 2^(c w) d_w, |d_w| <= 2^(c-1) (a digit above 2^(c-1) becomes d - 2^c and
// carries into the next window), so sum_i k_i P_i = sum_w 2^(c w) sum_{d=1..2^(c-1)} d * B[w][d] with
// B[w][d] = sum of +-P_i over the scalars whose w-th digit is +-d: half the buckets of the unsigned method.
//
//   k_points_to_mont   canonical affine points -> Montgomery form (once per point set: a proving key is fixed)
//   k_hist / k_scan / k_scatter   signed-digit recoding + counting sort of the point indices by |digit|, per window
//                      (atomics; the order inside a bucket is arbitrary, the group law does not care)
//   k_bucket_sum       one lane per (window, digit): mixed additions of its points into an XYZZ accumulator (10 field
//                      products each instead of 11), stored as a Jacobian point
//   k_segment_reduce   one lane per (window, run of seg_len buckets): sum_d d*B_d over the run by the running-sum trick plus
//                      a small scalar multiple for the run's offset
//   k_pair_reduce      tree reduction of the runs of a window
// The 2^(c w) combination of the window sums and the final inversion run on the host (a few hundred point operations).
// Exact integer arithmetic: the result is the same group element whatever the summation order.
#pragma once
#include "bls12_381.h"

namespace msm {

// the bucket accumulation runs on bounded, unreduced field values (bls12_381.h "loose arithmetic");
// -DBLS_CANONICAL_ACCUMULATION keeps every value reduced, for A/B measurements
#if defined(BLS_CANONICAL_ACCUMULATION)
template <class F> GL_HD bls::XyzzT<F> bucket_add(const bls::XyzzT<F> &p, const bls::AffineT<F> &q) { return bls::xyzz_add_mixed(p, q); }
template <class F> GL_HD bls::JacT<F> bucket_out(const bls::XyzzT<F> &p) { return bls::xyzz_to_jac(p); }
#else
template <class F> GL_HD bls::XyzzT<F> bucket_add(const bls::XyzzT<F> &p, const bls::AffineT<F> &q) { return bls::xyzz_add_mixed_loose(p, q); }
template <class F> GL_HD bls::JacT<F> bucket_out(const bls::XyzzT<F> &p) { return bls::xyzz_to_jac_loose(p); }
#endif

using bls::AffineT;
using bls::Field;
using bls::JacT;

constexpr int SCALAR_WORDS = 8;  // 256-bit scalars, little-endian 32-bit words
constexpr uint32_t HEAVY = 128;  // buckets with more than HEAVY << hs points are summed by a whole workgroup (k_heavy_sum);
                                 // hs ("heavy shift") is chosen by the host so that this is well above the mean bucket size
// lanes of that workgroup: the LDS tree holds one Jacobian point per lane (168 B for G1, 336 B for G2; 64 KB limit)
template <class F> constexpr int HEAVY_LANES = sizeof(JacT<F>) <= 168 ? 256 : 128;

__device__ __forceinline__ uint32_t digit(const uint32_t *k, int w, int c) {
  const int bit = w * c;
  if (bit >= 32 * SCALAR_WORDS) return 0;
  const int word = bit >> 5, sh = bit & 31;
  uint64_t v = k[word];
  if (word + 1 < SCALAR_WORDS) v |= (uint64_t)k[word + 1] << 32;
  return (uint32_t)(v >> sh) & ((1u << c) - 1);
}

template <class F>
__global__ __launch_bounds__(128) void k_points_to_mont(const uint32_t *__restrict__ xy, size_t n, AffineT<F> *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = bls::affine_from_canonical<F>(xy + 2 * Field<F>::WORDS * i);
}

// synthetic point set for benches and large-size tests: P_i = (a*i + b) * G, a and b < 2^16, written in the internal
// (Montgomery affine) form. Independent lanes: two small scalar multiples, one addition, one inversion each.
template <class F>
__global__ __launch_bounds__(64) void k_synthetic_points(AffineT<F> g, uint32_t a, uint32_t b, size_t n, AffineT<F> *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const JacT<F> G{g.x, g.y, Field<F>::one()};
  // (a*i + b) G = i * (a G) + b G, with i < 2^32
  const JacT<F> p = bls::jac_add(bls::jac_mul_small(bls::jac_mul_small(G, a), (uint32_t)i), bls::jac_mul_small(G, b));
  out[i] = bls::jac_to_affine(p);
}

// signed digit of window w, given the carry from the window below; updates the carry
__device__ __forceinline__ int signed_digit(const uint32_t *k, int w, int c, uint32_t &carry) {
  const uint32_t raw = digit(k, w, c) + carry;
  if (raw > (1u << (c - 1))) { carry = 1; return (int)raw - (1 << c); }
  carry = 0;
  return (int)raw;
}
// one lane per scalar, all its windows (the recoding carries from window to window)
__global__ void k_hist(const uint32_t *__restrict__ scalars, const uint8_t *__restrict__ inf, size_t n, int c, int windows,
                       size_t nbs, uint32_t *__restrict__ counts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (inf && inf[i])) return;
  uint32_t carry = 0;
  for (int w = 0; w < windows; w++) {
    const int d = signed_digit(scalars + SCALAR_WORDS * i, w, c, carry);
    if (d) atomicAdd(&counts[(size_t)w * nbs + (d < 0 ? -d : d)], 1u);
  }
}
// ---- the same counting sort for n >= 2^18 points, with the counters privatised in LDS ------------------------------
// k_hist / k_scatter issue one global atomic per (scalar, window): 0.57 G of them at 2^25 points, a quarter of the MSM
// (and on witness-like scalars they pile onto a few counters). Here a workgroup owns a tile of scalars of ONE
// window and keeps that window's <= 32784 counters in LDS (128 KB of the 160 KB): LDS atomics per scalar, one global
// atomic per non-empty counter and tile. The signed digits are computed once (the recoding carries through the
// windows) into digits[w][i]; 0 = no contribution (zero digit, or a point at infinity).
constexpr int TILE_NBS = 32784;  // 2^15 + 1 bucket indices padded to a multiple of 16 (c = 16, the widest window used)
__global__ void k_digits(const uint32_t *__restrict__ scalars, const uint8_t *__restrict__ inf, size_t n, int c, int windows,
                         int32_t *__restrict__ digits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool skip = inf && inf[i];
  uint32_t carry = 0;
  for (int w = 0; w < windows; w++) {
    const int d = signed_digit(scalars + SCALAR_WORDS * i, w, c, carry);
    digits[(size_t)w * n + i] = skip ? 0 : d;
  }
}
// grid = (tiles, windows)
__global__ __launch_bounds__(1024) void k_hist_tiled(const int32_t *__restrict__ digits, size_t n, size_t tile, size_t nbs,
                                                     uint32_t *__restrict__ counts) {
  __shared__ uint32_t h[TILE_NBS];
  const size_t w = blockIdx.y, lo = (size_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
  for (size_t b = threadIdx.x; b < nbs; b += 1024) h[b] = 0;
  __syncthreads();
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
    const int d = digits[w * n + i];
    if (d) atomicAdd(&h[d < 0 ? -d : d], 1u);
  }
  __syncthreads();
  for (size_t b = threadIdx.x; b < nbs; b += 1024)
    if (h[b]) atomicAdd(&counts[w * nbs + b], h[b]);
}
__global__ __launch_bounds__(1024) void k_scatter_tiled(const int32_t *__restrict__ digits, size_t n, size_t tile, size_t nbs,
                                                        uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted) {
  __shared__ uint32_t h[TILE_NBS];
  const size_t w = blockIdx.y, lo = (size_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
  for (size_t b = threadIdx.x; b < nbs; b += 1024) h[b] = 0;
  __syncthreads();
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
    const int d = digits[w * n + i];
    if (d) atomicAdd(&h[d < 0 ? -d : d], 1u);
  }
  __syncthreads();
  for (size_t b = threadIdx.x; b < nbs; b += 1024) {  // reserve this tile's slots of every bucket: h[b] <- first slot
    const uint32_t cnt = h[b];
    if (cnt) h[b] = atomicAdd(&cursor[w * nbs + b], cnt);
  }
  __syncthreads();
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
    const int d = digits[w * n + i];
    if (!d) continue;
    const uint32_t pos = atomicAdd(&h[d < 0 ? -d : d], 1u);
    sorted[w * n + pos] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
  }
}

// one workgroup of 1024 lanes per window: exclusive scan of its nbs counts -> offsets, cursor = offsets
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *__restrict__ counts, size_t nbs, uint32_t *__restrict__ offsets,
                                               uint32_t *__restrict__ cursor) {
  __shared__ uint32_t part[1024];
  const size_t base = (size_t)blockIdx.x * nbs;
  const uint32_t nb = (uint32_t)nbs, per = (nb + 1023) / 1024;
  const uint32_t lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
  uint32_t s = 0;
  for (uint32_t d = lo; d < hi; d++) s += counts[base + d];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (uint32_t d = lo; d < hi; d++) {
    offsets[base + d] = run;
    cursor[base + d] = run;
    run += counts[base + d];
  }
}
// sorted[w][pos] = point index, bit 31 set when the point enters its bucket negated
__global__ void k_scatter(const uint32_t *__restrict__ scalars, const uint8_t *__restrict__ inf, size_t n, int c, int windows,
                          size_t nbs, uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (inf && inf[i])) return;
  uint32_t carry = 0;
  for (int w = 0; w < windows; w++) {
    const int d = signed_digit(scalars + SCALAR_WORDS * i, w, c, carry);
    if (!d) continue;
    const uint32_t pos = atomicAdd(&cursor[(size_t)w * nbs + (d < 0 ? -d : d)], 1u);
    sorted[(size_t)w * n + pos] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
  }
}
// the point a sorted entry stands for (negated when bit 31 is set)
template <class F>
__device__ __forceinline__ AffineT<F> entry_point(const AffineT<F> *__restrict__ pts, uint32_t e) {
  AffineT<F> q = pts[e & 0x7fffffffu];
  if (e >> 31) q.y = bls::f_sub(Field<F>::zero(), q.y);
  return q;
}

// Buckets in order of decreasing size, so that the 64 lanes of a wave run (nearly) the same number of additions:
// with Poisson-distributed sizes a wave otherwise waits for its fullest bucket (~1.7x the mean at 16 points per bucket).
// Counting sort on min(count >> hs, HEAVY + 1): k_size_hist (LDS histogram per workgroup) -> k_size_scan -> k_size_scatter.
constexpr int SIZE_BINS = (int)HEAVY + 2;
__device__ __forceinline__ uint32_t size_bin(uint32_t cnt, int hs) { return cnt > (HEAVY << hs) ? HEAVY + 1 : cnt >> hs; }
__global__ __launch_bounds__(256) void k_size_hist(const uint32_t *__restrict__ counts, size_t total, int hs,
                                                   uint32_t *__restrict__ bins) {
  __shared__ uint32_t h[SIZE_BINS];
  for (int k = threadIdx.x; k < SIZE_BINS; k += 256) h[k] = 0;
  __syncthreads();
  const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (b < total) {
    const uint32_t cnt = counts[b];
    atomicAdd(&h[size_bin(cnt, hs)], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < SIZE_BINS; k += 256)
    if (h[k]) atomicAdd(&bins[k], h[k]);
}
// cursor[k] = number of buckets in bins larger than k (descending order: the largest sizes come first)
__global__ void k_size_scan(const uint32_t *__restrict__ bins, uint32_t *__restrict__ cursor) {
  if (threadIdx.x || blockIdx.x) return;
  uint32_t run = 0;
  for (int k = SIZE_BINS - 1; k >= 0; k--) { cursor[k] = run; run += bins[k]; }
}
__global__ __launch_bounds__(256) void k_size_scatter(const uint32_t *__restrict__ counts, size_t total, int hs,
                                                      uint32_t *__restrict__ cursor, uint32_t *__restrict__ order) {
  __shared__ uint32_t h[SIZE_BINS], base[SIZE_BINS];
  for (int k = threadIdx.x; k < SIZE_BINS; k += 256) h[k] = 0;
  __syncthreads();
  const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t bin = 0, rank = 0;
  if (b < total) {
    const uint32_t cnt = counts[b];
    bin = size_bin(cnt, hs);
    rank = atomicAdd(&h[bin], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < SIZE_BINS; k += 256) base[k] = h[k] ? atomicAdd(&cursor[k], h[k]) : 0;
  __syncthreads();
  if (b < total) order[base[bin] + rank] = (uint32_t)b;
}

// one lane per bucket, taken in `order`: bucket (w, d) = sum of its points
template <class F>
__global__ __launch_bounds__(128) void k_bucket_sum(const AffineT<F> *__restrict__ pts, const uint32_t *__restrict__ counts,
                                                    const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ sorted,
                                                    const uint32_t *__restrict__ order, size_t total, size_t n, size_t nbs, int hs,
                                                    JacT<F> *__restrict__ buckets) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const size_t b = order[t];
  const size_t w = b / nbs;
  bls::XyzzT<F> acc = bls::xyzz_inf<F>();
  const uint32_t cnt = counts[b];  // 0 for digit 0 (k_hist skips it)
  if (cnt > (HEAVY << hs)) return;  // k_heavy_sum writes this one
  const uint32_t *idx = sorted + w * n + offsets[b];
  for (uint32_t i = 0; i < cnt; i++) acc = bucket_add(acc, entry_point(pts, idx[i]));
  buckets[b] = bucket_out(acc);
}

// Skewed scalars (few distinct digits in a window; witness vectors full of 0 / 1: gnark witnesses are mostly bits) put
// a large share of the points into one bucket; a single lane would add them one after the other, and a single
// workgroup still leaves 255 CUs idle while it adds a million points. Heavy buckets are therefore cut into chunks of
// HEAVY_CHUNK points: k_heavy_list lists the buckets and their chunks, k_heavy_sum sums one chunk per workgroup
// (strided partial sums, then a tree in LDS), k_heavy_combine adds a bucket's chunk sums.
constexpr uint32_t HEAVY_CHUNK = 8192;
struct HeavyBucket { uint32_t bucket, first_chunk, n_chunks; };
struct HeavyChunk { uint32_t bucket, first, count; };  // points [first, first + count) of the bucket's sorted range
// head[0] = number of heavy buckets, head[1] = number of chunks (both zeroed by the host)
__global__ void k_heavy_list(const uint32_t *__restrict__ counts, size_t total_buckets, int hs, uint32_t *__restrict__ head,
                             HeavyBucket *__restrict__ heavy, HeavyChunk *__restrict__ chunks) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total_buckets) return;
  const uint32_t cnt = counts[b];
  if (cnt <= (HEAVY << hs)) return;
  const uint32_t nch = (cnt + HEAVY_CHUNK - 1) / HEAVY_CHUNK, base = atomicAdd(head + 1, nch);
  heavy[atomicAdd(head, 1u)] = {(uint32_t)b, base, nch};
  for (uint32_t j = 0; j < nch; j++) {
    const uint32_t first = j * HEAVY_CHUNK;
    chunks[base + j] = {(uint32_t)b, first, cnt - first < HEAVY_CHUNK ? cnt - first : HEAVY_CHUNK};
  }
}
template <class F>
__global__ __launch_bounds__(256) void k_heavy_sum(const AffineT<F> *__restrict__ pts, const uint32_t *__restrict__ offsets,
                                                   const uint32_t *__restrict__ sorted, size_t n, size_t nbs,
                                                   const uint32_t *__restrict__ head, const HeavyChunk *__restrict__ chunks,
                                                   JacT<F> *__restrict__ partial) {
  __shared__ JacT<F> part[HEAVY_LANES<F>];
  if (blockIdx.x >= head[1]) return;
  const HeavyChunk ch = chunks[blockIdx.x];
  const size_t w = ch.bucket / nbs;
  const uint32_t *idx = sorted + w * n + offsets[ch.bucket] + ch.first;
  bls::XyzzT<F> acc = bls::xyzz_inf<F>();
  for (uint32_t t = threadIdx.x; t < ch.count; t += HEAVY_LANES<F>) acc = bucket_add(acc, entry_point(pts, idx[t]));
  part[threadIdx.x] = bucket_out(acc);
  __syncthreads();
  for (int half = HEAVY_LANES<F> / 2; half >= 1; half >>= 1) {
    if ((int)threadIdx.x < half) part[threadIdx.x] = bls::jac_add(part[threadIdx.x], part[threadIdx.x + half]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = part[0];
}
template <class F>
__global__ __launch_bounds__(256) void k_heavy_combine(const uint32_t *__restrict__ head, const HeavyBucket *__restrict__ heavy,
                                                       const JacT<F> *__restrict__ partial, JacT<F> *__restrict__ buckets) {
  __shared__ JacT<F> part[HEAVY_LANES<F>];
  if (blockIdx.x >= head[0]) return;
  const HeavyBucket hb = heavy[blockIdx.x];
  JacT<F> acc = bls::jac_inf<F>();
  for (uint32_t t = threadIdx.x; t < hb.n_chunks; t += HEAVY_LANES<F>) acc = bls::jac_add(acc, partial[hb.first_chunk + t]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int half = HEAVY_LANES<F> / 2; half >= 1; half >>= 1) {
    if ((int)threadIdx.x < half) part[threadIdx.x] = bls::jac_add(part[threadIdx.x], part[threadIdx.x + half]);  // idle lanes hold infinity
    __syncthreads();
  }
  if (threadIdx.x == 0) buckets[hb.bucket] = part[0];
}

// Point operations on operands in memory, NOT inlined: the running-sum loop below holds two accumulators across iterations, and
// with both additions (each carrying a doubling for the equal-points case) inlined beside them the F_p^2 instance needed 873
// registers more than the 512 a lane has (1 664 B of scratch per lane: profiles/r03_msm_kernel_meta.csv). One addition at a time
// fits (k_pair_reduce<Fp2>: 401 + 145 registers, no spills). dst may be one of the operands (the result is complete before it is stored).
template <class F>
__device__ __noinline__ void jac_double_mem(JacT<F> *dst, const JacT<F> *a) { *dst = bls::jac_double(*a); }
template <class F>
__device__ __noinline__ bool jac_add_distinct_mem(JacT<F> *dst, const JacT<F> *a, const JacT<F> *b) {
  JacT<F> r;
  if (!bls::jac_add_distinct(*a, *b, r)) return false;
  *dst = r;
  return true;
}
template <class F>
__device__ __forceinline__ void jac_add_mem(JacT<F> *dst, const JacT<F> *a, const JacT<F> *b) {
  if (!jac_add_distinct_mem<F>(dst, a, b)) jac_double_mem<F>(dst, a);  // the same point twice: a doubling (its own call)
}

// grid = (segs / 64, windows), segs = nbs / seg_len: out[w][seg] = sum_{d in run} d * B[w][d]
// MEM (the F_p^2 instance): the two running sums of a lane live in LDS ([lane], 2 x 336 B x 64 lanes = 42 KB) and its multiple of
// the run total in the lane's output slot; every point operation is a call of the two functions above.
template <class F, bool MEM = (sizeof(JacT<F>) > 200)>
__global__ __launch_bounds__(64) void k_segment_reduce(const JacT<F> *__restrict__ buckets, size_t nbs, uint32_t seg_len,
                                                       JacT<F> *__restrict__ out) {
  const uint32_t segs = (uint32_t)(nbs / seg_len);
  const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = blockIdx.y;
  const uint32_t s = seg * seg_len;
  const JacT<F> *B = buckets + (size_t)w * nbs;
  if constexpr (MEM) {
    __shared__ JacT<F> st[2][64];
    if (seg >= segs) return;  // no barrier below: every lane works on its own slots
    JacT<F> *running = &st[0][threadIdx.x], *acc = &st[1][threadIdx.x], *r = &out[(size_t)w * segs + seg];
    *running = bls::jac_inf<F>();
    *acc = bls::jac_inf<F>();
    for (int d = (int)(s + seg_len) - 1; d >= (int)s; d--) {
      jac_add_mem<F>(running, running, &B[d]);
      jac_add_mem<F>(acc, acc, running);
    }
    // acc = sum (d - s + 1) B_d, running = T = sum B_d  ->  sum d B_d = acc + (s - 1) T
    if (s == 0) {
      *r = bls::jac_neg(*running);
    } else {
      *r = bls::jac_inf<F>();
      const uint32_t k = s - 1;
      for (int i = 31 - __clz(k | 1); i >= 0; i--) {  // double-and-add from the top set bit (same group element as jac_mul_small)
        jac_double_mem<F>(r, r);
        if ((k >> i) & 1) jac_add_mem<F>(r, r, running);
      }
    }
    jac_add_mem<F>(r, acc, r);
  } else {
    if (seg >= segs) return;
    JacT<F> running = bls::jac_inf<F>(), acc = bls::jac_inf<F>();
    for (int d = (int)(s + seg_len) - 1; d >= (int)s; d--) {
      running = bls::jac_add(running, B[d]);
      acc = bls::jac_add(acc, running);
    }
    // acc = sum (d - s + 1) B_d, running = T = sum B_d  ->  sum d B_d = acc + (s - 1) T
    const JacT<F> off = s == 0 ? bls::jac_neg(running) : bls::jac_mul_small(running, s - 1);
    out[(size_t)w * segs + seg] = bls::jac_add(acc, off);
  }
}
// data[w][i] += data[w][i + half] for i + half < m   (m items left, half = ceil(m / 2); grid = (half/64, windows))
template <class F>
__global__ __launch_bounds__(64) void k_pair_reduce(JacT<F> *__restrict__ data, uint32_t stride, uint32_t m, uint32_t half) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i + half >= m) return;
  JacT<F> *row = data + (size_t)blockIdx.y * stride;
  row[i] = bls::jac_add(row[i], row[i + half]);
}

}  // namespace msm
