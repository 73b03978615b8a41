// Goldilocks radix-2 NTT / iNTT / coset-LDE kernels for gfx950.
//
// Replaces plonky2_field's fft_with_options / ifft / coset_fft (un-vendored dependency; it is
// what `PolynomialBatch::from_values` runs inside CircuitData::prove — SURVEY.md §8(a) A3).
//
// Structure: decimation-in-frequency, natural order in -> bit-reversed order out, split into
// passes. One pass = L consecutive butterfly stages over index bits [q, q+L) executed entirely
// in LDS by one workgroup on a tile of 2^(L+c) elements, followed by the inter-pass twiddle
// omega_{2^(q+L)}^(lo * k). HBM (or L2/MALL) is touched once per pass: one coalesced read,
// one coalesced write. Twiddles come from a 3-level power table of omega_n (6144 entries,
// L2-resident); the per-pass local twiddles are staged in LDS once per workgroup.
#pragma once
#include "gl.h"

namespace ntt {

constexpr int LOG_TILE_MAX = 12;           // 4096 elements = 32 KiB of LDS per workgroup
constexpr int TILE_MAX = 1 << LOG_TILE_MAX;
constexpr int THREADS = 256;
constexpr int PT_BITS = 11;                // power table: 3 levels x 2048 entries
constexpr int PT_SIZE = 1 << PT_BITS;

// base^e for e < 2^33 from the 3-level table T[level][j] = base^(j << (11*level))
__device__ __forceinline__ uint64_t pow_table(const uint64_t *__restrict__ T, uint64_t e) {
  uint32_t e0 = (uint32_t)e & (PT_SIZE - 1), e1 = (uint32_t)(e >> PT_BITS) & (PT_SIZE - 1),
           e2 = (uint32_t)(e >> (2 * PT_BITS));
  uint64_t r = T[e0];
  if (e1) r = gl::mul(r, T[PT_SIZE + e1]);
  if (e2) r = gl::mul(r, T[2 * PT_SIZE + e2]);
  return r;
}

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) {
  return bits ? (__brev(x) >> (32 - bits)) : 0;
}

struct PassArgs {
  uint64_t *data;          // batch of polynomials, in place
  size_t stride;           // elements between polynomials
  const uint64_t *wtab;    // power table of omega_n (forward) or omega_n^-1 (inverse)
  const uint64_t *stab;    // power table of the coset shift (or its inverse); may be null
  int log_n;
  int q, L, c;             // stage bits [q, q+L); tile = 2^L rows x 2^c outer indices
  int first, last;         // first / last pass of the transform
  uint64_t scale;          // multiplied into every output of the last pass (1/n for inverse), 0 = none
  int coset_pre;           // forward coset: multiply input i by shift^i on load in the first pass
  int coset_post;          // inverse coset: multiply output (natural index) by shift^-i — only with natural-order epilogue
  // --- first-pass source redirection (register radix-16 kernels only) ---
  const uint64_t *src;     // if non-null the first pass reads src (polynomial b at src + b*src_stride) instead of data
  size_t src_stride;
  const uint64_t *ptab;    // per-block pre-scale table [block][2^log_n]; applied on the first-pass load
  size_t block_stride;     // blockIdx.z selects an output block: data + b*stride + z*block_stride
  int block_bits;          // ptab row = bitrev(z, block_bits)  (output block z holds coset rev(z))
  int natural_out;         // ROWS single-pass transforms only: store in natural order
  int staged_store;        // ROWS passes: stage the last round's 16 consecutive elements per thread through LDS so that a wave
                           // stores 512 contiguous bytes per instruction (ntt16.h)
};

// position of (r, cc) inside the LDS tile == order of the tile's elements in memory
template <bool ROWS_CONTIG>
__device__ __forceinline__ int lds_pos(int r, int cc, int L, int c) {
  return ROWS_CONTIG ? (cc << L) + r : (r << c) + cc;
}

// One DIF pass. ROWS_CONTIG: q == 0 (each sub-network is 2^L contiguous elements and the tile holds
// 2^c of them); otherwise q >= c and the tile is 2^L rows of 2^c contiguous elements.
template <bool ROWS_CONTIG>
__global__ __launch_bounds__(THREADS) void k_dif_pass(PassArgs a) {
  __shared__ uint64_t tile[TILE_MAX];
  __shared__ uint64_t ltw[TILE_MAX / 2];
  const int L = a.L, c = a.c, q = a.q;
  const int tile_elems = 1 << (L + c);
  const int tid = threadIdx.x;
  uint64_t *poly = a.data + (size_t)blockIdx.y * a.stride;
  const uint32_t o_base = blockIdx.x << c;  // first outer index of this tile
  const uint32_t qmask = (1u << q) - 1;

  // local twiddles omega_{2^L}^e = omega_n^(e << (log_n - L)), e < 2^(L-1)
  for (int e = tid; e < (1 << (L - 1)); e += THREADS)
    ltw[e] = pow_table(a.wtab, (uint64_t)e << (a.log_n - L));

  // load (coalesced: consecutive threads -> consecutive addresses)
  for (int p = tid; p < tile_elems; p += THREADS) {
    int r, cc;
    if (ROWS_CONTIG) { cc = p >> L; r = p & ((1 << L) - 1); }
    else { r = p >> c; cc = p & ((1 << c) - 1); }
    uint32_t o = o_base + cc;
    size_t idx = ((size_t)(o >> q) << (q + L)) | ((size_t)r << q) | (o & qmask);
    uint64_t v = poly[idx];
    if (a.coset_pre && a.first) v = gl::mul(v, pow_table(a.stab, idx));
    tile[p] = v;
  }
  __syncthreads();

  // L radix-2 DIF stages over r
  const int n_bfly = tile_elems >> 1;
  for (int t = 0; t < L; t++) {
    const int hb = L - 1 - t;  // bit of r that pairs
    const int half = 1 << hb;
    for (int b = tid; b < n_bfly; b += THREADS) {
      int cc, pr;
      if (ROWS_CONTIG) { cc = b >> (L - 1); pr = b & ((1 << (L - 1)) - 1); }
      else { pr = b >> c; cc = b & ((1 << c) - 1); }
      int j = pr & (half - 1);
      int r0 = ((pr >> hb) << (hb + 1)) | j;
      int p0 = lds_pos<ROWS_CONTIG>(r0, cc, L, c), p1 = lds_pos<ROWS_CONTIG>(r0 + half, cc, L, c);
      uint64_t u = tile[p0], v = tile[p1];
      tile[p0] = gl::add(u, v);
      tile[p1] = gl::mul(gl::sub(u, v), ltw[j << t]);
    }
    __syncthreads();
  }

  // inter-pass twiddle + store
  for (int p = tid; p < tile_elems; p += THREADS) {
    int r, cc;
    if (ROWS_CONTIG) { cc = p >> L; r = p & ((1 << L) - 1); }
    else { r = p >> c; cc = p & ((1 << c) - 1); }
    uint32_t o = o_base + cc;
    uint32_t lo = o & qmask;
    size_t idx = ((size_t)(o >> q) << (q + L)) | ((size_t)r << q) | lo;
    uint64_t v = tile[p];
    if (q > 0) {
      uint64_t e = (uint64_t)lo * bitrev(r, L);  // < 2^(q+L)
      if (e) v = gl::mul(v, pow_table(a.wtab, e << (a.log_n - q - L)));
    }
    if (a.last && a.scale) v = gl::mul(v, a.scale);
    poly[idx] = v;
  }
}

// Out-of-place bit-reversal permutation (+ optional per-index coset post-scale by stab^i).
__global__ void k_bitrev_copy(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst,
                              size_t src_stride, size_t dst_stride, int log_n,
                              const uint64_t *__restrict__ stab) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  const uint64_t *s = src + (size_t)blockIdx.y * src_stride;
  uint64_t *d = dst + (size_t)blockIdx.y * dst_stride;
  uint64_t v = s[bitrev((uint32_t)i, log_n)];
  if (stab) v = gl::mul(v, pow_table(stab, i));
  d[i] = v;
}

// zero-padded copy for the LDE: out[b][i] = i < n ? in[b][i] : 0
__global__ void k_pad_copy(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst,
                           size_t src_stride, size_t dst_stride, size_t n, size_t N) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  dst[(size_t)blockIdx.y * dst_stride + i] = i < n ? src[(size_t)blockIdx.y * src_stride + i] : 0;
}

}  // namespace ntt
