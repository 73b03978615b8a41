// NTT over the BLS12-381 scalar field (the F_r transforms of the Groth16 quotient).
// Decimation in frequency on a work array of Montgomery elements: natural order in, bit-reversed order out of the
// butterflies; the load / store kernels convert from / to the canonical 4 x u64 form of the API, apply the coset
// powers and 1/n, and undo the bit reversal. Every pass over HBM works on a 1024-element tile in LDS (40 KB):
//   k_colpass  g <= 7 consecutive stages with butterfly distances >= 1024: the tile is 2^g rows x 2^(10-g) adjacent
//              columns (segments of >= 320 contiguous bytes); the first one also does the load conversion
//   k_tile     the last ten stages on contiguous tiles; also does the store conversion
// so 2^20 and 2^22 points take three passes (5+5+10, 6+6+10 stages), 2^24 three (7+7+10).
#pragma once
#include "bls12_381_fr.h"

namespace frntt {

using blsfr::Fr;
constexpr int LOG_TILE = 10;

// out[i] = base^i for i < count (Montgomery); base given in Montgomery form
__global__ void k_powers(Fr base, size_t count, Fr *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = blsfr::fr_pow_u64(base, i);
}
// element p of the transform's input: Montgomery form of data[p], times the coset power
__device__ __forceinline__ Fr load_input(const uint32_t *__restrict__ data, const Fr *__restrict__ powers, size_t p) {
  Fr v = blsfr::fr_from_canonical(data + 8 * p);
  if (powers) v = blsfr::fr_mul(v, powers[p]);
  return v;
}
// what the store does with work[p]: data[rev(p)] = canonical(v * scale * powers[rev(p)])
struct StoreArgs {
  uint32_t *data;
  const Fr *powers;
  Fr scale;
  int use_scale, log_n;
};
__device__ __forceinline__ void store_output(const StoreArgs &st, size_t p, Fr v) {
  const size_t i = st.log_n ? (size_t)(__brevll((unsigned long long)p) >> (64 - st.log_n)) : 0;
  if (st.use_scale) v = blsfr::fr_mul(v, st.scale);
  if (st.powers) v = blsfr::fr_mul(v, st.powers[i]);
  blsfr::fr_to_canonical(v, st.data + 8 * i);
}

// g DIF stages s_hi, s_hi-1, ..., s_hi-g+1 (all with distance >= 2^LOG_TILE) in one pass: a workgroup owns 2^g rows
// (stride st = 2^(s_hi-g+1)) x C = 2^(LOG_TILE-g) adjacent columns. LOAD: the input comes from the canonical array.
template <bool LOAD>
__global__ __launch_bounds__(512) void k_colpass(Fr *__restrict__ work, const uint32_t *__restrict__ data, const Fr *__restrict__ powers,
                                                 size_t n, int s_hi, int g, const Fr *__restrict__ tw) {
  __shared__ Fr tile[1 << LOG_TILE];
  const int log_c = LOG_TILE - g;
  const size_t C = (size_t)1 << log_c, st = (size_t)1 << (s_hi - g + 1);
  const size_t groups = st >> log_c;  // column groups per block of 2^(s_hi+1) elements
  const size_t blk = blockIdx.x / groups, cg = blockIdx.x % groups;
  const size_t base = (blk << (s_hi + 1)) + (cg << log_c);
  for (size_t i = threadIdx.x; i < ((size_t)1 << LOG_TILE); i += blockDim.x) {
    const size_t idx = base + (i >> log_c) * st + (i & (C - 1));
    tile[i] = LOAD ? load_input(data, powers, idx) : work[idx];
  }
  __syncthreads();
  for (int t = g - 1; t >= 0; t--) {
    const size_t step = n >> (s_hi - g + 2 + t);  // n / (2 * st * 2^t)
    for (size_t b = threadIdx.x; b < ((size_t)1 << (LOG_TILE - 1)); b += blockDim.x) {
      const size_t c = b & (C - 1), rb = b >> log_c, jr = rb & (((size_t)1 << t) - 1);
      const size_t rlo = ((rb - jr) << 1) + jr, lo = (rlo << log_c) + c, hi = lo + ((size_t)1 << (t + log_c));
      const size_t j = jr * st + (cg << log_c) + c;
      const Fr u = tile[lo], v = tile[hi];
      tile[lo] = blsfr::fr_add(u, v);
      tile[hi] = blsfr::fr_mul(blsfr::fr_sub(u, v), tw[j * step]);
    }
    __syncthreads();
  }
  for (size_t i = threadIdx.x; i < ((size_t)1 << LOG_TILE); i += blockDim.x)
    work[base + (i >> log_c) * st + (i & (C - 1))] = tile[i];
}

// the last `stages` (<= LOG_TILE) DIF stages on contiguous tiles of 2^stages elements, in LDS.
// LOAD: input from the canonical array (transforms of <= 2^LOG_TILE points); STORE: output to the canonical array.
template <bool LOAD, bool STORE>
__global__ __launch_bounds__(512) void k_tile(Fr *__restrict__ work, const uint32_t *__restrict__ data, const Fr *__restrict__ powers,
                                              size_t n, int stages, size_t tw_step, const Fr *__restrict__ tw, StoreArgs sa) {
  __shared__ Fr tile[1 << LOG_TILE];
  const size_t tsize = (size_t)1 << stages, base = (size_t)blockIdx.x * tsize;
  for (size_t i = threadIdx.x; i < tsize; i += blockDim.x) tile[i] = LOAD ? load_input(data, powers, base + i) : work[base + i];
  __syncthreads();
  for (int s = stages - 1; s >= 0; s--) {
    const size_t half = (size_t)1 << s;
    // twiddle of the pair at offset j inside a block of 2*half: omega_{2 half}^j = omega_n^(j * n / (2 half))
    const size_t step = tw_step << (stages - 1 - s);
    for (size_t t = threadIdx.x; t < tsize / 2; t += blockDim.x) {
      const size_t j = t & (half - 1), lo = ((t - j) << 1) + j, hi = lo + half;
      const Fr u = tile[lo], v = tile[hi];
      tile[lo] = blsfr::fr_add(u, v);
      tile[hi] = blsfr::fr_mul(blsfr::fr_sub(u, v), tw[j * step]);
    }
    __syncthreads();
  }
  for (size_t i = threadIdx.x; i < tsize; i += blockDim.x) {
    if (STORE) store_output(sa, base + i, tile[i]);
    else work[base + i] = tile[i];
  }
}
// Groth16 quotient on the coset: a[i] = (a[i] * b[i] - c[i]) * den   (canonical words in and out)
__global__ void k_quotient_pointwise(uint32_t *__restrict__ a, const uint32_t *__restrict__ b, const uint32_t *__restrict__ c, Fr den,
                                     size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr A = blsfr::fr_from_canonical(a + 8 * i), B = blsfr::fr_from_canonical(b + 8 * i), C = blsfr::fr_from_canonical(c + 8 * i);
  blsfr::fr_to_canonical(blsfr::fr_mul(blsfr::fr_sub(blsfr::fr_mul(A, B), C), den), a + 8 * i);
}

}  // namespace frntt
