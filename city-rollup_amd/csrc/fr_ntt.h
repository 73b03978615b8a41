// NTT over the BLS12-381 scalar field (the F_r transforms of the Groth16 quotient).
// Decimation in frequency on a work array of Montgomery elements: natural order in, bit-reversed order out of the
// butterflies; the load / store kernels convert from / to the canonical 4 x u64 form of the API, apply the coset
// powers and 1/n, and undo the bit reversal. Stages with a butterfly distance of 512 elements or more run over HBM,
// two stages per launch (radix 4); the last ten stages run on 1024-element tiles in LDS (40 KB).
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
// work[i] = mont(data[i]) * (powers ? powers[i] : 1)
__global__ void k_load(const uint32_t *__restrict__ data, size_t n, const Fr *__restrict__ powers, Fr *__restrict__ work) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr v = blsfr::fr_from_canonical(data + 8 * i);
  if (powers) v = blsfr::fr_mul(v, powers[i]);
  work[i] = v;
}
// one DIF stage over HBM: pairs (j, j + half) inside blocks of 2*half; twiddle omega_n^(j * n/(2 half)) = tw[j * step]
__global__ void k_stage(Fr *__restrict__ work, size_t n, size_t half, size_t step, const Fr *__restrict__ tw) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / 2) return;
  const size_t j = t & (half - 1), lo = ((t - j) << 1) + j, hi = lo + half;
  const Fr u = work[lo], v = work[hi];
  work[lo] = blsfr::fr_add(u, v);
  work[hi] = blsfr::fr_mul(blsfr::fr_sub(u, v), tw[j * step]);
}
// two consecutive DIF stages (distances half and half/2) in one pass over HBM: lane t owns the four elements
// x, x + half/2, x + half, x + 3 half/2 of a block of 2*half
__global__ void k_stage2(Fr *__restrict__ work, size_t n, size_t half, size_t step, const Fr *__restrict__ tw) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / 4) return;
  const size_t q = half >> 1, j = t & (q - 1), x = ((t - j) << 2) + j;
  const Fr a = work[x], b = work[x + q], c = work[x + half], d = work[x + half + q];
  // stage `half`: (a, c) with twiddle index j, (b, d) with j + q
  const Fr a1 = blsfr::fr_add(a, c), c1 = blsfr::fr_mul(blsfr::fr_sub(a, c), tw[j * step]);
  const Fr b1 = blsfr::fr_add(b, d), d1 = blsfr::fr_mul(blsfr::fr_sub(b, d), tw[(j + q) * step]);
  // stage `half/2`: (a1, b1) and (c1, d1), both with twiddle index j at twice the step
  const Fr w = tw[j * 2 * step];
  work[x] = blsfr::fr_add(a1, b1);
  work[x + q] = blsfr::fr_mul(blsfr::fr_sub(a1, b1), w);
  work[x + half] = blsfr::fr_add(c1, d1);
  work[x + half + q] = blsfr::fr_mul(blsfr::fr_sub(c1, d1), w);
}
// the last `stages` (<= LOG_TILE) DIF stages on contiguous tiles of 2^stages elements, in LDS
__global__ __launch_bounds__(512) void k_tile(Fr *__restrict__ work, size_t n, int stages, size_t tw_step, const Fr *__restrict__ tw) {
  __shared__ Fr tile[1 << LOG_TILE];
  const size_t tsize = (size_t)1 << stages, base = (size_t)blockIdx.x * tsize;
  for (size_t i = threadIdx.x; i < tsize; i += blockDim.x) tile[i] = work[base + i];
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
  for (size_t i = threadIdx.x; i < tsize; i += blockDim.x) work[base + i] = tile[i];
}
// data[rev(p)] = canonical(work[p] * scale * (powers ? powers[rev(p)] : 1))
__global__ void k_store(const Fr *__restrict__ work, size_t n, int log_n, Fr scale, int use_scale, const Fr *__restrict__ powers,
                        uint32_t *__restrict__ data) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t i = log_n ? (size_t)(__brevll((unsigned long long)p) >> (64 - log_n)) : 0;
  Fr v = work[p];
  if (use_scale) v = blsfr::fr_mul(v, scale);
  if (powers) v = blsfr::fr_mul(v, powers[i]);
  blsfr::fr_to_canonical(v, data + 8 * i);
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
