// Z and partial-product polynomials of the PLONK permutation argument (SURVEY.md §3.3 step 5, §8(a) A7;
// plonky2 `wires_permutation_partial_products_and_zs`, un-vendored dependency).
//
// For row i (x = omega^i) and routed wire j:
//     num_j = w_ij + beta*k_j*x + gamma ,   den_j = w_ij + beta*sigma_j(x) + gamma
// The R quotients num_j/den_j are multiplied in chunks of `chunk` (= quotient_degree_factor): the running
// product over the chunks gives the npp = ceil(R/chunk)-1 partial products and Z(g x); Z(1) = 1.
//
// Two kernels, both with (challenge, proof) grid dimensions:
//   k_chunk_products  one lane per row: R numerators/denominators, ONE field inversion per row (Montgomery
//                     batch inversion of the chunk denominators), chunk products written in place of the
//                     partial products, the row product in place of Z;
//   k_scan_rows       one workgroup per (challenge, proof): exclusive prefix product of the row products
//                     -> Z, then the partial products of every row.
#pragma once
#include "gl.h"

namespace zs {

constexpr int MAX_CHUNKS = 32;

struct Args {
  const uint64_t *wires;        // [proof][num_wires][n]
  size_t wires_proof_stride;
  const uint64_t *const *sigmas;  // per proof: [num_routed][n] values over <omega_n>
  const uint64_t *k_is;         // [num_routed]
  const uint64_t *betas;        // betas[proof*chal_stride + c]
  const uint64_t *gammas;       // gammas[proof*chal_stride + c]
  size_t chal_stride;           // nc for two separate [proof][nc] arrays, 2*nc for the transcript's [proof][betas | gammas]
  uint64_t *out;                // [proof][nc*(1+npp)][n], committed order
  size_t out_proof_stride;
  const uint64_t *omega_tab;    // power table of omega_n (ntt::pow_table layout)
  size_t n;
  int num_routed, chunk, npp, nc;
};

__device__ __forceinline__ uint64_t pow_tab(const uint64_t *T, uint64_t e) {
  uint32_t e0 = (uint32_t)e & 2047, e1 = (uint32_t)(e >> 11) & 2047, e2 = (uint32_t)(e >> 22);
  uint64_t r = T[e0];
  if (e1) r = gl::mul(r, T[2048 + e1]);
  if (e2) r = gl::mul(r, T[4096 + e2]);
  return r;
}

// grid = (n/256, nc, B). NCH bounds the number of chunks (npp + 1) at compile time: with every loop over the chunk arrays fully
// unrolled they live in registers (the product shape has 10 chunks: 80 routed wires in chunks of 8); indexed by a run-time loop
// variable they were 784 bytes of scratch per lane. NCH = MAX_CHUNKS is the general kernel.
template <int NCH>
__global__ __launch_bounds__(256) void k_chunk_products(Args a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int c = blockIdx.y;
  const size_t proof = blockIdx.z;
  const uint64_t *w = a.wires + proof * a.wires_proof_stride + i;
  const uint64_t *sg = a.sigmas[proof] + i;
  const uint64_t beta = a.betas[proof * a.chal_stride + c], gamma = a.gammas[proof * a.chal_stride + c];
  const uint64_t bx = gl::mul(beta, pow_tab(a.omega_tab, i));
  const int nchunks = a.npp + 1;
  uint64_t num[NCH], den[NCH];
#pragma unroll
  for (int t = 0; t < NCH; t++) {
    if (t >= nchunks) break;
    uint64_t pn = 1, pd = 1;
    for (int j = t * a.chunk; j < a.num_routed && j < (t + 1) * a.chunk; j++) {
      uint64_t wv = w[(size_t)j * a.n];
      uint64_t base = gl::add(wv, gamma);
      pn = gl::mul(pn, gl::add(base, gl::mul(bx, a.k_is[j])));
      pd = gl::mul(pd, gl::add(base, gl::mul(beta, sg[(size_t)j * a.n])));
    }
    num[t] = pn;
    den[t] = pd;
  }
  // Montgomery batch inversion of den[0..nchunks): one inversion per row
  uint64_t pre[NCH];
  uint64_t acc = 1;
#pragma unroll
  for (int t = 0; t < NCH; t++)
    if (t < nchunks) { pre[t] = acc; acc = gl::mul(acc, den[t]); }
  uint64_t inv = gl::inv(acc);
  uint64_t *out = a.out + proof * a.out_proof_stride;
  uint64_t *Z = out + (size_t)c * a.n;
  uint64_t *PP = out + ((size_t)a.nc + (size_t)c * a.npp) * a.n;
  uint64_t row = 1;
#pragma unroll
  for (int t = NCH - 1; t >= 0; t--)
    if (t < nchunks) {
      uint64_t dinv = gl::mul(inv, pre[t]);
      inv = gl::mul(inv, den[t]);
      num[t] = gl::mul(num[t], dinv);  // chunk product t
    }
#pragma unroll
  for (int t = 0; t < NCH; t++)
    if (t < nchunks) {
      row = gl::mul(row, num[t]);
      if (t < a.npp) PP[(size_t)t * a.n + i] = num[t];
    }
  Z[i] = row;  // product of all chunks of this row
  // the last chunk product is recovered in k_scan_rows as row / prod(first npp chunks) — not needed:
  // pp_t only involve chunks 0..npp-1 and Z(g x) = Z(x) * row.
}

// grid = (1, nc, B), 256 threads; rows are split contiguously over the threads
__global__ __launch_bounds__(256) void k_scan_rows(Args a) {
  __shared__ uint64_t tot[256];
  const int t = threadIdx.x;
  const int c = blockIdx.y;
  const size_t proof = blockIdx.z;
  uint64_t *out = a.out + proof * a.out_proof_stride;
  uint64_t *Z = out + (size_t)c * a.n;
  uint64_t *PP = out + ((size_t)a.nc + (size_t)c * a.npp) * a.n;
  const size_t per = (a.n + 255) / 256;
  const size_t lo = (size_t)t * per < a.n ? (size_t)t * per : a.n, hi = lo + per < a.n ? lo + per : a.n;
  uint64_t p = 1;
  for (size_t i = lo; i < hi; i++) p = gl::mul(p, Z[i]);
  tot[t] = p;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {  // inclusive prefix product over chunk totals
    uint64_t v = tot[t];
    if (t >= d) v = gl::mul(v, tot[t - d]);
    __syncthreads();
    tot[t] = v;
    __syncthreads();
  }
  uint64_t z = t ? tot[t - 1] : 1;  // Z at the first row of this chunk
  for (size_t i = lo; i < hi; i++) {
    uint64_t row = Z[i];
    Z[i] = z;
    uint64_t acc = z;
    for (int k = 0; k < a.npp; k++) {
      acc = gl::mul(acc, PP[(size_t)k * a.n + i]);
      PP[(size_t)k * a.n + i] = acc;
    }
    z = gl::mul(z, row);
  }
}

}  // namespace zs
