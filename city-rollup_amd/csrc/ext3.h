// The two non-row-local primitives of a logUp (log-derivative lookup) argument over a cubic extension of Goldilocks,
// F_p[X]/(X^3 - m1 X - m0) with the modulus a PARAMETER (include/cityprover.h cp_cubic_batch_inverse_dev,
// cp_column_prefix_sum_dev): batched inversion of columns of extension elements, and the running sum of columns down the
// trace. The SHA-256 STARK the reference proves (smartgadget.rs:55-79: 912 extended columns) fills its lookup columns with
// exactly these two shapes — per row a sum of inverses, then a prefix sum over the rows; which columns and which modulus is
// the absent crate's business (starkyx 0.1.0, /root/reference/Cargo.toml:112) and arrives as data. zs.h has the same two
// shapes over F_p (Montgomery batch inversion per row; a workgroup scan) for plonky2's permutation argument.
#pragma once
#include "gl.h"

namespace ext3 {

struct E { uint64_t c[3]; };

// X * v in F_p[X]/(X^3 - m1 X - m0)
GL_HD E mulx(uint64_t m0, uint64_t m1, const E &v) { return E{{gl::mul(m0, v.c[2]), gl::add(v.c[0], gl::mul(m1, v.c[2])), v.c[1]}}; }
GL_HD E mul(uint64_t m0, uint64_t m1, const E &a, const E &b) {
  const E ax = mulx(m0, m1, a), axx = mulx(m0, m1, ax);
  E r;
  for (int i = 0; i < 3; i++) r.c[i] = gl::add(gl::add(gl::mul(a.c[i], b.c[0]), gl::mul(ax.c[i], b.c[1])), gl::mul(axx.c[i], b.c[2]));
  return r;
}
// a^-1 = adj / norm: the first row of the adjugate of the multiplication-by-a matrix (columns a, X a, X^2 a) solves
// M y = (1, 0, 0); norm = det M. Returns the adjugate row; the caller divides by the norm (batched).
GL_HD E adjugate_row(uint64_t m0, uint64_t m1, const E &a, uint64_t &norm) {
  const E c1 = mulx(m0, m1, a), c2 = mulx(m0, m1, c1);
  E k;
  k.c[0] = gl::sub(gl::mul(c1.c[1], c2.c[2]), gl::mul(c2.c[1], c1.c[2]));
  k.c[1] = gl::sub(gl::mul(c2.c[1], a.c[2]), gl::mul(a.c[1], c2.c[2]));
  k.c[2] = gl::sub(gl::mul(a.c[1], c1.c[2]), gl::mul(c1.c[1], a.c[2]));
  norm = gl::add(gl::add(gl::mul(a.c[0], k.c[0]), gl::mul(c1.c[0], k.c[1])), gl::mul(c2.c[0], k.c[2]));
  return k;
}

constexpr int GROUP = 8;  // elements per lane and field inversion (16 spills: 3 x 16 adjugate rows + norms + prefixes)

// lane <-> row (coalesced column accesses); blockIdx.y <-> a group of up to GROUP elements. cols: [3 * count][n]
__global__ __launch_bounds__(256) void k_batch_inverse(uint64_t *cols, size_t count, size_t n, uint64_t m0, uint64_t m1) {
  const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const size_t e0 = (size_t)blockIdx.y * GROUP;
  const int g = (int)(count - e0 < (size_t)GROUP ? count - e0 : (size_t)GROUP);
  E adj[GROUP];
  uint64_t norm[GROUP], pre[GROUP];
  // Montgomery's trick on the norms; a zero norm (the zero element) is stepped over and maps to zero
  uint64_t run = 1;
#pragma unroll
  for (int j = 0; j < GROUP; j++)
    if (j < g) {
      uint64_t *p = cols + 3 * (e0 + j) * n + row;
      adj[j] = adjugate_row(m0, m1, E{{p[0], p[n], p[2 * n]}}, norm[j]);
      pre[j] = run;
      if (norm[j]) run = gl::mul(run, norm[j]);
    }
  uint64_t inv = gl::inv(run);
#pragma unroll
  for (int j = GROUP - 1; j >= 0; j--)
    if (j < g) {
      uint64_t ni = 0;
      if (norm[j]) {
        ni = gl::mul(inv, pre[j]);
        inv = gl::mul(inv, norm[j]);
      }
      uint64_t *p = cols + 3 * (e0 + j) * n + row;
      p[0] = gl::mul(adj[j].c[0], ni);
      p[n] = gl::mul(adj[j].c[1], ni);
      p[2 * n] = gl::mul(adj[j].c[2], ni);
    }
}

// running sum of one column per workgroup: each thread sums a contiguous chunk, the 256 chunk sums are scanned in LDS, each
// thread writes its chunk's running sums from its offset. cols: [k][n]
__global__ __launch_bounds__(256) void k_prefix_sum(uint64_t *cols, size_t n, int exclusive) {
  __shared__ uint64_t part[256];
  uint64_t *col = cols + (size_t)blockIdx.x * n;
  const size_t chunk = (n + 255) / 256, lo = (size_t)threadIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  uint64_t s = 0;
  for (size_t i = lo; i < hi; i++) s = gl::add(s, col[i]);
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan
    const uint64_t v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] = gl::add(part[threadIdx.x], v);
    __syncthreads();
  }
  uint64_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (size_t i = lo; i < hi; i++) {
    const uint64_t v = col[i];
    if (exclusive) { col[i] = run; run = gl::add(run, v); }
    else { run = gl::add(run, v); col[i] = run; }
  }
}

}  // namespace ext3
