// Goldilocks field arithmetic for gfx950 device code (and host code: every function is
// __host__ __device__ so the C++ host side of the prover uses the same definitions).
//
// p = 2^64 - 2^32 + 1. Elements are canonical u64 in [0, p) at every kernel boundary
// (the wire format of plonky2's GoldilocksField); inside a kernel
// values may be carried "lazily" as any u64 congruent to the value.
//
// CDNA4 has no 64x64->128 multiply: the product is built from v_mad_u64_u32 and the
// reduction uses 2^64 == 2^32-1 and 2^96 == -1 (mod p), i.e. only adds/subs/shifts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GL_HD __host__ __device__ __forceinline__

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p

GL_HD uint64_t canon(uint64_t x) { return x >= P ? x - P : x; }

// a, b canonical -> canonical
GL_HD uint64_t add(uint64_t a, uint64_t b) {
  uint64_t s = a + b;
  // overflow past 2^64 or landing in [p, 2^64): subtract p (== add EPS mod 2^64)
  return (s < a || s >= P) ? s + EPS : s;
}
GL_HD uint64_t sub(uint64_t a, uint64_t b) {
  uint64_t d = a - b;
  return (a < b) ? d - EPS : d;
}
GL_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }

// lazy add: a any u64, b any u64 -> u64 congruent to a+b (not canonical)
GL_HD uint64_t add_lazy(uint64_t a, uint64_t b) {
  uint64_t s = a + b;
  if (s < a) {          // wrapped: + 2^64 == + EPS
    s += EPS;           // cannot wrap twice unless s >= 2^64-EPS, handled below
    if (s < EPS) s += EPS;
  }
  return s;
}

// 128-bit (hi:lo) -> canonical
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
  uint64_t hh = hi >> 32, hl = hi & EPS;
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= EPS;
  uint64_t t1 = (hl << 32) - hl;  // hl * (2^32 - 1)
  uint64_t r = t0 + t1;
  if (r < t1) r += EPS;
  return canon(r);
}

GL_HD void mul_wide(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b;
  hi = __umul64hi(a, b);
#else
  unsigned __int128 m = (unsigned __int128)a * b;
  lo = (uint64_t)m;
  hi = (uint64_t)(m >> 64);
#endif
}

GL_HD uint64_t mul(uint64_t a, uint64_t b) {
  uint64_t lo, hi;
  mul_wide(a, b, lo, hi);
  return reduce128(lo, hi);
}
GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }

GL_HD uint64_t pow7(uint64_t x) {
  uint64_t x2 = sqr(x), x4 = sqr(x2), x3 = mul(x, x2);
  return mul(x3, x4);
}

GL_HD uint64_t pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = mul(r, b);
    b = sqr(b);
    e >>= 1;
  }
  return r;
}
GL_HD uint64_t inv(uint64_t a) { return pow(a, P - 2); }

// quadratic extension F_p[X]/(X^2 - 7)
struct Ext {
  uint64_t a, b;
};
GL_HD Ext ext_add(Ext x, Ext y) { return {add(x.a, y.a), add(x.b, y.b)}; }
GL_HD Ext ext_sub(Ext x, Ext y) { return {sub(x.a, y.a), sub(x.b, y.b)}; }
GL_HD Ext ext_mul(Ext x, Ext y) {
  uint64_t bb = mul(x.b, y.b);
  // 7*bb
  uint64_t lo, hi;
  mul_wide(bb, 7, lo, hi);
  return {add(mul(x.a, y.a), reduce128(lo, hi)), add(mul(x.a, y.b), mul(x.b, y.a))};
}
GL_HD Ext ext_scale(Ext x, uint64_t s) { return {mul(x.a, s), mul(x.b, s)}; }

}  // namespace gl
