// Goldilocks field arithmetic for gfx950 device code (and host code: every function is
// __host__ __device__ so the C++ host side of the prover uses the same definitions).
//
// p = 2^64 - 2^32 + 1. Elements are canonical u64 in [0, p) at every kernel boundary
// (the wire format of plonky2's GoldilocksField); inside a kernel
// values may be carried "lazily" as any u64 congruent to the value.
//
// CDNA4 has no 64x64->128 multiply: the product is built from v_mad_u64_u32 and the
// reduction uses 2^64 == 2^32-1 and 2^96 == -1 (mod p).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GL_HD __host__ __device__ __forceinline__
// device code takes the hand-scheduled `asm` forms below; GL_PORTABLE (experiments: tools/ubench_leaf_latency.hip) builds the
// portable C forms for the device too
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_PORTABLE)
#define GL_DEVICE_ASM 1
#else
#define GL_DEVICE_ASM 0
#endif

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p

// x - p == x + EPS (mod 2^64): one 64-bit compare, one select, one 64-bit add (v_lshl_add_u64) on gfx950
GL_HD uint64_t canon(uint64_t x) { return x + (x >= P ? EPS : 0); }

GL_HD uint32_t lo32(uint64_t x) { return (uint32_t)x; }
GL_HD uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }
GL_HD uint64_t pack(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// The helpers below are phrased through unsigned __int128 so that LLVM legalises them into plain
// v_sub_co/v_subb_co (v_add_co/v_addc_co) carry chains: 5 instructions per modular subtraction,
// 7 per addition, no 64-bit compares/selects.
typedef unsigned __int128 u128;

// a, b canonical -> canonical.  a - b, plus p when it borrowed (p == -EPS mod 2^64)
GL_HD uint64_t sub(uint64_t a, uint64_t b) {
  u128 d = (u128)a - b;
  uint64_t borrowed = (uint64_t)(d >> 64);  // 0 or all ones
  return (uint64_t)d - (borrowed & EPS);
}
GL_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }
// a, b canonical -> canonical, five instructions on gfx950 (one v_lshl_add_u64, two 64-bit compares, a select, one more add; as
// a - (p - b) through carry chains it was seven). s = a + b mod 2^64. If it wrapped, the true sum is s + 2^64 == s + EPS, which
// cannot wrap again (s <= 2p - 2 - 2^64) and is below p; if it did not wrap but s >= p, then s - p == s + EPS (mod 2^64): both
// repairs add EPS.
GL_HD uint64_t add(uint64_t a, uint64_t b) {
  const uint64_t s = a + b;
  return s + (((s < a) | (s >= P)) ? EPS : 0);
}

// lazy add: a any u64, b any u64 -> u64 congruent to a+b (not canonical)
GL_HD uint64_t add_lazy(uint64_t a, uint64_t b) {
  uint64_t s = a + b;
  if (s < a) {          // wrapped: + 2^64 == + EPS
    s += EPS;
    if (s < EPS) s += EPS;
  }
  return s;
}

// A modular multiplication is 12 VALU instructions on gfx950: 4 v_mad_u64_u32 for the product (+ 2 moves, 1 select), 2 for
// `lo - w3`, one more multiply-add for `+ w2 (2^32 - 1)` (+ select, 64-bit add). Three pieces are written as `asm` because
// carry-outs are out of the compiler's reach (it zero-extends, adds and compares instead: 19 instructions; the permutation
// measures 2.06 -> 2.48 -> 2.62 G/s with them, tools/ubench_poseidon_variants.hip). The `s_nop 1` are the two wait states
// gfx950 needs between a VALU write of an SGPR and a VALU read of it; the compiler's hazard recogniser does not look inside
// `asm`. tests/test_gpu_parity.py drives `cp_field_mul` through every carry / borrow corner on the device.

// lo + top * (2^32 - 1) as a lazy u64 (top * 2^64 == top * (2^32 - 1))
GL_HD uint64_t fold_top(uint64_t lo, uint32_t top) {
#if GL_DEVICE_ASM
  // one multiply-add; its carry-out (one wrap of 2^64 == + EPS) selects the repair. The repaired sum cannot wrap again:
  // lo + top (2^32 - 1) < 2^65 - 2^33, so after one wrap it is below 2^64 - 2^33.
  uint64_t r, cy;
  uint32_t m;
  asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, %1" : "=&v"(r), "=&s"(cy), "=v"(m) : "v"(top), "v"(lo));
  return r + m;
#else
  const u128 s = (u128)lo + (((uint64_t)top << 32) - top);
  return (uint64_t)s + ((0 - (uint64_t)(s >> 64)) & EPS);
#endif
}

// 128-bit (hi:lo) -> lazy u64 (any representative):  lo - hi_hi + hi_lo*(2^32-1),
// every wrap of 2^64 repaid by -+EPS
GL_HD uint64_t reduce128_lazy(uint64_t lo, uint64_t hi) {
  const uint32_t w2 = lo32(hi), w3 = hi32(hi);
#if GL_DEVICE_ASM
  // lo - w3 borrows only when lo < w3 < 2^32 (a 2^-32 event for a product): the repair (- EPS for the 2^64 that was lent)
  // sits behind a wave-uniform branch instead of costing three instructions on every multiplication
  uint32_t tl, th, m;
  uint64_t bm;
  asm("v_sub_co_u32_e64 %0, %2, %3, %4\n\ts_nop 1\n\tv_subbrev_co_u32_e64 %1, %2, 0, %5, %2"
      : "=&v"(tl), "=&v"(th), "=&s"(bm)
      : "v"(lo32(lo)), "v"(w3), "v"(hi32(lo)));
  uint64_t t0 = pack(tl, th);
  if (__builtin_expect(bm != 0, 0)) {
    asm volatile("v_cndmask_b32 %0, 0, -1, %1" : "=v"(m) : "s"(bm));
    t0 -= m;
  }
#else
  const u128 t = (u128)lo - w3;
  const uint64_t t0 = (uint64_t)t - ((uint64_t)(t >> 64) & EPS);
#endif
  return fold_top(t0, w2);
}
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) { return canon(reduce128_lazy(lo, hi)); }

// 64x64 -> 128 as exactly four 32x32+64 multiply-adds (v_mad_u64_u32)
GL_HD void mul_wide(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
  const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
  const uint64_t p00 = (uint64_t)a0 * b0;
  const uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);  // < 2^64
#if GL_DEVICE_ASM
  // the third multiply-add takes the whole second one as its addend; its carry-out replaces two zero-extending moves
  // and a 64-bit addition
  uint64_t r, cr;
  uint32_t cbit;
  asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, 1, %1"
      : "=&v"(r), "=&s"(cr), "=&v"(cbit)
      : "v"(a1), "v"(b0), "v"(p01));
  lo = pack(lo32(p00), lo32(r));
  hi = (uint64_t)a1 * b1 + pack(hi32(r), cbit);
#else
  (void)p01;
  const u128 p = (u128)a * b;  // the host has the instruction
  lo = (uint64_t)p;
  hi = (uint64_t)(p >> 64);
#endif
}

GL_HD uint64_t mul(uint64_t a, uint64_t b) {
  uint64_t lo, hi;
  mul_wide(a, b, lo, hi);
  return reduce128(lo, hi);
}
GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }
// a*b + c as a lazy u64 (any representative < 2^64 of the class); a, b, c may be lazy themselves
GL_HD uint64_t mul_add_lazy(uint64_t a, uint64_t b, uint64_t c) {
  uint64_t lo, hi;
  mul_wide(a, b, lo, hi);
  lo += c;
  hi += lo < c;  // a*b <= (2^64-1)^2 keeps hi <= 2^64-2: no overflow
  return reduce128_lazy(lo, hi);
}

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
