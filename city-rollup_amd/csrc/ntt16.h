// Register radix-16 DIF passes for the Goldilocks NTT (gfx950).
//
// Same pass decomposition as ntt.h (one pass = L butterfly stages over index bits [q, q+L) of a
// 4096-element tile, then the inter-pass twiddle), but the stages run in REGISTERS: every thread
// owns 16 elements and executes up to four radix-2 DIF stages on them per round, so the tile
// crosses LDS once per round (<= 2 exchanges per pass) instead of once per stage, and the first
// load / last store go straight between HBM and registers.
//
// Twiddles inside a round split into (a) a power of omega_16, which in Goldilocks is +-2^(12k)
// (2 has order 192 mod p and omega_16 = 2^156): a shift, never a multiplication; and (b) one
// thread-uniform factor per stage, omega_{2^(f+b+1)}^(r_lo): one table lookup and three squarings
// per round. The last round of a pass (f = 0) needs no multiplications at all.
#pragma once
#include <type_traits>

#include "gl.h"
#include "ntt.h"

namespace ntt16 {

// x * 2^K mod p for a compile-time 0 <= K < 96, x canonical -> canonical
template <int K>
GL_HD uint64_t mul_pow2(uint64_t x) {
  static_assert(K >= 0 && K < 96, "shift out of range");
  if constexpr (K == 0) {
    return x;
  } else if constexpr (K < 64) {
    return gl::reduce128(x << K, x >> (64 - K));
  } else {
    constexpr int J = K - 64;  // x*2^K = (x << J) * 2^64 ;  x << J = y0 + y1*2^64
    uint64_t y0 = J ? (x << J) : x;
    uint64_t y1 = J ? (x >> (64 - J)) : 0;           // < 2^J <= 2^31
    uint64_t t = gl::reduce128(0, y0);               // y0 * 2^64
    return gl::sub(t, y1 << 32);                     // y1 * 2^128 == -y1 * 2^32
  }
}

// (u - v) * omega_16^E  (forward: omega_16 = 2^156, inverse: omega_16^-1 = 2^36), canonical
template <bool INV, int E>
GL_HD uint64_t diff_times_w16(uint64_t u, uint64_t v) {
  constexpr int SH = ((INV ? 36 : 156) * (E & 15)) % 192;
  constexpr bool NEG = SH >= 96;  // 2^96 == -1
  constexpr int K = SH % 96;
  uint64_t d = NEG ? gl::sub(v, u) : gl::sub(u, v);
  return mul_pow2<K>(d);
}

// One DIF stage on the 16 registers at field bit B (pairs m, m + 2^B), with the thread-uniform
// twiddle T (skipped when UNIT_T).
template <bool INV, int B, bool UNIT_T>
GL_HD void stage(uint64_t (&x)[16], uint64_t T) {
#pragma unroll
  for (int m = 0; m < 16; m++) {
    if (m & (1 << B)) continue;
    const int i = m & ((1 << B) - 1);  // position inside the half block
    uint64_t u = x[m], v = x[m + (1 << B)];
    x[m] = gl::add(u, v);
    uint64_t d;
    // omega_{2^(B+1)}^i = omega_16^(i << (3 - B))
    switch (i << (3 - B)) {
      case 0: d = diff_times_w16<INV, 0>(u, v); break;
      case 1: d = diff_times_w16<INV, 1>(u, v); break;
      case 2: d = diff_times_w16<INV, 2>(u, v); break;
      case 3: d = diff_times_w16<INV, 3>(u, v); break;
      case 4: d = diff_times_w16<INV, 4>(u, v); break;
      case 5: d = diff_times_w16<INV, 5>(u, v); break;
      case 6: d = diff_times_w16<INV, 6>(u, v); break;
      default: d = diff_times_w16<INV, 7>(u, v); break;
    }
    x[m + (1 << B)] = UNIT_T ? d : gl::mul(d, T);
  }
}

// NSTAGES DIF stages on the top bits of the 4-bit register field. T3 = omega_{2^(f+4)}^(r_lo).
// The thread-uniform factor of stage B is T_B = T3^(2^(3-B)), and it multiplies ALL the differences of its stage: every later butterfly
// pairs two values that carry the same T_B, so the factor commutes with the rest of the round. The stages therefore run with their
// shift twiddles only, and register m is multiplied ONCE at the end by T3^e(m), e(m) = sum over the round's stages of m_B 2^(3-B) —
// the powers T3^1 .. T3^(2^NSTAGES - 1) walked in order by one lazy product each: 14 + 15 products for four stages where the
// factor-per-stage form took 3 squarings + 32 (NTT16_STAGE_TWIDDLES restores it for the A/B). Same field elements, bit for bit.
template <bool INV, int NSTAGES, bool UNIT_T>
GL_HD void round16(uint64_t (&x)[16], uint64_t T3) {
#if defined(NTT16_STAGE_TWIDDLES)
  uint64_t T2 = UNIT_T ? 1 : gl::sqr(T3);
  stage<INV, 3, UNIT_T>(x, T3);
  if constexpr (NSTAGES >= 2) {
    stage<INV, 2, UNIT_T>(x, T2);
  }
  if constexpr (NSTAGES >= 3) {
    uint64_t T1 = UNIT_T ? 1 : gl::sqr(T2);
    stage<INV, 1, UNIT_T>(x, T1);
    if constexpr (NSTAGES >= 4) {
      uint64_t T0 = UNIT_T ? 1 : gl::sqr(T1);
      stage<INV, 0, UNIT_T>(x, T0);
    }
  }
#else
  stage<INV, 3, true>(x, 1);
  if constexpr (NSTAGES >= 2) stage<INV, 2, true>(x, 1);
  if constexpr (NSTAGES >= 3) stage<INV, 1, true>(x, 1);
  if constexpr (NSTAGES >= 4) stage<INV, 0, true>(x, 1);
  if constexpr (!UNIT_T) {
    uint64_t pw = T3;  // T3^e, lazy from the second on (a product takes any u64)
#pragma unroll
    for (int e = 1; e < (1 << NSTAGES); e++) {
      if (e > 1) {
        uint64_t lo, hi;
        gl::mul_wide(pw, T3, lo, hi);
        pw = gl::reduce128_lazy(lo, hi);
      }
#pragma unroll
      for (int m = 0; m < 16; m++) {
        int em = 0;  // e(m): bit B of m (B = 3 .. 4 - NSTAGES) weighs 2^(3 - B)
        for (int B = 3; B >= 4 - NSTAGES; B--) em += ((m >> B) & 1) << (3 - B);
        if (em == e) x[m] = gl::mul(x[m], pw);
      }
    }
  }
#endif
}

constexpr int LOG_TILE = 12;
constexpr int THREADS = 1 << (LOG_TILE - 4);  // 256

__device__ __forceinline__ int pad(int p) { return p + (p >> 4); }

// Two ways for the tile to cross LDS between rounds.
// HALF = false: all 4096 elements at once (34 KB per workgroup: four workgroups per CU, four waves per SIMD) and one barrier.
// HALF = true: one 32-bit half at a time (17 KB; low words, barrier, high words) — with 85-89 VGPRs that is five waves per
// SIMD, at the price of two more barriers and 32-bit LDS accesses. A pass that has the GPU to itself is bound by that
// occupancy (56-66 % of every wave's cycles are memory waits; with the LDS of a workgroup grown so that only three fit, the
// passes run 20-22 % slower; with HALF the 135 x 2^20 passes go 0.85 -> 0.74 ms and 1.05 -> 0.89 ms). The 2^12-2^15 transforms of
// a proof run beside the kernels of the other contexts, which hide the waits anyway; there the extra barriers only cost
// (three contexts x batch 32: 1 963 -> 1 918 proofs/s with HALF everywhere), so cityprover.hip picks HALF by transform size.
// x[m] sits at tile position Pw + (m << Fw) and is needed at Pr + (m << Fr). P has zeros in bits [F, F + 4) (that is where m
// goes), so pad(P + (m << F)) = pad(P) + pad(m << F): one address register per side and sixteen compile-time offsets.
template <bool HALF, typename T>
__device__ __forceinline__ void exchange(T *tile, uint64_t (&x)[16], int Pw, int Fw, int Pr, int Fr) {
  T *w = tile + pad(Pw), *r = tile + pad(Pr);
  if constexpr (HALF) {
    uint32_t lo[16];
#pragma unroll
    for (int m = 0; m < 16; m++) w[pad(m << Fw)] = gl::lo32(x[m]);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; m++) lo[m] = r[pad(m << Fr)];
    __syncthreads();  // every low word has been read before a high word takes its place
#pragma unroll
    for (int m = 0; m < 16; m++) w[pad(m << Fw)] = gl::hi32(x[m]);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = gl::pack(lo[m], r[pad(m << Fr)]);
  } else {
#pragma unroll
    for (int m = 0; m < 16; m++) w[pad(m << Fw)] = x[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = r[pad(m << Fr)];
  }
  // no barrier at the end: the next writes of a thread go to the positions it has just read (its own set)
}

// global element index of tile position `pos`
template <int L, int C, bool ROWS>
__device__ __forceinline__ size_t gidx(int pos, uint32_t tile_id, int q) {
  if (ROWS) return ((size_t)tile_id << LOG_TILE) + pos;  // q == 0: the tile is contiguous
  uint32_t r = pos >> C, o = (tile_id << C) + (pos & ((1 << C) - 1));
  return ((size_t)(o >> q) << (q + L)) | ((size_t)r << q) | (o & ((1u << q) - 1));
}

// One pass, L stages, tile = 2^L rows x 2^C outer indices (L + C == 12).
// HALF passes ask for five waves per SIMD: 96 VGPRs, which they fit without spilling (85 / 89) now that the LDS and row
// addresses are a base plus compile-time offsets; at six (80 VGPRs) 10-20 registers spill and the passes are slower again
// (0.79 / 1.02 ms per 135-column step against 0.74 / 0.89). The one-round column pass (L = 4: no exchange, 16 + 15 epilogue
// twiddles live at once) would spill a dozen registers at five and keeps the default.
#ifndef NTT16_MIN_WAVES
#define NTT16_MIN_WAVES 5
#endif
template <int L, int C, bool ROWS, bool INV, bool HALF>
__global__ __launch_bounds__(THREADS, (HALF && !(L == 4 && !ROWS)) ? NTT16_MIN_WAVES : 1) void k_dif_pass16(ntt::PassArgs a) {
  static_assert(L + C == LOG_TILE && L >= 4, "tile shape");
  using tile_t = typename std::conditional<HALF, uint32_t, uint64_t>::type;
  __shared__ tile_t tile[(1 << LOG_TILE) + (1 << (LOG_TILE - 4))];
  constexpr int B0 = ((L - 1) % 4) + 1;       // stages of the first round; the rest are full rounds
  constexpr int NROUNDS = 1 + (L - B0) / 4;
  const int t = threadIdx.x;
  const int q = a.q;
  uint64_t *poly = a.data + (size_t)blockIdx.y * a.stride + (size_t)blockIdx.z * a.block_stride;
  const uint64_t *in = (a.first && a.src) ? a.src + (size_t)blockIdx.y * a.src_stride : poly;
  const uint64_t *ptab =
      (a.first && a.ptab) ? a.ptab + ((size_t)ntt::bitrev(blockIdx.z, a.block_bits) << a.log_n) : nullptr;
  uint64_t x[16];

  // ---- round 0: field = r-bits [L-4, L) --------------------------------------------------
  {
    constexpr int f = L - 4;                  // r-bit position of the field
    constexpr int F = ROWS ? f : C + f;       // position in tile-position space
    const int P = ((t >> F) << (F + 4)) | (t & ((1 << F) - 1));
    // ROWS: the tile is contiguous — a wave-uniform base and a 32-bit position instead of a 64-bit index per element
    const uint64_t *in_t = ROWS ? in + ((size_t)blockIdx.x << LOG_TILE) : in;
    const uint64_t *ptab_t = (ROWS && ptab) ? ptab + ((size_t)blockIdx.x << LOG_TILE) : ptab;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      const size_t idx = gidx<L, C, ROWS>(P + (m << F), blockIdx.x, q);
      uint64_t v = ROWS ? in_t[P + (m << F)] : in[idx];
      if (ptab) v = gl::mul(v, ROWS ? ptab_t[P + (m << F)] : ptab[idx]);
      else if (a.coset_pre && a.first) v = gl::mul(v, ntt::pow_table(a.stab, idx));
      x[m] = v;
    }
    if constexpr (f == 0) {
      round16<INV, B0, true>(x, 1);
    } else {
      const int r = ROWS ? (P & ((1 << L) - 1)) : (P >> C);
      const uint32_t r_lo = r & ((1 << f) - 1);
      uint64_t T3 = ntt::pow_table(a.wtab, (uint64_t)r_lo << (a.log_n - f - 4));
      round16<INV, B0, false>(x, T3);
    }
  }
  // ---- middle / last rounds -----------------------------------------------------------------
  constexpr int Flast = ROWS ? 0 : C;
  constexpr int F0 = ROWS ? (L - 4) : C + (L - 4);
  int Plast = ((t >> F0) << (F0 + 4)) | (t & ((1 << F0) - 1));  // round 0's positions; the last round's after the loop
  int Fprev = F0;
#pragma unroll
  for (int k = 1; k < NROUNDS; k++) {
    const int f = L - B0 - 4 * k;             // compile-time after unrolling
    const int F = ROWS ? f : C + f;
    const int P = ((t >> F) << (F + 4)) | (t & ((1 << F) - 1));
    exchange<HALF>(tile, x, Plast, Fprev, P, F);
    if (f == 0) {
      round16<INV, 4, true>(x, 1);
    } else {
      const int r = ROWS ? (P & ((1 << L) - 1)) : (P >> C);
      const uint32_t r_lo = r & ((1 << f) - 1);
      uint64_t T3 = ntt::pow_table(a.wtab, (uint64_t)r_lo << (a.log_n - f - 4));
      round16<INV, 4, false>(x, T3);
    }
    Plast = P;
    Fprev = F;
  }
  // ---- epilogue: inter-pass twiddle, scaling, store -----------------------------------------
  // the thread holds r = r_base + m (m = 0..15) for one (cc): k_loc(m) = rev_L(r_base) + rev4(m) << (L-4)
  {
    const int P = Plast;
    if (!ROWS && q > 0) {
      const uint32_t cc = P & ((1 << C) - 1);
      const uint32_t r_base = P >> C;
      const uint32_t o = (blockIdx.x << C) + cc;
      const uint32_t lo = o & ((1u << q) - 1);
      if (lo) {
        // omega_{2^(q+L)}^(lo * k_loc) = W0 * G^rev4(m),  W0 = w^(lo*rev_L(r_base)),  G = w^(lo << (L-4))
        const int sh = a.log_n - q - L;
        uint64_t W0 = ntt::pow_table(a.wtab, ((uint64_t)lo * ntt::bitrev(r_base, L)) << sh);
        uint64_t G = ntt::pow_table(a.wtab, ((uint64_t)lo << (L - 4)) << sh);
        // walk j = 0..15 and apply to register m = rev4(j)
        uint64_t w = W0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int m = ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3);
          x[m] = gl::mul(x[m], w);
          if (j < 15) {  // the running power stays lazy: a product takes any u64
            uint64_t wl, wh;
            gl::mul_wide(w, G, wl, wh);
            w = gl::reduce128_lazy(wl, wh);
          }
        }
      }
    }
    if (a.last && a.scale) {
#pragma unroll
      for (int m = 0; m < 16; m++) x[m] = gl::mul(x[m], a.scale);
    }
    if (ROWS && a.natural_out) {
      // single-pass transform: position r holds X[rev_L(r)]; un-permute through LDS (HALF: low words, then high words), store coalesced
      __syncthreads();
      if constexpr (HALF) {
        uint32_t lo[16];
#pragma unroll
        for (int m = 0; m < 16; m++) tile[pad((int)ntt::bitrev((uint32_t)(P + m), L))] = gl::lo32(x[m]);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) lo[m] = tile[pad(t + (m << (LOG_TILE - 4)))];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) tile[pad((int)ntt::bitrev((uint32_t)(P + m), L))] = gl::hi32(x[m]);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) {
          int pos = t + (m << (LOG_TILE - 4));
          poly[((size_t)blockIdx.x << LOG_TILE) + pos] = gl::pack(lo[m], tile[pad(pos)]);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 16; m++) tile[pad((int)ntt::bitrev((uint32_t)(P + m), L))] = x[m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) {
          int pos = t + (m << (LOG_TILE - 4));
          poly[((size_t)blockIdx.x << LOG_TILE) + pos] = tile[pad(pos)];
        }
      }
      return;
    }
    if (ROWS && a.staged_store) {
      // The last round leaves every thread 16 CONSECUTIVE elements: stored directly, one instruction of a wave writes 8 bytes
      // into each of 64 different 128-byte lines. Through LDS (same positions, read back lane-contiguous) a wave writes 512
      // contiguous bytes per instruction.
      uint64_t *out_t = poly + ((size_t)blockIdx.x << LOG_TILE);
      __syncthreads();
      if constexpr (HALF) {
        uint32_t lo[16];
        tile_t *w = tile + pad(P);
#pragma unroll
        for (int m = 0; m < 16; m++) w[pad(m)] = gl::lo32(x[m]);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) lo[m] = tile[pad(t + (m << (LOG_TILE - 4)))];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) w[pad(m)] = gl::hi32(x[m]);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) {
          const int pos = t + (m << (LOG_TILE - 4));
          out_t[pos] = gl::pack(lo[m], tile[pad(pos)]);
        }
      } else {
        tile_t *w = tile + pad(P);
#pragma unroll
        for (int m = 0; m < 16; m++) w[pad(m)] = x[m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) {
          const int pos = t + (m << (LOG_TILE - 4));
          out_t[pos] = tile[pad(pos)];
        }
      }
    } else if (ROWS) {
      uint64_t *out_t = poly + ((size_t)blockIdx.x << LOG_TILE) + P;
#pragma unroll
      for (int m = 0; m < 16; m++) out_t[m << Flast] = x[m];
    } else {
#pragma unroll
      for (int m = 0; m < 16; m++) poly[gidx<L, C, ROWS>(P + (m << Flast), blockIdx.x, q)] = x[m];
    }
  }
}

// pre-scale table for the LDE: T[r][j] = (shift * omega_N^r)^j, r < 2^rate_bits, j < n
__global__ void k_fill_prescale(uint64_t *__restrict__ T, int log_n, int rate_bits,
                                const uint64_t *__restrict__ stab, const uint64_t *__restrict__ wtabN) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> (log_n + rate_bits)) return;
  uint64_t j = i & (((size_t)1 << log_n) - 1), r = i >> log_n;
  T[i] = gl::mul(ntt::pow_table(stab, j), ntt::pow_table(wtabN, (r * j) & (((uint64_t)1 << (log_n + rate_bits)) - 1)));
}

}  // namespace ntt16
