// The Fiat-Shamir transcript on the device (plonky2 `Challenger<F, PoseidonHash>`: duplex sponge
// over the Poseidon permutation, overwrite mode, rate 8, challenges popped from the end of the squeezed block).
//
// The transcript is a strictly sequential chain of ~115 permutations per proof and was hashed on the host, from caps and
// openings copied back after every phase: ten stream synchronisations per batch. Here every proof's challenger lives in
// device memory and a phase's observations / challenges are ONE small launch (`k_step`: twelve lanes per challenger on the
// cooperative permutation of poseidon_coop.h, five challengers per wave) reading what the previous kernels left in HBM and
// writing the challenges where the next kernels read them, so that a whole proof is enqueued without a host round trip; the
// small per-proof tables that depend on challenges (alpha powers, opening points, query indices) are built by the kernels
// below. All challengers of a batch advance in lockstep (same shape => same number of elements observed), which is what
// lets five of them share a wave. The host keeps its own HostChallenger for cp_verify and for the C-ABI transcript helpers.
#pragma once
#include "fri.h"
#include "poseidon_coop.h"

namespace tr {

using gl::Ext;

// plonky2 Challenger by value. The squeezed block is sponge_state[0..8) itself (nothing touches the state between a
// permutation and the next one), so the output buffer is just a count.
struct DevCh {
  uint64_t state[12];
  uint64_t in[8];
  int n_in, n_out;
};

struct Seg {  // `count` elements per proof at base + proof*proof_stride, or at table[proof] when table is set
  const uint64_t *base;
  const uint64_t *const *table;
  size_t proof_stride;
  uint32_t count;
};
constexpr int MAX_SEG = 6;
struct StepArgs {
  DevCh *ch;
  unsigned n_proofs;
  int n_seg;
  Seg seg[MAX_SEG];
  int n_chal;          // challenges drawn after the observations
  uint64_t *chal;      // chal[proof*chal_stride + i]
  size_t chal_stride;
};

// observe the segments in order, then draw n_chal challenges. grid = ceil(n_proofs / 5), block = 64 (one wave)
__global__ __launch_bounds__(64) void k_step(StepArgs a) {
  __shared__ __attribute__((aligned(16))) uint64_t sh[pcoop::STATES_PER_WAVE * pcoop::GROUP];
  const int lane = threadIdx.x;
  const int g = lane / pcoop::GROUP, e = lane - g * pcoop::GROUP;
  const bool lane_used = g < pcoop::STATES_PER_WAVE;
  const int gg = lane_used ? g : 0, ee = lane_used ? e : 0;
  const size_t p = (size_t)blockIdx.x * pcoop::STATES_PER_WAVE + gg;
  const bool active = lane_used && p < a.n_proofs;
  uint32_t coef[pcoop::GROUP];
#pragma unroll
  for (int j = 0; j < pcoop::GROUP; j++) coef[j] = pcoop::mds_coef(ee, j);
  DevCh *c = a.ch + (active ? p : 0);
  uint64_t x = active ? c->state[ee] : 0, buf = (active && ee < 8) ? c->in[ee] : 0;
  // lockstep: the counters of this wave's first challenger are those of all five (same shape, same history); only this wave
  // writes them
  int n_in = a.ch[(size_t)blockIdx.x * pcoop::STATES_PER_WAVE].n_in, n_out = a.ch[(size_t)blockIdx.x * pcoop::STATES_PER_WAVE].n_out;
  for (int s = 0; s < a.n_seg; s++) {
    const Seg sg = a.seg[s];
    const uint64_t *ptr = active ? (sg.table ? sg.table[p] : sg.base + p * sg.proof_stride) : nullptr;
    uint32_t off = 0, rem = sg.count;
    while (rem > 0) {
      const int take = (int)rem < 8 - n_in ? (int)rem : 8 - n_in;
      if (active && ee >= n_in && ee < n_in + take) buf = ptr[off + (uint32_t)(ee - n_in)];
      n_in += take;
      off += (uint32_t)take;
      rem -= (uint32_t)take;
      n_out = 0;  // observing invalidates what was squeezed
      if (n_in == 8) {
        if (ee < 8) x = buf;
        x = pcoop::permute(x, gg, ee, lane_used, sh, coef);
        n_in = 0;
        n_out = 8;
      }
    }
  }
  for (int i = 0; i < a.n_chal; i++) {
    if (n_in > 0 || n_out == 0) {
      if (ee < n_in) x = buf;
      x = pcoop::permute(x, gg, ee, lane_used, sh, coef);
      n_in = 0;
      n_out = 8;
    }
    n_out--;
    if (active && ee == n_out) a.chal[p * a.chal_stride + (size_t)i] = x;
  }
  if (active) {
    c->state[ee] = x;
    if (ee < 8) c->in[ee] = buf;
    if (ee == 0) { c->n_in = n_in; c->n_out = n_out; }
  }
}

// quotient: apow[(proof*nc + c)*n_terms + t] = alpha_c^t.   grid = (ceil(n_terms / 64), B*nc)
__global__ void k_alpha_powers(const uint64_t *__restrict__ alphas, int n_terms, uint64_t *__restrict__ apow) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  apow[(size_t)blockIdx.y * n_terms + t] = gl::pow(alphas[blockIdx.y], (uint64_t)t);
}

__device__ __forceinline__ Ext ext_inv(Ext x) {  // 1/(a + bX) = (a - bX) / (a^2 - 7 b^2)
  const uint64_t n = gl::sub(gl::mul(x.a, x.a), gl::mul(7, gl::mul(x.b, x.b)));
  const uint64_t ni = gl::inv(n);
  return Ext{gl::mul(x.a, ni), gl::mul(gl::neg(x.b), ni)};
}

// plonky2's two opening points from zeta: pts = [proof][zeta, g*zeta], then their inverses at + 2*B; flags |= 1 when zeta
// lies in the subgroup (zeta^n == 1: the prover cannot open there).   one thread per proof
__global__ void k_plonk_points(const uint64_t *__restrict__ zeta, unsigned n_proofs, uint64_t g, int degree_bits, Ext *__restrict__ pts,
                               unsigned *__restrict__ flags) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_proofs) return;
  const Ext z{zeta[2 * p], zeta[2 * p + 1]};
  Ext zn = z;
  for (int i = 0; i < degree_bits; i++) zn = gl::ext_mul(zn, zn);
  if (zn.a == 1 && zn.b == 0) atomicOr(flags + p, 1u);
  const Ext zg = gl::ext_scale(z, g);
  pts[2 * p] = z;
  pts[2 * p + 1] = zg;
  pts[2 * (size_t)n_proofs + 2 * p] = ext_inv(z);
  pts[2 * (size_t)n_proofs + 2 * p + 1] = ext_inv(zg);
}

// the final polynomial as the transcript and the proof want it: out[proof][i] = (re[i], im[i]), i < final_len, from the planar
// coefficient arrays [proof][re | im][coef_phys].   grid = (ceil(final_len / 64), B)
__global__ void k_pack_final(const uint64_t *__restrict__ coef, size_t coef_phys, size_t final_len, uint64_t *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= final_len) return;
  const uint64_t *re = coef + (size_t)blockIdx.y * 2 * coef_phys, *im = re + coef_phys;
  uint64_t *o = out + ((size_t)blockIdx.y * final_len + i) * 2;
  o[0] = re[i];
  o[1] = im[i];
}

// proof-of-work search state from the challenger (the witness goes into the next free input slot), and the initial `best`:
// the caller's witness when one is injected, 0 when pow_bits == 0, else "not found".   one thread per proof
__global__ void k_pow_prepare(const DevCh *__restrict__ ch, unsigned n_proofs, int pow_bits, const int *__restrict__ use_pow,
                              const uint64_t *__restrict__ pow_ov, fri::PowState *__restrict__ ps, unsigned long long *__restrict__ best,
                              unsigned long long *__restrict__ done) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_proofs) return;
  const DevCh &c = ch[p];
  for (int k = 0; k < 12; k++) ps[p].s[k] = (k < c.n_in) ? c.in[k] : c.state[k];
  ps[p].pos = c.n_in;
  ps[p].pad = 0;
  const unsigned long long b = (use_pow && use_pow[p]) ? pow_ov[p] : (pow_bits == 0 ? 0ull : ~0ull);
  best[p] = b;
  done[p] = b;
}

// resp: [proof][1 + nq] = the proof-of-work response, then the raw index challenges. idx[proof][i] = challenge % N;
// flags |= 2 when a SEARCHED witness does not give pow_bits leading zeros (cannot happen: the search checked it).
__global__ void k_query_indices(const uint64_t *__restrict__ resp, unsigned n_proofs, int nq, uint64_t N, int pow_bits,
                                const int *__restrict__ use_pow, uint64_t *__restrict__ idx, unsigned *__restrict__ flags) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_proofs * (unsigned)(nq + 1)) return;
  const unsigned p = t / (unsigned)(nq + 1), i = t % (unsigned)(nq + 1);
  const uint64_t v = resp[(size_t)p * (nq + 1) + i];
  if (i == 0) {
    if (!(use_pow && use_pow[p]) && pow_bits > 0 && (v >> (64 - pow_bits)) != 0) atomicOr(flags + p, 2u);
  } else {
    idx[(size_t)p * nq + (i - 1)] = v % N;
  }
}

}  // namespace tr
