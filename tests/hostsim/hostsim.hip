// TEST-ONLY: host-side instantiation of the __host__ __device__ arithmetic in city-rollup_amd/csrc
// so that the exact device formulas (lazy reductions, limb MDS, shift twiddles) can be unit-tested and
// run under sanitizers on the CPU. Never loaded by the product path.
#include "../../city-rollup_amd/csrc/gl.h"
#include "../../city-rollup_amd/csrc/poseidon.h"

extern "C" {
void hs_poseidon_permute(uint64_t *states, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint64_t s[12];
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    poseidon::permute(s);
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
  }
}
void hs_poseidon_permute_textbook(uint64_t *states, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint64_t s[12];
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    poseidon::permute_textbook(s);
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
  }
}
// carry normalisation of the plane-resident partial rounds: returns the limbs; value must be preserved mod p
void hs_renorm(const uint32_t *y, uint32_t *l) { poseidon::renorm(y[0], y[1], y[2], l[0], l[1], l[2]); }
uint64_t hs_mul(uint64_t a, uint64_t b) { return gl::mul(a, b); }
uint64_t hs_mul_lazy(uint64_t a, uint64_t b) { return poseidon::mul_lazy(a, b); }
uint64_t hs_add(uint64_t a, uint64_t b) { return gl::add(a, b); }
uint64_t hs_sub(uint64_t a, uint64_t b) { return gl::sub(a, b); }
void hs_mds_limb(const uint32_t *s, uint32_t *y) {
  uint32_t a[12], b[12];
  for (int i = 0; i < 12; i++) a[i] = s[i];
  poseidon::mds_limb(a, b);
  for (int i = 0; i < 12; i++) y[i] = b[i];
}
}
#include "../../city-rollup_amd/csrc/ntt16.h"
extern "C" {
#define MP(K) case K: return ntt16::mul_pow2<K>(x);
uint64_t hs_mul_pow2(uint64_t x, int k) {
  switch (k) {
    MP(0) MP(1) MP(5) MP(12) MP(24) MP(31) MP(32) MP(33) MP(36) MP(48) MP(60) MP(63) MP(64) MP(65) MP(72) MP(84) MP(95)
    default: return ~0ull;
  }
}
void hs_round16(uint64_t *x, int inverse) {
  uint64_t r[16];
  for (int i = 0; i < 16; i++) r[i] = x[i];
  if (inverse) ntt16::round16<true, 4, true>(r, 1); else ntt16::round16<false, 4, true>(r, 1);
  for (int i = 0; i < 16; i++) x[i] = r[i];
}
}
