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
