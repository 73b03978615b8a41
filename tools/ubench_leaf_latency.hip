// Lone-proof leaf hashing: 2^15 leaves of 135 columns = 512 waves on 1 024 SIMDs, each a chain of 17 permutations — bound by the
// latency of ONE wave's instruction stream, not by issue slots. This times merkle::k_leaf_hash_cols at that size (and at 2^20 for
// reference) so that the same source can be compiled with different scheduling strategies:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icity-rollup_amd/csrc -Itools tools/ubench_leaf_latency.hip -o tools/ubench_leaf_latency
//   ... -mllvm -amdgpu-sched-strategy=max-ilp -o tools/ubench_leaf_latency_ilp
// Prints ms per launch and a checksum of the digests (must agree between builds).
#include <cstdio>
#include <vector>
#include "merkle.h"
#include "ubench_poseidon_quad.h"
#include "poseidon_tables.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  CK(hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RC), POSEIDON_RC, sizeof POSEIDON_RC));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RCD), POSEIDON_RCD, sizeof POSEIDON_RCD));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDK), POSEIDON_DOMD_K, sizeof POSEIDON_DOMD_K));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDLAST), POSEIDON_DOMD_LAST, sizeof POSEIDON_DOMD_LAST));
  const int k = 135;
  for (int log_n : {15, 16, 17, 20}) {
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> h(n * k);
    uint64_t x = 0x243F6A8885A308D3ull;
    for (auto &v : h) { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; v = z % gl::P; }
    uint64_t *d_cols, *d_dig;
    CK(hipMalloc(&d_cols, n * k * 8));
    CK(hipMalloc(&d_dig, n * 32));
    CK(hipMemcpy(d_cols, h.data(), n * k * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((unsigned)((n + merkle::THREADS - 1) / merkle::THREADS), 1), block(merkle::THREADS);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(merkle::k_leaf_hash_cols<false>, grid, block, 0, 0, d_cols, n, k, n, d_dig, (size_t)0, (size_t)0, nullptr, 0, (size_t)0);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(merkle::k_leaf_hash_cols<false>, grid, block, 0, 0, d_cols, n, k, n, d_dig, (size_t)0, (size_t)0, nullptr, 0, (size_t)0);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> dig(n * 4);
    CK(hipMemcpy(dig.data(), d_dig, n * 32, hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (auto v : dig) sum = sum * 0x100000001B3ull + v;
    printf("2^%d leaves x %d columns: %.4f ms per launch  (%.2f us per chained permutation)  digest checksum %016llx\n", log_n, k, ms / reps,
           1e3 * ms / reps / 17.0, (unsigned long long)sum);
    // four lanes per leaf
    CK(hipMemset(d_dig, 0, n * 32));
    const dim3 qgrid((unsigned)((4 * n + 255) / 256), 1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(pquad::k_leaf_hash_cols_quad, qgrid, dim3(256), 0, 0, d_cols, n, k, n, d_dig, (size_t)0, (size_t)0);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(pquad::k_leaf_hash_cols_quad, qgrid, dim3(256), 0, 0, d_cols, n, k, n, d_dig, (size_t)0, (size_t)0);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    hipEventElapsedTime(&ms, e0, e1);
    CK(hipMemcpy(dig.data(), d_dig, n * 32, hipMemcpyDeviceToHost));
    sum = 0;
    for (auto v : dig) sum = sum * 0x100000001B3ull + v;
    printf("   quad: %.4f ms per launch  (%.2f us per chained permutation)  digest checksum %016llx\n", ms / reps, 1e3 * ms / reps / 17.0, (unsigned long long)sum);
    hipFree(d_cols); hipFree(d_dig);
  }
  return 0;
}
