// Lane-cooperative Poseidon permutation for the latency-bound small launches (the upper levels of the Merkle trees).
//
// With one lane per state (poseidon.h) a permutation is a chain of ~13 K instructions: a wave that is alone on its SIMD
// issues one instruction per 4-6 cycles, so a launch with fewer states than the chip has lanes takes ~30-40 us however
// small it is — ten such levels per tree. Here TWELVE lanes share one state, one element each: the twelve S-boxes of a full
// round run side by side, and the MDS row of every element is a 12-term dot product over the state, which the group
// exchanges through LDS (one ds_write_b64 + six ds_read_b128 per lane and round; five states per wave, lanes 60..63 idle).
// ~3-4 K instructions per lane and permutation instead of 13 K: a small level takes ~10 us instead of ~30-40 us, at ~3x the
// total lane-instructions — which is why only launches that cannot fill the chip use it (merkle_levels in cityprover.hip).
// Textbook round structure (constants, S-box, MDS), bit-exact with poseidon::permute.
#pragma once
#include "poseidon.h"

namespace pcoop {

constexpr int GROUP = 12;             // lanes per state
constexpr int STATES_PER_WAVE = 5;    // 60 of 64 lanes
constexpr int WAVES = 4;              // per workgroup (256 threads)
constexpr int STATES_PER_BLOCK = STATES_PER_WAVE * WAVES;
constexpr int RATE_ = 8;              // sponge rate

// lane e of a group computes y_e = sum_j MDS[e][j] s_j with MDS[e][j] = C[(j - e) mod 12] (+ 8 on [0][0])
__device__ __forceinline__ uint32_t mds_coef(int e, int j) {
  constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};  // == POSEIDON_MDS_CIRC (tests/test_tables.py)
  int d = j - e;
  if (d < 0) d += 12;
  return C[d] + ((e == 0 && j == 0) ? 8u : 0u);
}

// One permutation per 12-lane group. x: this lane's element (any u64 representative on entry), returns the canonical
// element. sh: the wave's LDS exchange area, STATES_PER_WAVE * 12 u64, 16-byte aligned. All 64 lanes of the wave must
// call this together (lanes 60..63 and groups without a state carry dummies).
__device__ __forceinline__ uint64_t permute(uint64_t x, int g, int e, bool lane_used, uint64_t *sh, const uint32_t (&coef)[GROUP]) {
  uint64_t *mine = sh + g * GROUP;
  // the round constant of this lane's element is a per-lane (divergent) load: fetched one round ahead, so that its
  // latency (a few hundred cycles from L2) hides under the S-box instead of heading every round
  uint64_t rc_next = poseidon::d_RC[e];
#pragma unroll 1
  for (int r = 0; r < poseidon::ROUNDS; r++) {
    const uint64_t rc_cur = rc_next;
    rc_next = poseidon::d_RC[(r + 1 < poseidon::ROUNDS ? r + 1 : r) * GROUP + e];
    x = poseidon::add_const_lazy(x, rc_cur);
    const bool full = r < poseidon::HALF_FULL || r >= poseidon::HALF_FULL + poseidon::PARTIAL;
    if (full || e == 0) x = poseidon::sbox_lazy(x);
    if (lane_used) mine[e] = x;  // lanes 60..63 shadow group 0 / element 0 and must not write
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // dot product in two 64-bit accumulators (low / high words of the lazy u64 elements): 12 x 41 x 2^32 < 2^41
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < GROUP; j++) {
      const uint64_t s = mine[j];
      lo += (uint64_t)(uint32_t)s * coef[j];
      hi += (uint64_t)(uint32_t)(s >> 32) * coef[j];
    }
    __builtin_amdgcn_wave_barrier();  // every lane has read the state before anyone overwrites it
    // value = lo + hi * 2^32 < 2^74
    const uint64_t l = lo + (hi << 32);
    const uint64_t h = (hi >> 32) + (l < lo);
    x = gl::reduce128_lazy(l, h);
  }
  return gl::canon(x);
}

// one tree level for launches that cannot fill the chip: parent[i] = two_to_one(child[2i], child[2i+1]), 12 lanes per parent.
// grid = (ceil(n_parents / 20), n_trees), block = 256
__global__ __launch_bounds__(256) void k_level_coop(const uint64_t *__restrict__ child, size_t n_parents, uint64_t *__restrict__ parent,
                                                    size_t child_tree_stride, size_t parent_tree_stride) {
  __shared__ __attribute__((aligned(16))) uint64_t sh[WAVES][STATES_PER_WAVE * GROUP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GROUP, e = lane - g * GROUP;
  const bool lane_used = g < STATES_PER_WAVE;
  const size_t i = (size_t)blockIdx.x * STATES_PER_BLOCK + wave * STATES_PER_WAVE + (lane_used ? g : 0);
  const bool active = lane_used && i < n_parents;
  child += (size_t)blockIdx.y * child_tree_stride;
  parent += (size_t)blockIdx.y * parent_tree_stride;
  uint32_t coef[GROUP];
#pragma unroll
  for (int j = 0; j < GROUP; j++) coef[j] = mds_coef(lane_used ? e : 0, j);
  uint64_t x = 0;
  if (active && e < 8) x = child[8 * i + e];
  x = permute(x, lane_used ? g : 0, lane_used ? e : 0, lane_used, sh[wave], coef);
  if (active && e < 4) parent[4 * i + e] = x;
}

// Up to MAX_FUSED consecutive tree levels in ONE launch: a workgroup owns 2^levels consecutive nodes of the child level —
// a whole subtree, so nothing crosses workgroups — and walks it upwards, the digests of the level just computed staying in
// LDS for the next step (every level is also written out: Merkle paths need them). The first step has at most 16 parents
// (of the 20 slots of a workgroup), the later ones 8, 4, 2, 1: what this buys is launches — the 7 to 11 upper levels of a
// commitment become 2 or 3 — on levels that are bound by the latency of one permutation, not by throughput.
// Level l of the tree (l = 0: the child level of this launch, n_child nodes per tree) sits at tree + off_l, off_(l+1) =
// off_l + 4 (n_child >> l); the level with cap_n nodes goes to caps + tree * 4 cap_n instead.
// grid = (n_child >> levels, n_trees), block = 64 x the waves the first step needs (ceil(2^(levels-1) / 5) <= 4)
constexpr int MAX_FUSED = 5;
__global__ __launch_bounds__(256) void k_levels_coop(uint64_t *__restrict__ D, size_t off_child, size_t n_child, int levels, size_t tree_stride,
                                                     uint64_t *__restrict__ caps, size_t cap_n) {
  __shared__ __attribute__((aligned(16))) uint64_t sh[WAVES][STATES_PER_WAVE * GROUP];
  __shared__ uint64_t node[2][4 << (MAX_FUSED - 1)];  // digests of the level being read / written, ping-pong
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GROUP, e = lane - g * GROUP;
  const bool lane_used = g < STATES_PER_WAVE;
  const int j = wave * STATES_PER_WAVE + (lane_used ? g : 0);  // parent slot of this lane's group
  uint64_t *tree = D + (size_t)blockIdx.y * tree_stride;
  uint32_t coef[GROUP];
#pragma unroll
  for (int k = 0; k < GROUP; k++) coef[k] = mds_coef(lane_used ? e : 0, k);
  size_t off = off_child, n = n_child;
  for (int l = 0; l < levels; l++) {
    const int m = 1 << (levels - 1 - l);  // parents this workgroup computes at this step (<= 16)
    const bool active = lane_used && j < m;
    const size_t np = n >> 1, i = (size_t)blockIdx.x * m + j;
    if (wave * STATES_PER_WAVE < m) {  // a wave without a parent at this step sits it out (the exchange inside is per wave)
      uint64_t x = 0;
      if (active && e < 8) x = l == 0 ? tree[off + 8 * i + e] : node[(l - 1) & 1][8 * j + e];
      x = permute(x, lane_used ? g : 0, lane_used ? e : 0, lane_used, sh[wave], coef);
      if (active && e < 4) {
        node[l & 1][4 * j + e] = x;
        if (np == cap_n) caps[(size_t)blockIdx.y * 4 * cap_n + 4 * i + e] = x;
        else tree[off + 4 * n + 4 * i + e] = x;
      }
    }
    off += 4 * n;
    n = np;
    __syncthreads();
  }
}

// FRI layer leaves for launches that cannot fill the chip (same layout as fri::k_leaf_hash_fri): leaf j = the `arity`
// extension values at bit-reversed positions [arity*j, arity*(j+1)), flattened (a0, b0, a1, b1, ...), hashed by the
// overwrite-mode sponge — 2*arity/8 chained permutations, 12 lanes per leaf. vals: [proof][re | im][n_vals].
// grid = (ceil(n_leaves / 20), B), block = 256
__global__ __launch_bounds__(256) void k_leaf_hash_fri_coop(const uint64_t *__restrict__ vals, size_t vals_stride, size_t n_vals, size_t n_leaves,
                                                            int arity, uint64_t *__restrict__ digests, size_t dig_stride) {
  __shared__ __attribute__((aligned(16))) uint64_t sh[WAVES][STATES_PER_WAVE * GROUP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GROUP, e = lane - g * GROUP;
  const bool lane_used = g < STATES_PER_WAVE;
  const size_t j = (size_t)blockIdx.x * STATES_PER_BLOCK + wave * STATES_PER_WAVE + (lane_used ? g : 0);
  const bool active = lane_used && j < n_leaves;
  const uint64_t *re = vals + (size_t)blockIdx.y * vals_stride, *im = re + n_vals;
  uint32_t coef[GROUP];
#pragma unroll
  for (int k = 0; k < GROUP; k++) coef[k] = mds_coef(lane_used ? e : 0, k);
  const int len = 2 * arity;
  // element `idx` of the flattened leaf
  auto elem = [&](int idx) -> uint64_t { return (idx & 1) ? im[arity * j + (idx >> 1)] : re[arity * j + (idx >> 1)]; };
  uint64_t x = 0;
  if (len <= 4) {  // hash_or_noop: a leaf of at most four elements is its own digest
    if (active && e < len) x = elem(e);
  } else {
    uint64_t nxt = (active && e < RATE_ && e < len) ? elem(e) : 0;
    for (int e0 = 0; e0 < len; e0 += RATE_) {
      if (e < RATE_ && e0 + e < len) x = nxt;  // overwrite the rate part with this chunk (a short last chunk keeps the rest)
      const int n0 = e0 + RATE_;
      nxt = (active && e < RATE_ && n0 + e < len) ? elem(n0 + e) : 0;  // fetched under the permutation
      x = permute(x, lane_used ? g : 0, lane_used ? e : 0, lane_used, sh[wave], coef);
    }
  }
  if (active && e < 4) digests[(size_t)blockIdx.y * dig_stride + 4 * j + e] = x;
}

// Column-major leaves (merkle::k_leaf_hash_cols) for commitments of a few thousand leaves: with one lane per leaf such a launch is
// a handful of waves, each a chain of ceil(k / 8) permutations at the 38 us a lone wave needs for one (a 912-column leaf of the
// SHA-256 STARK: 114 of them, 4.3 ms however few rows there are); twelve lanes per leaf run the chain at 13 us a permutation. At
// fourteen times the instructions this pays while the launch is latency-bound: up to 8 192 leaves (cityprover.hip
// merkle_cols_batch). Leaf i of tree t = column values cols[t * tree_cols_stride + j * col_stride + i], j < leaf_len, then n_salt
// salt elements (SALT); leaf_len + n_salt > 4 (shorter leaves are their own digest: the lane-per-leaf kernel does those).
// grid = (ceil(n_leaves / 20), n_trees), block = 256
template <bool SALT>
__global__ __launch_bounds__(256) void k_leaf_hash_cols_coop(const uint64_t *__restrict__ cols, size_t n_leaves, int leaf_len, size_t col_stride,
                                                             uint64_t *__restrict__ digests, size_t tree_cols_stride, size_t tree_dig_stride,
                                                             const uint64_t *__restrict__ salt, int n_salt, size_t salt_tree_stride) {
  __shared__ __attribute__((aligned(16))) uint64_t sh[WAVES][STATES_PER_WAVE * GROUP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / GROUP, e = lane - g * GROUP;
  const bool lane_used = g < STATES_PER_WAVE;
  const size_t i = (size_t)blockIdx.x * STATES_PER_BLOCK + wave * STATES_PER_WAVE + (lane_used ? g : 0);
  const bool active = lane_used && i < n_leaves;
  cols += (size_t)blockIdx.y * tree_cols_stride;
  if (SALT) salt += (size_t)blockIdx.y * salt_tree_stride;
  uint32_t coef[GROUP];
#pragma unroll
  for (int k = 0; k < GROUP; k++) coef[k] = mds_coef(lane_used ? e : 0, k);
  const int len = SALT ? leaf_len + n_salt : leaf_len;
  auto elem = [&](int j) -> uint64_t {
    if (SALT && j >= leaf_len) return salt[(size_t)(j - leaf_len) * col_stride + i];
    return cols[(size_t)j * col_stride + i];
  };
  uint64_t x = 0;
  uint64_t nxt = (active && e < RATE_ && e < len) ? elem(e) : 0;
  for (int e0 = 0; e0 < len; e0 += RATE_) {
    if (e < RATE_ && e0 + e < len) x = nxt;  // overwrite the rate part with this chunk (a short last chunk keeps the rest)
    const int n0 = e0 + RATE_;
    nxt = (active && e < RATE_ && n0 + e < len) ? elem(n0 + e) : 0;  // fetched under the permutation
    x = permute(x, lane_used ? g : 0, lane_used ? e : 0, lane_used, sh[wave], coef);
  }
  if (active && e < 4) digests[(size_t)blockIdx.y * tree_dig_stride + 4 * i + e] = x;
}

}  // namespace pcoop
