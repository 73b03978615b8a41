// Poseidon sponge / Merkle-tree kernels for gfx950.
//
// Replaces plonky2's `MerkleTree::new` + `PoseidonHash::{hash_no_pad,two_to_one}` as used for
// the three polynomial-batch oracles and the FRI layer trees inside CircuitData::prove
// (hash call sites in-tree: city_crypto/src/hash/traits/hasher.rs:77-159).
//
// HBM layout: leaves are COLUMN-major (poly-major LDE output, bit-reversed index order), so lane i
// reading element j of leaf i is a fully coalesced 512-B wave access and no transpose is ever
// materialised. Digests are stored level by level, 4 x u64 per node (32-B per lane, coalesced).
#pragma once
#include "poseidon.h"

namespace merkle {

constexpr int THREADS = 256;

// leaf digests: digest[i] = hash_or_noop(leaf i)
// blockIdx.y = tree index (batched commitments: tree t reads cols + t*tree_cols_stride)
// SALT (zero-knowledge circuits, FRI `hiding`): the leaf is the leaf_len column values followed by n_salt salt
// elements read from a second column-major array (salt + t*salt_tree_stride, column stride = col_stride).
// The levels a workgroup can finish on its own (round 4): a workgroup's 256 consecutive leaves are a subtree, so the 128 parents
// of the first level, the 64 of the second and the 32 of the third are computed by the same workgroup from digests that never
// leave the CU (LDS) - 2 + 1 + 1 wave-permutations, the first two levels with every lane busy: the same issue slots the
// lane-per-parent kernel k_level spends on them, without three launches (each ending in a tail that cannot fill the chip) and
// without reading the digests back from HBM. Deeper levels would run on a fraction of one wave - a quarter of the lanes, then an
// eighth - at the full latency of a permutation each; those stay with k_level / the cooperative kernels. Every level is still
// written out (Merkle paths need them). `child`: the 256 digests of the level below in LDS ([node][4]); `first_parent`: index of
// this workgroup's first node in the level being computed.
template <int FUSE>
__device__ __forceinline__ void fused_levels(uint64_t *sd, const uint64_t (&own)[4], uint64_t *__restrict__ digests, size_t level0_nodes,
                                             size_t first_node) {
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) sd[4 * t + k] = own[k];
  size_t off = 0, nodes = level0_nodes;
#pragma unroll 1  // one copy of the permutation's code for all levels
  for (int l = 1; l <= FUSE; l++) {
    off += 4 * nodes;  // words of the levels below
    nodes >>= 1;
    const int cnt = THREADS >> l;
    uint64_t s[poseidon::W];
    __syncthreads();
    if (t < cnt) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = sd[8 * t + k];
    }
    __syncthreads();  // every child pair is in registers before a parent overwrites LDS
    if (t < cnt) {
#pragma unroll
      for (int k = 8; k < poseidon::W; k++) s[k] = 0;
      poseidon::permute(s);
      uint64_t *d = digests + off + 4 * ((first_node >> l) + t);
#pragma unroll
      for (int k = 0; k < 4; k++) { d[k] = s[k]; sd[4 * t + k] = s[k]; }
    }
  }
}

// leaf digests: digest[i] = hash_or_noop(leaf i)
// blockIdx.y = tree index (batched commitments: tree t reads cols + t*tree_cols_stride)
// SALT (zero-knowledge circuits, FRI `hiding`): the leaf is the leaf_len column values followed by n_salt salt
// elements read from a second column-major array (salt + t*salt_tree_stride, column stride = col_stride).
// FUSE > 0: the workgroup also computes the next FUSE levels of its subtree (fused_levels); n_leaves must then be a multiple of
// THREADS and the levels must lie below the cap.
template <bool SALT, int FUSE = 0>
__global__ __launch_bounds__(THREADS, 5) void k_leaf_hash_cols(const uint64_t *__restrict__ cols,
                                                            size_t n_leaves, int leaf_len,
                                                            size_t col_stride,
                                                            uint64_t *__restrict__ digests,
                                                            size_t tree_cols_stride,
                                                            size_t tree_dig_stride,
                                                            const uint64_t *__restrict__ salt, int n_salt,
                                                            size_t salt_tree_stride) {
  __shared__ uint64_t sd[FUSE > 0 ? THREADS * 4 : 1];
  size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (FUSE == 0 && i >= n_leaves) return;
  cols += (size_t)blockIdx.y * tree_cols_stride;
  digests += (size_t)blockIdx.y * tree_dig_stride;
  if (SALT) salt += (size_t)blockIdx.y * salt_tree_stride;
  const int total = SALT ? leaf_len + n_salt : leaf_len;
  auto elem = [&](int j) -> uint64_t {
    if (SALT && j >= leaf_len) return salt[(size_t)(j - leaf_len) * col_stride + i];
    return cols[(size_t)j * col_stride + i];
  };
  uint64_t s[poseidon::W];
#pragma unroll
  for (int k = 0; k < poseidon::W; k++) s[k] = 0;
  if (total <= 4) {
    for (int j = 0; j < total; j++) s[j] = elem(j);
  } else {
    int j = 0;
    for (; j + poseidon::RATE <= total; j += poseidon::RATE) {
#pragma unroll
      for (int k = 0; k < poseidon::RATE; k++) s[k] = elem(j + k);
      poseidon::permute(s);
    }
    if (j < total) {  // partial last chunk overwrites only its own lanes (overwrite-mode sponge)
#pragma unroll
      for (int k = 0; k < poseidon::RATE; k++)
        if (j + k < total) s[k] = elem(j + k);
      poseidon::permute(s);
    }
  }
  uint64_t *d = digests + 4 * i;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
  if (FUSE > 0) {
    const uint64_t own[4] = {s[0], s[1], s[2], s[3]};
    fused_levels<FUSE>(sd, own, digests, n_leaves, (size_t)blockIdx.x * THREADS);
  }
}

// same for row-major leaves (host-API convenience path)
__global__ __launch_bounds__(THREADS) void k_leaf_hash_rows(const uint64_t *__restrict__ rows,
                                                            size_t n_leaves, int leaf_len,
                                                            uint64_t *__restrict__ digests,
                                                            int force_hash) {
  size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= n_leaves) return;
  const uint64_t *row = rows + i * (size_t)leaf_len;
  uint64_t s[poseidon::W];
#pragma unroll
  for (int k = 0; k < poseidon::W; k++) s[k] = 0;
  if (leaf_len <= 4 && !force_hash) {
    for (int j = 0; j < leaf_len; j++) s[j] = row[j];
  } else {
    for (int j = 0; j < leaf_len; j += poseidon::RATE) {
#pragma unroll
      for (int k = 0; k < poseidon::RATE; k++)
        if (j + k < leaf_len) s[k] = row[j + k];
      poseidon::permute(s);
    }
  }
  uint64_t *d = digests + 4 * i;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}

// one tree level: parent[i] = two_to_one(child[2i], child[2i+1])
__global__ __launch_bounds__(THREADS) void k_level(const uint64_t *__restrict__ child,
                                                   size_t n_parents,
                                                   uint64_t *__restrict__ parent,
                                                   size_t child_tree_stride,
                                                   size_t parent_tree_stride) {
  size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= n_parents) return;
  child += (size_t)blockIdx.y * child_tree_stride;
  parent += (size_t)blockIdx.y * parent_tree_stride;
  uint64_t s[poseidon::W];
  const uint64_t *c = child + 8 * i;
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = c[k];
#pragma unroll
  for (int k = 8; k < poseidon::W; k++) s[k] = 0;
  poseidon::permute(s);
  uint64_t *d = parent + 4 * i;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}

// 1 + FUSE levels in one launch: the level of n_parents nodes from the digests below it (at D + child_off), then FUSE more by
// the same workgroup (fused_levels). All levels written into the per-tree digest array D (level array layout). n_parents must be
// a multiple of THREADS and every level written must lie below the cap.
template <int FUSE>
__global__ __launch_bounds__(THREADS, 5) void k_level_fused(uint64_t *__restrict__ D, size_t child_off, size_t n_parents, size_t tree_stride) {
  __shared__ uint64_t sd[THREADS * 4];
  D += (size_t)blockIdx.y * tree_stride;
  const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  uint64_t s[poseidon::W];
  const uint64_t *c = D + child_off + 8 * i;
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = c[k];
#pragma unroll
  for (int k = 8; k < poseidon::W; k++) s[k] = 0;
  poseidon::permute(s);
  uint64_t *lvl = D + child_off + 8 * n_parents;  // this level sits right behind its children
  uint64_t *d = lvl + 4 * i;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
  const uint64_t own[4] = {s[0], s[1], s[2], s[3]};
  fused_levels<FUSE>(sd, own, lvl, n_parents, (size_t)blockIdx.x * THREADS);
}

// two_to_one over separate left/right arrays
__global__ __launch_bounds__(THREADS) void k_two_to_one(const uint64_t *__restrict__ left,
                                                        const uint64_t *__restrict__ right,
                                                        size_t count, uint64_t *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= count) return;
  uint64_t s[poseidon::W];
#pragma unroll
  for (int k = 0; k < 4; k++) { s[k] = left[4 * i + k]; s[4 + k] = right[4 * i + k]; }
#pragma unroll
  for (int k = 8; k < poseidon::W; k++) s[k] = 0;
  poseidon::permute(s);
#pragma unroll
  for (int k = 0; k < 4; k++) out[4 * i + k] = s[k];
}

// raw permutation over an array of states (count x 12, row-major)
__global__ __launch_bounds__(THREADS) void k_permute(uint64_t *__restrict__ states, size_t count) {
  size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= count) return;
  uint64_t s[poseidon::W];
  uint64_t *p = states + poseidon::W * i;
#pragma unroll
  for (int k = 0; k < poseidon::W; k++) s[k] = p[k];
  poseidon::permute(s);
#pragma unroll
  for (int k = 0; k < poseidon::W; k++) p[k] = s[k];
}

}  // namespace merkle
