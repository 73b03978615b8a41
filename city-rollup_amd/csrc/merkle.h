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
template <bool SALT>
__global__ __launch_bounds__(THREADS) void k_leaf_hash_cols(const uint64_t *__restrict__ cols,
                                                            size_t n_leaves, int leaf_len,
                                                            size_t col_stride,
                                                            uint64_t *__restrict__ digests,
                                                            size_t tree_cols_stride,
                                                            size_t tree_dig_stride,
                                                            const uint64_t *__restrict__ salt, int n_salt,
                                                            size_t salt_tree_stride) {
  size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= n_leaves) return;
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
