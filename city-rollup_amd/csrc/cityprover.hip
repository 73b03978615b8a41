// libcityprover_hip.so — C ABI (include/cityprover.h) over the gfx950 kernels.
// No CPU fallback: every compute entry point needs a live HIP device and fails loudly without one.
#include "core.h"
#include "merkle.h"
#include "ntt.h"
#include "ntt16.h"
#include "poseidon.h"
#include "poseidon_coop.h"

namespace {

// 3-level power table of `base`, cached per ctx
int get_pow_table(cp_ctx *ctx, uint64_t base, const uint64_t **out) {
  auto it = ctx->pow_tables.find(base);
  if (it != ctx->pow_tables.end()) {
    *out = it->second.dev;
    return CP_OK;
  }
  std::vector<uint64_t> h(3 * ntt::PT_SIZE);
  uint64_t b = base;
  for (int lvl = 0; lvl < 3; lvl++) {
    uint64_t acc = 1;
    for (int j = 0; j < ntt::PT_SIZE; j++) {
      h[lvl * ntt::PT_SIZE + j] = acc;
      acc = gl::mul(acc, b);
    }
    b = acc;  // base^(2048^(lvl+1))
  }
  PowTable t;
  HIP_TRY(ctx, dev_malloc(ctx->device, (void **)&t.dev, h.size() * sizeof(uint64_t)));
  HIP_TRY(ctx, hipMemcpyAsync(t.dev, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // h goes out of scope
  ctx->pow_tables[base] = t;
  *out = t.dev;
  return CP_OK;
}

uint64_t root_of_unity(int log_n, bool inverse) {
  return inverse ? GL_ROOTS_INV[log_n] : GL_ROOTS[log_n];
}

// LDE pre-scale table T[r][j] = (shift * omega_N^r)^j  (r < 2^rate_bits, j < n), cached per ctx
int get_prescale_table(cp_ctx *ctx, int log_n, int rate_bits, uint64_t shift, const uint64_t **out) {
  cp_ctx::PreKey key{log_n, rate_bits, shift};
  auto it = ctx->prescale_tables.find(key);
  if (it != ctx->prescale_tables.end()) {
    *out = it->second;
    return CP_OK;
  }
  const uint64_t *stab, *wtabN;
  CP_TRY(get_pow_table(ctx, shift, &stab));
  CP_TRY(get_pow_table(ctx, root_of_unity(log_n + rate_bits, false), &wtabN));
  size_t N = (size_t)1 << (log_n + rate_bits);
  uint64_t *T = nullptr;
  HIP_TRY(ctx, dev_malloc(ctx->device, (void **)&T, N * sizeof(uint64_t)));
  LAUNCH(ctx, "lde_fill_prescale", ntt16::k_fill_prescale, dim3((unsigned)((N + 255) / 256)), dim3(256), T,
         log_n, rate_bits, stab, wtabN);
  ctx->prescale_tables[key] = T;
  *out = T;
  return CP_OK;
}

// Power-on self-test of the arithmetic that is written below the compiler (gl.h: three `asm` blocks with hand-placed wait
// states): a dozen products through every carry / borrow path and one permutation, against the same headers compiled for the
// host (portable C paths). Run once per context; a device that disagrees never gets to prove anything.
__global__ void k_self_test(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, uint64_t *__restrict__ out, int n_mul) {
  const int i = threadIdx.x;
  if (i < n_mul) out[i] = gl::mul(a[i], b[i]);
  if (i == 63) {
    uint64_t s[poseidon::W];
    for (int k = 0; k < poseidon::W; k++) s[k] = a[k];
    poseidon::permute(s);
    for (int k = 0; k < poseidon::W; k++) out[n_mul + k] = s[k];
  }
}
int power_on_self_test(cp_ctx *ctx) {
  constexpr int N = 16;
  const uint64_t M = ~0ull, P = gl::P;
  // canonical and lazy operands; (2^48, 3 2^48), (2^32 u, 2^32 k) and a small low product take the borrow of lo - w3
  const uint64_t a[N] = {0, 1, P - 1, P - 1, M, M, 1ull << 48, 0xFFFFFFFFull << 32, 0x123456789ABCDEF1ull, 0xFFFFFFFF00000000ull,
                         0x8000000080000000ull, 0xFFFFFFFEFFFFFFFFull, 0xdeadbeefull << 32, 0x9E3779B97F4A7C15ull, 3, P + 5};
  const uint64_t b[N] = {M, M, P - 1, 2, M, P, 3ull << 48, 0xFFFFFFFFull << 32, 0xF0E1D2C3B4A59687ull, 0xFFFFFFFF00000001ull,
                         0x8000000080000000ull, 0xFFFFFFFEFFFFFFFFull, 0xfeedfaceull << 32, 0xD1B54A32D192ED03ull, M / 3, M - 7};
  uint64_t want[N + poseidon::W], got[N + poseidon::W];
  for (int i = 0; i < N; i++) want[i] = gl::mul(a[i], b[i]);
  {
    uint64_t s[poseidon::W];
    for (int k = 0; k < poseidon::W; k++) s[k] = a[k];
    poseidon::permute_host(s);  // the integer round structure against the device's double-precision layers
    for (int k = 0; k < poseidon::W; k++) want[N + k] = s[k];
  }
  CP_TRY(ensure_scratch(ctx, (3 * N + poseidon::W) * sizeof(uint64_t)));
  uint64_t *da = (uint64_t *)ctx->scratch, *db = da + N, *dout = db + N;
  CP_TRY(cp_h2d(ctx, da, a, sizeof a));
  CP_TRY(cp_h2d(ctx, db, b, sizeof b));
  LAUNCH(ctx, "self_test", k_self_test, dim3(1), dim3(64), da, db, dout, N);
  CP_TRY(cp_d2h(ctx, got, dout, sizeof got));
  if (hostu::fault_fires(CP_FAULT_SELFTEST)) got[7] ^= 1;  // cp_fault_inject: what a mis-executing device would look like
  for (int i = 0; i < N + poseidon::W; i++)
    if (got[i] != want[i])
      return set_error(ctx, CP_ERR_INTERNAL, "device arithmetic self-test failed (%s %d): this GPU does not execute the field kernels correctly",
                       i < N ? "product" : "permutation word", i < N ? i : i - N);
  return CP_OK;
}

int upload_constants(cp_ctx *ctx) {
  HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RC), POSEIDON_RC, sizeof POSEIDON_RC));
  HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_RCD), POSEIDON_RCD, sizeof POSEIDON_RCD));
  HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDK), POSEIDON_DOMD_K, sizeof POSEIDON_DOMD_K));
  HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_DDLAST), POSEIDON_DOMD_LAST, sizeof POSEIDON_DOMD_LAST));
  return CP_OK;
}


// ---- NTT driver ---------------------------------------------------------------------------

// DIF passes, natural in -> bit-reversed out, in place.
struct DifExtra {
  const uint64_t *src = nullptr;  // first pass reads from here
  size_t src_stride = 0;
  const uint64_t *ptab = nullptr;
  size_t block_stride = 0;
  int n_blocks = 1, block_bits = 0;
  bool natural_out = false;
};

int run_dif(cp_ctx *ctx, uint64_t *data, int log_n, size_t batch, size_t stride, bool inverse,
            uint64_t scale, const uint64_t *stab_pre, const DifExtra &ex = DifExtra()) {
  if (log_n == 0) return CP_OK;
  const uint64_t *wtab;
  CP_TRY(get_pow_table(ctx, root_of_unity(log_n, inverse), &wtab));
  // pass plan: top passes take L = min(remaining, cap) bits; the last pass has q = 0
  int q = log_n;
  bool first = true;
  while (q > 0) {
    int L, c;
    // choose L so that every pass keeps c <= q_after (coalescing rule q >= c) or is the last
    int remaining = q;
    if (remaining <= ntt::LOG_TILE_MAX) {
      L = remaining;  // last pass, q_after = 0
    } else {
      // keep the contiguous last pass at 12 bits when what is left in front of it is >= 4 bits
      // (register radix-16 kernels need L >= 4); otherwise split evenly
      int front = remaining - ntt::LOG_TILE_MAX;
      if (front >= 4 && front <= ntt::LOG_TILE_MAX - 4) L = front;  // c = 12 - L >= 4: 128-B segments
      else {
        int passes = (remaining + 9) / 10;
        L = (remaining + passes - 1) / passes;
      }
    }
    int q_after = q - L;
    c = ntt::LOG_TILE_MAX - L;
    int outer_bits = log_n - L;  // number of outer-index bits
    if (c > outer_bits) c = outer_bits;
    if (q_after > 0 && c > q_after) c = q_after;
    ntt::PassArgs a;
    a.data = data;
    a.stride = stride;
    a.wtab = wtab;
    a.stab = stab_pre;
    a.log_n = log_n;
    a.q = q_after;
    a.L = L;
    a.c = c;
    a.first = first;
    a.last = (q_after == 0);
    a.scale = scale;
    a.coset_pre = stab_pre != nullptr;
    a.coset_post = 0;
    a.src = ex.src;
    a.src_stride = ex.src_stride;
    a.ptab = ex.ptab;
    a.block_stride = ex.block_stride;
    a.block_bits = ex.block_bits;
    a.natural_out = ex.natural_out ? 1 : 0;
    const int staged = (int)CP_KNOB(ctx, "NTT_STAGED_STORE", 1);
    a.staged_store = (staged && q_after == 0 && log_n >= 16) ? 1 : 0;
    dim3 grid((unsigned)(((size_t)1 << outer_bits) >> c), (unsigned)batch, (unsigned)ex.n_blocks);
    bool done = false;
    const bool need16 = ex.src || ex.ptab || ex.n_blocks > 1 || ex.natural_out;
    if (L >= 4 && L + c == ntt16::LOG_TILE && (need16 || !getenv("CITYPROVER_NTT_V1"))) {
      // register radix-16 pass (ntt16.h)
      // a pass of a big transform has the GPU to itself and is occupancy-bound: half-tile exchange, five waves per SIMD; the
      // 2^12-2^15 transforms of a proof run beside other contexts' kernels: whole-tile exchange, fewer barriers (ntt16.h)
      const bool half = log_n >= 16;
#define PASS16_(LL, ROWS_, NAME)                                                                                              \
  if (inverse) {                                                                                                              \
    if (half) LAUNCH(ctx, NAME, (ntt16::k_dif_pass16<LL, 12 - LL, ROWS_, true, true>), grid, dim3(ntt16::THREADS), a);        \
    else LAUNCH(ctx, NAME, (ntt16::k_dif_pass16<LL, 12 - LL, ROWS_, true, false>), grid, dim3(ntt16::THREADS), a);            \
  } else {                                                                                                                    \
    if (half) LAUNCH(ctx, NAME, (ntt16::k_dif_pass16<LL, 12 - LL, ROWS_, false, true>), grid, dim3(ntt16::THREADS), a);       \
    else LAUNCH(ctx, NAME, (ntt16::k_dif_pass16<LL, 12 - LL, ROWS_, false, false>), grid, dim3(ntt16::THREADS), a);           \
  }
#define PASS16(LL)                                 \
  case LL:                                         \
    if (q_after == 0) {                            \
      PASS16_(LL, true, "ntt16_rows")              \
    } else {                                       \
      PASS16_(LL, false, "ntt16_cols")             \
    }                                              \
    done = true;                                   \
    break;
      switch (L) {
        PASS16(4) PASS16(5) PASS16(6) PASS16(7) PASS16(8) PASS16(9) PASS16(10) PASS16(11) PASS16(12)
        default: break;
      }
#undef PASS16
#undef PASS16_
    }
    if (!done && need16) return set_error(ctx, CP_ERR_INTERNAL, "radix-16 pass required but unavailable (L=%d c=%d)", L, c);
    if (!done) {
      if (q_after == 0)
        LAUNCH(ctx, "ntt_dif_pass_rows", ntt::k_dif_pass<true>, grid, dim3(ntt::THREADS), a);
      else
        LAUNCH(ctx, "ntt_dif_pass_cols", ntt::k_dif_pass<false>, grid, dim3(ntt::THREADS), a);
    }
    q = q_after;
    first = false;
  }
  return CP_OK;
}

int bitrev_copy(cp_ctx *ctx, const uint64_t *src, uint64_t *dst, size_t src_stride,
                size_t dst_stride, int log_n, size_t batch, const uint64_t *stab) {
  size_t n = (size_t)1 << log_n;
  dim3 grid(blocks_for(n, 256), (unsigned)batch);
  LAUNCH(ctx, "ntt_bitrev_copy", ntt::k_bitrev_copy, grid, dim3(256), src, dst, src_stride, dst_stride,
         log_n, stab);
  return CP_OK;
}

// u64 words of digest storage per tree: every level below the cap (>= the leaf level itself)
size_t merkle_words_per_tree(size_t n_leaves, int cap_height) {
  size_t cap_n = (size_t)1 << cap_height;
  size_t nodes = n_leaves > cap_n ? 2 * n_leaves - 2 * cap_n : n_leaves;
  return nodes * 4;
}

// How many levels the leaf-hash workgroups (CITYPROVER_MERKLE_FUSE) and the lane-per-parent level kernel
// (CITYPROVER_MERKLE_LEVEL_FUSE) compute on top of their own, 0 .. 3 (merkle.h fused_levels). Measured on the headline step
// (profiles/r04_merkle_fuse_matrix.jsonl, VERDICT r3 "next" #2; ms per step, two rounds on one box): leaf 0 / 1 / 2 / 3 levels =
// 7.98, 7.93 / 7.84, 7.86 / 7.89, 8.02 / 8.01, 8.06; the level kernel fused 1 or 2 deep 7.97-8.02. ONE fused level wins a little
// (its 128 parents keep two of the workgroup's four waves fully busy for one permutation: 0.18 ms of separate launches become
// 0.06-0.13 ms inside the leaf hash); deeper loses: the leaf hash runs at the issue rate of its instructions with exactly five
// waves per SIMD, and every wave parked at the barrier of a fused level is an issue slot nobody uses (three levels: 0.92 M node
// permutations cost 0.4-0.57 ms inside the leaf hash against 0.33 ms as launches of their own). Defaults: 1 and 0.
int merkle_fuse_levels(cp_ctx *ctx) {
  const int v = (int)CP_KNOB(ctx, "MERKLE_FUSE", 1);
  return v < 0 ? 0 : v > 3 ? 3 : v;
}
int merkle_level_fuse_levels(cp_ctx *ctx) {
  const int v = (int)CP_KNOB(ctx, "MERKLE_LEVEL_FUSE", 0);
  return v < 0 ? 0 : v > 3 ? 3 : v;
}
// levels a 256-node workgroup may add on top of a level of `nodes` nodes: below the cap, and only whole workgroups
int fusable_levels(size_t nodes, size_t cap_n, int want) {
  if (nodes % merkle::THREADS) return 0;
  int f = 0;
  while (f < want && (nodes >> (f + 1)) > cap_n) f++;
  return f;
}

// Levels above the leaf digests for `n_trees` trees laid out per tree at D + t*per_tree (level 0 first). `done`: levels the
// leaf-hash launch has computed already (fused_levels).
int merkle_levels(cp_ctx *ctx, uint64_t *D, size_t per_tree, size_t n_leaves, size_t n_trees,
                  int cap_height, uint64_t *caps, int done = 0) {
  size_t cap_n = (size_t)1 << cap_height;
  size_t n = n_leaves, off = 0;
  for (int l = 0; l < done; l++, n >>= 1) off += n * 4;
  while (n > cap_n) {
    size_t np = n / 2;
    const uint64_t *child = D + off;
    uint64_t *parent;
    size_t pstride;
    if (np == cap_n) { parent = caps; pstride = cap_n * 4; }
    else { parent = D + off + n * 4; pstride = per_tree; }
    // a launch that cannot fill the chip is bound by the latency of ONE permutation: twelve lanes per state then
    // (poseidon_coop.h); lane-per-state otherwise. CITYPROVER_COOP_MAX overrides the switch (0 = never) for measurements.
    const size_t coop_max = (size_t)CP_KNOB(ctx, "COOP_MAX", 16384);
    const int fuse_max = (int)CP_KNOB(ctx, "COOP_FUSE", pcoop::MAX_FUSED);
    if (np * n_trees <= coop_max && fuse_max >= 1) {
      // ... and several such levels go into one launch (a workgroup walks a whole subtree): everything up to the cap
      while (n > cap_n) {
        int levels = 0;
        while (levels < fuse_max && levels < pcoop::MAX_FUSED && (n >> levels) > cap_n) levels++;
        const unsigned waves = ((1u << (levels - 1)) + pcoop::STATES_PER_WAVE - 1) / pcoop::STATES_PER_WAVE;  // for the first, widest step
        LAUNCH(ctx, "merkle_levels_coop", pcoop::k_levels_coop, dim3((unsigned)(n >> levels), (unsigned)n_trees), dim3(64 * waves), D, off, n,
               levels, per_tree, caps, cap_n);
        for (int l = 0; l < levels; l++, n >>= 1) off += n * 4;
      }
      break;
    }
    // a level that fills the chip, lane per parent: with up to three more levels by the same workgroups when they are whole and
    // stay below the cap (and those levels would not rather go to the cooperative kernels: they are at least as wide as its switch)
    int fuse = np == cap_n ? 0 : fusable_levels(np, cap_n, merkle_level_fuse_levels(ctx));
    while (fuse > 0 && (np >> fuse) * n_trees < coop_max) fuse--;
    if (fuse > 0) {
      const dim3 g((unsigned)(np / merkle::THREADS), (unsigned)n_trees), b(merkle::THREADS);
      if (fuse == 1) LAUNCH(ctx, "merkle_level_fused", merkle::k_level_fused<1>, g, b, D, off, np, per_tree);
      else if (fuse == 2) LAUNCH(ctx, "merkle_level_fused", merkle::k_level_fused<2>, g, b, D, off, np, per_tree);
      else LAUNCH(ctx, "merkle_level_fused", merkle::k_level_fused<3>, g, b, D, off, np, per_tree);
      for (int l = 0; l <= fuse; l++, n >>= 1) off += n * 4;
      continue;
    }
    if (np * n_trees <= coop_max)
      LAUNCH(ctx, "merkle_level_coop", pcoop::k_level_coop, dim3(blocks_for(np, pcoop::STATES_PER_BLOCK), (unsigned)n_trees), dim3(256), child,
             np, parent, per_tree, pstride);
    else
      LAUNCH(ctx, "merkle_level", merkle::k_level, dim3(blocks_for(np, merkle::THREADS), (unsigned)n_trees),
             dim3(merkle::THREADS), child, np, parent, per_tree, pstride);
    off += n * 4;
    n = np;
  }
  if (n_leaves == cap_n) {
    HIP_TRY(ctx, hipMemcpy2DAsync(caps, cap_n * 32, D, per_tree * 8, cap_n * 32, n_trees,
                                  hipMemcpyDeviceToDevice, ctx->stream));
  }
  return CP_OK;
}

// n_trees Merkle trees over column-major leaves; tree t reads cols + t*tree_cols_stride.
int merkle_cols_batch(cp_ctx *ctx, const uint64_t *cols, size_t n_leaves, size_t leaf_len,
                      size_t col_stride, size_t n_trees, size_t tree_cols_stride, int cap_height,
                      uint64_t *digests, uint64_t *caps, const uint64_t *salt = nullptr, int n_salt = 0,
                      size_t salt_tree_stride = 0) {
  size_t per_tree = merkle_words_per_tree(n_leaves, cap_height);
  uint64_t *D = digests;
  if (!D) {
    CP_TRY(ensure_scratch(ctx, n_trees * per_tree * sizeof(uint64_t)));
    D = (uint64_t *)ctx->scratch;
  }
  const dim3 grid(blocks_for(n_leaves, merkle::THREADS), (unsigned)n_trees), block(merkle::THREADS);
  // a few thousand leaves: the launch is bound by the latency of one lane's chain of permutations — twelve lanes per leaf then
  // (poseidon_coop.h; CITYPROVER_COOP_LEAF_MAX = largest number of leaves over all trees that takes this form, 0 = never)
  const size_t coop_leaf_max = (size_t)CP_KNOB(ctx, "COOP_LEAF_MAX", 8192);
  const bool salted = salt && n_salt > 0;
  if (n_leaves * n_trees <= coop_leaf_max && leaf_len + (salted ? (size_t)n_salt : 0) > 4) {
    const dim3 cgrid(blocks_for(n_leaves, pcoop::STATES_PER_BLOCK), (unsigned)n_trees);
    if (salted)
      LAUNCH(ctx, "leaf_hash_cols_coop", pcoop::k_leaf_hash_cols_coop<true>, cgrid, dim3(256), cols, n_leaves, (int)leaf_len, col_stride, D,
             tree_cols_stride, per_tree, salt, n_salt, salt_tree_stride);
    else
      LAUNCH(ctx, "leaf_hash_cols_coop", pcoop::k_leaf_hash_cols_coop<false>, cgrid, dim3(256), cols, n_leaves, (int)leaf_len, col_stride, D,
             tree_cols_stride, per_tree, (const uint64_t *)nullptr, 0, (size_t)0);
    return merkle_levels(ctx, D, per_tree, n_leaves, n_trees, cap_height, caps);
  }
  // the first levels by the leaf-hash workgroups themselves (merkle.h fused_levels) when the levels they would write are wide
  // enough to belong to the lane-per-parent kernel anyway
  const size_t coop_max = (size_t)CP_KNOB(ctx, "COOP_MAX", 16384);
  int fuse = fusable_levels(n_leaves, (size_t)1 << cap_height, merkle_fuse_levels(ctx));
  while (fuse > 0 && (n_leaves >> fuse) * n_trees < coop_max) fuse--;
#define CP_LEAF_LAUNCH(SALTED, F, SP, NS, SS)                                                                                          \
  LAUNCH(ctx, "leaf_hash_cols", (merkle::k_leaf_hash_cols<SALTED, F>), grid, block, cols, n_leaves, (int)leaf_len, col_stride, D, \
         tree_cols_stride, per_tree, SP, NS, SS)
  if (salted) {
    if (fuse == 3) CP_LEAF_LAUNCH(true, 3, salt, n_salt, salt_tree_stride);
    else if (fuse == 2) CP_LEAF_LAUNCH(true, 2, salt, n_salt, salt_tree_stride);
    else if (fuse == 1) CP_LEAF_LAUNCH(true, 1, salt, n_salt, salt_tree_stride);
    else CP_LEAF_LAUNCH(true, 0, salt, n_salt, salt_tree_stride);
  } else {
    if (fuse == 3) CP_LEAF_LAUNCH(false, 3, (const uint64_t *)nullptr, 0, (size_t)0);
    else if (fuse == 2) CP_LEAF_LAUNCH(false, 2, (const uint64_t *)nullptr, 0, (size_t)0);
    else if (fuse == 1) CP_LEAF_LAUNCH(false, 1, (const uint64_t *)nullptr, 0, (size_t)0);
    else CP_LEAF_LAUNCH(false, 0, (const uint64_t *)nullptr, 0, (size_t)0);
  }
#undef CP_LEAF_LAUNCH
  return merkle_levels(ctx, D, per_tree, n_leaves, n_trees, cap_height, caps, fuse);
}

bool valid_merkle_shape(size_t n_leaves, int cap_height) {
  if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return false;
  if (cap_height < 0 || cap_height > 40) return false;
  return ((size_t)1 << cap_height) <= n_leaves;
}

}  // namespace

namespace { void batches_orphan(cp_ctx *ctx); }  // fri_prove.inc

extern "C" {

int cp_abi_version(void) { return CP_ABI_VERSION; }

int cp_fault_inject(int kind, long after) {
  if (kind != CP_FAULT_THREAD && kind != CP_FAULT_ALLOC && kind != CP_FAULT_SELFTEST && kind != CP_FAULT_DEVMEM) return set_error(nullptr, CP_ERR_INVALID_ARG, "unknown fault kind %d", kind);
  hostu::fault_counter(kind).store(after < 0 ? -1 : after);
  return CP_OK;
}

int cp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *cp_last_error(cp_ctx *ctx) { return ctx ? ctx->error.c_str() : g_tls_error.c_str(); }

cp_ctx *cp_ctx_create(int device) try {
  int n = cp_device_count();
  if (n <= 0) {
    set_error(nullptr, CP_ERR_NO_DEVICE,
              "no HIP device visible: libcityprover_hip has no CPU fallback");
    return nullptr;
  }
  if (device < 0 || device >= n) {
    set_error(nullptr, CP_ERR_INVALID_ARG, "device %d out of range (have %d)", device, n);
    return nullptr;
  }
  cp_ctx *ctx = new (std::nothrow) cp_ctx();
  if (!ctx) {
    set_error(nullptr, CP_ERR_OOM, "out of host memory");
    return nullptr;
  }
  ctx->device = device;
  auto fail = [&](const char *what, hipError_t e) {
    set_error(nullptr, CP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    delete ctx;
    return (cp_ctx *)nullptr;
  };
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail("hipSetDevice", e);
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return fail("hipStreamCreate", e);
  if (upload_constants(ctx) != CP_OK || power_on_self_test(ctx) != CP_OK) {
    std::string msg = ctx->error;
    hipStreamDestroy(ctx->stream);
    delete ctx;
    g_tls_error = msg;
    return nullptr;
  }
  return ctx;
} catch (...) {
  exception_status(nullptr);
  return nullptr;
}

void cp_ctx_destroy(cp_ctx *ctx) {
  if (!ctx) return;
  batches_orphan(ctx);  // batch handles may be destroyed after their context (their buffers are not the context's)
  for (cp_ctx *lane : ctx->lanes) cp_ctx_destroy(lane);
  ctx->lanes.clear();
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  prof_flush(ctx);
  for (auto e : ctx->prof_pool) hipEventDestroy(e);
  for (auto &kv : ctx->pow_tables) hipFree(kv.second.dev);
  for (auto &kv : ctx->prescale_tables) hipFree(kv.second);
  for (auto &kv : ctx->l0_tables) hipFree(kv.second);
  for (auto &kv : ctx->air_sel_tables) hipFree(kv.second);
  for (auto &ch : ctx->arena.chunks) hipFree(ch.first);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->wires_stage) hipFree(ctx->wires_stage);
  for (auto &kv : ctx->fr_twiddles) hipFree(kv.second);
  if (ctx->fr_work) hipFree(ctx->fr_work);
  for (auto &t : ctx->fr_pow)
    if (t.tab) hipFree(t.tab);
  if (ctx->msm_ws) hipFree(ctx->msm_ws);
  if (ctx->pin) hipHostFree(ctx->pin);
  if (ctx->noncanonical_flag) hipHostFree(ctx->noncanonical_flag);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

int cp_ctx_set_lanes(cp_ctx *ctx, int lanes) try {
  CHECK_CTX(ctx);
  if (ctx->parent) return set_error(ctx, CP_ERR_INVALID_ARG, "a lane has no lanes of its own");
  if (lanes < 1 || lanes > 8) return set_error(ctx, CP_ERR_INVALID_ARG, "lanes must be 1..8 (got %d)", lanes);
  while ((int)ctx->lanes.size() < lanes) {
    cp_ctx *lane = cp_ctx_create(ctx->device);
    if (!lane) return set_error(ctx, CP_ERR_HIP, "lane context: %s", cp_last_error(nullptr));
    lane->parent = ctx;
    lane->transcript_mode = ctx->transcript_mode;
    ctx->lanes.push_back(lane);
  }
  ctx->n_lanes = lanes;
  return CP_OK;
} CP_CATCH(ctx)

int cp_ctx_set_option(cp_ctx *ctx, const char *name, long value) try {
  CHECK_CTX(ctx);
  if (!name) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL argument");
  if (ctx->parent) return set_error(ctx, CP_ERR_INVALID_ARG, "a lane takes its options from its parent");
  bool known = false;
  for (const char *const *k = knob_names(); *k; k++) known = known || strcmp(*k, name) == 0;
  if (!known) return set_error(ctx, CP_ERR_INVALID_ARG, "unknown option \"%s\" (see INTEGRATION.md section 5b)", name);
  for (auto &kv : ctx->options)
    if (kv.first == name) { kv.second = value; return CP_OK; }
  ctx->options.emplace_back(name, value);
  return CP_OK;
} CP_CATCH(ctx)

int cp_ctx_set_device_transcript(cp_ctx *ctx, int mode) try {
  CHECK_CTX(ctx);
  if (mode < -1 || mode > 1) return set_error(ctx, CP_ERR_INVALID_ARG, "mode must be -1 (automatic), 0 (host) or 1 (device)");
  ctx->transcript_mode = mode;
  for (cp_ctx *l : ctx->lanes) l->transcript_mode = mode;
  return CP_OK;
} CP_CATCH(ctx)

int cp_dev_alloc(cp_ctx *ctx, size_t bytes, void **out) try {
  CHECK_CTX(ctx);
  if (!out) return set_error(ctx, CP_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return CP_OK;
  HIP_TRY(ctx, dev_malloc(ctx->device, out, bytes));
  return CP_OK;
} CP_CATCH(ctx)
int cp_dev_free(cp_ctx *ctx, void *ptr) try {
  CHECK_CTX(ctx);
  if (!ptr) return CP_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipFree(ptr));
  return CP_OK;
} CP_CATCH(ctx)
int cp_host_alloc(cp_ctx *ctx, size_t bytes, void **out) try {
  CHECK_CTX(ctx);
  if (!out) return set_error(ctx, CP_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return CP_OK;
  if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return set_error(ctx, CP_ERR_OOM, "hipHostMalloc of %zu bytes failed", bytes); }
  return CP_OK;
} CP_CATCH(ctx)
int cp_host_free(cp_ctx *ctx, void *ptr) try {
  CHECK_CTX(ctx);
  if (!ptr) return CP_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipHostFree(ptr));
  return CP_OK;
} CP_CATCH(ctx)
int cp_h2d(cp_ctx *ctx, void *dst, const void *src, size_t bytes) try {
  CHECK_CTX(ctx);
  if (bytes == 0) return CP_OK;
  if (!dst || !src) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CP_OK;
} CP_CATCH(ctx)
}  // extern "C"
// Field elements that arrive from the host must be canonical (< p). Scanning 2^16 rows x 418 columns on one host core took 7 ms
// of a 34 ms STARK proof (tools/stark_upload_probe.py); the same scan on the device, behind the upload, is lost in the noise. The
// kernel sets a page-locked word; the caller reads it after the next synchronisation of the stream it already needs (the cap
// download of a commitment) and only then looks for the offending index on the host, for the error message.
__global__ __launch_bounds__(256) void k_flag_noncanonical(const uint64_t *__restrict__ v, size_t n, uint32_t *__restrict__ flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad |= v[i] >= gl::P;
  if (bad) *flag = 1;
}
// enqueue the scan of v[0, n) (device memory) on the context's stream; the flag accumulates over several calls until it is read
int canonical_check_enqueue(cp_ctx *ctx, const uint64_t *v_dev, size_t n, bool reset) {
  if (!ctx->noncanonical_flag) {
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->noncanonical_flag, 64, hipHostMallocDefault));
    *ctx->noncanonical_flag = 0;
  }
  if (reset) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // no earlier scan may still be writing
    __atomic_store_n(ctx->noncanonical_flag, 0u, __ATOMIC_SEQ_CST);
  }
  if (n == 0) return CP_OK;
  const size_t blocks = std::min<size_t>(blocks_for(n, 256), 4096);
  LAUNCH(ctx, "canonical_check", k_flag_noncanonical, dim3((unsigned)blocks), dim3(256), v_dev, n, ctx->noncanonical_flag);
  return CP_OK;
}
// after the stream has been synchronised: did any scan since the last reset meet an element >= p?
bool canonical_check_failed(cp_ctx *ctx) { return ctx->noncanonical_flag && __atomic_load_n(ctx->noncanonical_flag, __ATOMIC_SEQ_CST) != 0; }
extern "C" {
int cp_d2h(cp_ctx *ctx, void *dst, const void *src, size_t bytes) try {
  CHECK_CTX(ctx);
  if (bytes == 0) return CP_OK;
  if (!dst || !src) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CP_OK;
} CP_CATCH(ctx)
int cp_d2d(cp_ctx *ctx, void *dst, const void *src, size_t bytes) try {
  CHECK_CTX(ctx);
  if (bytes == 0) return CP_OK;
  if (!dst || !src) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return CP_OK;
} CP_CATCH(ctx)
int cp_sync(cp_ctx *ctx) try {
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CP_OK;
} CP_CATCH(ctx)
int cp_event_create(cp_ctx *ctx, void **event_out) try {
  CHECK_CTX(ctx);
  if (!event_out) return set_error(ctx, CP_ERR_INVALID_ARG, "event_out is NULL");
  hipEvent_t ev;
  HIP_TRY(ctx, hipEventCreate(&ev));
  *event_out = (void *)ev;
  return CP_OK;
} CP_CATCH(ctx)
int cp_event_destroy(cp_ctx *ctx, void *event) try {
  CHECK_CTX(ctx);
  if (event) HIP_TRY(ctx, hipEventDestroy((hipEvent_t)event));
  return CP_OK;
} CP_CATCH(ctx)
int cp_event_record(cp_ctx *ctx, void *event) try {
  CHECK_CTX(ctx);
  if (!event) return set_error(ctx, CP_ERR_INVALID_ARG, "event is NULL");
  HIP_TRY(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
  return CP_OK;
} CP_CATCH(ctx)
int cp_event_elapsed_ms(cp_ctx *ctx, void *start, void *stop, float *ms_out) try {
  CHECK_CTX(ctx);
  if (!start || !stop || !ms_out) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL argument");
  HIP_TRY(ctx, hipEventSynchronize((hipEvent_t)stop));
  HIP_TRY(ctx, hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop));
  return CP_OK;
} CP_CATCH(ctx)

int cp_profile_begin(cp_ctx *ctx) try {
  CHECK_CTX(ctx);
  prof_flush(ctx);
  ctx->prof_acc.clear();
  ctx->profiling = true;
  return CP_OK;
} CP_CATCH(ctx)
int cp_profile_end(cp_ctx *ctx, char *json_out, size_t cap) try {
  CHECK_CTX(ctx);
  ctx->profiling = false;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  prof_flush(ctx);
  std::string js = "{";
  bool first = true;
  for (auto &kv : ctx->prof_acc) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s\"%s\": {\"launches\": %llu, \"total_ms\": %.6f}", first ? "" : ", ",
             kv.first.c_str(), (unsigned long long)kv.second.first, kv.second.second);
    js += buf;
    first = false;
  }
  js += "}";
  if (json_out && cap) {
    if (js.size() + 1 > cap) return set_error(ctx, CP_ERR_INVALID_ARG, "profile buffer too small (%zu needed)", js.size() + 1);
    memcpy(json_out, js.c_str(), js.size() + 1);
  }
  return CP_OK;
} CP_CATCH(ctx)

// ---- NTT ------------------------------------------------------------------------------------

int cp_ntt_dev(cp_ctx *ctx, uint64_t *data, int log_n, size_t batch, size_t stride, unsigned flags,
               uint64_t coset_shift) try {
  CHECK_CTX(ctx);
  if (batch == 0) return CP_OK;
  if (!data) return set_error(ctx, CP_ERR_INVALID_ARG, "data is NULL");
  if (log_n < 0 || log_n > 32) return set_error(ctx, CP_ERR_INVALID_ARG, "log_n %d out of range [0,32]", log_n);
  size_t n = (size_t)1 << log_n;
  if (stride < n) return set_error(ctx, CP_ERR_INVALID_ARG, "stride %zu < n %zu", stride, n);
  if (batch > 65535) return set_error(ctx, CP_ERR_INVALID_ARG, "batch %zu > 65535", batch);
  if (flags & ~(CP_NTT_INVERSE | CP_NTT_BITREV_OUT | CP_NTT_COSET | CP_NTT_BITREV_IN))
    return set_error(ctx, CP_ERR_INVALID_ARG, "unknown flags 0x%x", flags);
  const bool inverse = flags & CP_NTT_INVERSE, coset = flags & CP_NTT_COSET;
  if (coset && (coset_shift == 0 || coset_shift >= gl::P))
    return set_error(ctx, CP_ERR_INVALID_ARG, "coset shift must be a non-zero canonical element");
  if (log_n == 0) return CP_OK;

  const uint64_t *stab = nullptr;
  if (coset) CP_TRY(get_pow_table(ctx, inverse ? gl::inv(coset_shift) : coset_shift, &stab));

  const bool need_tmp = (flags & CP_NTT_BITREV_IN) || !(flags & CP_NTT_BITREV_OUT) || (coset && inverse);
  uint64_t *tmp = nullptr;
  if (need_tmp) {
    CP_TRY(ensure_scratch(ctx, batch * n * sizeof(uint64_t)));
    tmp = (uint64_t *)ctx->scratch;
  }
  if (flags & CP_NTT_BITREV_IN) {  // un-permute the input first (v1: explicit pass)
    CP_TRY(bitrev_copy(ctx, data, tmp, stride, n, log_n, batch, nullptr));
    // one strided device-to-device copy for the whole batch (a single contiguous copy when stride == n)
    HIP_TRY(ctx, hipMemcpy2DAsync(data, stride * sizeof(uint64_t), tmp, n * sizeof(uint64_t), n * sizeof(uint64_t), batch,
                                  hipMemcpyDeviceToDevice, ctx->stream));
  }
  uint64_t scale = 0;
  if (inverse) scale = gl::inv((uint64_t)n % gl::P);
  // a 4096-point transform is one workgroup-resident pass: natural order comes out of its LDS epilogue
  if (log_n == ntt16::LOG_TILE && !(flags & CP_NTT_BITREV_OUT) && !(coset && inverse)) {
    DifExtra ex;
    ex.natural_out = true;
    return run_dif(ctx, data, log_n, batch, stride, inverse, scale, (coset && !inverse) ? stab : nullptr, ex);
  }
  CP_TRY(run_dif(ctx, data, log_n, batch, stride, inverse, scale, (coset && !inverse) ? stab : nullptr));
  if (!(flags & CP_NTT_BITREV_OUT)) {
    CP_TRY(bitrev_copy(ctx, data, tmp, stride, n, log_n, batch, (coset && inverse) ? stab : nullptr));
    // one strided device-to-device copy for the whole batch (a single contiguous copy when stride == n)
    HIP_TRY(ctx, hipMemcpy2DAsync(data, stride * sizeof(uint64_t), tmp, n * sizeof(uint64_t), n * sizeof(uint64_t), batch,
                                  hipMemcpyDeviceToDevice, ctx->stream));
  } else if (coset && inverse) {
    return set_error(ctx, CP_ERR_UNSUPPORTED, "inverse coset NTT with bit-reversed output is not supported");
  }
  return CP_OK;
} CP_CATCH(ctx)

int cp_ntt(cp_ctx *ctx, uint64_t *data_host, int log_n, size_t batch, unsigned flags,
           uint64_t coset_shift) try {
  CHECK_CTX(ctx);
  if (batch == 0) return CP_OK;
  if (!data_host) return set_error(ctx, CP_ERR_INVALID_ARG, "data is NULL");
  if (log_n < 0 || log_n > 32) return set_error(ctx, CP_ERR_INVALID_ARG, "log_n %d out of range [0,32]", log_n);
  size_t n = (size_t)1 << log_n, bytes = batch * n * sizeof(uint64_t);
  uint64_t *d = nullptr;
  HIP_TRY(ctx, dev_malloc(ctx->device, (void **)&d, bytes));
  int rc = cp_h2d(ctx, d, data_host, bytes);
  if (rc == CP_OK) rc = cp_ntt_dev(ctx, d, log_n, batch, n, flags, coset_shift);
  if (rc == CP_OK) rc = cp_d2h(ctx, data_host, d, bytes);
  hipStreamSynchronize(ctx->stream);
  hipFree(d);
  return rc;
} CP_CATCH(ctx)

int cp_lde_dev(cp_ctx *ctx, const uint64_t *coeffs, size_t in_stride, int log_n, int rate_bits,
               size_t batch, uint64_t coset_shift, unsigned flags, uint64_t *out, size_t out_stride) try {
  CHECK_CTX(ctx);
  if (batch == 0) return CP_OK;
  if (!coeffs || !out) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  if (log_n < 0 || rate_bits < 0 || log_n + rate_bits > 32)
    return set_error(ctx, CP_ERR_INVALID_ARG, "log_n %d + rate_bits %d out of range", log_n, rate_bits);
  size_t n = (size_t)1 << log_n, N = n << rate_bits;
  if (in_stride < n || out_stride < N) return set_error(ctx, CP_ERR_INVALID_ARG, "stride too small");
  if (batch > 65535) return set_error(ctx, CP_ERR_INVALID_ARG, "batch %zu > 65535", batch);
  if (flags & ~CP_NTT_BITREV_OUT) return set_error(ctx, CP_ERR_INVALID_ARG, "unsupported flags 0x%x", flags);
  if (coset_shift == 0 || coset_shift >= gl::P)
    return set_error(ctx, CP_ERR_INVALID_ARG, "coset shift must be a non-zero canonical element");
  // Product shape (n = 4096): the zero-padded size-N transform is 2^rate_bits independent coset NTTs of
  // size n (the first rate_bits stages only copy), each one workgroup-resident pass reading the
  // coefficients once and writing its block of the bit-reversed LDE directly.
  if (log_n == ntt16::LOG_TILE && (flags & CP_NTT_BITREV_OUT) && rate_bits <= 6) {
    const uint64_t *ptab;
    CP_TRY(get_prescale_table(ctx, log_n, rate_bits, coset_shift, &ptab));
    DifExtra ex;
    ex.src = coeffs;
    ex.src_stride = in_stride;
    ex.ptab = ptab;
    ex.block_stride = n;
    ex.n_blocks = 1 << rate_bits;
    ex.block_bits = rate_bits;
    return run_dif(ctx, out, log_n, batch, out_stride, false, 0, nullptr, ex);
  }
  dim3 grid(blocks_for(N, 256), (unsigned)batch);
  LAUNCH(ctx, "lde_pad_copy", ntt::k_pad_copy, grid, dim3(256), coeffs, out, in_stride, out_stride, n, N);
  return cp_ntt_dev(ctx, out, log_n + rate_bits, batch, out_stride,
                    (flags & CP_NTT_BITREV_OUT) | CP_NTT_COSET, coset_shift);
} CP_CATCH(ctx)

// ---- field self-test --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_field_mul(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b,
                                                   uint64_t *__restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = gl::mul(a[i], b[i]);
}

int cp_field_mul(cp_ctx *ctx, const uint64_t *a_host, const uint64_t *b_host, uint64_t *out_host, size_t count) try {
  CHECK_CTX(ctx);
  if (count == 0) return CP_OK;
  if (!a_host || !b_host || !out_host) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  if (count > ((size_t)1 << 28)) return set_error(ctx, CP_ERR_INVALID_ARG, "count too large");
  const size_t bytes = count * sizeof(uint64_t);
  CP_TRY(ensure_scratch(ctx, 3 * bytes));
  uint64_t *a = (uint64_t *)ctx->scratch, *b = a + count, *o = b + count;
  CP_TRY(cp_h2d(ctx, a, a_host, bytes));
  CP_TRY(cp_h2d(ctx, b, b_host, bytes));
  LAUNCH(ctx, "field_mul", k_field_mul, dim3(blocks_for(count, 256)), dim3(256), a, b, o, count);
  return cp_d2h(ctx, out_host, o, bytes);
} CP_CATCH(ctx)

// ---- Poseidon -------------------------------------------------------------------------------

int cp_poseidon_permute_dev(cp_ctx *ctx, uint64_t *states, size_t count) try {
  CHECK_CTX(ctx);
  if (count == 0) return CP_OK;
  if (!states) return set_error(ctx, CP_ERR_INVALID_ARG, "states is NULL");
  LAUNCH(ctx, "poseidon_permute", merkle::k_permute, dim3(blocks_for(count, merkle::THREADS)),
         dim3(merkle::THREADS), states, count);
  return CP_OK;
} CP_CATCH(ctx)

int cp_poseidon_permute(cp_ctx *ctx, uint64_t *states_host, size_t count) try {
  CHECK_CTX(ctx);
  if (count == 0) return CP_OK;
  if (!states_host) return set_error(ctx, CP_ERR_INVALID_ARG, "states is NULL");
  size_t bytes = count * 12 * sizeof(uint64_t);
  CP_TRY(ensure_scratch(ctx, bytes));
  CP_TRY(cp_h2d(ctx, ctx->scratch, states_host, bytes));
  CP_TRY(cp_poseidon_permute_dev(ctx, (uint64_t *)ctx->scratch, count));
  return cp_d2h(ctx, states_host, ctx->scratch, bytes);
} CP_CATCH(ctx)

int cp_hash_no_pad(cp_ctx *ctx, const uint64_t *in_host, size_t count, size_t len,
                   uint64_t *digests_host) try {
  CHECK_CTX(ctx);
  if (count == 0) return CP_OK;
  if (!digests_host || (!in_host && len)) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  if (len > (1u << 30)) return set_error(ctx, CP_ERR_INVALID_ARG, "len too large");
  size_t in_bytes = count * len * sizeof(uint64_t), out_bytes = count * 32;
  size_t in_al = (in_bytes + 255) & ~(size_t)255;
  CP_TRY(ensure_scratch(ctx, in_al + out_bytes));
  uint64_t *din = (uint64_t *)ctx->scratch, *dout = (uint64_t *)((char *)ctx->scratch + in_al);
  if (in_bytes) CP_TRY(cp_h2d(ctx, din, in_host, in_bytes));
  LAUNCH(ctx, "leaf_hash_rows", merkle::k_leaf_hash_rows, dim3(blocks_for(count, merkle::THREADS)),
         dim3(merkle::THREADS), din, count, (int)len, dout, 1);
  return cp_d2h(ctx, digests_host, dout, out_bytes);
} CP_CATCH(ctx)

int cp_two_to_one(cp_ctx *ctx, const uint64_t *left_host, const uint64_t *right_host, size_t count,
                  uint64_t *out_host) try {
  CHECK_CTX(ctx);
  if (count == 0) return CP_OK;
  if (!left_host || !right_host || !out_host) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  size_t b = count * 32;
  CP_TRY(ensure_scratch(ctx, 3 * b));
  uint64_t *l = (uint64_t *)ctx->scratch, *r = l + count * 4, *o = r + count * 4;
  CP_TRY(cp_h2d(ctx, l, left_host, b));
  CP_TRY(cp_h2d(ctx, r, right_host, b));
  LAUNCH(ctx, "two_to_one", merkle::k_two_to_one, dim3(blocks_for(count, merkle::THREADS)),
         dim3(merkle::THREADS), l, r, count, o);
  return cp_d2h(ctx, out_host, o, b);
} CP_CATCH(ctx)

// ---- Merkle ---------------------------------------------------------------------------------

int cp_merkle_cols_dev(cp_ctx *ctx, const uint64_t *cols, size_t n_leaves, size_t leaf_len,
                       size_t col_stride, int cap_height, uint64_t *digests_dev, uint64_t *cap_dev) try {
  CHECK_CTX(ctx);
  if (!cols || !cap_dev) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  if (!valid_merkle_shape(n_leaves, cap_height))
    return set_error(ctx, CP_ERR_INVALID_ARG, "n_leaves %zu must be a power of two >= 2^cap_height (%d)",
                     n_leaves, cap_height);
  if (leaf_len == 0 || leaf_len > (1u << 20)) return set_error(ctx, CP_ERR_INVALID_ARG, "leaf_len %zu out of range", leaf_len);
  if (col_stride < n_leaves) return set_error(ctx, CP_ERR_INVALID_ARG, "col_stride < n_leaves");
  size_t cap_n = (size_t)1 << cap_height;
  return merkle_cols_batch(ctx, cols, n_leaves, leaf_len, col_stride, 1, 0, cap_height,
                           (digests_dev && n_leaves > cap_n) ? digests_dev : nullptr, cap_dev);
} CP_CATCH(ctx)

int cp_merkle_cap(cp_ctx *ctx, const uint64_t *rows_host, size_t n_leaves, size_t leaf_len,
                  int cap_height, uint64_t *cap_host) try {
  CHECK_CTX(ctx);
  if (!rows_host || !cap_host) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  if (!valid_merkle_shape(n_leaves, cap_height))
    return set_error(ctx, CP_ERR_INVALID_ARG, "n_leaves %zu must be a power of two >= 2^cap_height (%d)",
                     n_leaves, cap_height);
  if (leaf_len == 0 || leaf_len > (1u << 20)) return set_error(ctx, CP_ERR_INVALID_ARG, "leaf_len %zu out of range", leaf_len);
  size_t cap_n = (size_t)1 << cap_height;
  size_t rows_bytes = n_leaves * leaf_len * sizeof(uint64_t);
  uint64_t *rows = nullptr, *cap = nullptr;
  HIP_TRY(ctx, dev_malloc(ctx->device, (void **)&rows, rows_bytes));
  hipError_t e = dev_malloc(ctx->device, (void **)&cap, cap_n * 32);
  if (e != hipSuccess) { hipFree(rows); return set_error(ctx, CP_ERR_OOM, "hipMalloc: %s", hipGetErrorString(e)); }
  int rc = cp_h2d(ctx, rows, rows_host, rows_bytes);
  size_t per_tree = merkle_words_per_tree(n_leaves, cap_height);
  if (rc == CP_OK) rc = ensure_scratch(ctx, per_tree * sizeof(uint64_t));
  if (rc == CP_OK) {
    uint64_t *level0 = (uint64_t *)ctx->scratch;
    hipLaunchKernelGGL(merkle::k_leaf_hash_rows, dim3(blocks_for(n_leaves, merkle::THREADS)),
                       dim3(merkle::THREADS), 0, ctx->stream, rows, n_leaves, (int)leaf_len, level0, 0);
    if (hipGetLastError() != hipSuccess) rc = set_error(ctx, CP_ERR_HIP, "leaf hash launch failed");
    if (rc == CP_OK) rc = merkle_levels(ctx, level0, per_tree, n_leaves, 1, cap_height, cap);
  }
  if (rc == CP_OK) rc = cp_d2h(ctx, cap_host, cap, cap_n * 32);
  hipStreamSynchronize(ctx->stream);
  hipFree(rows);
  hipFree(cap);
  return rc;
} CP_CATCH(ctx)

// ---- commit ---------------------------------------------------------------------------------

int cp_commit_batch_dev(cp_ctx *ctx, const uint64_t *values, size_t k, size_t n_trees, int log_n,
                        int rate_bits, int cap_height, uint64_t *coeffs_dev, uint64_t *lde_dev,
                        uint64_t *digests_dev, uint64_t *caps_dev) try {
  CHECK_CTX(ctx);
  if (!values || !lde_dev || !caps_dev) return set_error(ctx, CP_ERR_INVALID_ARG, "NULL pointer");
  if (k == 0 || n_trees == 0 || k * n_trees > 65535 || n_trees > 65535)
    return set_error(ctx, CP_ERR_INVALID_ARG, "k %zu x n_trees %zu out of range", k, n_trees);
  if (log_n < 0 || rate_bits < 0 || log_n + rate_bits > 32)
    return set_error(ctx, CP_ERR_INVALID_ARG, "log_n/rate_bits out of range");
  size_t n = (size_t)1 << log_n, N = n << rate_bits, polys = k * n_trees;
  if (!valid_merkle_shape(N, cap_height)) return set_error(ctx, CP_ERR_INVALID_ARG, "cap_height %d too large", cap_height);
  uint64_t *coeffs = coeffs_dev;
  uint64_t *own = nullptr;
  if (!coeffs) {
    HIP_TRY(ctx, dev_malloc(ctx->device, (void **)&own, polys * n * sizeof(uint64_t)));
    coeffs = own;
  }
  int rc;
  if (log_n == ntt16::LOG_TILE) {
    // one launch: values -> natural-order coefficients (read from `values`, written to `coeffs`)
    DifExtra ex;
    ex.src = values;
    ex.src_stride = n;
    ex.natural_out = true;
    rc = run_dif(ctx, coeffs, log_n, polys, n, true, gl::inv((uint64_t)n), nullptr, ex);
  } else {
    rc = cp_d2d(ctx, coeffs, values, polys * n * sizeof(uint64_t));
    if (rc == CP_OK) rc = cp_ntt_dev(ctx, coeffs, log_n, polys, n, CP_NTT_INVERSE, 0);
  }
  if (rc == CP_OK)
    rc = cp_lde_dev(ctx, coeffs, n, log_n, rate_bits, polys, 7, CP_NTT_BITREV_OUT, lde_dev, N);
  if (rc == CP_OK)
    rc = merkle_cols_batch(ctx, lde_dev, N, k, N, n_trees, k * N, cap_height, digests_dev, caps_dev);
  if (own) {
    hipStreamSynchronize(ctx->stream);
    hipFree(own);
  }
  return rc;
} CP_CATCH(ctx)

int cp_commit_dev(cp_ctx *ctx, const uint64_t *values, size_t k, int log_n, int rate_bits,
                  int cap_height, uint64_t *coeffs_dev, uint64_t *lde_dev, uint64_t *digests_dev,
                  uint64_t *cap_dev) try {
  return cp_commit_batch_dev(ctx, values, k, 1, log_n, rate_bits, cap_height, coeffs_dev, lde_dev,
                             digests_dev, cap_dev);
} CP_CATCH(ctx)

}  // extern "C"

// ---- circuits + proof tail (transcript, openings, FRI, proof bytes) -------------------------------
#include "fri.h"
#include "transcript.h"
#include "zs.h"
#include "quotient.h"
#include "prover_tail.inc"
#include "batcher.inc"
#include "verify.inc"
#include "fri_prove.inc"
#include "circuit_file.inc"
#include "air.h"
#include "ext3.h"
#include "stark.inc"
// (the BLS12-381 / Groth16 side is its own translation unit: bls.hip)
