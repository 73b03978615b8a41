/*
 * cityprover — C ABI of the MI355X-native Plonky2 prover backend for city-rollup.
 *
 * This is the drop-in boundary: the entry points a
 * Rust `extern "C"` shim inside a patched `plonky2` crate binds so that
 *   plonky2::plonk::circuit_data::CircuitData::prove / ::verify
 * (called at e.g. city_common_circuit/src/proof_minifier/pm_core.rs:143-154,
 *  city_common_circuit/src/treeprover/aggregation/state_transition/mod.rs:260-304,
 *  city_rollup_circuit/src/block_circuits/ops/l2_transfer/circuit.rs:209-235)
 * and the primitives underneath it run on the GPU. Plain pointers and sizes only.
 *
 * Groups of entry points: context / memory / events / profiling; primitives (NTT, LDE, Poseidon, Merkle,
 * commitments); circuits (cp_circuit_load, cp_circuit_set_gates); proving (cp_prove, cp_prove_batch{,_host},
 * cp_prove_batch_zk_host, cp_prove_tail*, cp_zs_partial_products_dev) and cp_verify; and, for the Groth16 wrap
 * (SURVEY.md §8(a) A12), the BLS12-381 G1 / G2 multi-scalar multiplications (cp_msm_bls12381_*) and the scalar-field NTT
 * (cp_ntt_bls12381_fr*); and the generic polynomial-commitment seams under all of it — cp_batch_* (`PolynomialBatch`) and
 * cp_fri_prove / cp_fri_verify (`prove_openings` / `verify_fri_proof` over arbitrary oracles and opening batches).
 *
 * Conventions
 *  - every plonky2 field element is a canonical Goldilocks u64 (little-endian on the wire), p = 2^64-2^32+1;
 *    BLS12-381 coordinates and scalars are little-endian u64 limbs of the canonical value (6 resp. 4 limbs)
 *  - all functions return 0 on success, a negative cp_status otherwise, and NEVER abort or throw:
 *    every entry point is a function-try-block that maps std::bad_alloc to CP_ERR_OOM and anything else to
 *    CP_ERR_INTERNAL, and the internal worker threads (transcript hashing, cp_ctx_set_lanes, the Groth16 side chain)
 *    fall back to the calling thread when a thread cannot be created (tests: cp_fault_inject);
 *    the message is available from cp_last_error() (the Rust shim turns it into anyhow::bail!,
 *    matching the `anyhow::Result` convention of city_rollup_circuit/src/worker/traits.rs:16-43)
 *  - a cp_ctx is bound to one device and is thread-compatible (one caller at a time); multi-GPU =
 *    one ctx per device, one host thread / process each (city_rollup_core_worker/src/lib.rs:131-145
 *    scales the same way: one worker process per consumer)
 *  - "_dev" entry points take device pointers (operands already resident in HBM, obtained from
 *    cp_dev_alloc or from any other HIP allocation on the same device, e.g. a torch tensor);
 *    the plain entry points take host pointers and stage through HBM.
 */
#ifndef CITYPROVER_H
#define CITYPROVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CP_ABI_VERSION 4

typedef enum cp_status {
  CP_OK = 0,
  CP_ERR_INVALID_ARG = -1,
  CP_ERR_NO_DEVICE = -2,
  CP_ERR_HIP = -3,
  CP_ERR_OOM = -4,
  CP_ERR_UNSUPPORTED = -5,
  CP_ERR_INTERNAL = -6,
  CP_ERR_VERIFY = -7 /* cp_verify: the proof is well-formed input but does not verify */
} cp_status;

typedef struct cp_ctx cp_ctx;

/* ---- library / context ------------------------------------------------------------- */
int cp_abi_version(void);
/* number of visible HIP devices; 0 when there is none (never an error) */
int cp_device_count(void);
/* Create a context on `device`. Returns NULL on failure (see cp_last_error(NULL)). Fails loudly
 * when no GPU is present: there is no CPU fallback in this library. */
cp_ctx *cp_ctx_create(int device);
void cp_ctx_destroy(cp_ctx *ctx);
/* Internal pipelining for single-threaded callers (the reference's worker loop is one thread per process,
 * city_rollup_core_worker/src/lib.rs:131-145): with lanes > 1, cp_prove_batch_host / cp_prove_batch_zk_host split a
 * batch of >= 4 * lanes proofs among `lanes` internal contexts (own stream, workspace and staging; the circuits stay
 * shared) driven by short-lived threads, so the host phases of one part overlap the kernels of the others. Same proof
 * bytes; ~20 % more proofs/s than lanes = 1 for one caller at B = 32. Callers that already run
 * several contexts from several threads should leave it at 1 (the default). lanes: 1..8. */
int cp_ctx_set_lanes(cp_ctx *ctx, int lanes);
/* Where the Fiat-Shamir transcripts of a proving call are hashed (plonky2's Challenger: a sequential chain of ~115 Poseidon
 * permutations per proof). 1: on the device — a whole batch is enqueued without a host round trip, no host thread hashes;
 * 0: on the host, one synchronisation per phase — faster for a LONE proof (a CPU core out-runs one wave on a sequential chain);
 * -1 (default): by batch size (device from EIGHT proofs per batch up). Same proof bytes in every
 * mode. Lanes inherit the setting. */
int cp_ctx_set_device_transcript(cp_ctx *ctx, int mode);
/* One of the library's measurement switches for THIS context (and its lanes) instead of process-wide: `name` is the part after
 * CITYPROVER_ of the environment variables INTEGRATION.md section 5b lists (DEVICE_TRANSCRIPT, QUOT_ALL_MAX, QUOT_FLIP, QUOT_GROUP,
 * QUOT_TILE, COOP_MAX, COOP_FUSE, COOP_LEAF_MAX, COOP_FRI_MAX, MERKLE_FUSE, MERKLE_LEVEL_FUSE, NTT_STAGED_STORE, AIR_TARGET_WAVES,
 * AIR_LDS_SLOTS, AIR_POINTS_PER_LANE). Order of precedence: this call, the environment variable (read once per process), the
 * built-in default. The switches select among forms that give the same bytes; two contexts of one process may differ. */
int cp_ctx_set_option(cp_ctx *ctx, const char *name, long value);
/* last error message of `ctx`, or of the calling thread when ctx == NULL. Never NULL. */
const char *cp_last_error(cp_ctx *ctx);
/* Fault injection for tests of the error paths (the reference has none; a backend that lives inside
 * the worker process must turn every failure into a status): the `after`-th (0 = the next) event of `kind` in this
 * process fails once - CP_FAULT_THREAD: creation of an internal worker thread (the library must carry on on the calling
 * thread: same results); CP_FAULT_ALLOC: a host allocation checkpoint inside a proving / verifying call throws
 * std::bad_alloc (the call must return CP_ERR_OOM and leave the context usable); CP_FAULT_SELFTEST: the power-on self-test
 * of the device arithmetic that cp_ctx_create runs (a dozen field products through every carry path and one Poseidon
 * permutation, against the host's portable code) sees a wrong answer - cp_ctx_create must return NULL with the reason;
 * CP_FAULT_DEVMEM: a device allocation of the library is refused by "the runtime" as out of memory (the library must give
 * the batch-handle pool back and try again: the call succeeds when the pool held anything, else CP_ERR_OOM).
 * after < 0 disarms. */
enum { CP_FAULT_THREAD = 0, CP_FAULT_ALLOC = 1, CP_FAULT_SELFTEST = 2, CP_FAULT_DEVMEM = 3 };
int cp_fault_inject(int kind, long after);

/* ---- device memory & stream ---------------------------------------------------------- */
int cp_dev_alloc(cp_ctx *ctx, size_t bytes, void **out);
int cp_dev_free(cp_ctx *ctx, void *ptr);
/* Page-locked host memory for buffers the caller hands to cp_prove / cp_prove_batch_host (the witness
 * generator's wire matrices): copies from it are DMA transfers that overlap other contexts' work; copies
 * from ordinary pageable memory are staged by the runtime and serialise across contexts. */
int cp_host_alloc(cp_ctx *ctx, size_t bytes, void **out);
int cp_host_free(cp_ctx *ctx, void *ptr);
int cp_h2d(cp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int cp_d2h(cp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int cp_d2d(cp_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
int cp_sync(cp_ctx *ctx);
/* HIP events on the context's stream (the stream every kernel of this ctx is launched on) */
int cp_event_create(cp_ctx *ctx, void **event_out);
int cp_event_destroy(cp_ctx *ctx, void *event);
int cp_event_record(cp_ctx *ctx, void *event);
int cp_event_elapsed_ms(cp_ctx *ctx, void *start, void *stop, float *ms_out); /* syncs on `stop` */

/* Per-kernel timing with HIP events on the context stream: between cp_profile_begin and
 * cp_profile_end every kernel launch of this ctx is bracketed by an event pair. cp_profile_end
 * syncs and writes a JSON object {"<kernel>": {"launches": n, "total_ms": t}, ...} (NUL-terminated)
 * into json_out. (Counterpart of the reference's per-job TraceTimer,
 * city_common/src/logging/trace_timer.rs:32-70, at kernel granularity.) */
int cp_profile_begin(cp_ctx *ctx);
int cp_profile_end(cp_ctx *ctx, char *json_out, size_t cap);

/* ---- NTT over Goldilocks -------------------------------------------------------------
 * Replaces plonky2_field's `fft_with_options` / `ifft_with_options` / `PolynomialCoeffs::
 * coset_fft_with_options` as used by `PolynomialBatch::from_values` inside CircuitData::prove
 * (SURVEY.md §8(a) A3). omega_n = 7^((p-1)/n).
 *
 * data: `batch` polynomials of n = 2^log_n elements, polynomial b at data + b*stride (stride >= n).
 * flags: */
#define CP_NTT_INVERSE 1u        /* inverse transform (includes the 1/n scaling) */
#define CP_NTT_BITREV_OUT 2u     /* leave the output in bit-reversed index order (Merkle leaf order) */
#define CP_NTT_COSET 4u          /* forward only: evaluate on shift*<omega> (input coefficient i is  \
                                    multiplied by shift^i first); inverse only: divide coefficient i \
                                    by shift^i afterwards */
#define CP_NTT_BITREV_IN 8u      /* the input is in bit-reversed index order */
int cp_ntt_dev(cp_ctx *ctx, uint64_t *data_dev, int log_n, size_t batch, size_t stride,
               unsigned flags, uint64_t coset_shift);
int cp_ntt(cp_ctx *ctx, uint64_t *data_host, int log_n, size_t batch, unsigned flags,
           uint64_t coset_shift);

/* Low-degree extension: coefficients (n = 2^log_n each, polynomial b at coeffs + b*in_stride) ->
 * evaluations on coset_shift*<omega_{n<<rate_bits}>, polynomial b at out + b*out_stride
 * (out_stride >= n<<rate_bits). Output order: natural, or bit-reversed with CP_NTT_BITREV_OUT.
 * (plonky2 `PolynomialCoeffs::lde` + `coset_fft_with_options`) */
int cp_lde_dev(cp_ctx *ctx, const uint64_t *coeffs_dev, size_t in_stride, int log_n, int rate_bits,
               size_t batch, uint64_t coset_shift, unsigned flags, uint64_t *out_dev,
               size_t out_stride);

/* ---- Goldilocks field, element-wise (diagnostic) ------------------------------------------
 * out[i] = a[i] * b[i] mod p with the device's multiplication (csrc/gl.h `mul`: hand-scheduled carry handling), canonical
 * out; a, b: ANY u64 (the kernels carry values lazily), host pointers. No call site in the reference — this is the
 * built-in self-test hook for the one arithmetic primitive every kernel shares (plonky2 `GoldilocksField::mul`). */
int cp_field_mul(cp_ctx *ctx, const uint64_t *a_host, const uint64_t *b_host, uint64_t *out_host, size_t count);

/* ---- Poseidon-Goldilocks -------------------------------------------------------------
 * Replaces plonky2 `PoseidonPermutation::permute`, `PoseidonHash::{hash_no_pad, two_to_one}`
 * (reference call sites city_crypto/src/hash/traits/hasher.rs:77-159).
 * states: count x 12 u64, permuted in place. */
int cp_poseidon_permute_dev(cp_ctx *ctx, uint64_t *states_dev, size_t count);
int cp_poseidon_permute(cp_ctx *ctx, uint64_t *states_host, size_t count);
/* hash_no_pad of `count` inputs of `len` felts each (row-major), digests out: count x 4 */
int cp_hash_no_pad(cp_ctx *ctx, const uint64_t *in_host, size_t count, size_t len,
                   uint64_t *digests_host);
/* two_to_one over `count` pairs: left/right count x 4 -> out count x 4 */
int cp_two_to_one(cp_ctx *ctx, const uint64_t *left_host, const uint64_t *right_host, size_t count,
                  uint64_t *out_host);

/* ---- Merkle tree with cap ------------------------------------------------------------
 * Replaces plonky2 `MerkleTree::new(leaves, cap_height)`: leaf digest =
 * hash_or_noop(row) (rows of <= 4 felts are zero-padded, not hashed), node = two_to_one,
 * the tree stops at the 2^cap_height-entry cap.
 *
 * Leaves are given COLUMN-MAJOR (the HBM layout of a polynomial batch): element j of leaf i is
 * cols[j*col_stride + i]; leaf order is the caller's (the prover passes bit-reversed LDE columns).
 * digests_dev (nullable): all levels below the cap, level 0 (n_leaves digests) first, then
 * n_leaves/2, ... ; total (2*n_leaves - 2^(cap_height+1)) x 4 u64. cap_dev: 2^cap_height x 4. */
int cp_merkle_cols_dev(cp_ctx *ctx, const uint64_t *cols_dev, size_t n_leaves, size_t leaf_len,
                       size_t col_stride, int cap_height, uint64_t *digests_dev,
                       uint64_t *cap_dev);
/* host convenience, ROW-MAJOR leaves (n_leaves x leaf_len) as plonky2's API takes them */
int cp_merkle_cap(cp_ctx *ctx, const uint64_t *rows_host, size_t n_leaves, size_t leaf_len,
                  int cap_height, uint64_t *cap_host);

/* ---- PolynomialBatch::from_values (commit) --------------------------------------------
 * values: k polynomials of n = 2^log_n evaluations over <omega_n> (natural order), poly-major.
 * Produces coefficient form (k x n), the rate-2^rate_bits coset LDE on 7*<omega_{n<<rate_bits}>
 * in bit-reversed order (k x N, poly-major == column-major leaves), the Merkle digests and the cap.
 * Any of coeffs_dev / digests_dev may be NULL. (SURVEY.md §3.3 step 3, §8(a) A3+A4.) */
int cp_commit_dev(cp_ctx *ctx, const uint64_t *values_dev, size_t k, int log_n, int rate_bits,
                  int cap_height, uint64_t *coeffs_dev, uint64_t *lde_dev, uint64_t *digests_dev,
                  uint64_t *cap_dev);

/* Batched form: n_trees independent commitments of k polynomials each (e.g. the same oracle of
 * n_trees proofs in flight). values: n_trees*k polynomials, tree t owns polynomials [t*k, (t+1)*k).
 * lde_dev: n_trees*k x N; digests_dev (nullable): per tree (2N - 2^(cap_height+1)) x 4 (or N x 4 when
 * N == 2^cap_height); caps_dev: n_trees x 2^cap_height x 4. */
int cp_commit_batch_dev(cp_ctx *ctx, const uint64_t *values_dev, size_t k, size_t n_trees, int log_n,
                        int rate_bits, int cap_height, uint64_t *coeffs_dev, uint64_t *lde_dev,
                        uint64_t *digests_dev, uint64_t *caps_dev);

/* ---- circuits and the proof tail ----------------------------------------------------------
 * cp_shape = the scalars of plonky2 `CommonCircuitData` the prover needs (the same fields the
 * reference serialises at city_common_circuit/src/verify_template/ser_data.rs:55-123).
 * standard_recursion_config as used by every worker circuit: degree_bits 12, num_constants 5
 * (2 gate constants + 3 selectors), num_routed_wires 80, num_wires 135, num_challenges 2,
 * num_partial_products 9, quotient_degree_factor 8, rate_bits 3, cap_height 4, pow_bits 16,
 * num_query_rounds 28, arity_bits {4,4}. */
typedef struct cp_shape {
  int degree_bits, num_constants, num_routed_wires, num_wires, num_challenges, num_partial_products,
      quotient_degree_factor;
  int rate_bits, cap_height, pow_bits, num_query_rounds;
  int n_arity;
  int arity_bits[8];
  int zero_knowledge; /* CircuitConfig::zero_knowledge (standard_recursion_zk_config, the user-side signature
                         circuits: city_common_circuit/src/circuits/zk_signature/inner.rs:50,116-117): FRI `hiding`, i.e.
                         the Merkle leaves of the wires / Z / quotient oracles end with CP_SALT_SIZE random elements */
  int num_public_inputs; /* CommonCircuitData::num_public_inputs. The prove entry points and cp_verify refuse any other
                            count (plonky2 `validate_proof_with_pis_shape`): hash_no_pad is an unpadded sponge, so without
                            this check a proof would also verify under a zero-extended public-input vector. Proofs of
                            circuits that differ only in this field may share a batch. */
} cp_shape;
#define CP_SALT_SIZE 4

typedef struct cp_circuit cp_circuit;

/* Load a circuit: commits its constants+sigmas polynomials ((num_constants+num_routed_wires) x n
 * VALUES over <omega_n>, host pointer) once and keeps coefficients / LDE / Merkle tree resident in
 * HBM for the circuit's lifetime — the counterpart of building `CircuitData` once in
 * `CRWorkerToolboxRootCircuits::new` (city_rollup_circuit/src/worker/toolbox/root.rs:75-139).
 * Returns NULL on failure (cp_last_error(ctx)). */
cp_circuit *cp_circuit_load(cp_ctx *ctx, const cp_shape *shape, const uint64_t circuit_digest[4],
                            const uint64_t *cs_values_host,
                            const uint64_t *k_is_host /* num_routed_wires coset shifts
                                                         (CommonCircuitData::k_is); NULL = 7^i */);

/* A7 — permutation argument: Z and partial-product polynomials (plonky2
 * `wires_permutation_partial_products_and_zs`) for n_proofs proofs of one shape.
 * wires_values_dev: [proof][num_wires][n]; betas/gammas: [proof][num_challenges] (host);
 * out_dev: [proof][num_challenges*(1+num_partial_products)][n] values over <omega_n>, in the order the
 * batch is committed (Z polynomials first, then the partial products of challenge 0, 1, ...). */
int cp_zs_partial_products_dev(cp_ctx *ctx, size_t n_proofs, cp_circuit *const *circuits,
                               const uint64_t *wires_values_dev, const uint64_t *betas_host,
                               const uint64_t *gammas_host, uint64_t *out_dev);
void cp_circuit_destroy(cp_circuit *circuit);

/* ---- circuit files: the CircuitData import bridge ------------------------------------
 * A ".cpcirc" file is one built circuit as flat data: cp_shape, circuit_digest, the gate list with its selector
 * groups, k_is, the positions of the public inputs in the wire matrix, and the constants + sigmas polynomials (values
 * over <omega_n>, or coefficients as `PolynomialBatch::polynomials` holds them). The Rust side writes it once per
 * circuit after `CircuitBuilder::build` (rust/plonky2-hwa-patch: `CircuitData::dump_cityprover`; build sites
 * city_common_circuit/src/proof_minifier/pm_core.rs:96-105, city_rollup_circuit/src/worker/toolbox/root.rs:75-139);
 * the byte layout is in csrc/circuit_file.inc and INTEGRATION.md section 4. Files carry a version and an FNV-1a
 * checksum; a file of another version, a truncated or a corrupt one is refused. */
/* load: cp_circuit_load + cp_circuit_set_gates (+ the public-input targets) from a file. NULL on failure. */
cp_circuit *cp_circuit_load_file(cp_ctx *ctx, const char *path);
/* save: writes `circuit` (values form, with k_is and, when set, the public-input targets) */
int cp_circuit_save_file(cp_circuit *circuit, const char *path);
/* Parses and checks a file WITHOUT a GPU (structure, sizes, canonical elements, checksum); any out pointer may be NULL.
 * flags_out: bit 0 = polynomials in coefficient form, bit 1 = k_is present, bit 2 = public-input targets present. */
int cp_circuit_file_info(const char *path, cp_shape *shape_out, uint64_t digest_out[4], size_t *n_gates_out,
                         int *num_selectors_out, unsigned *flags_out);
/* `ProverOnlyCircuitData::public_inputs` as (row, wire) pairs: where public input j sits in the wire matrix. */
int cp_circuit_set_public_input_targets(cp_circuit *circuit, const uint32_t *row_wire_pairs, size_t n_targets);
/* public_inputs_out[j] = wires[wire_j][row_j] (what `CircuitData::prove` reads back from the witness) */
int cp_circuit_public_inputs_from_wires(cp_circuit *circuit, const uint64_t *wires_values_host, uint64_t *public_inputs_out);
/* the shape and digest a circuit was loaded with */
int cp_circuit_shape(cp_circuit *circuit, cp_shape *shape_out, uint64_t digest_out[4]);
/* 1 when proofs of the two circuits may share one cp_prove_batch* call (same shape up to the number of public inputs, same
 * gate list and selector grouping: "one batch = one shape"), 0 when not, negative on a NULL argument. What a batching
 * worker asks before it merges ready jobs of different circuits (cp_batcher asks the same internally). */
int cp_circuits_batch_compatible(const cp_circuit *a, const cp_circuit *b);
/* the circuit's constants_sigmas_cap (2^cap_height x 4), i.e. VerifierOnlyCircuitData */
int cp_circuit_cs_cap(cp_circuit *circuit, uint64_t *cap_out_host);

/* Everything of `CircuitData::prove` after the polynomials are known (SURVEY.md §3.3 steps 3-4, 6,
 * 8-11): commits wires / Z+partial-products / quotient chunks, runs the Fiat-Shamir transcript,
 * opens every polynomial at zeta (Z polynomials also at g*zeta), builds the FRI opening proof
 * (batch polynomial, commit phase, proof of work with the SMALLEST valid witness unless
 * use_pow_override, query rounds) and serialises `ProofWithPublicInputs` in bincode (malloc'd;
 * release with cp_free).
 *   wires_values_dev : num_wires x n evaluations (device)
 *   zs_pp_values_dev : num_challenges*(1+num_partial_products) x n evaluations, Z polynomials first
 *   quotient_coeffs_dev : num_challenges*quotient_degree_factor x n COEFFICIENTS (degree-n chunks)
 * The Z / quotient computation itself (gate constraints) is upstream of this
 * entry point. */
int cp_prove_tail(cp_circuit *circuit, const uint64_t *public_inputs_host, size_t n_public_inputs,
                  const uint64_t *wires_values_dev, const uint64_t *zs_pp_values_dev,
                  const uint64_t *quotient_coeffs_dev, int use_pow_override, uint64_t pow_override,
                  uint8_t **proof_out, size_t *proof_len);
/* Batched form — the throughput path. Proves n_proofs independent proofs of ONE shape in one call
 * (circuits may differ, e.g. the 20 op-leaf jobs of a block): every device step is a single launch
 * over the batch, every Fiat-Shamir step advances n_proofs transcripts. Polynomial inputs are laid
 * out [proof][poly][n] contiguously. use_pow_override / pow_override may be NULL (= none). On success
 * proofs_out[i] / proof_lens[i] hold malloc'd bincode proofs (cp_free each). All circuits must belong
 * to `ctx`. */
int cp_prove_tail_batch(cp_ctx *ctx, size_t n_proofs, cp_circuit *const *circuits,
                        const uint64_t *const *public_inputs_host, const size_t *n_public_inputs,
                        const uint64_t *wires_values_dev, const uint64_t *zs_pp_values_dev,
                        const uint64_t *quotient_coeffs_dev, const int *use_pow_override,
                        const uint64_t *pow_override, uint8_t **proofs_out, size_t *proof_lens);
/* ---- gates and the whole proof ---------------------------------------------------------------
 * A8 needs the circuit's gate set: CommonCircuitData::gates (in order: a gate's index is its
 * position) with its selector group (SelectorsInfo: selector_indices[gate], groups[selector]).
 * Supported: the whole city-common gate set (city_common_circuit/src/builder/pad_circuit.rs:31-55)
 * and all eight in-tree u32 gates (city_common_circuit/src/u32/gates/). The first num_selectors
 * "constants" columns are the selector polynomials; a gate's own constants follow them. */
enum {
  CP_GATE_NOOP = 0, CP_GATE_CONSTANT = 1, CP_GATE_PUBLIC_INPUT = 2, CP_GATE_ARITHMETIC = 3, CP_GATE_POSEIDON = 4,
  /* in-tree city-rollup gates (city_common_circuit/src/u32/gates/): */
  CP_GATE_COMPARISON = 5,      /* comparison.rs:96-200      param = num_bits, param2 = num_chunks */
  CP_GATE_U32_ARITHMETIC = 6,  /* arithmetic_u32.rs:90-150  param = num_ops */
  CP_GATE_U32_RANGE_CHECK = 7, /* range_check_u32.rs:57-80  param = num_input_limbs */
  CP_GATE_U32_ADD_MANY = 8,    /* add_many_u32.rs:93-140    param = num_ops, param2 = num_addends */
  CP_GATE_U32_SUBTRACTION = 9, /* subtraction_u32.rs:89-125 param = num_ops */
  CP_GATE_U32_INTERLEAVE = 10, /* interleave_u32.rs:90-128  param = num_ops */
  CP_GATE_UNINTERLEAVE_TO_U32 = 11, /* uninterleave_to_u32.rs:82-130 param = num_ops */
  CP_GATE_UNINTERLEAVE_TO_B32 = 12, /* uninterleave_to_b32.rs:82-131 param = num_ops */
  /* remaining upstream plonky2 0.2.2 gates of the city-common set: */
  CP_GATE_ARITHMETIC_EXT = 13, /* ArithmeticExtensionGate   param = num_ops */
  CP_GATE_MUL_EXT = 14,        /* MulExtensionGate          param = num_ops */
  CP_GATE_BASE_SUM = 15,       /* BaseSumGate<B>            param = num_limbs, param2 = B */
  CP_GATE_RANDOM_ACCESS = 16,  /* RandomAccessGate          param = bits (<= 4), param2 = num_copies, param3 = num_extra_constants */
  CP_GATE_REDUCING = 17,       /* ReducingGate              param = num_coeffs */
  CP_GATE_REDUCING_EXT = 18,   /* ReducingExtensionGate     param = num_coeffs */
  CP_GATE_POSEIDON_MDS = 19,   /* PoseidonMdsGate */
  CP_GATE_COSET_INTERPOLATION = 20, /* CosetInterpolationGate param = subgroup_bits (<= 5), param2 = degree */
  /* not in the city-common set; completes plonky2 0.2.2's standard gates except the lookup pair: */
  CP_GATE_EXPONENTIATION = 21  /* ExponentiationGate        param = num_power_bits */
};
typedef struct cp_gate {
  int type;           /* CP_GATE_* */
  int selector_index; /* selector polynomial of this gate's group */
  int group_start, group_end; /* gate indices [start, end) sharing that selector */
  int param;          /* Constant: num_consts, Arithmetic / U32Arithmetic: num_ops, Comparison: num_bits, ... */
  int param2, param3; /* second / third constructor parameter where the gate has one, else 0 */
} cp_gate;
int cp_circuit_set_gates(cp_circuit *circuit, const cp_gate *gates, size_t n_gates, int num_selectors);

/* The whole of `CircuitData::prove` after witness generation (SURVEY.md §8(a) A1 minus A2):
 * wires -> Z / partial products (A7) -> quotient chunks (A8) -> openings, FRI, proof bytes, for
 * n_proofs proofs of one shape / gate set. wires_values_dev: [proof][num_wires][n] evaluations.
 * Witness generation (A2) stays on the CPU upstream of this call. */
int cp_prove_batch(cp_ctx *ctx, size_t n_proofs, cp_circuit *const *circuits,
                   const uint64_t *const *public_inputs_host, const size_t *n_public_inputs,
                   const uint64_t *wires_values_dev, const int *use_pow_override,
                   const uint64_t *pow_override, uint8_t **proofs_out, size_t *proof_lens);
/* The same with the wire matrices in HOST memory — what the Rust shim calls right after plonky2's witness generation
 * (`generate_partial_witness(...).full_witness().wire_values`, the matrix `CircuitData::prove` builds before
 * `PolynomialBatch::from_values`; call sites: SURVEY.md §8(a) A1). wires_values_host[p]: [num_wires][n]. The library
 * stages them through a device buffer owned by the context. cp_prove = one proof (SURVEY.md §8(b) `cp_prove`). */
int cp_prove_batch_host(cp_ctx *ctx, size_t n_proofs, cp_circuit *const *circuits,
                        const uint64_t *const *public_inputs_host, const size_t *n_public_inputs,
                        const uint64_t *const *wires_values_host, const int *use_pow_override,
                        const uint64_t *pow_override, uint8_t **proofs_out, size_t *proof_lens);
int cp_prove(cp_circuit *circuit, const uint64_t *wires_values_host, const uint64_t *public_inputs_host,
             size_t n_public_inputs, int use_pow_override, uint64_t pow_override, uint8_t **proof_out,
             size_t *proof_len);
/* Zero-knowledge circuits (shape.zero_knowledge = 1). plonky2 draws the leaf salts from its RNG inside
 * `PolynomialBatch::from_values(.., blinding = true, ..)`; here the CALLER supplies them, so the choice of RNG stays
 * on the Rust side and a run is reproducible: salts_host[p] = [3 oracles: wires, Z/partial products, quotient]
 * [CP_SALT_SIZE][N = n << rate_bits] uniformly random canonical field elements, indexed by LEAF (the order is
 * immaterial for random data). Everything else — the blinding rows `CircuitBuilder::blind_and_pad` adds — is part
 * of the circuit and the witness. The plain entry points refuse zero-knowledge circuits and vice versa. */
int cp_prove_batch_zk_host(cp_ctx *ctx, size_t n_proofs, cp_circuit *const *circuits,
                           const uint64_t *const *public_inputs_host, const size_t *n_public_inputs,
                           const uint64_t *const *wires_values_host, const uint64_t *const *salts_host,
                           const int *use_pow_override, const uint64_t *pow_override, uint8_t **proofs_out,
                           size_t *proof_lens);
/* Group commit for callers that prove ONE job at a time. The reference's worker loop does exactly that
 * (`SimpleActorWorker::process_next_job`, city_rollup_core_worker/src/actors/simple.rs:32-56 -> `prove_base` ->
 * `CircuitData::prove`) and scales by running more loops; when several loops live in one process as threads, a batcher
 * merges their concurrent calls: cp_batcher_prove is thread-safe and BLOCKING, the first caller to find the device free
 * leads — it proves its own request together with every pending request of the same shape and gate set (at most
 * max_batch) in one cp_prove_batch_host — and calls that arrive while a batch runs form the next one. Proof bytes are
 * those of cp_prove. A failing request (cp_prove's status and message, the latter through cp_last_error(NULL) of the
 * calling thread) does not fail the requests it was batched with: they are proved again singly.
 * linger_us > 0 lets a leader that found fewer than max_batch requests wait that long for more (0: never wait).
 * With cp_ctx_set_lanes(ctx, L > 1) BEFORE cp_batcher_create, up to L batches run at once, one per lane, so that the host
 * phases of one overlap the kernels of another (the HIP runtime maps the streams of one process onto GPU_MAX_HW_QUEUES = 4
 * hardware queues by default: with more than three lanes export GPU_MAX_HW_QUEUES=8 before the first cp_* call). While a
 * batcher exists, nothing else may prove on its context.
 * cp_batcher_destroy waits for running batches; calling it with callers still inside cp_batcher_prove is an error. */
typedef struct cp_batcher cp_batcher;
typedef struct cp_batcher_stats {
  uint64_t calls;          /* cp_batcher_prove calls accepted */
  uint64_t batches;        /* cp_prove_batch_host launches they were merged into */
  uint64_t proofs;         /* requests that went through those batches (== calls once all have returned) */
  uint64_t largest_batch;
  uint64_t retried_singly; /* batches that failed as a whole and were proved request by request */
} cp_batcher_stats;
cp_batcher *cp_batcher_create(cp_ctx *ctx, size_t max_batch, unsigned linger_us);
int cp_batcher_prove(cp_batcher *batcher, cp_circuit *circuit, const uint64_t *wires_values_host,
                     const uint64_t *public_inputs_host, size_t n_public_inputs, int use_pow_override,
                     uint64_t pow_override, uint8_t **proof_out, size_t *proof_len);
int cp_batcher_get_stats(cp_batcher *batcher, cp_batcher_stats *out);
void cp_batcher_destroy(cp_batcher *batcher);
/* plonky2 `CircuitData::verify` (reference call site: city_common_circuit/src/proof_minifier/
 * pm_chain.rs:264-268): transcript, vanishing identity at zeta, proof of work, every query round's
 * Merkle paths, fri_combine_initial, fold chain and final polynomial. Runs on the host (a few thousand
 * permutations). 0 = accepted; CP_ERR_VERIFY with cp_last_error() naming the first failing check. */
int cp_verify(cp_circuit *circuit, const uint8_t *proof, size_t proof_len);
void cp_free(void *ptr);
/* Audit hook — the query phase of cp_verify (per query round: the Merkle paths of the wires / Z / quotient oracles and
 * of every FRI layer, fri_combine_initial, the fold chain, the final polynomial) with the challenges SUPPLIED instead of
 * derived from the transcript, and without a circuit (the constants/sigmas path is not checked). The reference proofs
 * of qbench_data/example.bin come without their circuits, but alpha, zeta and the FRI betas can be recovered from the
 * proof bytes by algebra (tests/reference_challenges.py): this holds the very code cp_verify runs against plonky2's own
 * output. fri_betas: n_arity x 2; x_indices: num_query_rounds leaf indices. Host arithmetic only, no context. */
int cp_verify_fri_queries_with_challenges(const cp_shape *shape, const uint8_t *proof, size_t proof_len,
                                          const uint64_t alpha[2], const uint64_t zeta[2], const uint64_t *fri_betas,
                                          const uint64_t *x_indices);

/* ---- FRI primitives, the kernels the opening proof is built from ------------------------
 * cp_fri_combine_dev: comp[c] = sum_{j<k} alpha^j f_j[c] over F_p^2 for k base-field vectors of length n (polynomial j at
 *   polys_dev + j*n) — plonky2's `ReducingFactor::reduce_polys_base`, the batch polynomial of the FRI opening proof.
 *   comp_dev: n extension elements (2 u64 each).
 * cp_fri_fold_dev: one reduction layer in coefficient space, out[j] = sum_{i < 2^arity_bits} beta^i c[(j << arity_bits) + i]
 *   (= plonky2's `compute_evaluation` of every coset at beta, without leaving coefficient space).
 *   coeffs_dev: [re | im][n_in]; out_dev: [re | im][n_in >> arity_bits]. */
int cp_fri_combine_dev(cp_ctx *ctx, const uint64_t *polys_dev, size_t k, size_t n, const uint64_t alpha[2],
                       uint64_t *comp_dev);
int cp_fri_fold_dev(cp_ctx *ctx, const uint64_t *coeffs_dev, size_t n_in, int arity_bits, const uint64_t beta[2],
                    uint64_t *out_dev);

/* ---- the two plonky2-level seams every FRI-based prover goes through -----------
 * `CircuitData::prove` is one client of plonky2's polynomial commitment; the SHA-256 STARK the sighash circuit proves
 * three times per block is another: starkyx `ByteStark::prove` (city_common_circuit/src/hash/accelerator/sha256/
 * smartgadget.rs:518-524, called from city_rollup_circuit/src/sighash_circuits/sighash.rs:132-146) is built on plonky2's
 * own config and FRI (`plonky2::{stark::config::GenericCombinedConfig, Plonky2Air}`, smartgadget.rs:48-49): it commits
 * its 418 + 912 trace columns (smartgadget.rs:55-79) with `PolynomialBatch::from_values` and opens them with
 * `PolynomialBatch::prove_openings`. These entry points are those two functions with the data resident on the device;
 * cp_prove itself runs on the same code (csrc/fri_engine.inc). The AIR (constraint evaluation) stays with the caller.
 *
 * cp_poly_batch = plonky2 `PolynomialBatch`: k polynomials of degree < n = 2^degree_bits in coefficient form, their
 * rate-2^rate_bits LDE on the coset 7<omega_N> in bit-reversed order, and the Merkle tree over the LDE rows (leaf i =
 * the k values at bit-reversed position i, followed by the salt when blinding) with its 2^cap_height-entry cap.
 * Owned by the caller (cp_batch_destroy); bound to the context that made it.
 *
 * Lifetime and streams. Every cp_batch_* / cp_fri_prove call has drained the library's stream when it returns.
 * cp_batch_destroy does NOT synchronise the device: the buffers of a destroyed handle go to a per-device pool and are
 * handed to the next commitment of the same shape at once (a hipFree per buffer would stall every other context of the
 * process). The one way a caller's own kernels can touch these buffers is cp_batch_device_ptrs: a handle whose pointers
 * were handed out is NOT recycled - its buffers are released with hipFree, which waits for all work on the device - but
 * the caller must still have finished (or synchronised) every stream that uses the pointers before it calls
 * cp_batch_destroy, exactly as with hipFree of its own memory. A handle may be destroyed after its context
 * (cp_ctx_destroy leaves its buffers alone); every other call on such a handle except cp_batch_info returns
 * CP_ERR_INVALID_ARG.
 * The pool keeps at most CITYPROVER_BATCH_POOL_MB per device (default 4 096; 0 = off), is emptied whenever any allocation
 * of the library would otherwise fail with out-of-memory, and can be emptied by hand (cp_batch_pool_trim). */
typedef struct cp_poly_batch cp_poly_batch;
#define CP_BATCH_FROM_COEFFS 1u /* polys are coefficients (PolynomialBatch::from_coeffs), not values over <omega_n> */
/* polys_host: k x n, polynomial-major. salts_host: NULL = blinding off; else CP_SALT_SIZE x N uniformly random canonical
 * elements indexed by leaf (plonky2 draws them inside from_values(.., blinding = true, ..); here the caller's RNG does). */
int cp_batch_commit(cp_ctx *ctx, const uint64_t *polys_host, size_t k, int degree_bits, int rate_bits, int cap_height,
                    unsigned flags, const uint64_t *salts_host, cp_poly_batch **batch_out);
/* the same from device memory (polys_dev: k x n; salts_dev: CP_SALT_SIZE x N or NULL); the inputs are not retained.
 * Device-resident inputs must be canonical (< p): they are the output of the library's own kernels or of the caller's
 * evaluator and are not re-checked. The host variant checks polynomials and salts — on the device, behind the upload (one host
 * core scanning 418 x 2^16 elements took a fifth of a STARK proof): an element >= p is reported when the call returns
 * (CP_ERR_INVALID_ARG, naming the element), nothing is committed and no handle is made. cp_stark_prove checks a host trace
 * the same way. */
int cp_batch_commit_dev(cp_ctx *ctx, const uint64_t *polys_dev, size_t k, int degree_bits, int rate_bits, int cap_height,
                        unsigned flags, const uint64_t *salts_dev, cp_poly_batch **batch_out);
void cp_batch_destroy(cp_poly_batch *batch);
/* the pool of `device`: bytes / buffers parked, hits / misses of commitments, times it was emptied; pooled = 0 when the
 * device index has no pool (allocations then go straight to the runtime). Needs no context. */
typedef struct cp_batch_pool_info { size_t bytes, buffers, hits, misses, trims, cap_bytes; int pooled; } cp_batch_pool_info;
int cp_batch_pool_stats(int device, cp_batch_pool_info *out);
/* hand everything the pool of ctx's device holds back to the runtime (released_out may be NULL) */
int cp_batch_pool_trim(cp_ctx *ctx, size_t *released_out);
/* any out pointer may be NULL; n_salt_out: CP_SALT_SIZE for a blinded batch, else 0 */
int cp_batch_info(const cp_poly_batch *batch, size_t *k_out, int *degree_bits_out, int *rate_bits_out, int *cap_height_out,
                  int *n_salt_out);
/* `merkle_tree.cap`: 2^cap_height x 4 */
int cp_batch_cap(cp_poly_batch *batch, uint64_t *cap_out_host);
/* out[j] = polynomials[first + j].to_extension().eval(point), j < count (what `OpeningSet::new` / starky's
 * `StarkOpeningSet::new` compute from a commitment); point, out: F_p^2 elements as 2 u64 */
int cp_batch_eval_ext(cp_poly_batch *batch, size_t first, size_t count, const uint64_t point[2], uint64_t *out_host);
/* `polynomials[first .. first + count)`: the coefficient vectors (count x n, polynomial-major) — what a host-side
 * `PolynomialBatch::polynomials` holds; a copy, the handle stays in buffer recycling */
int cp_batch_coeffs(cp_poly_batch *batch, size_t first, size_t count, uint64_t *out_host);
/* `PolynomialBatch::get_lde_values(index * step, ..)` for `count` consecutive indices: row r of out_host
 * (count x k, row-major) = the k LDE values at NATURAL position (first_index + r) * step of the coset (salt excluded) —
 * what a CPU constraint evaluator reads while it builds the quotient. */
int cp_batch_lde_rows(cp_poly_batch *batch, size_t first_index, size_t count, size_t step, uint64_t *out_host);
/* `merkle_tree.leaves[first_leaf .. first_leaf + count)`: the Merkle leaves themselves, in LEAF order (leaf i = the LDE row at
 * natural position bit_reverse(i)), salt included — count x (k + n_salt) row-major. What a host-side `PolynomialBatch` keeps
 * so that plonky2's own `get_lde_values` / `get_lde_values_packed` keep working on a batch committed on the device. */
int cp_batch_leaves(cp_poly_batch *batch, size_t first_leaf, size_t count, uint64_t *out_host);
/* device views for callers that evaluate their constraints on the device: coefficient array (k x n) and bit-reversed
 * LDE (k x N), polynomial-major, valid until cp_batch_destroy — which the caller may only call once its own streams are
 * done with them (see "Lifetime and streams" above; the handle is taken out of buffer recycling by this call) */
int cp_batch_device_ptrs(cp_poly_batch *batch, const uint64_t **coeffs_dev_out, const uint64_t **lde_dev_out);

/* plonky2 `Challenger<F, PoseidonHash>` by value: the duplex sponge state, the elements observed since the last
 * permutation (input_buffer, n_input <= 8) and the squeezed elements not yet handed out (output_buffer; challenges are
 * popped from its END, n_output <= 8). Crosses the ABI in both directions so that the caller's transcript continues
 * exactly where plonky2's would. */
typedef struct cp_challenger_state {
  uint64_t sponge_state[12];
  uint64_t input_buffer[8];
  uint64_t output_buffer[8];
  uint32_t n_input, n_output;
} cp_challenger_state;
/* plonky2 `FriParams` (config.rate_bits, config.cap_height, config.proof_of_work_bits, config.num_query_rounds,
 * degree_bits, reduction_arity_bits; `hiding` is carried by the batches / cp_fri_oracle_info::blinding) */
typedef struct cp_fri_params {
  int degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds;
  int n_arity;
  int arity_bits[8];
} cp_fri_params;
/* `FriBatchInfo { point, polynomials }` with the polynomial list as runs: polynomials first .. first+count-1 of oracle
 * `oracle`, in list order (a FriPolynomialInfo list is the concatenation of its runs) */
typedef struct cp_fri_poly_range { uint32_t oracle, first, count; } cp_fri_poly_range;
typedef struct cp_fri_batch {
  uint64_t point[2];
  const cp_fri_poly_range *ranges;
  size_t n_ranges;
} cp_fri_batch;
/* `PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params, timing) -> FriProof`.
 * oracles: n_oracles (<= 8) batches of one degree, rate and cap height — `FriInstanceInfo::oracles`; batches:
 * `FriInstanceInfo::batches`. The caller has observed the opened values already (plonky2 `observe_openings`); this
 * draws alpha, commits the folded layers (observing each cap, drawing each beta), observes the final polynomial, grinds
 * the proof of work (the SMALLEST witness whose response has pow_bits leading zeros, unless use_pow_override) and
 * draws the query indices — `challenger` comes back in the state plonky2's would be in. fri_proof_out: bincode
 * `FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }` (malloc'd; cp_free). */
int cp_fri_prove(cp_ctx *ctx, cp_poly_batch *const *oracles, size_t n_oracles, const cp_fri_batch *batches, size_t n_batches,
                 const cp_fri_params *params, cp_challenger_state *challenger, int use_pow_override, uint64_t pow_override,
                 uint8_t **fri_proof_out, size_t *fri_proof_len);
/* The verifier side (plonky2 `Challenger::fri_challenges` + `verify_fri_proof`; starkyx verifies the proof it has just
 * made natively, smartgadget.rs:524): from the oracle caps, the opened values of every batch (opened_values[b]: n_polys
 * of batch b x 2 u64, in list order — `FriOpenings`), the challenger state after `observe_openings` and the FriProof
 * bytes. Checks the proof shape, the proof of work, and for every query round the Merkle paths of the initial oracles,
 * fri_combine_initial, the fold chain with its layer paths and the final polynomial. Host arithmetic only: needs no
 * context and no GPU. 0 = accepted; CP_ERR_VERIFY with cp_last_error(NULL) naming the first failing check. */
typedef struct cp_fri_oracle_info { uint32_t num_polys, blinding; } cp_fri_oracle_info;
int cp_fri_verify(const cp_fri_params *params, const cp_fri_oracle_info *oracles, size_t n_oracles,
                  const uint64_t *const *oracle_caps, const cp_fri_batch *batches, size_t n_batches,
                  const uint64_t *const *opened_values, cp_challenger_state *challenger, const uint8_t *fri_proof,
                  size_t fri_proof_len);
/* Transcript helpers with plonky2's semantics (observe_elements / get_n_challenges), for hosts that keep the challenger
 * in this form between calls (tests, the C++ harness); a Rust caller converts its own `Challenger` instead. */
int cp_challenger_observe(cp_challenger_state *challenger, const uint64_t *elements, size_t count);
int cp_challenger_challenges(cp_challenger_state *challenger, uint64_t *out, size_t count);

/* ---- the STARK's own two steps as GENERIC device machinery (second slice) ------
 * starkyx `ByteStark::prove` (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524; 418 free + 912
 * extended columns, :55-79; row count :310-312; called from city_rollup_circuit/src/sighash_circuits/sighash.rs:132-146)
 * does two things between its commitments that are the AIR itself: it fills the extended (lookup-argument) columns and it
 * evaluates every constraint on the quotient coset. The AIR lives in the un-vendored crate starkyx 0.1.0 and cannot be
 * restated - but it does not have to be: starkyx evaluates its constraints through a generic parser trait, and the crate is
 * already a [patch] target of the workspace (/root/reference/Cargo.toml:131-132), so a fork can RECORD the constraints as a
 * flat straight-line program (rust/starkyx-patch/recording_parser.rs, uncompiled) and hand it over. What is built here is
 * the machinery such a program runs on, with nothing of any particular AIR in it:
 *   cp_air_program            a straight-line program over F_p: loads (local row / next row / public / global / challenge
 *                             slots / constants), add, sub, mul, neg (+ inv and column stores in "map" programs), and
 *                             constraint sinks (all rows / transition / first row / last row)
 *   cp_air_quotient_commit    plonky2-starky's `compute_quotient_polys` + `PolynomialBatch::from_coeffs`: the program at every
 *                             point of the quotient coset straight from cp_poly_batch LDEs, alpha-combined per challenge,
 *                             / Z_H, coset iNTT, split into degree-n chunks, committed - nothing leaves the device
 *   cp_air_map_dev            a map program over the n rows of value columns, writing new columns (the row-local part of
 *                             filling extended columns: denominators, products, row sums)
 *   cp_cubic_batch_inverse_dev, cp_column_prefix_sum_dev   the two non-row-local primitives of a logUp argument over a cubic
 *                             extension F_p[X]/(X^3 - m1 X - m0) whose modulus is a parameter: batched inversion and the running
 *                             sum of columns down the trace
 *   cp_stark_prove / cp_stark_verify   the whole prover strung from these and the two seams above (commit -> challenges ->
 *                             extended columns -> commit -> quotient -> commit -> openings -> FRI) and its verifier
 * Protocol restated from plonky2's starky (UPSTREAM-MEMORY, as SURVEY.md Appendix B): transition constraints are multiplied
 * by (x - g^(n-1)), first / last row constraints by the Lagrange basis polynomials L_0 / L_(n-1); per challenge alpha the
 * constraints are folded as acc = acc * alpha + c in program order; quotient = acc / Z_H on the coset 7<omega_(n 2^q)>,
 * q = quotient_degree_bits <= rate_bits, split into 2^q chunks per challenge. PARITY OF A13 STAYS UNPINNED: the reference holds
 * no STARK vector (smartgadget.rs:505-513 asserts digests only), and starkyx's own transcript order is not in the tree. */
typedef struct cp_air_op { uint32_t op, a, b, c; } cp_air_op; /* c: reserved, 0 */
enum {
  CP_AIR_LOCAL = 0,     /* value = column a of the current row (columns of all trace oracles concatenated, in oracle order) */
  CP_AIR_NEXT = 1,      /* value = column a of the next row (cyclically) */
  CP_AIR_PUBLIC = 2,    /* value = public input a */
  CP_AIR_GLOBAL = 3,    /* value = global value a */
  CP_AIR_CHALLENGE = 4, /* value = challenge a (the challenges drawn between trace rounds) */
  CP_AIR_CONST = 5,     /* value = consts[a] */
  CP_AIR_ADD = 6,       /* value = value[a] + value[b]; a, b: indices of EARLIER ops that define a value */
  CP_AIR_SUB = 7,
  CP_AIR_MUL = 8,
  CP_AIR_NEG = 9,       /* value = -value[a] */
  CP_AIR_INV = 10,      /* map programs only: value = value[a]^-1 (0 -> 0) */
  CP_AIR_ASSERT_ZERO = 11,            /* constraint value[a] = 0 on every row (defines no value) */
  CP_AIR_ASSERT_ZERO_TRANSITION = 12, /* on every row but the last */
  CP_AIR_ASSERT_ZERO_FIRST_ROW = 13,
  CP_AIR_ASSERT_ZERO_LAST_ROW = 14,
  CP_AIR_STORE = 15     /* map programs only: output column a <- value[b] (defines no value) */
};
enum { CP_AIR_CONSTRAINTS = 0, CP_AIR_MAP = 1 };
typedef struct cp_air_program_desc {
  int kind;                /* CP_AIR_CONSTRAINTS (sinks, no INV / STORE) or CP_AIR_MAP (INV / STORE, no sinks) */
  const cp_air_op *ops;
  size_t n_ops;            /* <= 2^24 */
  const uint64_t *consts;  /* canonical */
  size_t n_consts;
  uint32_t n_columns;      /* width of a row (local / next) */
  uint32_t n_public, n_global, n_challenge;
  uint32_t n_out_columns;  /* map programs: the columns STORE may write (each at most once); else 0 */
} cp_air_program_desc;
typedef struct cp_air_program cp_air_program;
/* Validates (operand order, index ranges, canonical constants, kind) and compiles the program for the device: dead values
 * dropped, the constraints cut into independent segments so that a small trace still fills the chip, temporaries assigned by
 * liveness to a few per-lane slots. NULL on failure (cp_last_error(ctx)). Immutable afterwards; bound to ctx's device. */
cp_air_program *cp_air_program_create(cp_ctx *ctx, const cp_air_program_desc *desc);
void cp_air_program_destroy(cp_air_program *program);
typedef struct cp_air_program_info {
  size_t n_ops, n_live_ops;      /* as given / after dead-value elimination */
  size_t n_constraints;          /* sinks */
  uint32_t max_constraint_degree; /* in the row variables, selectors included (transition / first / last add one) */
  uint32_t n_segments_max;       /* how many independent segments the constraints can be cut into */
  uint32_t n_slots;              /* per-lane temporaries the compiled single-segment form needs */
  size_t n_instructions;         /* device instructions of the single-segment form */
} cp_air_program_info;
int cp_air_program_get_info(const cp_air_program *program, cp_air_program_info *out);
/* The program on ONE row over F_p^2 (host arithmetic, no GPU): what a verifier does at zeta. local / next / publics / globals /
 * challenges: extension elements (2 u64 each). CP_AIR_CONSTRAINTS programs: out = one extension value per constraint, in program
 * order, NOT multiplied by any selector; sink_kinds_out (optional) = the CP_AIR_ASSERT_* code of each. */
int cp_air_program_eval_ext(const cp_air_program *program, const uint64_t *local, const uint64_t *next, const uint64_t *publics,
                            const uint64_t *globals, const uint64_t *challenges, uint64_t *out, uint32_t *sink_kinds_out);
/* starky `compute_quotient_polys` + `from_coeffs`: oracles = the trace commitments (one degree, rate and cap height; their
 * columns concatenated are the program's row), quotient_degree_bits <= rate_bits with max constraint degree <= 2^q + 1;
 * publics / globals / challenges / alphas: host arrays of canonical elements (n_alphas = starky's num_challenges, <= 4).
 * quotient_out: a cp_poly_batch of n_alphas * 2^q polynomials (challenge-major, chunk-minor), same rate and cap height. */
int cp_air_quotient_commit(cp_ctx *ctx, const cp_air_program *program, cp_poly_batch *const *oracles, size_t n_oracles,
                           int quotient_degree_bits, const uint64_t *publics, const uint64_t *globals,
                           const uint64_t *challenges, const uint64_t *alphas, size_t n_alphas, cp_poly_batch **quotient_out);
/* A map program over n rows. in_cols_dev: n_columns x n values (column-major, natural row order; the next row of the last
 * row is row 0); out_cols_dev: n_out_columns x n, only the stored columns are written. publics / globals / challenges: host. */
int cp_air_map_dev(cp_ctx *ctx, const cp_air_program *program, const uint64_t *in_cols_dev, uint64_t *out_cols_dev, size_t n,
                   const uint64_t *publics, const uint64_t *globals, const uint64_t *challenges);
/* count x n cubic-extension elements, element e of row i = (cols[3e][i], cols[3e+1][i], cols[3e+2][i]) with columns n apart,
 * replaced by their inverses in F_p[X]/(X^3 - modulus[1] X - modulus[0]) (0 -> 0). One field inversion per lane and 8
 * elements (Montgomery's trick on the norms). */
int cp_cubic_batch_inverse_dev(cp_ctx *ctx, const uint64_t modulus[2], uint64_t *cols_dev, size_t count, size_t n);
/* k columns of n rows (columns n apart), each replaced by its running sum down the rows: inclusive (out[i] = sum_{j<=i} in[j])
 * or exclusive (out[i] = sum_{j<i} in[j], out[0] = 0). Addition in a cubic (or any) extension is componentwise, so the running
 * sum of an extension column is this on its component columns. */
int cp_column_prefix_sum_dev(cp_ctx *ctx, uint64_t *cols_dev, size_t k, size_t n, int exclusive);

/* One step of filling the extended columns (all on the device, in order): */
enum { CP_STARK_STEP_MAP = 0, CP_STARK_STEP_CUBIC_INVERSE = 1, CP_STARK_STEP_PREFIX_SUM = 2 };
typedef struct cp_stark_step {
  int kind;
  /* MAP: `program` (n_columns = n_trace_columns + n_extended_columns: a row is the execution trace followed by the extended
   *   columns as filled so far; n_out_columns = n_extended_columns; n_challenge = n_round_challenges)
   * CUBIC_INVERSE: extended columns [first, first + 3 * count) as `count` cubic elements, inverted, modulus as above
   * PREFIX_SUM: extended columns [first, first + count), flags bit 0 = exclusive */
  const cp_air_program *program;
  uint32_t first, count, flags;
  uint64_t modulus[2];
} cp_stark_step;
typedef struct cp_stark_desc {
  int degree_bits;           /* n = 2^degree_bits rows */
  int quotient_degree_bits;  /* q */
  uint32_t num_challenges;   /* alphas (starky: config.num_challenges), 1..4 */
  cp_fri_params fri;         /* fri.degree_bits must equal degree_bits */
  uint32_t n_trace_columns;     /* the execution trace the caller hands in */
  uint32_t n_extended_columns;  /* second trace round, filled by `steps` after the round challenges; 0 = none */
  uint32_t n_round_challenges;  /* drawn after the first trace cap is observed */
  uint32_t n_public, n_global;
  const cp_stark_step *steps;
  size_t n_steps;
  const cp_air_program *constraints; /* n_columns = n_trace_columns + n_extended_columns */
} cp_stark_desc;
/* The prover. trace_values: n_trace_columns x n, column-major, natural row order (host, or device with trace_on_device).
 * `challenger` comes in with whatever the caller's protocol observes first (public inputs, a configuration digest) and goes
 * back as the verifier's will be. Order: observe trace cap; [draw round challenges; fill and commit the extended columns;
 * observe cap]; draw alphas; quotient, commit, observe cap; draw zeta (F_p^2); open every trace column at zeta and g zeta,
 * the quotient chunks at zeta; observe the openings (zeta batch: trace, extended, quotient; then the g zeta batch);
 * `prove_openings`. proof_out (malloc'd; cp_free) in bincode conventions:
 *   trace_caps: Vec<Vec<[u64;4]>> (1 or 2) | quotient_cap: Vec<[u64;4]> | local_values, next_values, quotient_polys: Vec<[u64;2]> |
 *   FriProof (as cp_fri_prove) */
int cp_stark_prove(cp_ctx *ctx, const cp_stark_desc *desc, const uint64_t *trace_values, int trace_on_device,
                   const uint64_t *publics, const uint64_t *globals, cp_challenger_state *challenger, int use_pow_override,
                   uint64_t pow_override, uint8_t **proof_out, size_t *proof_len);
/* The verifier (host arithmetic only; desc->steps is not read, and the program handle is used for its host form only):
 * transcript, constraints at zeta from the openings against Z_H(zeta) * sum_i zeta^(n i) t_i(zeta), then `verify_fri_proof`.
 * 0 = accepted; CP_ERR_VERIFY with cp_last_error(NULL) naming the first failing check. */
int cp_stark_verify(const cp_stark_desc *desc, const uint64_t *publics, const uint64_t *globals, cp_challenger_state *challenger,
                    const uint8_t *proof, size_t proof_len);

/* ---- BLS12-381 G1 multi-scalar multiplication ---------------------------------
 * The G1 MSMs of the Groth16 wrap proof: replaces the CPU MSM inside `gnark_plonky2_wrapper::wrap_plonky2_proof`
 * (reference call sites: city_rollup_circuit/src/worker/toolbox/root.rs:296-304,
 * city_rollup_core_worker/src/lib.rs:121; the arithmetic itself is gnark-crypto's, a Go dependency outside the tree).
 * Scalars: 4 little-endian u64 per scalar (any 256-bit integer). Points: affine x || y, 6 + 6 little-endian u64 of the
 * canonical (non-Montgomery) coordinates; points_inf: optional byte flags, non-zero = the point at infinity.
 * Result: affine canonical coordinates + infinity flag. First kernels of row A12 (the witness solver and the
 * proof assembly are not built). */
int cp_msm_bls12381_g1(cp_ctx *ctx, const uint64_t *scalars_host, const uint64_t *points_xy_host,
                       const uint8_t *points_inf_host, size_t n, uint64_t out_xy[12], int *out_is_infinity);
/* Device-resident variant for a fixed point set (a proving key): convert once to the library's internal form
 * (points_mont_dev: n * CP_G1_AFFINE_BYTES bytes, caller-allocated), then run any number of MSMs against it. */
#define CP_G1_AFFINE_BYTES 112
int cp_msm_bls12381_g1_prepare_dev(cp_ctx *ctx, const uint64_t *points_xy_dev, size_t n, void *points_mont_dev);
/* Bench / test helper: fills points_mont_dev with P_i = (a*i + b) * G for i < n (a, b in [1, 65535]), G = any curve
 * point given by its affine canonical coordinates — distinct points without a 100 MB upload, and a closed form for the
 * expected MSM: (sum_i k_i (a i + b)) * G. */
int cp_msm_bls12381_g1_synthetic_points_dev(cp_ctx *ctx, const uint64_t generator_xy[12], uint32_t a, uint32_t b,
                                            size_t n, void *points_mont_dev);
int cp_msm_bls12381_g1_dev(cp_ctx *ctx, const uint64_t *scalars_dev, const void *points_mont_dev,
                           const uint8_t *points_inf_dev, size_t n, uint64_t out_xy[12], int *out_is_infinity);
/* The same over G2 (the twist y^2 = x^3 + 4(1+u) over F_p^2 = F_p[u]/(u^2+1)): the B-query MSM of Groth16.
 * A coordinate is c0 + c1*u, 6 + 6 little-endian u64; a point is x.c0, x.c1, y.c0, y.c1 = 24 u64. */
#define CP_G2_AFFINE_BYTES 224
int cp_msm_bls12381_g2(cp_ctx *ctx, const uint64_t *scalars_host, const uint64_t *points_xy_host,
                       const uint8_t *points_inf_host, size_t n, uint64_t out_xy[24], int *out_is_infinity);
int cp_msm_bls12381_g2_prepare_dev(cp_ctx *ctx, const uint64_t *points_xy_dev, size_t n, void *points_mont_dev);
int cp_msm_bls12381_g2_synthetic_points_dev(cp_ctx *ctx, const uint64_t generator_xy[24], uint32_t a, uint32_t b,
                                            size_t n, void *points_mont_dev);
int cp_msm_bls12381_g2_dev(cp_ctx *ctx, const uint64_t *scalars_dev, const void *points_mont_dev,
                           const uint8_t *points_inf_dev, size_t n, uint64_t out_xy[24], int *out_is_infinity);

/* ---- NTT over the BLS12-381 scalar field F_r (the transforms of Groth16's quotient) ----
 * Elements: 4 little-endian u64 of the canonical value (< r). In place, natural order in and out,
 * omega_n = 7^((r-1)/n). flags: CP_NTT_INVERSE (exact inverse, 1/n included), CP_NTT_COSET with coset_shift
 * (4 u64, non-zero): forward = evaluations on shift*<omega_n>, inverse = its inverse. log_n <= 28. */
int cp_ntt_bls12381_fr(cp_ctx *ctx, uint64_t *data_host, int log_n, unsigned flags, const uint64_t *coset_shift);
int cp_ntt_bls12381_fr_dev(cp_ctx *ctx, uint64_t *data_dev, int log_n, unsigned flags, const uint64_t *coset_shift);

/* Groth16 quotient polynomial over F_r (the H part of the proof's C element; inside gnark's `groth16.Prove`, which
 * `gnark_plonky2_wrapper::wrap_plonky2_proof` runs: city_rollup_circuit/src/worker/toolbox/root.rs:296-304).
 * a, b, c: n = 2^log_n evaluations of the R1CS products (A w), (B w), (C w) on <omega_n>, canonical, natural order.
 * Computes h = (a(x) b(x) - c(x)) / (x^n - 1) through the coset 7<omega_n>:
 *   a, b, c <- iNTT;  <- coset NTT;  a <- (a o b - c) / (7^n - 1);  a <- coset iNTT.
 * h's n coefficients replace a; b and c are overwritten with intermediates. For a satisfied R1CS (a o b = c on the
 * domain) h is the exact quotient, of degree <= n - 2. Parity unpinned by the reference (no vectors). */
int cp_groth16_quotient_bls12381_dev(cp_ctx *ctx, uint64_t *a_dev, uint64_t *b_dev, uint64_t *c_dev, int log_n);
int cp_groth16_quotient_bls12381(cp_ctx *ctx, uint64_t *a_host, const uint64_t *b_host, const uint64_t *c_host,
                                 int log_n);

/* ---- Groth16 proof assembly (gnark's groth16.Prove after the witness solver) ----
 * The proving key as device-resident point sets in the library's internal form (cp_msm_bls12381_g1/g2_prepare_dev),
 * wires ordered public first (the constant-one wire included), then private:
 *   a_g1[i], b_g1[i], b_g2[i], i < n_wires       the A / B query ([u_i(tau)]_1, [v_i(tau)]_1, [v_i(tau)]_2)
 *   a_inf / b_inf                                 optional device byte flags: 1 = that entry is the point at infinity
 *   k_g1[i], i < n_private                        [(beta u_i + alpha v_i + w_i)(tau) / delta]_1 of the private wires
 *   z_g1[j], j < 2^log_domain - 1                 [tau^j (tau^n - 1) / delta]_1
 * and alpha, beta, delta as canonical affine coordinates. */
typedef struct cp_groth16_pk {
  size_t n_wires, n_private;
  int log_domain;
  const void *a_g1, *b_g1, *b_g2, *k_g1, *z_g1;
  const uint8_t *a_inf, *b_inf;
  uint64_t alpha_g1[12], beta_g1[12], delta_g1[12];
  uint64_t beta_g2[24], delta_g2[24];
} cp_groth16_pk;
/* witness_dev: n_wires x 4 u64 (canonical scalars). a/b/c_evals_dev: 2^log_domain evaluations of A w, B w, C w on the
 * domain (canonical; overwritten - see cp_groth16_quotient_bls12381_dev). r, s: the prover's blinding scalars, supplied
 * by the caller (the RNG stays on the host side; runs are reproducible). Outputs: affine canonical A (G1), B (G2), C (G1):
 *   A = alpha + sum w_i A_i + r delta,  B = beta + sum w_i B_i + s delta,
 *   C = sum_private w_i K_i + sum h_j Z_j + s A + r B1 - r s delta.
 * Five MSMs + the quotient on the device, the remaining point operations on the host. Parity unpinned by the reference
 * (no proving key or vectors in the tree): tests check against the trapdoor of a setup they generate. */
int cp_groth16_prove_bls12381(cp_ctx *ctx, const cp_groth16_pk *pk, const uint64_t *witness_dev, uint64_t *a_evals_dev,
                              uint64_t *b_evals_dev, uint64_t *c_evals_dev, const uint64_t r[4], const uint64_t s[4],
                              uint64_t out_a[12], uint64_t out_b[24], uint64_t out_c[12]);


/* The proof in the form the worker stores and the chain consumes: `CityGroth16ProofData { pi_a, pi_b_a0, pi_b_a1, pi_c }`
 * = 4 x 48 bytes (city_rollup_common/src/block_template/data.rs:6-34; produced at
 * city_rollup_circuit/src/worker/toolbox/root.rs:296-315). Each element is a compressed point: x little-endian, flags in
 * the two top bits of the last byte (0x80: y is the larger root, 0x40: infinity); pi_b is one G2 point whose
 * x = a0 + a1 u is split into pi_b_a0 (no flags) and pi_b_a1 (flags). Byte order, flag position and the a0 / a1 order
 * are pinned on the reference's two samples (data.rs:72-73: all eight elements decompress to points of the r-torsion
 * subgroups); the choice of root behind 0x80 follows arkworks (y > -y; F_p^2 by c1, then c0). Host arithmetic only:
 * these two need no context and no GPU. Inputs: affine canonical coordinates as cp_groth16_prove_bls12381 returns them. */
int cp_groth16_proof_pack_city(const uint64_t a_xy[12], const uint64_t b_xy[24], const uint64_t c_xy[12], uint8_t out[192]);
/* the inverse (decompression: one square root per element); refuses x that is not on the curve / twist */
int cp_groth16_proof_unpack_city(const uint8_t in[192], uint64_t a_xy[12], uint64_t b_xy[24], uint64_t c_xy[12]);

#ifdef __cplusplus
}
#endif
#endif /* CITYPROVER_H */
