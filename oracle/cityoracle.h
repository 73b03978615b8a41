/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see goldilocks.h). CPU restatement of the
 * plonky2 0.2.2 proving primitives that city-rollup reaches through
 * `CircuitData::prove`.
 *
 * Parity status: Poseidon permutation / sponge / two_to_one / Merkle path
 * direction are PINNED against the reference's own known-answer data
 * (tests/golden/, see tests/test_oracle_golden.py). NTT/LDE have no vectors in
 * the reference tree; they are exact integer arithmetic and are pinned only
 * indirectly (FRI fold consistency of the reference proofs in
 * qbench_data/example.bin, tests/test_oracle_fri_reference.py).
 */
#ifndef CITY_ORACLE_H
#define CITY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- field (exported for ctypes-based tests) ---- */
uint64_t or_gl_add(uint64_t a, uint64_t b);
uint64_t or_gl_sub(uint64_t a, uint64_t b);
uint64_t or_gl_mul(uint64_t a, uint64_t b);
uint64_t or_gl_mul_slow(uint64_t a, uint64_t b);
uint64_t or_gl_inv(uint64_t a);
uint64_t or_gl_pow(uint64_t a, uint64_t e);
uint64_t or_gl_root_of_unity(int log_n);

/* ---- Poseidon-Goldilocks (width 12, x^7, 4+22+4 rounds) ---- */
/* The 360 round constants, regenerated from the published procedure
 * (ChaCha8 seeded with 0, uniform in [0,p)). out[360]. */
void or_poseidon_round_constants(uint64_t *out);
/* MDS: circulant first row (12) and diagonal (12). */
void or_poseidon_mds(uint64_t *circ, uint64_t *diag);
/* 1: the MDS layer without 128-bit products (plonky2's portable form: 32-bit halves, u64 dot products) in every
 * permutation of the oracle; 0 (default): the textbook layer. Same results (tests/test_oracle_golden.py). */
void or_set_fast_poseidon(int on);
void or_poseidon_permute(uint64_t state[12]);
void or_poseidon_permute_many(uint64_t *states, size_t count);
/* poseidon_simd.c — eight states per AVX-512 permutation, for bench.py's cpu_baseline legs ONLY ("port-simd"): off by default, so
 * that the checker of the tests is the scalar textbook form; tests/test_oracle_simd.py holds the two against each other.
 * or_set_simd_poseidon(1) routes the leaf hashes and levels of or_merkle_tree_cols (and everything built on it: commitments,
 * whole proofs) and or_poseidon_permute_many through it when the CPU has AVX-512 F / DQ / VL / BW; else it stays off. */
int or_simd_available(void);
void or_set_simd_poseidon(int on);
int or_simd_poseidon_enabled(void);
void or_poseidon_permute_x8(uint64_t st[12][8]);
void or_simd_leaf_hash_cols_x8(const uint64_t *cols, size_t leaf_len, size_t col_stride, size_t i0, uint64_t *digests_out);
void or_simd_two_to_one_x8(const uint64_t *children, uint64_t *parents_out);
void or_simd_permute_aos_x8(uint64_t *states);

/* PoseidonHash::hash_no_pad — overwrite-mode sponge, rate 8
 * (call sites: city_crypto/src/hash/traits/hasher.rs:82-95). */
void or_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]);
/* hash_or_noop: n <= 4 -> zero-padded copy, else hash_no_pad. */
void or_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]);
/* PoseidonHash::two_to_one (city_crypto/src/hash/traits/hasher.rs:77-80). */
void or_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);

/* ---- Merkle tree with cap (plonky2 MerkleTree::new) ----
 * leaves: n_leaves rows of leaf_len felts, row-major, in TREE order (the caller
 * applies any bit-reversal). digests_out (optional, may be NULL) receives all
 * levels below the cap, level 0 (leaf digests) first: n_leaves*4, then
 * n_leaves/2*4, ... down to 2^(cap_height+1) nodes. cap_out: 2^cap_height * 4. */
void or_merkle_tree(const uint64_t *leaves, size_t n_leaves, size_t leaf_len,
                    int cap_height, uint64_t *digests_out, uint64_t *cap_out);
/* same, leaves given column-major: element j of leaf i at cols[j*col_stride + i] */
void or_merkle_tree_cols(const uint64_t *cols, size_t n_leaves, size_t leaf_len,
                         size_t col_stride, int cap_height, uint64_t *digests_out,
                         uint64_t *cap_out);
/* Verify a Merkle path against a cap: returns 1 if ok. Path direction follows
 * city_crypto/src/hash/merkle/core.rs:200-213 (bit i of index == 0 -> H(cur, sib)). */
int or_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t index,
                     const uint64_t *siblings, size_t n_siblings,
                     const uint64_t *cap, int cap_height);

/* ---- NTT over Goldilocks (plonky2 fft / ifft / coset LDE semantics) ----
 * natural order in, natural order out; omega = g^((p-1)/n), g = 7. */
void or_ntt(uint64_t *a, int log_n);
void or_intt(uint64_t *a, int log_n);
/* coefficients (n = 2^log_n) -> evaluations on shift*<omega_{n*2^rate_bits}>,
 * natural order, out has n << rate_bits entries. */
void or_coset_lde(const uint64_t *coeffs, int log_n, int rate_bits, uint64_t shift,
                  uint64_t *out);
/* in-place bit-reversal permutation of n = 2^log_n u64 */
void or_bit_reverse(uint64_t *a, int log_n);
/* O(n^2) DFT for small-size cross-checks */
void or_dft_naive(const uint64_t *in, uint64_t *out, int log_n);

/* ---- PolynomialBatch::from_values ----
 * values: k polys, each n = 2^log_n values in natural order (poly-major).
 * coeffs_out (k*n, optional), lde_out (k * n<<rate_bits, poly-major, BIT-REVERSED
 * index order == Merkle leaf order; optional), cap_out (2^cap_height*4). */
void or_commit_batch(const uint64_t *values, size_t k, int log_n, int rate_bits,
                     int cap_height, uint64_t *coeffs_out, uint64_t *lde_out,
                     uint64_t *digests_out, uint64_t *cap_out);

/* ---- transcript / openings / FRI / proof bytes (plonky2_tail.c) ---- */
typedef struct {
  uint64_t state[12];
  uint64_t in[8];
  int n_in;
  uint64_t out[8];
  int n_out;
} or_challenger;
void or_ch_init(or_challenger *c);
void or_ch_observe(or_challenger *c, const uint64_t *elems, size_t n);
uint64_t or_ch_challenge(or_challenger *c);

/* circuit shape: CommonCircuitData scalars the prover tail needs
 * (fields mirrored from city_common_circuit/src/verify_template/ser_data.rs:55-123) */
typedef struct {
  int degree_bits, num_constants, num_routed_wires, num_wires, num_challenges, num_partial_products,
      quotient_degree_factor;
  int rate_bits, cap_height, pow_bits, num_query_rounds;
  int n_arity;
  int arity_bits[8];
  int zero_knowledge; /* CircuitConfig::zero_knowledge: FRI `hiding` — the leaves of the wires / Z / quotient oracles
                         carry SALT_SIZE = 4 random elements (plonky2 PlonkOracle::{WIRES,ZS_PARTIAL_PRODUCTS,QUOTIENT}.blinding) */
} or_shape;
#define OR_SALT_SIZE 4

typedef struct {
  uint64_t betas[8], gammas[8], alphas[8];
  uint64_t zeta[2];
  uint64_t fri_betas[8][2];
  uint64_t pow_response;
  uint64_t query_indices[64];
} or_tail_debug;

/* Everything of CircuitData::prove after the polynomials are known. cs_values: (constants+sigmas) x n
 * values; wires_values: num_wires x n; zs_pp_values: num_challenges*(1+num_partial_products) x n
 * (Z polynomials first); quotient_coeffs: num_challenges*quotient_degree_factor x n COEFFICIENTS.
 * Produces bincode ProofWithPublicInputs bytes (malloc'd, free with or_free). */
int or_prove_tail(const or_shape *sh, const uint64_t circuit_digest[4], const uint64_t *public_inputs,
                  size_t n_pi, const uint64_t *cs_values, const uint64_t *wires_values,
                  const uint64_t *zs_pp_values, const uint64_t *quotient_coeffs, int use_pow_override,
                  uint64_t pow_override, uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg);
/* Verifier side of the same: transcript, PoW, all Merkle paths, fri_combine_initial vs the openings,
 * fold consistency, final polynomial. cs_cap: the circuit's constants_sigmas cap. 0 = accepted. */
int or_verify_tail(const or_shape *sh, const uint64_t circuit_digest[4], const uint64_t *cs_cap,
                   const uint8_t *proof, size_t len, or_tail_debug *dbg);
void or_fri_compute_evaluation(uint64_t x, size_t x_index_within_coset, int arity_bits,
                               const uint64_t *evals, const uint64_t beta[2], uint64_t out[2]);
uint64_t or_fri_query_point(size_t x_index, int log_n);
void or_free(void *p);

/* ---- the generic seams under the whole-proof functions (plonky2_tail.c): plonky2 `PolynomialBatch` (from_values /
 * from_coeffs, optional salt = OR_SALT_SIZE x N leaf-ordered elements), `PolynomialBatch::prove_openings` over any
 * oracles and opening batches, `verify_fri_proof`. Checker of cp_batch_* / cp_fri_prove / cp_fri_verify. ---- */
typedef struct or_batch or_batch;
or_batch *or_batch_commit(const uint64_t *polys, size_t k, int log_n, int rate_bits, int cap_height, int from_coeffs,
                          const uint64_t *salt);
void or_batch_free(or_batch *b);
const uint64_t *or_batch_cap(const or_batch *b);
const uint64_t *or_batch_coeffs(const or_batch *b);
const uint64_t *or_batch_lde(const or_batch *b);
void or_batch_eval_ext(const or_batch *b, size_t first, size_t count, const uint64_t point[2], uint64_t *out);
void or_batch_lde_rows(const or_batch *b, size_t first_index, size_t count, size_t step, uint64_t *out);
typedef struct { int degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds, n_arity; int arity_bits[8]; } or_fri_params;
typedef struct { uint32_t oracle, first, count; } or_fri_range;
typedef struct { uint64_t point[2]; const or_fri_range *ranges; size_t n_ranges; } or_fri_batch;
/* c: the transcript after the openings were observed; advanced as plonky2's would be. out: bincode FriProof (or_free). */
int or_fri_prove(const or_batch *const *oracles, size_t n_oracles, const or_fri_batch *batches, size_t n_batches,
                 const or_fri_params *params, or_challenger *c, int use_pow_override, uint64_t pow_override,
                 uint8_t **out, size_t *len, or_tail_debug *dbg);
/* opened[b]: the claimed values of batch b (n_polys x 2 u64, list order). 0 = accepted (c advanced), else the failing check. */
int or_fri_verify(const or_fri_params *params, const uint32_t *num_polys, const uint32_t *blinding, size_t n_oracles,
                  const uint64_t *const *caps, const or_fri_batch *batches, size_t n_batches, const uint64_t *const *opened,
                  or_challenger *c, const uint8_t *proof, size_t len, or_tail_debug *dbg);

void or_batch_shape(const or_batch *b, size_t *k, int *log_n, int *rate_bits, int *cap_height);

/* ---- generic AIR machinery (stark_air.c): checker of cp_air_* / cp_cubic_batch_inverse_dev / cp_column_prefix_sum_dev /
 * cp_stark_prove / cp_stark_verify. An AIR is DATA here - a straight-line program over F_p with the op codes of
 * include/cityprover.h (CP_AIR_*) - because the SHA-256 STARK's own AIR lives in an absent crate (starkyx 0.1.0,
 * /root/reference/Cargo.toml:112); the protocol around it follows plonky2's starky (UPSTREAM-MEMORY). Parity unpinned. ---- */
typedef struct { uint32_t op, a, b, c; } or_air_op;
typedef struct {
  int map; /* 0: constraint program (sinks), 1: map program (INV / STORE) */
  const or_air_op *ops;
  size_t n_ops;
  const uint64_t *consts;
  size_t n_consts;
  uint32_t n_columns, n_public, n_global, n_challenge, n_out_columns;
} or_air_program;
size_t or_air_check(const or_air_program *p); /* 0 = well-formed, else 1 + index of the first bad op */
size_t or_air_num_constraints(const or_air_program *p);
void or_air_eval_row(const or_air_program *p, const uint64_t *local, const uint64_t *next, const uint64_t *publics,
                     const uint64_t *globals, const uint64_t *challenges, uint64_t *vals);
/* unfiltered constraint values on one row over F_p^2 (inputs: extension elements), program order */
void or_air_eval_ext(const or_air_program *p, const uint64_t *local, const uint64_t *next, const uint64_t *publics,
                     const uint64_t *globals, const uint64_t *challenges, uint64_t *out, uint32_t *kinds_out);
void or_air_map(const or_air_program *p, const uint64_t *in_cols, uint64_t *out_cols, size_t n, const uint64_t *publics,
                const uint64_t *globals, const uint64_t *challenges);
/* F_p[X]/(X^3 - m[1] X - m[0]) */
void or_cubic_mul(const uint64_t m[2], const uint64_t a[3], const uint64_t b[3], uint64_t out[3]);
void or_cubic_inverse(const uint64_t m[2], const uint64_t a[3], uint64_t out[3]);
void or_cubic_batch_inverse(const uint64_t m[2], uint64_t *cols, size_t count, size_t n);
void or_column_prefix_sum(uint64_t *cols, size_t k, size_t n, int exclusive);
/* starky compute_quotient_polys: n_alphas * 2^qdb coefficient vectors of length n, challenge-major */
int or_air_quotient(const or_air_program *p, const or_batch *const *oracles, size_t n_oracles, int qdb, const uint64_t *publics,
                    const uint64_t *globals, const uint64_t *challenges, const uint64_t *alphas, size_t n_alphas, uint64_t *out);
typedef struct {
  int kind; /* 0 map, 1 cubic inverse, 2 prefix sum */
  const or_air_program *program;
  uint32_t first, count, flags;
  uint64_t modulus[2];
} or_stark_step;
typedef struct {
  int degree_bits, quotient_degree_bits;
  uint32_t num_challenges;
  or_fri_params fri;
  uint32_t n_trace_columns, n_extended_columns, n_round_challenges, n_public, n_global;
  const or_stark_step *steps;
  size_t n_steps;
  const or_air_program *constraints;
} or_stark_desc;
int or_stark_prove(const or_stark_desc *d, const uint64_t *trace_values, const uint64_t *publics, const uint64_t *globals,
                   or_challenger *c, int use_pow_override, uint64_t pow_override, uint8_t **proof_out, size_t *proof_len);
/* 0 accepted; 1 malformed; 2 constraints fail at zeta; negative: or_fri_verify's code */
int or_stark_verify(const or_stark_desc *d, const uint64_t *publics, const uint64_t *globals, or_challenger *c, const uint8_t *proof,
                    size_t len);

/* ---- gates / quotient (plonky2_quotient.c) ---- */
enum { OR_GATE_NOOP = 0, OR_GATE_CONSTANT = 1, OR_GATE_PUBLIC_INPUT = 2, OR_GATE_ARITHMETIC = 3, OR_GATE_POSEIDON = 4,
       OR_GATE_COMPARISON = 5, OR_GATE_U32_ARITHMETIC = 6, OR_GATE_U32_RANGE_CHECK = 7,
       /* plonky2_gates.c: */
       OR_GATE_U32_ADD_MANY = 8, OR_GATE_U32_SUBTRACTION = 9, OR_GATE_U32_INTERLEAVE = 10, OR_GATE_UNINTERLEAVE_TO_U32 = 11,
       OR_GATE_UNINTERLEAVE_TO_B32 = 12, OR_GATE_ARITHMETIC_EXT = 13, OR_GATE_MUL_EXT = 14, OR_GATE_BASE_SUM = 15,
       OR_GATE_RANDOM_ACCESS = 16, OR_GATE_REDUCING = 17, OR_GATE_REDUCING_EXT = 18, OR_GATE_POSEIDON_MDS = 19,
       OR_GATE_COSET_INTERPOLATION = 20, OR_GATE_EXPONENTIATION = 21 };
typedef struct {
  int type;           /* OR_GATE_* */
  int selector_index; /* which selector polynomial (constants column) carries this gate's group */
  int group_start, group_end; /* the gate indices sharing that selector */
  int param;          /* Constant: num_consts; Arithmetic / U32Arithmetic: num_ops; Comparison: num_bits; RangeCheck: limbs */
  int param2;         /* Comparison: num_chunks; AddMany: num_addends; BaseSum: B; RandomAccess: num_copies; CosetInterpolation: degree */
  int param3;         /* RandomAccess: num_extra_constants */
} or_gate;
typedef struct {
  int n_gates;
  or_gate gates[32];  /* in CommonCircuitData::gates order: the gate's index is its position */
  int num_selectors;  /* the first num_selectors "constants" columns are selector polynomials */
  uint64_t k_is[256]; /* num_routed_wires coset shifts */
} or_gates;
int or_gates_num_constraints(const or_gates *g);
/* A8: quotient chunk polynomials (coefficients), num_challenges*quotient_degree_factor x n, from the
 * bit-reversed LDEs of the three oracles. */
int or_quotient_polys(const or_shape *sh, const or_gates *G, const uint64_t pi_hash[4], const uint64_t *cs_lde,
                      const uint64_t *wires_lde, const uint64_t *zs_lde, const uint64_t *betas,
                      const uint64_t *gammas, const uint64_t *alphas, uint64_t *out_coeffs);
/* verifier: Z_H(zeta) * sum_i zeta^(n i) t_i(zeta) == vanishing(zeta) from the openings. 0 = holds. */
int or_check_vanishing(const or_shape *sh, const or_gates *G, const uint64_t pi_hash[4], const uint64_t zeta[2],
                       const uint64_t *op_constants, const uint64_t *op_sigmas, const uint64_t *op_wires,
                       const uint64_t *op_zs, const uint64_t *op_zs_next, const uint64_t *op_pps,
                       const uint64_t *op_quotient, const uint64_t *betas, const uint64_t *gammas,
                       const uint64_t *alphas);
/* wires -> proof (A7 + A8 + tail) */
int or_prove_full(const or_shape *sh, const or_gates *G, const uint64_t circuit_digest[4],
                  const uint64_t *public_inputs, size_t n_pi, const uint64_t *cs_values,
                  const uint64_t *wires_values, int use_pow_override, uint64_t pow_override,
                  uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg);

/* zero-knowledge variant: salts = [3 oracles: wires, zs/partial products, quotient][OR_SALT_SIZE][N] random field
 * elements in LEAF order (the prover appends them to the Merkle leaves; plonky2 draws them from its RNG). Requires
 * sh->zero_knowledge != 0. */
int or_prove_full_zk(const or_shape *sh, const or_gates *G, const uint64_t circuit_digest[4],
                     const uint64_t *public_inputs, size_t n_pi, const uint64_t *cs_values,
                     const uint64_t *wires_values, const uint64_t *salts, int use_pow_override, uint64_t pow_override,
                     uint8_t **proof_out, size_t *proof_len, or_tail_debug *dbg);

/* A7: Z and partial-product polynomials of the permutation argument (values over <omega_n>).
 * wires_values: num_wires x n (only the routed ones are read); sigma_values: num_routed_wires x n;
 * out: num_challenges*(1+num_partial_products) x n in committed order (Z's first). */
void or_zs_partial_products(const or_shape *sh, const uint64_t *wires_values, const uint64_t *sigma_values,
                            const uint64_t *k_is, const uint64_t *betas, const uint64_t *gammas,
                            uint64_t *out);

/* ---- BLS12-381 G1 (bls12_381.c): checker for the Groth16-wrap MSM kernels ----
 * field elements: 6 little-endian u64 limbs of the canonical value; points: affine x||y (12 limbs) + infinity flag;
 * scalars: 4 little-endian u64 limbs. */
void or_bls_constants(uint64_t p[6], uint64_t r[4], uint64_t gen_xy[12]);
int or_bls_g1_on_curve(const uint64_t xy[12]);
void or_bls_g1_add(const uint64_t a_xy[12], int a_inf, const uint64_t b_xy[12], int b_inf, uint64_t out_xy[12], int *out_inf);
void or_bls_g1_mul(const uint64_t xy[12], int inf, const uint64_t k[4], uint64_t out_xy[12], int *out_inf);
void or_bls_g1_msm(const uint64_t *scalars, const uint64_t *points_xy, const uint8_t *points_inf, size_t n,
                   uint64_t out_xy[12], int *out_inf);

/* G2 (twist over F_p^2): points = x.c0, x.c1, y.c0, y.c1 (24 limbs) + infinity flag */
void or_bls_g2_generator(uint64_t xy[24]);
int or_bls_g2_on_curve(const uint64_t xy[24]);
void or_bls_g2_add(const uint64_t a_xy[24], int a_inf, const uint64_t b_xy[24], int b_inf, uint64_t out_xy[24], int *out_inf);
void or_bls_g2_mul(const uint64_t xy[24], int inf, const uint64_t k[4], uint64_t out_xy[24], int *out_inf);
void or_bls_g2_msm(const uint64_t *scalars, const uint64_t *points_xy, const uint8_t *points_inf, size_t n,
                   uint64_t out_xy[24], int *out_inf);

/* scalar field F_r of BLS12-381 and its NTT (Groth16's quotient transforms); elements = 4 LE u64, canonical */
int or_fr_is_canonical(const uint64_t a[4]);
void or_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void or_fr_root_of_unity(int log_n, uint64_t out[4]);
void or_fr_dft_naive(const uint64_t *in, uint64_t *out, int log_n);
void or_fr_ntt(uint64_t *data, int log_n, int inverse, const uint64_t *shift);
/* Groth16 quotient h = (a b - c) / (x^n - 1) from evaluations on <omega_n> (through the coset 7<omega_n>); h -> a */
void or_groth16_quotient(uint64_t *a, uint64_t *b, uint64_t *c, int log_n);

/* number of worker threads the oracle uses for the batch entry points
 * (or_poseidon_permute_many, or_merkle_tree*, or_commit_batch); default 1 */
void or_set_threads(int n);
int or_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
