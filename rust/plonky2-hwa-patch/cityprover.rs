//! plonky2/src/plonk/cityprover.rs — NEW FILE of the `plonky2-hwa` fork (feature `cityprover`).
//!
//! The Rust half of the import bridge: flattens a built `CircuitData` into the flat form
//! `libcityprover_hip.so` takes, forwards `CircuitData::prove` (after witness generation) and `::verify` to it, and dumps
//! circuits / witnesses to the `.cpcirc` / `.cpwit` files the native q-bench harness replays
//! (tools/cityprover_qbench, tools/qbench/pack.h in the cityprover repository).
//!
//! Written against plonky2 0.2.2 (QEDProtocol/plonky2-hwa rev 6a8ca008, the revision city-rollup pins:
//! Cargo.toml:101-102, patched in at Cargo.toml:128-132) FROM MEMORY of its public API — this image has no cargo,
//! no rustc and no copy of the crate, so NOTHING here has been compiled. Field and method names to re-check when
//! applying: `CommonCircuitData::{config, fri_params, gates, selectors_info, quotient_degree_factor, num_constants,
//! num_public_inputs, k_is, num_partial_products}`, `ProverOnlyCircuitData::{constants_sigmas_commitment, public_inputs,
//! circuit_digest}`, `PolynomialBatch::polynomials`, `MatrixWitness::wire_values` (wire-major: `wire_values[wire][row]`),
//! `Target::Wire(Wire { row, column })`.
use std::io::Write;
use std::ops::Range;
use std::path::Path;
use std::sync::{Mutex, OnceLock};

use anyhow::{bail, ensure, Result};
use cityprover_sys::{Batcher, Circuit as GpuCircuit, Context, CpGate, CpShape};

use crate::field::goldilocks_field::GoldilocksField;
use crate::field::types::{Field, PrimeField64};
use crate::gates::gate::GateRef;
use crate::iop::target::Target;
use crate::iop::witness::MatrixWitness;
use crate::plonk::circuit_data::{CommonCircuitData, ProverOnlyCircuitData, VerifierOnlyCircuitData};
use crate::plonk::config::PoseidonGoldilocksConfig;
use crate::plonk::proof::ProofWithPublicInputs;

type F = GoldilocksField;
type C = PoseidonGoldilocksConfig;
const D: usize = 2;

/// One context per process, on the GPU named by `CITYPROVER_DEVICE` (default 0): the worker is one process per
/// consumer (city_rollup_core_worker/src/lib.rs:104-146); N GPUs = N worker processes on the same queue.
pub fn context() -> Result<&'static Mutex<Context>> {
    static CTX: OnceLock<Mutex<Context>> = OnceLock::new();
    if let Some(c) = CTX.get() {
        return Ok(c);
    }
    let device = std::env::var("CITYPROVER_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0usize);
    let ctx = Context::new(device)?;
    if let Some(l) = std::env::var("CITYPROVER_LANES").ok().and_then(|s| s.parse().ok()) {
        ctx.set_lanes(l)?;
    }
    Ok(CTX.get_or_init(|| Mutex::new(ctx)))
}

/// `CITYPROVER_BATCH=<max_batch>` (with `CITYPROVER_LINGER_US`, default 300): the process runs SEVERAL worker loops as threads
/// (N x `SimpleActorWorker::run_worker` over one toolbox, city_rollup_core_worker/src/actors/simple.rs:32-56) and their
/// one-proof `prove` calls are merged into batches by a `cp_batcher` instead of queueing on the context's mutex — measured at
/// the rate of an explicitly batching worker. The batcher proves on the context's LANES (at least two are
/// forced: `CITYPROVER_LANES`, default 4), so that the context itself stays free for circuit loads under the mutex.
pub fn batcher() -> Result<Option<&'static Batcher>> {
    static B: OnceLock<Option<Batcher>> = OnceLock::new();
    if let Some(b) = B.get() {
        return Ok(b.as_ref());
    }
    let Some(max_batch) = std::env::var("CITYPROVER_BATCH").ok().and_then(|s| s.parse::<usize>().ok()) else {
        return Ok(B.get_or_init(|| None).as_ref());
    };
    let linger = std::env::var("CITYPROVER_LINGER_US").ok().and_then(|s| s.parse().ok()).unwrap_or(300u32);
    let ctx = context()?.lock().unwrap();
    let lanes = std::env::var("CITYPROVER_LANES").ok().and_then(|s| s.parse().ok()).unwrap_or(4usize).max(2);
    ctx.set_lanes(lanes)?;
    let b = Batcher::new(&ctx, max_batch, linger)?;
    Ok(B.get_or_init(|| Some(b)).as_ref())
}

/// `gate.0.id()` is the Debug form of the gate struct, e.g. "ArithmeticGate { num_ops: 20 }",
/// "BaseSumGate { num_limbs: 63 } + Base: 2", "RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2, _phantom: .. }<D=2>".
/// Gate type ids and parameter order: include/cityprover.h (CP_GATE_*), INTEGRATION.md §4.
pub fn cp_gate_of(gate: &GateRef<F, D>, selector: usize, group: &Range<usize>) -> Result<CpGate> {
    let id = gate.0.id();
    let field = |name: &str| -> i32 {
        id.split(&format!("{name}: "))
            .nth(1)
            .and_then(|t| t.split(|c: char| !c.is_ascii_digit()).next().and_then(|d| d.parse().ok()))
            .unwrap_or(0)
    };
    let name = id.split(|c: char| c == ' ' || c == '<' || c == '(' || c == '{').next().unwrap_or("");
    let (ty, p1, p2, p3) = match name {
        "NoopGate" => (0, 0, 0, 0),
        "ConstantGate" => (1, field("num_consts"), 0, 0),
        "PublicInputGate" => (2, 0, 0, 0),
        "ArithmeticGate" => (3, field("num_ops"), 0, 0),
        "PoseidonGate" => (4, 0, 0, 0),
        "ComparisonGate" => (5, field("num_bits"), field("num_chunks"), 0),
        "U32ArithmeticGate" => (6, field("num_ops"), 0, 0),
        "U32RangeCheckGate" => (7, field("num_input_limbs"), 0, 0),
        "U32AddManyGate" => (8, field("num_ops"), field("num_addends"), 0),
        "U32SubtractionGate" => (9, field("num_ops"), 0, 0),
        "U32InterleaveGate" => (10, field("num_ops"), 0, 0),
        "UninterleaveToU32Gate" => (11, field("num_ops"), 0, 0),
        "UninterleaveToB32Gate" => (12, field("num_ops"), 0, 0),
        "ArithmeticExtensionGate" => (13, field("num_ops"), 0, 0),
        "MulExtensionGate" => (14, field("num_ops"), 0, 0),
        "BaseSumGate" => (15, field("num_limbs"), field("Base"), 0),
        "RandomAccessGate" => (16, field("bits"), field("num_copies"), field("num_extra_constants")),
        "ReducingGate" => (17, field("num_coeffs"), 0, 0),
        "ReducingExtensionGate" => (18, field("num_coeffs"), 0, 0),
        "PoseidonMdsGate" => (19, 0, 0, 0),
        "CosetInterpolationGate" => (20, field("subgroup_bits"), field("degree"), 0),
        "ExponentiationGate" => (21, field("num_power_bits"), 0, 0),
        other => bail!("gate {other} ({id}) has no GPU constraint kernel: keep the CPU prover for this circuit"),
    };
    Ok(CpGate {
        type_: ty,
        selector_index: selector as i32,
        group_start: group.start as i32,
        group_end: group.end as i32,
        param: p1,
        param2: p2,
        param3: p3,
    })
}

/// Everything `cp_circuit_load` + `cp_circuit_set_gates` need, as flat data.
pub struct FlatCircuit {
    pub shape: CpShape,
    pub digest: [u64; 4],
    pub gates: Vec<CpGate>,
    pub num_selectors: usize,
    pub k_is: Vec<u64>,
    /// (row, wire) of each public input in the wire matrix (`ProverOnlyCircuitData::public_inputs`)
    pub public_input_targets: Vec<(u32, u32)>,
    /// constants (selectors first) then sigmas, COEFFICIENT form, `[num_constants + num_routed_wires][n]`
    pub cs_coeffs: Vec<u64>,
}

pub fn flatten_circuit(common: &CommonCircuitData<F, D>, prover_only: &ProverOnlyCircuitData<F, C, D>) -> Result<FlatCircuit> {
    let cfg = &common.config;
    let fri = &common.fri_params;
    ensure!(common.num_lookup_polys == 0, "lookup tables are not supported by the GPU prover");
    ensure!(fri.reduction_arity_bits.len() <= 8, "more than 8 FRI reduction layers");
    let mut arity_bits = [0i32; 8];
    for (i, a) in fri.reduction_arity_bits.iter().enumerate() {
        arity_bits[i] = *a as i32;
    }
    let shape = CpShape {
        degree_bits: common.degree_bits() as i32,
        num_constants: common.num_constants as i32, // selectors + gate constants
        num_routed_wires: cfg.num_routed_wires as i32,
        num_wires: cfg.num_wires as i32,
        num_challenges: cfg.num_challenges as i32,
        num_partial_products: common.num_partial_products as i32,
        quotient_degree_factor: common.quotient_degree_factor as i32,
        rate_bits: fri.config.rate_bits as i32,
        cap_height: fri.config.cap_height as i32,
        pow_bits: fri.config.proof_of_work_bits as i32,
        num_query_rounds: fri.config.num_query_rounds as i32,
        n_arity: fri.reduction_arity_bits.len() as i32,
        arity_bits,
        zero_knowledge: cfg.zero_knowledge as i32,
        num_public_inputs: common.num_public_inputs as i32,
    };
    let sel = &common.selectors_info;
    let gates = common
        .gates
        .iter()
        .enumerate()
        .map(|(i, g)| {
            let s = sel.selector_indices[i];
            cp_gate_of(g, s, &sel.groups[s])
        })
        .collect::<Result<Vec<_>>>()?;
    let digest_felts = prover_only.circuit_digest.elements; // HashOut<F>
    let digest = [
        digest_felts[0].to_canonical_u64(),
        digest_felts[1].to_canonical_u64(),
        digest_felts[2].to_canonical_u64(),
        digest_felts[3].to_canonical_u64(),
    ];
    let public_input_targets = prover_only
        .public_inputs
        .iter()
        .map(|t| match t {
            Target::Wire(w) => Ok((w.row as u32, w.column as u32)),
            Target::VirtualTarget { .. } => bail!("a public input is a virtual target"),
        })
        .collect::<Result<Vec<_>>>()?;
    let n = 1usize << shape.degree_bits;
    let polys = &prover_only.constants_sigmas_commitment.polynomials;
    ensure!(polys.len() == (shape.num_constants + shape.num_routed_wires) as usize, "unexpected number of constants/sigmas polynomials");
    let mut cs_coeffs = Vec::with_capacity(polys.len() * n);
    for p in polys {
        ensure!(p.coeffs.len() == n, "constants/sigmas polynomial of unexpected length");
        cs_coeffs.extend(p.coeffs.iter().map(|c| c.to_canonical_u64()));
    }
    Ok(FlatCircuit {
        shape,
        digest,
        gates,
        num_selectors: sel.num_selectors(),
        k_is: common.k_is.iter().map(|k| k.to_canonical_u64()).collect(),
        public_input_targets,
        cs_coeffs,
    })
}

fn fnv1a64(data: &[u8]) -> u64 {
    data.iter().fold(0xcbf2_9ce4_8422_2325u64, |h, b| (h ^ *b as u64).wrapping_mul(0x0000_0100_0000_01b3))
}

impl FlatCircuit {
    /// `.cpcirc`, coefficient form (flags = 1 | 2 | 4); layout: csrc/circuit_file.inc of the cityprover repository.
    pub fn write_file(&self, path: &Path) -> Result<()> {
        let mut b: Vec<u8> = Vec::new();
        b.extend_from_slice(b"CPCIRCv1");
        b.extend_from_slice(&1u32.to_le_bytes());
        b.extend_from_slice(&(1u32 | 2 | 4).to_le_bytes());
        b.extend_from_slice(&0u64.to_le_bytes()); // total, patched below
        let s = &self.shape;
        let mut ints = vec![
            s.degree_bits, s.num_constants, s.num_routed_wires, s.num_wires, s.num_challenges, s.num_partial_products,
            s.quotient_degree_factor, s.rate_bits, s.cap_height, s.pow_bits, s.num_query_rounds, s.n_arity,
        ];
        ints.extend_from_slice(&s.arity_bits);
        ints.extend_from_slice(&[s.zero_knowledge, s.num_public_inputs, 0, 0]);
        for v in ints {
            b.extend_from_slice(&v.to_le_bytes());
        }
        for d in self.digest {
            b.extend_from_slice(&d.to_le_bytes());
        }
        b.extend_from_slice(&(self.num_selectors as u32).to_le_bytes());
        b.extend_from_slice(&(self.gates.len() as u32).to_le_bytes());
        for g in &self.gates {
            for v in [g.type_, g.selector_index, g.group_start, g.group_end, g.param, g.param2, g.param3] {
                b.extend_from_slice(&v.to_le_bytes());
            }
        }
        if self.gates.len() % 2 == 1 {
            b.extend_from_slice(&[0u8; 4]);
        }
        for k in &self.k_is {
            b.extend_from_slice(&k.to_le_bytes());
        }
        for (row, wire) in &self.public_input_targets {
            b.extend_from_slice(&row.to_le_bytes());
            b.extend_from_slice(&wire.to_le_bytes());
        }
        for c in &self.cs_coeffs {
            b.extend_from_slice(&c.to_le_bytes());
        }
        let total = (b.len() + 8) as u64;
        b[16..24].copy_from_slice(&total.to_le_bytes());
        let sum = fnv1a64(&b);
        b.extend_from_slice(&sum.to_le_bytes());
        std::fs::File::create(path)?.write_all(&b)?;
        Ok(())
    }
}

/// `.cpwit`: the wire matrix + public inputs of one proof and, optionally, the proof the CPU prover made from them
/// (the harness then injects its `pow_witness` and requires identical bytes). Layout: tools/qbench/pack.h.
pub fn write_witness_file(path: &Path, digest: [u64; 4], witness: &MatrixWitness<F>, public_inputs: &[F], cpu_proof: Option<&[u8]>) -> Result<()> {
    let num_wires = witness.wire_values.len();
    let n = witness.wire_values[0].len();
    let mut b: Vec<u8> = Vec::new();
    b.extend_from_slice(b"CPWITNv1");
    b.extend_from_slice(&1u32.to_le_bytes());
    b.extend_from_slice(&(cpu_proof.is_some() as u32).to_le_bytes());
    for d in digest {
        b.extend_from_slice(&d.to_le_bytes());
    }
    for v in [num_wires as u32, n.trailing_zeros(), public_inputs.len() as u32, 0u32] {
        b.extend_from_slice(&v.to_le_bytes());
    }
    for p in public_inputs {
        b.extend_from_slice(&p.to_canonical_u64().to_le_bytes());
    }
    for column in &witness.wire_values {
        for v in column {
            b.extend_from_slice(&v.to_canonical_u64().to_le_bytes());
        }
    }
    if let Some(p) = cpu_proof {
        b.extend_from_slice(&(p.len() as u64).to_le_bytes());
        b.extend_from_slice(p);
        b.resize(b.len() + (8 - p.len() % 8) % 8, 0);
    }
    let sum = fnv1a64(&b);
    b.extend_from_slice(&sum.to_le_bytes());
    std::fs::File::create(path)?.write_all(&b)?;
    Ok(())
}

/// The GPU twin of one `CircuitData`, created on first use and kept for the circuit's lifetime.
pub struct GpuHandle {
    pub circuit: GpuCircuit,
}

pub fn load_gpu_circuit(
    common: &CommonCircuitData<F, D>,
    prover_only: &ProverOnlyCircuitData<F, C, D>,
    verifier_only: &VerifierOnlyCircuitData<C, D>,
) -> Result<GpuHandle> {
    let flat = flatten_circuit(common, prover_only)?;
    // zero-knowledge circuits (standard_recursion_zk_config: the user-side signature circuits) need leaf salts from plonky2's RNG;
    // `prove_gpu` below calls the plain entry points, which refuse them — such a circuit stays on the CPU prover
    ensure!(flat.shape.zero_knowledge == 0, "zero-knowledge circuit: salts are drawn by the CPU prover");
    if let Ok(dir) = std::env::var("CITYPROVER_DUMP_DIR") {
        // one file per circuit, named by its digest: the input of the native q-bench harness
        let name = format!("{:016x}{:016x}.cpcirc", flat.digest[0], flat.digest[1]);
        flat.write_file(&Path::new(&dir).join(name))?;
    }
    // the library takes VALUES over the subgroup: one FFT per polynomial (only here, once per circuit)
    let n = 1usize << flat.shape.degree_bits;
    let mut values = Vec::with_capacity(flat.cs_coeffs.len());
    for p in &prover_only.constants_sigmas_commitment.polynomials {
        values.extend(p.clone().fft().values.iter().map(|v| v.to_canonical_u64()));
    }
    debug_assert_eq!(values.len(), flat.cs_coeffs.len());
    let _ = n;
    let ctx = context()?.lock().unwrap();
    let circuit = GpuCircuit::load(&ctx, &flat.shape, flat.digest, &values, &flat.k_is, &flat.gates, flat.num_selectors)?;
    // first parity check on a real circuit: the GPU's constants_sigmas_cap must be plonky2's
    let cap = circuit.constants_sigmas_cap()?;
    for (mine, theirs) in cap.iter().zip(verifier_only.constants_sigmas_cap.0.iter()) {
        let t = theirs.elements.map(|e| e.to_canonical_u64());
        ensure!(*mine == t, "constants_sigmas_cap computed on the GPU differs from plonky2's");
    }
    Ok(GpuHandle { circuit })
}

/// `prove_with_partition_witness` after witness generation: the wire matrix goes to the GPU, the bincode bytes come back.
/// `Ok(None)`: the backend REFUSED the request (CP_ERR_INVALID_ARG / CP_ERR_UNSUPPORTED) — the caller proves on the CPU; any
/// other failure (device error, out of memory) is an `Err`, as it is for the CPU prover.
pub fn prove_gpu_or_refuse(handle: &GpuHandle, witness: &MatrixWitness<F>, public_inputs: &[F], pow_witness: Option<u64>) -> Result<Option<ProofWithPublicInputs<F, C, D>>> {
    match prove_gpu(handle, witness, public_inputs, pow_witness) {
        Ok(p) => Ok(Some(p)),
        Err(e) if cityprover_sys::is_refusal(&e) => {
            log::warn!("cityprover: request refused, proving on the CPU: {e}");
            Ok(None)
        }
        Err(e) => Err(e),
    }
}

pub fn prove_gpu(handle: &GpuHandle, witness: &MatrixWitness<F>, public_inputs: &[F], pow_witness: Option<u64>) -> Result<ProofWithPublicInputs<F, C, D>> {
    let wires: Vec<u64> = witness.wire_values.iter().flat_map(|col| col.iter().map(|v| v.to_canonical_u64())).collect();
    let pis: Vec<u64> = public_inputs.iter().map(|v| v.to_canonical_u64()).collect();
    let bytes = match batcher()? {
        // several worker threads: no lock — the batcher merges whatever calls are in flight
        Some(b) => b.prove(&handle.circuit, &wires, &pis, pow_witness)?,
        None => {
            let ctx = context()?.lock().unwrap();
            ctx.prove_batch(&[&handle.circuit], &[&wires], &[&pis], &[pow_witness])?.pop().expect("one proof")
        }
    };
    // the bytes ARE the bincode the worker stores (city_redis_store/src/lib.rs:71-83)
    Ok(bincode::deserialize(&bytes)?)
}


// ---- the two generic seams: PolynomialBatch::from_values and PolynomialBatch::prove_openings ---------------------------------
// Every FRI-based prover in the process goes through these two functions of plonky2 — `CircuitData::prove` (which the hook
// above replaces wholesale) and starkyx's `ByteStark::prove` (the SHA-256 STARK of the sighash circuit:
// city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524, 418 + 912 columns, three proofs per block). With the
// hooks of hooks.patch, a `PolynomialBatch` built by `from_values` carries a GPU twin, and `prove_openings` runs on the device
// when every oracle has one. UNCOMPILED, from memory of plonky2 0.2.2: names to re-check — `PolynomialBatch::{polynomials,
// merkle_tree, degree_log, rate_bits, blinding}`, `MerkleTree::{leaves, digests, cap}`, `FriInstanceInfo::{oracles, batches}`,
// `FriBatchInfo::{point, polynomials}`, `FriPolynomialInfo::{oracle_index, polynomial_index}`, `Challenger::{sponge_state,
// input_buffer, output_buffer}` (private fields: this module lives inside the crate), `FriParams::{config, degree_bits,
// reduction_arity_bits, hiding}`, `SALT_SIZE`.
use cityprover_sys::ffi::{CpChallengerState, CpFriParams, CpFriPolyRange};
use cityprover_sys::{FriBatch, PolyBatch};

use crate::field::extension::quadratic::QuadraticExtension;
use crate::field::polynomial::{PolynomialCoeffs, PolynomialValues};
use crate::fri::oracle::PolynomialBatch;
use crate::fri::proof::FriProof;
use crate::fri::structure::FriInstanceInfo;
use crate::fri::FriParams;
use crate::hash::hash_types::HashOut;
use crate::hash::merkle_tree::{MerkleCap, MerkleTree};
use crate::hash::poseidon::PoseidonHash;
use crate::iop::challenger::Challenger;

/// Smallest commitment worth a round trip: below this the CPU is faster than the copies (override: CITYPROVER_MIN_COMMIT).
fn min_commit_elements() -> usize {
    std::env::var("CITYPROVER_MIN_COMMIT").ok().and_then(|s| s.parse().ok()).unwrap_or(1 << 18)
}

/// `PolynomialBatch::from_values` on the GPU: iNTT, coset LDE, Merkle tree. Returns the host-side `PolynomialBatch` plonky2
/// expects — `polynomials` (coefficients) and `merkle_tree.{leaves, cap}` fetched from the device, so that `get_lde_values`,
/// `get_lde_values_packed` and `eval` of CPU callers (a STARK's constraint evaluation) keep working — with the device twin
/// attached. `merkle_tree.digests` stays EMPTY: the only consumer of the inner nodes is `prove_openings`, which runs on the
/// device when the twin is there; when it cannot (`prove_openings_gpu` -> `None`), the hook rebuilds the host tree first
/// (`with_host_tree`). `None`: not worth it / not possible (blinding needs plonky2's RNG: the salts are drawn HERE, on the
/// Rust side, and handed over).
pub fn batch_from_values_gpu(values: &[PolynomialValues<F>], rate_bits: usize, blinding: bool, cap_height: usize) -> Result<Option<PolynomialBatch<F, C, D>>> {
    let k = values.len();
    if k == 0 {
        return Ok(None);
    }
    let n = values[0].len();
    let flat: Vec<u64> = values.iter().flat_map(|p| p.values.iter().map(|v| v.to_canonical_u64())).collect();
    batch_commit_gpu(&flat, k, n, rate_bits, blinding, cap_height, false, None)
}

/// `PolynomialBatch::from_coeffs` on the GPU (the quotient of a STARK prover goes through this constructor, its traces through
/// `from_values`: both must have twins for `prove_openings` to run on the device — ADVICE r3). `polynomials` are padded to a
/// common power-of-two length by the caller (plonky2 asserts it).
pub fn batch_from_coeffs_gpu(polynomials: &[PolynomialCoeffs<F>], rate_bits: usize, blinding: bool, cap_height: usize) -> Result<Option<PolynomialBatch<F, C, D>>> {
    let k = polynomials.len();
    if k == 0 {
        return Ok(None);
    }
    let n = polynomials[0].len();
    if polynomials.iter().any(|p| p.len() != n) {
        return Ok(None);
    }
    let flat: Vec<u64> = polynomials.iter().flat_map(|p| p.coeffs.iter().map(|v| v.to_canonical_u64())).collect();
    batch_commit_gpu(&flat, k, n, rate_bits, blinding, cap_height, true, Some(polynomials))
}

fn batch_commit_gpu(flat: &[u64], k: usize, n: usize, rate_bits: usize, blinding: bool, cap_height: usize, from_coeffs: bool,
                    host_polys: Option<&[PolynomialCoeffs<F>]>) -> Result<Option<PolynomialBatch<F, C, D>>> {
    if k * n < min_commit_elements() || !n.is_power_of_two() {
        return Ok(None);
    }
    let degree_bits = n.trailing_zeros() as usize;
    let big_n = n << rate_bits;
    let salts: Option<Vec<u64>> = blinding.then(|| (0..crate::plonk::plonk_common::SALT_SIZE * big_n).map(|_| F::rand().to_canonical_u64()).collect());
    let ctx = context()?.lock().unwrap();
    let twin = match PolyBatch::commit(&ctx, flat, k, degree_bits, rate_bits, cap_height, from_coeffs, salts.as_deref()) {
        Ok(t) => t,
        Err(e) if cityprover_sys::is_refusal(&e) => return Ok(None),
        Err(e) => return Err(e),
    };
    // host mirror
    let cap = MerkleCap(twin.cap()?.into_iter().map(|h| HashOut { elements: h.map(F::from_canonical_u64) }).collect());
    let width = k + if blinding { crate::plonk::plonk_common::SALT_SIZE } else { 0 };
    let mut leaves = Vec::with_capacity(big_n);
    const SLICE: usize = 1 << 14; // leaves per copy
    for first in (0..big_n).step_by(SLICE) {
        let cnt = SLICE.min(big_n - first);
        let rows = twin.leaves(first, cnt)?;
        leaves.extend(rows.chunks_exact(width).map(|r| r.iter().map(|v| F::from_canonical_u64(*v)).collect::<Vec<_>>()));
    }
    // coefficients: the caller's own for from_coeffs, else one copy (cp_batch_coeffs: k x n u64; the handle's device pointers
    // are NOT taken, so its buffers stay in the library's recycling pool)
    let polynomials = match host_polys {
        Some(p) => p.to_vec(),
        None => twin.coeffs()?.chunks_exact(n).map(|p| PolynomialCoeffs::new(p.iter().map(|v| F::from_canonical_u64(*v)).collect())).collect(),
    };
    Ok(Some(PolynomialBatch {
        polynomials,
        merkle_tree: MerkleTree { leaves, digests: Vec::new(), cap },
        degree_log: degree_bits,
        rate_bits,
        blinding,
        gpu: Some(std::sync::Arc::new(twin)),
    }))
}

/// A device twin for an oracle that was committed on the CPU (below CITYPROVER_MIN_COMMIT, or by a constructor without a hook),
/// made on demand so that one small oracle does not send a whole `prove_openings` back to the CPU: its coefficients and — for a
/// blinded oracle — the salts its host leaves end in. The twin's cap must equal the host tree's. `None`: refused by the backend.
fn twin_on_demand(ctx: &cityprover_sys::Context, o: &PolynomialBatch<F, C, D>) -> Result<Option<PolyBatch>> {
    let k = o.polynomials.len();
    let n = 1usize << o.degree_log;
    if k == 0 || o.polynomials.iter().any(|p| p.len() != n) {
        return Ok(None);
    }
    let big_n = n << o.rate_bits;
    let flat: Vec<u64> = o.polynomials.iter().flat_map(|p| p.coeffs.iter().map(|v| v.to_canonical_u64())).collect();
    let salt_size = crate::plonk::plonk_common::SALT_SIZE;
    let salts: Option<Vec<u64>> = o.blinding.then(|| {
        // cp_batch_commit takes salts as SALT_SIZE x N indexed by leaf; host leaf i ends in its SALT_SIZE salt elements
        let mut s = vec![0u64; salt_size * big_n];
        for (i, leaf) in o.merkle_tree.leaves.iter().enumerate() {
            for j in 0..salt_size {
                s[j * big_n + i] = leaf[k + j].to_canonical_u64();
            }
        }
        s
    });
    let twin = match PolyBatch::commit(ctx, &flat, k, o.degree_log, o.rate_bits, o.merkle_tree.cap.height(), true, salts.as_deref()) {
        Ok(t) => t,
        Err(e) if cityprover_sys::is_refusal(&e) => return Ok(None),
        Err(e) => return Err(e),
    };
    let cap: Vec<[u64; 4]> = twin.cap()?;
    let same = cap.len() == o.merkle_tree.cap.0.len()
        && cap.iter().zip(&o.merkle_tree.cap.0).all(|(a, b)| a.iter().zip(&b.elements).all(|(x, y)| *x == y.to_canonical_u64()));
    ensure!(same, "cityprover: the device commitment of a CPU-committed oracle has a different cap");
    Ok(Some(twin))
}

/// The host Merkle tree of a device-committed oracle, rebuilt from the leaves the mirror holds: what the CPU path of
/// `prove_openings` needs (`merkle_tree.prove(index)` walks `digests`). Used by the hook when the device path declines.
pub fn with_host_tree(o: &PolynomialBatch<F, C, D>) -> PolynomialBatch<F, C, D> {
    PolynomialBatch {
        polynomials: o.polynomials.clone(),
        merkle_tree: MerkleTree::new(o.merkle_tree.leaves.clone(), o.merkle_tree.cap.height()),
        degree_log: o.degree_log,
        rate_bits: o.rate_bits,
        blinding: o.blinding,
        gpu: None,
    }
}

fn challenger_to_c(ch: &Challenger<F, PoseidonHash>) -> CpChallengerState {
    let mut s: CpChallengerState = unsafe { std::mem::zeroed() };
    for (d, v) in s.sponge_state.iter_mut().zip(ch.sponge_state.as_ref()) {
        *d = v.to_canonical_u64();
    }
    for (d, v) in s.input_buffer.iter_mut().zip(&ch.input_buffer) {
        *d = v.to_canonical_u64();
    }
    for (d, v) in s.output_buffer.iter_mut().zip(&ch.output_buffer) {
        *d = v.to_canonical_u64();
    }
    s.n_input = ch.input_buffer.len() as u32;
    s.n_output = ch.output_buffer.len() as u32;
    s
}

fn challenger_from_c(s: &CpChallengerState, ch: &mut Challenger<F, PoseidonHash>) {
    for (d, v) in ch.sponge_state.as_mut().iter_mut().zip(s.sponge_state) {
        *d = F::from_canonical_u64(v);
    }
    ch.input_buffer = s.input_buffer[..s.n_input as usize].iter().map(|v| F::from_canonical_u64(*v)).collect();
    ch.output_buffer = s.output_buffer[..s.n_output as usize].iter().map(|v| F::from_canonical_u64(*v)).collect();
}

/// `PolynomialBatch::prove_openings` on the GPU when every oracle carries a device twin; `None` otherwise (the caller runs
/// plonky2's own code). The polynomial lists of the instance are compressed into runs; the bytes that come back are the
/// bincode of `FriProof<F, PoseidonHash, D>`.
pub fn prove_openings_gpu(
    instance: &FriInstanceInfo<F, D>,
    oracles: &[&PolynomialBatch<F, C, D>],
    challenger: &mut Challenger<F, PoseidonHash>,
    fri_params: &FriParams,
) -> Result<Option<FriProof<F, PoseidonHash, D>>> {
    if oracles.len() > 8 || fri_params.reduction_arity_bits.len() > 8 || oracles.iter().all(|o| o.gpu.is_none()) {
        return Ok(None); // nothing lives on the device: the CPU path as it is
    }
    // oracles without a twin (committed on the CPU: small, or through an unhooked constructor) get one on demand
    let ctx = context()?.lock().unwrap();
    let mut made: Vec<Option<PolyBatch>> = Vec::with_capacity(oracles.len());
    for o in oracles {
        made.push(match &o.gpu {
            Some(_) => None,
            None => match twin_on_demand(&ctx, o)? {
                Some(t) => Some(t),
                None => return Ok(None),
            },
        });
    }
    let twins: Vec<&PolyBatch> = oracles.iter().zip(&made).map(|(o, m)| m.as_ref().unwrap_or_else(|| o.gpu.as_deref().unwrap())).collect();
    let batches: Vec<FriBatch> = instance
        .batches
        .iter()
        .map(|b| {
            let mut ranges: Vec<CpFriPolyRange> = Vec::new();
            for p in &b.polynomials {
                match ranges.last_mut() {
                    Some(r) if r.oracle as usize == p.oracle_index && (r.first + r.count) as usize == p.polynomial_index => r.count += 1,
                    _ => ranges.push(CpFriPolyRange { oracle: p.oracle_index as u32, first: p.polynomial_index as u32, count: 1 }),
                }
            }
            let pt: QuadraticExtension<F> = b.point;
            FriBatch { point: [pt.0[0].to_canonical_u64(), pt.0[1].to_canonical_u64()], ranges }
        })
        .collect();
    let mut arity_bits = [0i32; 8];
    for (d, a) in arity_bits.iter_mut().zip(&fri_params.reduction_arity_bits) {
        *d = *a as i32;
    }
    let params = CpFriParams {
        degree_bits: fri_params.degree_bits as i32,
        rate_bits: fri_params.config.rate_bits as i32,
        cap_height: fri_params.config.cap_height as i32,
        pow_bits: fri_params.config.proof_of_work_bits as i32,
        num_query_rounds: fri_params.config.num_query_rounds as i32,
        n_arity: fri_params.reduction_arity_bits.len() as i32,
        arity_bits,
    };
    let mut state = challenger_to_c(challenger);
    let bytes = match cityprover_sys::fri_prove(&ctx, &twins, &batches, &params, &mut state, None) {
        Ok(b) => b,
        Err(e) if cityprover_sys::is_refusal(&e) => return Ok(None),
        Err(e) => return Err(e),
    };
    challenger_from_c(&state, challenger);
    Ok(Some(bincode::deserialize(&bytes)?))
}
