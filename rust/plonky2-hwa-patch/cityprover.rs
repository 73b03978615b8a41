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
