//! A RECORDING implementation of starkyx's constraint parser: instead of evaluating an AIR's constraints on field elements
//! it writes every parser call down as one op of a flat straight-line program (include/cityprover.h `cp_air_op`), which the
//! HIP library then evaluates at every point of the quotient coset (`cp_air_quotient_commit`) — nothing of the AIR is
//! restated anywhere, the AIR itself produces the program by being run once against this parser.
//!
//! Where it goes: a fork of `starkyx` 0.1.0 (git QEDProtocol/starkyx @ a53ea106 — already a `[patch]` target of the workspace,
//! /root/reference/Cargo.toml:131-132), new module `src/plonky2/stark/gpu.rs`; the one call that changes is the quotient step of
//! `StarkyProver::prove` (what `ByteStark::prove` runs: city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-522).
//!
//! NOT COMPILED in the build image (no cargo / rustc, and the starkyx crate is not in the tree). Written from memory of
//! starkyx's `air::parser::AirParser` / `air::RAir` / `plonky2::stark::prover::StarkyProver`; rust/README.md lists every name
//! relied upon. What IS checked here without Rust: the op codes and the `cp_air_op` / `cp_air_program_desc` layouts against the
//! header (tests/test_rust_bridge.py), and the program semantics end to end on the device through a Python recorder of the
//! same shape (tests/air_programs.py `Builder` / `RecField`: a toy AIR with a cubic-extension lookup recorded, proved, verified).

use cityprover_sys::ffi::{self, CpAirOp, CpAirProgramDesc};
use cityprover_sys::{AirProgram, Context, PolyBatch};

use crate::air::parser::AirParser;
use crate::air::RAir;
use crate::math::prelude::*;

type F = GoldilocksField;

/// `Var` of the recording parser: the index of the op that defines the value.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct Rec(pub u32);

pub struct RecordingParser {
    ops: Vec<CpAirOp>,
    consts: Vec<u64>,
    const_index: std::collections::HashMap<u64, u32>,
    local: Vec<Rec>,
    next: Vec<Rec>,
    challenges: Vec<Rec>,
    globals: Vec<Rec>,
    publics: Vec<Rec>,
}

impl RecordingParser {
    /// Slices are recorded up front: one load op per column / challenge / global / public, so that `local_slice()` etc. can
    /// hand out `&[Rec]` like any other parser (loads nobody uses are dropped by the library's dead-value elimination).
    pub fn new(n_columns: usize, n_challenges: usize, n_globals: usize, n_publics: usize) -> Self {
        let mut p = Self { ops: Vec::new(), consts: Vec::new(), const_index: Default::default(), local: vec![], next: vec![], challenges: vec![], globals: vec![], publics: vec![] };
        p.local = (0..n_columns).map(|c| p.emit(ffi::CP_AIR_LOCAL, c as u32, 0)).collect();
        p.next = (0..n_columns).map(|c| p.emit(ffi::CP_AIR_NEXT, c as u32, 0)).collect();
        p.challenges = (0..n_challenges).map(|c| p.emit(ffi::CP_AIR_CHALLENGE, c as u32, 0)).collect();
        p.globals = (0..n_globals).map(|c| p.emit(ffi::CP_AIR_GLOBAL, c as u32, 0)).collect();
        p.publics = (0..n_publics).map(|c| p.emit(ffi::CP_AIR_PUBLIC, c as u32, 0)).collect();
        p
    }

    fn emit(&mut self, op: i32, a: u32, b: u32) -> Rec {
        self.ops.push(CpAirOp { op: op as u32, a, b, c: 0 });
        Rec((self.ops.len() - 1) as u32)
    }

    /// Runs the AIR once against the recorder (`RAir::eval`, the very call the CPU prover makes per row) and compiles the result.
    pub fn record<A: RAir<Self>>(air: &A, ctx: &Context, n_columns: usize, n_challenges: usize, n_globals: usize, n_publics: usize) -> anyhow::Result<AirProgram> {
        let mut p = Self::new(n_columns, n_challenges, n_globals, n_publics);
        air.eval(&mut p);
        let desc = CpAirProgramDesc {
            kind: ffi::CP_AIR_CONSTRAINTS,
            ops: p.ops.as_ptr(),
            n_ops: p.ops.len(),
            consts: p.consts.as_ptr(),
            n_consts: p.consts.len(),
            n_columns: n_columns as u32,
            n_public: n_publics as u32,
            n_global: n_globals as u32,
            n_challenge: n_challenges as u32,
            n_out_columns: 0,
        };
        AirProgram::create(ctx, &desc)
    }
}

impl AirParser for RecordingParser {
    type Field = F;
    type Var = Rec;

    fn local_slice(&self) -> &[Rec] { &self.local }
    fn next_slice(&self) -> &[Rec] { &self.next }
    fn challenge_slice(&self) -> &[Rec] { &self.challenges }
    fn global_slice(&self) -> &[Rec] { &self.globals }
    fn public_slice(&self) -> &[Rec] { &self.publics }

    fn constraint(&mut self, c: Rec) { self.emit(ffi::CP_AIR_ASSERT_ZERO, c.0, 0); }
    fn constraint_transition(&mut self, c: Rec) { self.emit(ffi::CP_AIR_ASSERT_ZERO_TRANSITION, c.0, 0); }
    fn constraint_first_row(&mut self, c: Rec) { self.emit(ffi::CP_AIR_ASSERT_ZERO_FIRST_ROW, c.0, 0); }
    fn constraint_last_row(&mut self, c: Rec) { self.emit(ffi::CP_AIR_ASSERT_ZERO_LAST_ROW, c.0, 0); }

    fn constant(&mut self, value: F) -> Rec {
        let v = value.to_canonical_u64();
        let ix = match self.const_index.get(&v) {
            Some(ix) => *ix,
            None => {
                let ix = self.consts.len() as u32;
                self.consts.push(v);
                self.const_index.insert(v, ix);
                ix
            }
        };
        self.emit(ffi::CP_AIR_CONST, ix, 0)
    }

    fn add(&mut self, a: Rec, b: Rec) -> Rec { self.emit(ffi::CP_AIR_ADD, a.0, b.0) }
    fn sub(&mut self, a: Rec, b: Rec) -> Rec { self.emit(ffi::CP_AIR_SUB, a.0, b.0) }
    fn neg(&mut self, a: Rec) -> Rec { self.emit(ffi::CP_AIR_NEG, a.0, 0) }
    fn mul(&mut self, a: Rec, b: Rec) -> Rec { self.emit(ffi::CP_AIR_MUL, a.0, b.0) }
    // add_const / sub_const / mul_const / zero / one / sum / assert_eq and the extension-field helpers of `CubicParser` /
    // `PolynomialParser` are DEFAULT methods of the traits, written in terms of the five calls above: a cubic product becomes
    // nine recorded multiplications and its additions, exactly as the CPU parser would have executed them.
}
// marker impls: `impl<E: CubicParameters<F>> CubicParser<E> for RecordingParser {}`, `impl PolynomialParser for RecordingParser {}`

/// The quotient step of `StarkyProver::prove` on the device. `trace_twins`: the device twins of the trace commitments
/// (`PolynomialBatch::gpu`, set by the `from_values` hook of rust/plonky2-hwa-patch); returns the committed quotient chunks
/// (`num_challenges * 2^quotient_degree_bits` polynomials, challenge-major) as a twin for `prove_openings`.
/// `None`: a trace round without a twin, or a refusal of the backend — the caller carries on with the CPU evaluator.
pub fn quotient_on_gpu(
    ctx: &Context,
    program: &AirProgram,
    trace_twins: &[Option<&PolyBatch>],
    quotient_degree_bits: usize,
    publics: &[F],
    globals: &[F],
    challenges: &[F],
    alphas: &[F],
) -> anyhow::Result<Option<PolyBatch>> {
    let twins: Option<Vec<&PolyBatch>> = trace_twins.iter().copied().collect();
    let Some(twins) = twins else { return Ok(None) };
    let u = |v: &[F]| v.iter().map(|x| x.to_canonical_u64()).collect::<Vec<u64>>();
    match cityprover_sys::air_quotient_commit(ctx, program, &twins, quotient_degree_bits, &u(publics), &u(globals), &u(challenges), &u(alphas)) {
        Ok(q) => Ok(Some(q)),
        Err(e) if cityprover_sys::is_refusal(&e) => Ok(None),
        Err(e) => Err(e),
    }
}

// The hunk in `StarkyProver::prove` (src/plonky2/stark/prover.rs), after the alphas are drawn:
//
//     #[cfg(feature = "cityprover")]
//     let quotient_commitment = match gpu::quotient_on_gpu(ctx, &stark.gpu_program(ctx)?, &twins_of(&trace_commitments),
//                                                          quotient_degree_bits, public_inputs, &global_values, &challenges, &alphas)? {
//         Some(twin) => gpu::mirror_from_twin(twin, rate_bits, cap_height)?,    // host mirror as in plonky2-hwa-patch::batch_commit_gpu
//         None => cpu_quotient_commitment(...),                                 // the code that is there today
//     };
//
// `stark.gpu_program` caches `RecordingParser::record(&stark.air, ..)` in a `OnceLock` next to the AIR: recorded once per AIR.
