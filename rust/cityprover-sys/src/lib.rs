//! cityprover-sys — Rust side of the drop-in boundary.
//!
//! `ffi` is the whole C ABI of `libcityprover_hip.so`, generated from `include/cityprover.h` by
//! `tools/gen_rust_ffi.py`. The types below are the thin safe layer the patched `plonky2` crate uses
//! (rust/plonky2-hwa-patch): one [`Context`] per worker process and GPU, one [`Circuit`] per built
//! `CircuitData`, `prove` = everything of `CircuitData::prove` after witness generation, `verify` =
//! `CircuitData::verify`. Every error of the library becomes an `anyhow::Error`, the convention of
//! `city_rollup_circuit/src/worker/traits.rs:16-43`; the library itself never aborts or unwinds.
//!
//! NOT COMPILED in the build image (no cargo / rustc there): this is source for a machine with Rust.
pub mod ffi;

use std::ffi::{CStr, CString};
use std::os::raw::c_int;
use std::path::Path;
use std::ptr;

use anyhow::{anyhow, bail, Result};

pub use ffi::{CpGate, CpShape};

fn last_error(ctx: *mut ffi::CpCtx) -> String {
    unsafe { CStr::from_ptr(ffi::cp_last_error(ctx)) }.to_string_lossy().into_owned()
}

fn check(ctx: *mut ffi::CpCtx, rc: c_int) -> Result<()> {
    if rc == ffi::CP_OK {
        return Ok(());
    }
    bail!("cityprover[{rc}]: {}", last_error(ctx))
}

/// One context = one GPU + one stream. Thread-compatible: one caller at a time (the worker loop is one
/// thread per process, city_rollup_core_worker/src/lib.rs:131-145).
pub struct Context {
    raw: *mut ffi::CpCtx,
}
unsafe impl Send for Context {}

impl Context {
    pub fn device_count() -> usize {
        unsafe { ffi::cp_device_count() }.max(0) as usize
    }

    /// Fails loudly when no GPU is visible: the library has no CPU fallback.
    pub fn new(device: usize) -> Result<Self> {
        let abi = unsafe { ffi::cp_abi_version() };
        if abi != ffi::CP_ABI_VERSION {
            bail!("libcityprover_hip.so has ABI version {abi}, this crate was generated for {}", ffi::CP_ABI_VERSION);
        }
        let raw = unsafe { ffi::cp_ctx_create(device as c_int) };
        if raw.is_null() {
            bail!("cp_ctx_create({device}): {}", last_error(ptr::null_mut()));
        }
        Ok(Self { raw })
    }

    /// Internal pipelining of one `prove_batch` call for a single-threaded caller (2..3 pays).
    pub fn set_lanes(&self, lanes: usize) -> Result<()> {
        check(self.raw, unsafe { ffi::cp_ctx_set_lanes(self.raw, lanes as c_int) })
    }

    pub fn raw(&self) -> *mut ffi::CpCtx {
        self.raw
    }

    /// `CircuitData::prove` after witness generation for a batch of proofs of ONE shape / gate set
    /// (circuits may differ). `wires[p]` = `MatrixWitness::wire_values` flattened wire-major
    /// (`[num_wires][n]` canonical u64), `public_inputs[p]` = the values of `prover_only.public_inputs`.
    /// `pow_witness[p]`: `Some(w)` injects a proof-of-work witness (e.g. the CPU prover's, whose parallel
    /// search returns ANY valid witness), `None` = the smallest valid one.
    /// Returns the bincode bytes of `ProofWithPublicInputs`, i.e. exactly what the worker stores
    /// (city_redis_store/src/lib.rs:71-83).
    pub fn prove_batch(
        &self,
        circuits: &[&Circuit],
        wires: &[&[u64]],
        public_inputs: &[&[u64]],
        pow_witness: &[Option<u64>],
    ) -> Result<Vec<Vec<u8>>> {
        let n = circuits.len();
        if wires.len() != n || public_inputs.len() != n || pow_witness.len() != n {
            bail!("prove_batch: argument lengths differ");
        }
        for (c, w) in circuits.iter().zip(wires) {
            let want = (c.shape.num_wires as usize) << c.shape.degree_bits;
            if w.len() != want {
                bail!("prove_batch: wire matrix of {} elements, the circuit needs {want}", w.len());
            }
        }
        let cs: Vec<*mut ffi::CpCircuit> = circuits.iter().map(|c| c.raw).collect();
        let ws: Vec<*const u64> = wires.iter().map(|w| w.as_ptr()).collect();
        let pis: Vec<*const u64> = public_inputs.iter().map(|p| p.as_ptr()).collect();
        let npis: Vec<usize> = public_inputs.iter().map(|p| p.len()).collect();
        let use_pow: Vec<c_int> = pow_witness.iter().map(|w| w.is_some() as c_int).collect();
        let pow: Vec<u64> = pow_witness.iter().map(|w| w.unwrap_or(0)).collect();
        let mut out: Vec<*mut u8> = vec![ptr::null_mut(); n];
        let mut lens = vec![0usize; n];
        check(self.raw, unsafe {
            ffi::cp_prove_batch_host(
                self.raw, n, cs.as_ptr(), pis.as_ptr(), npis.as_ptr(), ws.as_ptr(), use_pow.as_ptr(), pow.as_ptr(),
                out.as_mut_ptr(), lens.as_mut_ptr(),
            )
        })?;
        Ok(out
            .into_iter()
            .zip(lens)
            .map(|(p, len)| {
                let v = unsafe { std::slice::from_raw_parts(p, len) }.to_vec();
                unsafe { ffi::cp_free(p.cast()) };
                v
            })
            .collect())
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { ffi::cp_ctx_destroy(self.raw) }
    }
}

/// A circuit resident on the GPU: the constants + sigmas commitment (coefficients, LDE, Merkle tree), the
/// gate list with its selector groups, k_is — the counterpart of a built `CircuitData`
/// (`CRWorkerToolboxRootCircuits::new`, city_rollup_circuit/src/worker/toolbox/root.rs:75-139).
/// Must not outlive its `Context`.
pub struct Circuit {
    raw: *mut ffi::CpCircuit,
    pub shape: CpShape,
    pub digest: [u64; 4],
}
unsafe impl Send for Circuit {}
// The device data of a circuit is read-only after `load` and the C side documents sharing between lanes / batcher callers
// (include/cityprover.h, cp_batcher): `&Circuit` may be used from several threads. Without this, a `CircuitData` holding a
// GPU twin would stop being `Sync`, which plonky2's rayon users rely on.
unsafe impl Sync for Circuit {}

impl Circuit {
    /// From flat data (see `plonky2-hwa-patch`: `flatten_circuit`). `cs_values`: the constants (selectors first)
    /// and sigma polynomials as VALUES over the subgroup, `[num_constants + num_routed_wires][n]`.
    pub fn load(ctx: &Context, shape: &CpShape, digest: [u64; 4], cs_values: &[u64], k_is: &[u64], gates: &[CpGate], num_selectors: usize) -> Result<Self> {
        let want = ((shape.num_constants + shape.num_routed_wires) as usize) << shape.degree_bits;
        if cs_values.len() != want || k_is.len() != shape.num_routed_wires as usize {
            bail!("Circuit::load: constants/sigmas or k_is have the wrong length");
        }
        let raw = unsafe { ffi::cp_circuit_load(ctx.raw, shape, digest.as_ptr(), cs_values.as_ptr(), k_is.as_ptr()) };
        if raw.is_null() {
            bail!("cp_circuit_load: {}", last_error(ctx.raw));
        }
        let c = Self { raw, shape: *shape, digest };
        check(ctx.raw, unsafe { ffi::cp_circuit_set_gates(raw, gates.as_ptr(), gates.len(), num_selectors as c_int) })?;
        Ok(c)
    }

    /// From a `.cpcirc` file (written by `CircuitData::dump_cityprover` of the patched plonky2, or by the library).
    pub fn load_file(ctx: &Context, path: &Path) -> Result<Self> {
        let p = CString::new(path.to_string_lossy().as_bytes())?;
        let raw = unsafe { ffi::cp_circuit_load_file(ctx.raw, p.as_ptr()) };
        if raw.is_null() {
            bail!("cp_circuit_load_file({}): {}", path.display(), last_error(ctx.raw));
        }
        let mut shape: CpShape = unsafe { std::mem::zeroed() };
        let mut digest = [0u64; 4];
        check(ctx.raw, unsafe { ffi::cp_circuit_shape(raw, &mut shape, digest.as_mut_ptr()) })?;
        Ok(Self { raw, shape, digest })
    }

    pub fn save_file(&self, path: &Path) -> Result<()> {
        let p = CString::new(path.to_string_lossy().as_bytes())?;
        let rc = unsafe { ffi::cp_circuit_save_file(self.raw, p.as_ptr()) };
        if rc != ffi::CP_OK {
            bail!("cp_circuit_save_file[{rc}]: {}", last_error(ptr::null_mut()));
        }
        Ok(())
    }

    /// `VerifierOnlyCircuitData::constants_sigmas_cap` as the GPU computed it (2^cap_height x 4): the shim compares it
    /// with plonky2's own cap at load time — the first end-to-end parity check on a real circuit (fingerprints P8).
    pub fn constants_sigmas_cap(&self) -> Result<Vec<[u64; 4]>> {
        let n = 1usize << self.shape.cap_height;
        let mut flat = vec![0u64; 4 * n];
        let rc = unsafe { ffi::cp_circuit_cs_cap(self.raw, flat.as_mut_ptr()) };
        if rc != ffi::CP_OK {
            bail!("cp_circuit_cs_cap[{rc}]: {}", last_error(ptr::null_mut()));
        }
        Ok(flat.chunks_exact(4).map(|c| [c[0], c[1], c[2], c[3]]).collect())
    }

    /// `CircuitData::verify` on bincode bytes: `Ok(())` = accepted.
    pub fn verify(&self, proof: &[u8]) -> Result<()> {
        let rc = unsafe { ffi::cp_verify(self.raw, proof.as_ptr(), proof.len()) };
        if rc == ffi::CP_OK {
            return Ok(());
        }
        Err(anyhow!("cityprover[{rc}]: {}", last_error(ptr::null_mut())))
    }

    pub fn raw(&self) -> *mut ffi::CpCircuit {
        self.raw
    }
}

impl Drop for Circuit {
    fn drop(&mut self) {
        unsafe { ffi::cp_circuit_destroy(self.raw) }
    }
}

/// Error classes the caller may want to tell apart: a request the GPU backend REFUSES (unsupported gate, zero-knowledge
/// circuit on a plain entry point, malformed argument) is not a failed proof — the caller falls back to the CPU prover.
pub fn is_refusal(err: &anyhow::Error) -> bool {
    let s = err.to_string();
    s.starts_with(&format!("cityprover[{}]", ffi::CP_ERR_INVALID_ARG)) || s.starts_with(&format!("cityprover[{}]", ffi::CP_ERR_UNSUPPORTED))
}

/// plonky2 `PolynomialBatch` resident on the GPU (`cp_poly_batch`): coefficients, bit-reversed LDE and Merkle tree of k
/// polynomials. The generic seam every FRI-based prover goes through — `CircuitData::prove` and starkyx's `ByteStark::prove`
/// (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524) alike.
pub struct PolyBatch {
    raw: *mut ffi::CpPolyBatch,
    pub k: usize,
    pub degree_bits: usize,
    pub rate_bits: usize,
    pub cap_height: usize,
    pub blinding: bool,
}
unsafe impl Send for PolyBatch {}
unsafe impl Sync for PolyBatch {}

impl PolyBatch {
    /// `PolynomialBatch::from_values` (`from_coeffs` with `coeffs = true`). `polys`: k x n, polynomial-major canonical u64.
    /// `salts`: `Some(CP_SALT_SIZE x N)` = blinding with the caller's randomness.
    pub fn commit(ctx: &Context, polys: &[u64], k: usize, degree_bits: usize, rate_bits: usize, cap_height: usize, coeffs: bool, salts: Option<&[u64]>) -> Result<Self> {
        if polys.len() != k << degree_bits {
            bail!("PolyBatch::commit: {} elements for {k} polynomials of 2^{degree_bits}", polys.len());
        }
        if let Some(s) = salts {
            if s.len() != ffi::CP_SALT_SIZE << (degree_bits + rate_bits) {
                bail!("PolyBatch::commit: salts have the wrong length");
            }
        }
        let mut raw = ptr::null_mut();
        check(ctx.raw, unsafe {
            ffi::cp_batch_commit(
                ctx.raw, polys.as_ptr(), k, degree_bits as c_int, rate_bits as c_int, cap_height as c_int,
                if coeffs { ffi::CP_BATCH_FROM_COEFFS } else { 0 }, salts.map_or(ptr::null(), |s| s.as_ptr()), &mut raw,
            )
        })?;
        Ok(Self { raw, k, degree_bits, rate_bits, cap_height, blinding: salts.is_some() })
    }

    fn ok(&self, rc: c_int) -> Result<()> {
        if rc == ffi::CP_OK {
            return Ok(());
        }
        bail!("cityprover[{rc}]: {}", last_error(ptr::null_mut()))
    }

    /// `merkle_tree.cap`
    pub fn cap(&self) -> Result<Vec<[u64; 4]>> {
        let mut flat = vec![0u64; 4 << self.cap_height];
        self.ok(unsafe { ffi::cp_batch_cap(self.raw, flat.as_mut_ptr()) })?;
        Ok(flat.chunks_exact(4).map(|c| [c[0], c[1], c[2], c[3]]).collect())
    }

    /// `polynomials[first..first+count].map(|p| p.to_extension().eval(point))`
    pub fn eval_ext(&self, first: usize, count: usize, point: [u64; 2]) -> Result<Vec<[u64; 2]>> {
        let mut flat = vec![0u64; 2 * count];
        self.ok(unsafe { ffi::cp_batch_eval_ext(self.raw, first, count, point.as_ptr(), flat.as_mut_ptr()) })?;
        Ok(flat.chunks_exact(2).map(|c| [c[0], c[1]]).collect())
    }

    /// `merkle_tree.leaves[first..first+count]`, row-major, salt included
    pub fn leaves(&self, first: usize, count: usize) -> Result<Vec<u64>> {
        let w = self.k + if self.blinding { ffi::CP_SALT_SIZE } else { 0 };
        let mut out = vec![0u64; count * w];
        self.ok(unsafe { ffi::cp_batch_leaves(self.raw, first, count, out.as_mut_ptr()) })?;
        Ok(out)
    }

    /// `polynomials`: all k coefficient vectors (k x n, polynomial-major); a copy — the handle's device pointers are not taken
    pub fn coeffs(&self) -> Result<Vec<u64>> {
        let mut out = vec![0u64; self.k << self.degree_bits];
        self.ok(unsafe { ffi::cp_batch_coeffs(self.raw, 0, self.k, out.as_mut_ptr()) })?;
        Ok(out)
    }

    pub fn raw(&self) -> *mut ffi::CpPolyBatch {
        self.raw
    }
}

impl Drop for PolyBatch {
    fn drop(&mut self) {
        unsafe { ffi::cp_batch_destroy(self.raw) }
    }
}

/// One opening batch (`FriBatchInfo`): the point and the polynomial list as (oracle, first, count) runs.
pub struct FriBatch {
    pub point: [u64; 2],
    pub ranges: Vec<ffi::CpFriPolyRange>,
}

/// `PolynomialBatch::prove_openings`: bincode `FriProof` bytes; `challenger` is advanced as plonky2's would be.
pub fn fri_prove(ctx: &Context, oracles: &[&PolyBatch], batches: &[FriBatch], params: &ffi::CpFriParams, challenger: &mut ffi::CpChallengerState, pow_witness: Option<u64>) -> Result<Vec<u8>> {
    let os: Vec<*mut ffi::CpPolyBatch> = oracles.iter().map(|o| o.raw).collect();
    let bs: Vec<ffi::CpFriBatch> = batches.iter().map(|b| ffi::CpFriBatch { point: b.point, ranges: b.ranges.as_ptr(), n_ranges: b.ranges.len() }).collect();
    let (mut out, mut len) = (ptr::null_mut::<u8>(), 0usize);
    check(ctx.raw, unsafe {
        ffi::cp_fri_prove(ctx.raw, os.as_ptr(), os.len(), bs.as_ptr(), bs.len(), params, challenger, pow_witness.is_some() as c_int, pow_witness.unwrap_or(0), &mut out, &mut len)
    })?;
    let v = unsafe { std::slice::from_raw_parts(out, len) }.to_vec();
    unsafe { ffi::cp_free(out.cast()) };
    Ok(v)
}

/// `verify_fri_proof` (+ `fri_challenges`): host arithmetic, no GPU.
pub fn fri_verify(params: &ffi::CpFriParams, oracles: &[ffi::CpFriOracleInfo], caps: &[&[u64]], batches: &[FriBatch], opened: &[&[u64]], challenger: &mut ffi::CpChallengerState, proof: &[u8]) -> Result<()> {
    let cs: Vec<*const u64> = caps.iter().map(|c| c.as_ptr()).collect();
    let ov: Vec<*const u64> = opened.iter().map(|o| o.as_ptr()).collect();
    let bs: Vec<ffi::CpFriBatch> = batches.iter().map(|b| ffi::CpFriBatch { point: b.point, ranges: b.ranges.as_ptr(), n_ranges: b.ranges.len() }).collect();
    let rc = unsafe { ffi::cp_fri_verify(params, oracles.as_ptr(), oracles.len(), cs.as_ptr(), bs.as_ptr(), bs.len(), ov.as_ptr(), challenger, proof.as_ptr(), proof.len()) };
    if rc == ffi::CP_OK {
        return Ok(());
    }
    Err(anyhow!("cityprover[{rc}]: {}", last_error(ptr::null_mut())))
}

/// Group commit for worker loops that prove one job per call (`SimpleActorWorker::process_next_job`,
/// city_rollup_core_worker/src/actors/simple.rs:32-56) when several of them run as THREADS of one process: their
/// concurrent `prove` calls are merged into `cp_prove_batch_host` launches (`cp_batcher_*` in include/cityprover.h).
/// `Sync`: `prove` may be called from any number of threads; it blocks until the proof is there. Same bytes as
/// [`Context::prove_batch`]. Call `Context::set_lanes` first to let several batches overlap. While a `Batcher`
/// exists nothing else may prove on its context, and it must not outlive the context or the circuits it is handed.
pub struct Batcher {
    raw: *mut ffi::CpBatcher,
}
unsafe impl Send for Batcher {}
unsafe impl Sync for Batcher {}

impl Batcher {
    /// `linger_us`: how long a caller that found the device free waits for company (0 = never; a few hundred pays
    /// once there are more callers than `max_batch`).
    pub fn new(ctx: &Context, max_batch: usize, linger_us: u32) -> Result<Self> {
        let raw = unsafe { ffi::cp_batcher_create(ctx.raw, max_batch, linger_us) };
        if raw.is_null() {
            bail!("cp_batcher_create: {}", last_error(ptr::null_mut()));
        }
        Ok(Self { raw })
    }

    pub fn prove(&self, circuit: &Circuit, wires: &[u64], public_inputs: &[u64], pow_witness: Option<u64>) -> Result<Vec<u8>> {
        let want = (circuit.shape.num_wires as usize) << circuit.shape.degree_bits;
        if wires.len() != want {
            bail!("Batcher::prove: wire matrix of {} elements, the circuit needs {want}", wires.len());
        }
        let (mut out, mut len) = (ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe {
            ffi::cp_batcher_prove(
                self.raw, circuit.raw, wires.as_ptr(), public_inputs.as_ptr(), public_inputs.len(),
                pow_witness.is_some() as c_int, pow_witness.unwrap_or(0), &mut out, &mut len,
            )
        };
        if rc != ffi::CP_OK {
            // the message is the calling thread's: the context's buffer belongs to whoever runs a batch on it
            bail!("cityprover[{rc}]: {}", last_error(ptr::null_mut()));
        }
        let v = unsafe { std::slice::from_raw_parts(out, len) }.to_vec();
        unsafe { ffi::cp_free(out.cast()) };
        Ok(v)
    }

    pub fn stats(&self) -> ffi::CpBatcherStats {
        let mut st: ffi::CpBatcherStats = unsafe { std::mem::zeroed() };
        unsafe { ffi::cp_batcher_get_stats(self.raw, &mut st) };
        st
    }
}

impl Drop for Batcher {
    fn drop(&mut self) {
        unsafe { ffi::cp_batcher_destroy(self.raw) }
    }
}

/// `CityGroth16ProofData` bytes (pi_a | pi_b_a0 | pi_b_a1 | pi_c, 4 x 48) from affine coordinates
/// (city_rollup_common/src/block_template/data.rs:27-34).
pub fn groth16_pack_city(a_xy: &[u64; 12], b_xy: &[u64; 24], c_xy: &[u64; 12]) -> Result<[u8; 192]> {
    let mut out = [0u8; 192];
    let rc = unsafe { ffi::cp_groth16_proof_pack_city(a_xy.as_ptr(), b_xy.as_ptr(), c_xy.as_ptr(), out.as_mut_ptr()) };
    if rc != ffi::CP_OK {
        bail!("cp_groth16_proof_pack_city[{rc}]: {}", last_error(ptr::null_mut()));
    }
    Ok(out)
}

/// A straight-line AIR program compiled for the device (`cp_air_program`, include/cityprover.h "the STARK's own two steps as
/// GENERIC device machinery"): what rust/starkyx-patch/recording_parser.rs produces by running an AIR once against a recorder.
pub struct AirProgram {
    raw: *mut ffi::CpAirProgram,
}
unsafe impl Send for AirProgram {}
unsafe impl Sync for AirProgram {}

impl AirProgram {
    pub fn create(ctx: &Context, desc: &ffi::CpAirProgramDesc) -> Result<Self> {
        let raw = unsafe { ffi::cp_air_program_create(ctx.raw, desc) };
        if raw.is_null() {
            bail!("cityprover[{}]: {}", ffi::CP_ERR_INVALID_ARG, last_error(ctx.raw));
        }
        Ok(Self { raw })
    }

    pub fn info(&self) -> ffi::CpAirProgramInfo {
        let mut i: ffi::CpAirProgramInfo = unsafe { std::mem::zeroed() };
        unsafe { ffi::cp_air_program_get_info(self.raw, &mut i) };
        i
    }

    pub fn raw(&self) -> *mut ffi::CpAirProgram {
        self.raw
    }
}

impl Drop for AirProgram {
    fn drop(&mut self) {
        unsafe { ffi::cp_air_program_destroy(self.raw) }
    }
}

/// starky `compute_quotient_polys` + `PolynomialBatch::from_coeffs` on the device: the program at every point of the quotient
/// coset straight from the trace commitments, alpha-folded, / Z_H, coset iNTT, committed. Returns the quotient chunks
/// (`alphas.len() << quotient_degree_bits` polynomials, challenge-major) as a batch for `fri_prove`.
pub fn air_quotient_commit(ctx: &Context, program: &AirProgram, oracles: &[&PolyBatch], quotient_degree_bits: usize, publics: &[u64], globals: &[u64], challenges: &[u64], alphas: &[u64]) -> Result<PolyBatch> {
    let os: Vec<*mut ffi::CpPolyBatch> = oracles.iter().map(|o| o.raw).collect();
    let mut raw = ptr::null_mut();
    check(ctx.raw, unsafe {
        ffi::cp_air_quotient_commit(ctx.raw, program.raw, os.as_ptr(), os.len(), quotient_degree_bits as c_int, publics.as_ptr(), globals.as_ptr(), challenges.as_ptr(), alphas.as_ptr(), alphas.len(), &mut raw)
    })?;
    let first = oracles[0];
    Ok(PolyBatch { raw, k: alphas.len() << quotient_degree_bits, degree_bits: first.degree_bits, rate_bits: first.rate_bits, cap_height: first.cap_height, blinding: false })
}
