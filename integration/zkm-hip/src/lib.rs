//! `HipProver`: Ziren's `MachineProver` (crates/stark/src/prover.rs:30-184) on an MI355X through libzkm_hip.so.
//!
//! Plug-in points, nothing else in Ziren changes:
//!   * `ZKMProverComponents::CoreProver = HipProver<CoreSC, MipsAir<..>>` (crates/prover/src/components.rs:6-35) — `HipProverComponents` below;
//!   * `run_test::<HipProver<_, _>>(program)` in the core machine's tests (crates/core/machine/src/utils/prove.rs:614-656).
//!
//! The SDK-facing pieces (`HipProverComponents`, the `ZKM_PROVER=hip` client) live in integration/sdk-hip/, which depends on this crate.
//!
//! NOT compiled where it was written (no Rust toolchain in that image); the C ABI underneath is exercised call for call by
//! tests/c_abi/consumer.c and by the Python mirror ziren_amd/prover.py. ffi.rs is generated from include/zkm_hip.h.

pub mod decode;
pub mod ffi;
pub mod recorder;
pub mod tracegen;

use std::ffi::{c_int, CStr};
use std::ptr::null_mut;

use hashbrown::HashMap;
use p3_air::Air;
use p3_challenger::DuplexChallenger;
use p3_field::PrimeField32;
use p3_koala_bear::KoalaBear;
use p3_matrix::{dense::RowMajorMatrix, Matrix};
use p3_uni_stark::SymbolicAirBuilder;
use zkm_stark::{
    air::MachineAir, koala_bear_poseidon2::KoalaBearPoseidon2, DebugConstraintBuilder, MachineProof, MachineProver, MachineProvingKey, MachineRecord,
    ShardMainData, ShardProof,
    Challenger, Com, StarkGenericConfig, StarkMachine, StarkProvingKey, StarkVerifyingKey, Val,
};

use recorder::RecordedChip;

type SC = KoalaBearPoseidon2;
type F = KoalaBear;

#[derive(Debug, thiserror::Error)]
#[error("libzkm_hip: {0}")]
pub struct HipProverError(pub String);

fn last_error() -> HipProverError {
    HipProverError(unsafe { CStr::from_ptr(ffi::zkm_last_error()) }.to_string_lossy().into_owned())
}
fn check(rc: c_int) -> Result<(), HipProverError> { if rc == 0 { Ok(()) } else { Err(last_error()) } }

/// One GPU: a `zkm_ctx` (its own stream, memory pool and kernels). `Send + Sync`: the library serialises calls on a context.
pub struct HipContext(*mut ffi::ZkmCtx);
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}
impl Drop for HipContext { fn drop(&mut self) { unsafe { ffi::zkm_ctx_destroy(self.0) } } }

/// `DeviceMatrix`: a column-major matrix resident in HBM.
pub struct HipMatrix {
    ctx: *mut ffi::ZkmCtx,
    h: *mut ffi::ZkmMatrix,
    height: usize,
    width: usize,
    /// host copy for `Matrix::row` (debugging reads only): downloaded once, on the first row asked for
    host: std::sync::OnceLock<Vec<u32>>,
}
unsafe impl Send for HipMatrix {}
unsafe impl Sync for HipMatrix {}
impl Drop for HipMatrix { fn drop(&mut self) { unsafe { ffi::zkm_matrix_free(self.ctx, self.h) } } }
impl HipMatrix {
    pub(crate) fn from_handle(ctx: *mut ffi::ZkmCtx, h: *mut ffi::ZkmMatrix, height: usize, width: usize) -> Self {
        Self { ctx, h, height, width, host: std::sync::OnceLock::new() }
    }
}
impl Matrix<F> for HipMatrix {
    fn width(&self) -> usize { self.width }
    fn height(&self) -> usize { self.height }
    type Row<'a> = std::vec::IntoIter<F>;
    // Rows of a device matrix are not read on the host on the proving path (`open` works on the handles). A debugging read downloads the
    // matrix once and serves every later row from that copy; a failed download panics like an out-of-range row would.
    fn row(&self, r: usize) -> Self::Row<'_> {
        let host = self.host.get_or_init(|| {
            let mut host = vec![0u32; self.height * self.width];
            check(unsafe { ffi::zkm_matrix_download(self.ctx, self.h, host.as_mut_ptr()) }).expect("zkm_matrix_download");
            host
        });
        host[r * self.width..(r + 1) * self.width].iter().map(|&w| unsafe { core::mem::transmute::<u32, F>(w) }).collect::<Vec<_>>().into_iter()
    }
}

/// `DeviceProverData`: owned by the library inside zkm_main_data; the Rust side only carries the handle.
pub struct HipMainData {
    ctx: *mut ffi::ZkmCtx,
    h: *mut ffi::ZkmMainData,
    /// chip names in the order `zkm_commit` received them: `zkm_open` wants the descriptors in that same order
    caller_names: Vec<String>,
}
unsafe impl Send for HipMainData {}
unsafe impl Sync for HipMainData {}
impl Drop for HipMainData { fn drop(&mut self) { if !self.h.is_null() { unsafe { ffi::zkm_main_data_free(self.ctx, self.h) } } } }

/// `DeviceProvingKey`: the host key (vk fields, chip ordering) plus the preprocessed traces / LDEs / tree on the device.
pub struct HipProvingKey { host: StarkProvingKey<SC>, ctx: *mut ffi::ZkmCtx, h: *mut ffi::ZkmPk, _prep: Vec<HipMatrix> }
unsafe impl Send for HipProvingKey {}
unsafe impl Sync for HipProvingKey {}
impl Drop for HipProvingKey { fn drop(&mut self) { unsafe { ffi::zkm_pk_free(self.ctx, self.h) } } }
impl MachineProvingKey<SC> for HipProvingKey {
    fn preprocessed_commit(&self) -> Com<SC> { self.host.commit.clone() }
    fn pc_start(&self) -> Val<SC> { self.host.pc_start }
    fn initial_global_cumulative_sum(&self) -> zkm_stark::septic_digest::SepticDigest<Val<SC>> { self.host.initial_global_cumulative_sum }
    fn observe_into(&self, challenger: &mut Challenger<SC>) {
        // host-side, as StarkProvingKey::observe_into (machine.rs:79-86); zkm_pk_observe_into is its twin on a ZkmChallenger
        self.host.observe_into(challenger)
    }
}

pub struct HipProver<A> where A: MachineAir<F> {
    machine: StarkMachine<SC, A>,
    ctx: HipContext,
    /// every chip of the machine, recorded once: name -> bytecode + lookups (the memory ZkmChipDesc points into)
    recorded: HashMap<String, RecordedChip>,
}

impl<A> HipProver<A>
where
    A: MachineAir<F> + for<'a> Air<SymbolicAirBuilder<F>>,
{
    fn desc(&self, name: &str, prep_index: i32) -> ffi::ZkmChipDesc {
        let c = &self.recorded[name];
        ffi::ZkmChipDesc {
            name: c.name.as_ptr(), main_width: c.main_width, prep_width: c.prep_width, prep_index,
            log_quotient_degree: c.log_quotient_degree, local_only: c.local_only as u32, commit_scope_global: c.commit_scope_global as u32,
            num_constraints: c.num_constraints, lookups: c.lookups.as_ptr(), lookups_len: c.lookups.len() as u32,
            program: c.program.as_ptr(), program_len: c.program.len() as u32,
        }
    }

    fn upload(&self, m: &RowMajorMatrix<F>) -> Result<HipMatrix, HipProverError> {
        let mut h = null_mut();
        // asynchronous: slabbed DMA + transpose on the library's upload streams; consumers wait per matrix on the device
        check(unsafe { ffi::zkm_matrix_upload_async(self.ctx.0, m.values.as_ptr() as *const u32, m.height(), m.width(), &mut h) })?;
        Ok(HipMatrix::from_handle(self.ctx.0, h, m.height(), m.width()))
    }

    /// The ahead-of-time compiled quotient kernel of a chip, if one was shipped: `ziren_amd/codegen.py` (`specialize_many`, run by
    /// `__graft_entry__.build()`) keeps `<ZKM_HIP_KERNEL_DIR>/manifest.json`, a JSON object {sha256 of the program words: file name}; the
    /// file is one code object (.hsaco) or, for a program long enough to be cut into several kernels (KeccakSponge), their container
    /// (.parts) — zkm_ctx_register_quotient_kernel takes either. A chip without one runs through the library's bytecode interpreter
    /// (4x slower on the quotient phase, same field values): no compiler is needed at run time either way. A kernel the library
    /// refuses is an error, not a silent fall-back.
    fn register_aot_kernels(&self) -> Result<(), HipProverError> {
        use sha2::{Digest, Sha256};
        let dir = std::path::PathBuf::from(env!("ZKM_HIP_KERNEL_DIR"));
        let Ok(text) = std::fs::read_to_string(dir.join("manifest.json")) else { return Ok(()) };
        let manifest: std::collections::HashMap<String, String> =
            serde_json::from_str(&text).map_err(|e| HipProverError(format!("{}: {e}", dir.join("manifest.json").display())))?;
        for c in self.recorded.values() {
            let bytes: Vec<u8> = c.program.iter().flat_map(|w| w.to_le_bytes()).collect();
            let key = Sha256::digest(&bytes).iter().map(|b| format!("{b:02x}")).collect::<String>();
            let Some(file) = manifest.get(&key) else { continue };
            let obj = std::fs::read(dir.join(file)).map_err(|e| HipProverError(format!("{file}: {e}")))?;
            check(unsafe { ffi::zkm_ctx_register_quotient_kernel(self.ctx.0, c.program.as_ptr(), c.program.len() as u32, obj.as_ptr() as *const _, obj.len()) })?;
        }
        Ok(())
    }
}

impl<A> MachineProver<SC, A> for HipProver<A>
where
    A: MachineAir<F> + for<'a> Air<SymbolicAirBuilder<F>> + 'static,
    // ... plus the bounds CpuProver carries (prover.rs:210-221): Air<ProverConstraintFolder>, Air<VerifierConstraintFolder>, Air<LookupBuilder>
    A::Record: MachineRecord,
{
    type DeviceMatrix = HipMatrix;
    type DeviceProverData = HipMainData;
    type DeviceProvingKey = HipProvingKey;
    type Error = HipProverError;

    fn new(machine: StarkMachine<SC, A>) -> Self {
        let device = std::env::var("ZKM_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx = null_mut();
        check(unsafe { ffi::zkm_ctx_create(device, &mut ctx) }).expect("zkm_ctx_create (there is no CPU fallback)");
        let recorded = machine.chips().iter().map(|chip| (chip.name(), recorder::record_chip(chip))).collect();
        let p = Self { machine, ctx: HipContext(ctx), recorded };
        p.register_aot_kernels().expect("registering the shipped quotient kernels");
        p
    }

    fn machine(&self) -> &StarkMachine<SC, A> { &self.machine }

    fn setup(&self, program: &A::Program) -> (Self::DeviceProvingKey, StarkVerifyingKey<SC>) {
        let (pk, vk) = self.machine.setup(program);      // host side, as CpuProver::setup (prover.rs:238-243)
        (self.pk_to_device(&pk), vk)
    }

    fn pk_from_vk(&self, program: &A::Program, vk: &StarkVerifyingKey<SC>) -> Self::DeviceProvingKey {
        self.pk_to_device(&self.machine.setup_core(program, vk.initial_global_cumulative_sum).0)
    }

    fn pk_to_device(&self, pk: &StarkProvingKey<SC>) -> Self::DeviceProvingKey {
        let prep: Vec<HipMatrix> = pk.traces.iter().map(|t| self.upload(t).expect("upload")).collect();
        let handles: Vec<*const ffi::ZkmMatrix> = prep.iter().map(|m| m.h as *const _).collect();
        let local_only: Vec<u32> = pk.local_only.iter().map(|&b| b as u32).collect();
        let igcs: Vec<u32> = pk.initial_global_cumulative_sum.0.x.0.iter().chain(pk.initial_global_cumulative_sum.0.y.0.iter())
            .map(|&w| unsafe { core::mem::transmute::<F, u32>(w) }).collect();
        let mut h = null_mut();
        check(unsafe {
            ffi::zkm_pk_setup(self.ctx.0, prep.len(), handles.as_ptr(), local_only.as_ptr(), core::mem::transmute::<F, u32>(pk.pc_start), igcs.as_ptr(),
                              self.config().pcs().fri_config().log_blowup as u32, &mut h)
        }).expect("zkm_pk_setup");
        let mut root = [0u32; 8];
        unsafe { ffi::zkm_pk_commitment(h, root.as_mut_ptr()) };
        debug_assert_eq!(root.map(|w| unsafe { core::mem::transmute::<u32, F>(w) }), <[F; 8]>::from(pk.commit.clone()), "device and host preprocessed commitments differ");
        HipProvingKey { host: pk.clone(), ctx: self.ctx.0, h, _prep: prep }
    }

    fn pk_to_host(&self, pk: &Self::DeviceProvingKey) -> StarkProvingKey<SC> { pk.host.clone() }

    /// prover.rs:70-109, with one difference: a chip the library builds from events (tracegen::device_trace) gets no host trace — an empty
    /// matrix of its width stands in, and `commit` builds the real one on the device. Only for the core machine (A::Record = ExecutionRecord).
    fn generate_traces(&self, record: &A::Record) -> Result<Vec<(String, RowMajorMatrix<Val<SC>>)>, A::Error> {
        let on_device = (record as &dyn std::any::Any).downcast_ref::<zkm_core_executor::ExecutionRecord>().is_some();
        self.shard_chips(record)
            .map(|chip| {
                let name = chip.name();
                if on_device && tracegen::DEVICE_BUILT.contains(&name.as_str()) {
                    Ok((name, RowMajorMatrix::new(vec![], chip.width())))
                } else {
                    chip.generate_trace(record, &mut A::Record::default()).map(|t| (name, t))
                }
            })
            .collect()
    }

    /// prover.rs:111-115 / :258-292. Device-built chips are generated from the record's events; the others' host traces go up tallest first
    /// so their upload overlaps the commit's kernels. Byte lookups keep the reference's flow (generate_dependencies has counted every chip's
    /// into record.byte_lookups, the Byte chip's trace is a host trace), so the device generators are not asked to count them again.
    fn commit(&self, record: &A::Record, traces: Vec<(String, RowMajorMatrix<Val<SC>>)>) -> ShardMainData<SC, Self::DeviceMatrix, Self::DeviceProverData> {
        let core = (record as &dyn std::any::Any).downcast_ref::<zkm_core_executor::ExecutionRecord>();
        let mut order: Vec<usize> = (0..traces.len()).collect();
        order.sort_by_key(|&i| std::cmp::Reverse(traces[i].1.height()));
        let mut dev: Vec<Option<HipMatrix>> = (0..traces.len()).map(|_| None).collect();
        for i in order {
            let (name, host) = &traces[i];
            let born = match core {
                Some(r) if host.values.is_empty() => {
                    let fixed = r.shape.as_ref().and_then(|s| s.log2_height_by_name(name)).map(|h| h as i32).unwrap_or(-1);
                    tracegen::device_trace(self.ctx.0, name, r, fixed, null_mut()).expect("device trace generation")
                }
                _ => None,
            };
            dev[i] = Some(match born { Some(m) => m, None => self.upload(host).expect("upload") });
        }
        let dev: Vec<HipMatrix> = dev.into_iter().map(Option::unwrap).collect();
        let names: Vec<std::ffi::CString> = traces.iter().map(|(n, _)| std::ffi::CString::new(n.as_str()).unwrap()).collect();
        let name_ptrs: Vec<*const std::ffi::c_char> = names.iter().map(|n| n.as_ptr()).collect();
        let handles: Vec<*const ffi::ZkmMatrix> = dev.iter().map(|m| m.h as *const _).collect();
        let public_values: Vec<F> = record.public_values();
        let (mut root, mut order_out, mut h) = ([0u32; 8], vec![0u32; traces.len()], null_mut());
        check(unsafe {
            ffi::zkm_commit(self.ctx.0, traces.len(), name_ptrs.as_ptr(), handles.as_ptr(), public_values.as_ptr() as *const u32, public_values.len(),
                            self.config().pcs().fri_config().log_blowup as u32, root.as_mut_ptr(), order_out.as_mut_ptr(), &mut h)
        }).expect("zkm_commit");
        let caller_names: Vec<String> = traces.iter().map(|(n, _)| n.clone()).collect();
        // chip_ordering: name -> position in commit order (prover.rs:264-271); order_out[position] = caller index
        let chip_ordering = order_out.iter().enumerate().map(|(pos, &idx)| (traces[idx as usize].0.clone(), pos)).collect();
        // ShardMainData.traces are in commit order (prover.rs:283-291)
        let mut by_caller: Vec<Option<HipMatrix>> = dev.into_iter().map(Some).collect();
        let sorted = order_out.iter().map(|&i| by_caller[i as usize].take().unwrap()).collect();
        ShardMainData::new(sorted, root.map(|w| unsafe { core::mem::transmute::<u32, F>(w) }).into(), HipMainData { ctx: self.ctx.0, h, caller_names }, chip_ordering, public_values)
    }

    /// prover.rs:118-126 / :298-653: everything between the main commitment and the ShardProof happens inside zkm_open.
    fn open(&self, pk: &Self::DeviceProvingKey, mut data: ShardMainData<SC, Self::DeviceMatrix, Self::DeviceProverData>,
            challenger: &mut Challenger<SC>) -> Result<ShardProof<SC>, Self::Error> {
        // descriptors in the caller order of zkm_commit (the library maps them through its own chip_ordering), each with the key's
        // preprocessed index of its chip (pk.chip_ordering, machine.rs:58-75)
        let caller_names = std::mem::take(&mut data.main_data.caller_names);
        let descs: Vec<ffi::ZkmChipDesc> = caller_names.iter()
            .map(|n| self.desc(n, pk.host.chip_ordering.get(n).map(|&i| i as i32).unwrap_or(-1))).collect();
        let fri = self.config().pcs().fri_config();
        let cfg = ffi::ZkmFriConfig { log_blowup: fri.log_blowup as u32, num_queries: fri.num_queries as u32, proof_of_work_bits: fri.proof_of_work_bits as u32 };
        let mut ch = challenger_to_ffi(challenger);
        // zkm_open neither frees the main data nor advances the transcript when the stream does not fit (it reports the length it
        // needs in `len`), so that one failure is retried once with the right size; `data` frees the handle when it drops
        let mut proof = vec![0u32; 1usize << 22];
        let mut len = 0usize;
        let mut rc = unsafe {
            ffi::zkm_open(self.ctx.0, pk.h, data.main_data.h, descs.as_ptr(), &cfg, self.machine.num_pv_elts() as u32, &mut ch,
                          proof.as_mut_ptr(), proof.len(), &mut len)
        };
        if rc != 0 && len > proof.len() {
            proof.resize(len, 0);
            ch = challenger_to_ffi(challenger);
            rc = unsafe {
                ffi::zkm_open(self.ctx.0, pk.h, data.main_data.h, descs.as_ptr(), &cfg, self.machine.num_pv_elts() as u32, &mut ch,
                              proof.as_mut_ptr(), proof.len(), &mut len)
            };
        }
        check(rc)?;
        proof.truncate(len);
        challenger_from_ffi(&ch, challenger);
        Ok(decode::decode_shard_proof(&proof, &caller_names))
    }

    /// prover.rs:140-148, CPU body :660-693: dependencies, the key into the transcript, then per record `generate_traces` -> `commit`
    /// -> `open` on a clone of the challenger. The records go through one after the other, not through rayon: a `zkm_ctx` serialises
    /// its calls anyway (one context per GPU; `ZKMProverOpts::gpu` sets shard_batch_size = 1 for the same reason, opts.rs:83-110),
    /// and inside `commit` the upload of the next trace already overlaps the kernels of the previous one.
    fn prove(&self, pk: &Self::DeviceProvingKey, mut records: Vec<A::Record>, challenger: &mut Challenger<SC>,
             opts: <A::Record as MachineRecord>::Config) -> Result<MachineProof<SC>, Self::Error>
    where
        A: for<'a> Air<DebugConstraintBuilder<'a, Val<SC>, <SC as StarkGenericConfig>::Challenge>>,
    {
        self.machine.generate_dependencies(&mut records, &opts, None).map_err(|e| HipProverError(format!("generate_dependencies: {e:?}")))?;
        pk.observe_into(challenger);
        let mut shard_proofs = Vec::with_capacity(records.len());
        for record in records {
            let named_traces = self.generate_traces(&record).map_err(|e| HipProverError(format!("generate_traces: {e:?}")))?;
            let shard_data = self.commit(&record, named_traces);
            shard_proofs.push(self.open(pk, shard_data, &mut challenger.clone())?);
        }
        Ok(MachineProof { shard_proofs })
    }
}

/// DuplexChallenger<KoalaBear, Perm, 16, 8> <-> zkm_challenger (same fields as ChallengerPublicValues,
/// crates/recursion/circuit/src/challenger.rs:62-66,117-150).
fn challenger_to_ffi(c: &Challenger<SC>) -> ffi::ZkmChallenger {
    let mut out = ffi::ZkmChallenger { sponge_state: [0; 16], num_inputs: 0, input_buffer: [0; 16], num_outputs: 0, output_buffer: [0; 16] };
    let w = |x: F| unsafe { core::mem::transmute::<F, u32>(x) };
    for (d, s) in out.sponge_state.iter_mut().zip(c.sponge_state.iter()) { *d = w(*s); }
    out.num_inputs = c.input_buffer.len() as u32;
    for (d, s) in out.input_buffer.iter_mut().zip(c.input_buffer.iter()) { *d = w(*s); }
    out.num_outputs = c.output_buffer.len() as u32;
    for (d, s) in out.output_buffer.iter_mut().zip(c.output_buffer.iter()) { *d = w(*s); }
    out
}
fn challenger_from_ffi(src: &ffi::ZkmChallenger, c: &mut Challenger<SC>) {
    let f = |x: u32| unsafe { core::mem::transmute::<u32, F>(x) };
    for (d, s) in c.sponge_state.iter_mut().zip(src.sponge_state.iter()) { *d = f(*s); }
    c.input_buffer = src.input_buffer[..src.num_inputs as usize].iter().map(|&x| f(x)).collect();
    c.output_buffer = src.output_buffer[..src.num_outputs as usize].iter().map(|&x| f(x)).collect();
}

#[allow(dead_code)]
fn _duplex_is_the_challenger(_: &DuplexChallenger<F, zkm_stark::koala_bear_poseidon2::InnerPerm, 16, 8>) {}
#[allow(dead_code)]
fn _canonical(x: F) -> u32 { x.as_canonical_u32() }
