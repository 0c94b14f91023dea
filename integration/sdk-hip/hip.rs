//! `HipProver`: the SDK prover behind `ProverClient::hip()` / `ZKM_PROVER=hip`. Goes into crates/sdk/src/provers/hip.rs.
//!
//! Modelled on the two local provers of the SDK: like `CudaProver` (crates/sdk/src/provers/cuda.rs:19-60) it keeps a CPU
//! `ZKMProver<DefaultProverComponents>` for everything that is not proving (execute, verify, the `Prover<DefaultProverComponents>`
//! bound `ProverClient` asks for, crates/sdk/src/lib.rs:45-49) next to the GPU prover; unlike it there is no server process in between
//! (the CUDA prover talks to a container over gRPC): the GPU prover is a `ZKMProver<HipProverComponents>` in this process, so
//! `prove_impl` is `CpuProver::prove_impl` (crates/sdk/src/provers/cpu.rs:84-186) line for line with `self.hip_prover` in place of
//! `self.prover`, and the options are `ZKMProverOpts::hip()` (one shard in flight per context; patches/opts.rs.patch).
//!
//! NOT compiled where it was written (no Rust toolchain in that image).

use anyhow::Result;
use zkm_core_executor::ZKMContext;
use zkm_core_machine::io::ZKMStdin;
use zkm_prover::components::{DefaultProverComponents, HipProverComponents};
use zkm_prover::{ZKMProver, ZKM_CIRCUIT_VERSION};
use zkm_stark::{MachineProver, ZKMProverOpts};

use crate::install::try_install_circuit_artifacts;
use crate::{
    provers::ProofOpts, Prover, ZKMProof, ZKMProofKind, ZKMProofWithPublicValues, ZKMProvingKey, ZKMVerifyingKey,
};

use super::ProverType;

/// An implementation of [crate::ProverClient] that proves on an AMD Instinct GPU through libzkm_hip.so.
pub struct HipProver {
    /// execute / verify / key formats: unchanged CPU code
    pub(crate) cpu_prover: ZKMProver<DefaultProverComponents>,
    /// core, compress and shrink machines on the GPU; wrap on the CPU (HipProverComponents)
    pub(crate) hip_prover: ZKMProver<HipProverComponents>,
}

impl HipProver {
    /// Creates a new [HipProver] on the device `ZKM_HIP_DEVICE` names (default 0). Panics when no HIP device is visible: the library has
    /// no CPU fallback, `ProverClient::cpu()` is the CPU prover.
    pub fn new() -> Self {
        Self { cpu_prover: ZKMProver::new(), hip_prover: ZKMProver::new() }
    }

    /// The options every proof of this prover runs with unless the caller passes its own (crates/stark/src/opts.rs:83-110 is the CUDA
    /// analogue): one shard in flight per context, shard size left at the default.
    pub fn default_opts() -> ProofOpts {
        ProofOpts { zkm_prover_opts: ZKMProverOpts::hip(), timeout: None }
    }

    fn wrap(&self, kind: ZKMProofKind, proof: ZKMProof, public_values: zkm_primitives::io::ZKMPublicValues) -> ZKMProofWithPublicValues {
        debug_assert!(matches!(
            (&kind, &proof),
            (ZKMProofKind::Core, ZKMProof::Core(_)) | (ZKMProofKind::Compressed, ZKMProof::Compressed(_)) | (ZKMProofKind::Plonk, ZKMProof::Plonk(_))
                | (ZKMProofKind::Groth16, ZKMProof::Groth16(_)) | (ZKMProofKind::CompressToGroth16, ZKMProof::Groth16(_))
        ));
        ZKMProofWithPublicValues { proof, public_values, zkm_version: self.version().to_string() }
    }
}

impl Default for HipProver {
    fn default() -> Self { Self::new() }
}

impl Prover<DefaultProverComponents> for HipProver {
    fn id(&self) -> ProverType {
        ProverType::Hip
    }

    /// The host key (same bytes as the CPU prover's: `ZKMProvingKey` holds `pk_to_host`, crates/prover/src/lib.rs:288-303). The device
    /// key is rebuilt from it per proof (`pk_to_device`: one upload + commit of the preprocessed traces, milliseconds).
    fn setup(&self, elf: &[u8]) -> (ZKMProvingKey, ZKMVerifyingKey) {
        let (pk, _, _, vk) = self.hip_prover.setup(elf);
        (pk, vk)
    }

    fn zkm_prover(&self) -> &ZKMProver<DefaultProverComponents> {
        &self.cpu_prover
    }

    fn prove_impl<'a>(
        &'a self,
        pk: &ZKMProvingKey,
        stdin: ZKMStdin,
        opts: ProofOpts,
        context: ZKMContext<'a>,
        kind: ZKMProofKind,
        _elf_id: Option<String>,
    ) -> Result<(ZKMProofWithPublicValues, u64)> {
        let opts = if opts.zkm_prover_opts == ZKMProverOpts::default() { Self::default_opts() } else { opts };
        if kind == ZKMProofKind::CompressToGroth16 {
            // cpu.rs:34-67 with the shrink machine on the GPU
            let mut stdin = stdin;
            assert_eq!(stdin.buffer.len(), 1);
            let public_values = bincode::deserialize(stdin.buffer.last().unwrap())?;
            assert_eq!(stdin.proofs.len(), 1);
            let (proof, _) = stdin.proofs.pop().unwrap();
            let shrink_proof = self.hip_prover.shrink(proof, opts.zkm_prover_opts)?;
            let outer_proof = self.hip_prover.wrap_bn254(shrink_proof, opts.zkm_prover_opts)?;
            let artifacts = if zkm_prover::build::zkm_dev_mode() {
                zkm_prover::build::try_build_groth16_bn254_artifacts_dev(&outer_proof.vk, &outer_proof.proof)
            } else {
                try_install_circuit_artifacts("groth16", ZKM_CIRCUIT_VERSION)
            };
            let proof = self.hip_prover.wrap_groth16_bn254(outer_proof, &artifacts);
            return Ok((self.wrap(kind, ZKMProof::Groth16(proof), public_values), 0));
        }

        let program = self.hip_prover.get_program(&pk.elf).unwrap();
        let pk_d = self.hip_prover.core_prover.pk_to_device(&pk.pk);

        // core proof: every shard through HipProver::commit / open (crates/core/machine/src/utils/prove.rs:484-497)
        let proof = self.hip_prover.prove_core(&pk_d, program, &stdin, opts.zkm_prover_opts, context)?;
        let cycles = proof.cycles;
        if kind == ZKMProofKind::Core {
            return Ok((self.wrap(kind, ZKMProof::Core(proof.proof.0), proof.public_values), cycles));
        }

        let deferred_proofs = stdin.proofs.iter().map(|(reduce_proof, _)| reduce_proof.clone()).collect();
        let public_values = proof.public_values.clone();

        // the recursion tree: compress-machine shards through the same library (crates/prover/src/lib.rs:617-957)
        let reduce_proof = self.hip_prover.compress(&pk.vk, proof, deferred_proofs, opts.zkm_prover_opts)?;
        if kind == ZKMProofKind::Compressed {
            return Ok((self.wrap(kind, ZKMProof::Compressed(Box::new(reduce_proof)), public_values), cycles));
        }

        let compress_proof = self.hip_prover.shrink(reduce_proof, opts.zkm_prover_opts)?;
        let outer_proof = self.hip_prover.wrap_bn254(compress_proof, opts.zkm_prover_opts)?;

        if kind == ZKMProofKind::Plonk {
            let artifacts = if zkm_prover::build::zkm_dev_mode() {
                zkm_prover::build::try_build_plonk_bn254_artifacts_dev(&outer_proof.vk, &outer_proof.proof)
            } else {
                try_install_circuit_artifacts("plonk", ZKM_CIRCUIT_VERSION)
            };
            let proof = self.hip_prover.wrap_plonk_bn254(outer_proof, &artifacts);
            return Ok((self.wrap(kind, ZKMProof::Plonk(proof), public_values), cycles));
        } else if kind == ZKMProofKind::Groth16 {
            let artifacts = if zkm_prover::build::zkm_dev_mode() {
                zkm_prover::build::try_build_groth16_bn254_artifacts_dev(&outer_proof.vk, &outer_proof.proof)
            } else {
                try_install_circuit_artifacts("groth16", ZKM_CIRCUIT_VERSION)
            };
            let proof = self.hip_prover.wrap_groth16_bn254(outer_proof, &artifacts);
            return Ok((self.wrap(kind, ZKMProof::Groth16(proof), public_values), cycles));
        }
        unreachable!()
    }
}
