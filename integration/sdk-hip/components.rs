//! `HipProverComponents`: the sibling of `DefaultProverComponents` (crates/prover/src/components.rs:28-35) that puts the KoalaBear /
//! Poseidon2 machines on the MI355X. Goes into crates/prover/src/components.rs (the trait and the `Air` types live in that crate) behind
//! the `hip` feature, with `zkm-hip = { path = "../hip", optional = true }` in crates/prover/Cargo.toml.
//!
//! CoreSC and InnerSC are the same config type (`KoalaBearPoseidon2`, crates/prover/src/lib.rs:99-103); they differ only in the FRI
//! parameters the machine was built with (core and compress: `default()` = log_blowup 1 / 84 queries, shrink: `compressed()` = 2 / 42;
//! crates/prover/src/lib.rs:189-199, crates/stark/src/kb31_poseidon2.rs:203-241), which `HipProver` reads from the machine's config at
//! every call. The wrap machine commits with a BN254 Poseidon2 Merkle tree (OuterSC, crates/recursion/core/src/stark/config.rs:70-83):
//! a different hasher and field of digests, not on this path — it stays on the CPU prover.
//!
//! NOT compiled where it was written (no Rust toolchain in that image).

use zkm_core_machine::mips::MipsAir;
use zkm_hip::HipProver;
use zkm_stark::{CpuProver, StarkGenericConfig};

use crate::components::ZKMProverComponents;
use crate::{CompressAir, CoreSC, InnerSC, OuterSC, ShrinkAir, WrapAir};

pub struct HipProverComponents;

impl ZKMProverComponents for HipProverComponents {
    type CoreProver = HipProver<MipsAir<<CoreSC as StarkGenericConfig>::Val>>;
    type CompressProver = HipProver<CompressAir<<InnerSC as StarkGenericConfig>::Val>>;
    type ShrinkProver = HipProver<ShrinkAir<<InnerSC as StarkGenericConfig>::Val>>;
    type WrapProver = CpuProver<OuterSC, WrapAir<<OuterSC as StarkGenericConfig>::Val>>;
}
