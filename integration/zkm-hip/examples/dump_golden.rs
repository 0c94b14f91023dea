//! One bounded file for the day a box with cargo and the vendored Plonky3 fork exists: linked against the real Ziren
//! workspace it writes bytes the REFERENCE produced into tests/golden/from_reference/, and tests/test_reference_goldens.py then holds the
//! in-repo oracle and the GPU path against them — the one thing `parity: partial` waits for. NOT compiled where it was written (no Rust
//! toolchain in that image): names follow the reference tree as read (file:line beside each use).
//!
//!   cargo run --release --example dump_golden -- <repo>/tests/golden/from_reference
//!
//! Files are text, one record per line: `name w0 w1 ...` in decimal. A field element is written as the word Plonky3 keeps in memory
//! (Montgomery u32, `KoalaBear` is #[repr(transparent)]) — the representation of include/zkm_hip.h. Inputs are given by formula, not stored:
//!   value(seed, r, c) = ((seed * 2654435761 + r * 40503 + c * 9973 + 12345) mod 2^32) mod p        (canonical)
//!
//! 1. pcs_size_gaps.txt — `TwoAdicFriPcs::commit` over the size-gap batch of crates/recursion/circuit/src/fri.rs:580-624 at power-of-two
//!    heights (4 x 1024, 5 x 64, 6 x 8 rows, 8 columns; seeds 1..15): the commitment, and `open_batch(6)` on the committed LDEs
//!    (opened rows + sibling digests). Pins coset LDE, row bit-reversal, the sponge, compression and mixed-height injection.
//! 2. simple_program_shard_<k>.txt — every `ShardProof` `CpuProver` makes for `simple_program()`
//!    (crates/core/executor/src/programs.rs:15-22) through `run_test_core` without a shape config (utils/prove.rs:628-656), as the flat
//!    stream of INTEGRATION.md section 3 with `caller index := position` and the chip names beside it. Pins the transcript, LogUp,
//!    quotient, openings, FRI betas / query indices / proof-of-work and the stream order — fri.rs:626-815's checks are contained in it.
//! 3. airs.txt — for every chip of the core machine (`MipsAir`) and of the compress machine (`RecursionAir<_, 3>`): widths, lookup counts by
//!    kind, and the value of EVERY main constraint, in evaluation order, at one fixed point (`get_symbolic_constraints`,
//!    crates/stark/src/machine.rs:377-390, chip.rs:65-90; variables by the formula above: main local / next = value(7, 0 / 1, c),
//!    preprocessed = value(11, 0 / 1, c), public value i = value(13, 0, i), is_first_row / is_last_row / is_transition = value(17, 0, 0 / 1 / 2)).
//!    Pins the hand-transcribed AIRs of ziren_amd/chips.py and recursion.py: a missing, extra, reordered or altered constraint changes the
//!    list (the order fixes the powers of alpha, prover.rs:447-456). tests/test_reference_goldens.py::test_recorded_airs_equal_the_references.
use std::{fmt::Write as _, fs, path::PathBuf};

use p3_commit::{Mmcs, Pcs};
use p3_field::AbstractField;
use p3_koala_bear::KoalaBear;
use p3_matrix::dense::RowMajorMatrix;
use zkm_core_executor::{programs::tests::simple_program, Executor, ZKMCoreOpts};
use zkm_core_machine::{io::ZKMStdin, mips::MipsAir, utils::run_test_core};
use zkm_stark::{
    inner_perm, koala_bear_poseidon2::KoalaBearPoseidon2, AirOpenedValues, CpuProver, InnerChallenger, InnerDft, InnerHash, InnerCompress, InnerPcs,
    InnerValMmcs, ShardProof, StarkGenericConfig,
};

type F = KoalaBear;
const P: u64 = 0x7f00_0001;

fn word(x: F) -> u32 { unsafe { core::mem::transmute::<F, u32>(x) } }      // the Montgomery word
fn value(seed: u32, r: u32, c: u32) -> F {
    let x = seed.wrapping_mul(2_654_435_761).wrapping_add(r.wrapping_mul(40_503)).wrapping_add(c.wrapping_mul(9_973)).wrapping_add(12_345);
    F::from_canonical_u32((x as u64 % P) as u32)
}
fn matrix(seed: u32, h: usize, w: usize) -> RowMajorMatrix<F> {
    RowMajorMatrix::new((0..h * w).map(|i| value(seed, (i / w) as u32, (i % w) as u32)).collect(), w)
}
fn record(out: &mut String, name: &str, words: impl IntoIterator<Item = u32>) {
    write!(out, "{name}").unwrap();
    for w in words { write!(out, " {w}").unwrap(); }
    out.push('\n');
}

fn pcs_size_gaps() -> String {
    let perm = inner_perm();                                                                       // crates/stark/src/kb31_poseidon2.rs:48
    let mmcs = InnerValMmcs::new(InnerHash::new(perm.clone()), InnerCompress::new(perm.clone()));
    let pcs = InnerPcs::new(InnerDft::default(), mmcs.clone(), zkm_stark::inner_fri_config());     // :69, log_blowup 1
    let shapes: Vec<usize> = [vec![1024; 4], vec![64; 5], vec![8; 6]].concat();
    let batch = shapes.iter().enumerate().map(|(i, &h)| {
        (<InnerPcs as Pcs<_, InnerChallenger>>::natural_domain_for_degree(&pcs, h), matrix(i as u32 + 1, h, 8))
    }).collect::<Vec<_>>();
    let (commit, data) = <InnerPcs as Pcs<_, InnerChallenger>>::commit(&pcs, batch);
    let (opened, proof) = mmcs.open_batch(6, &data);
    let mut out = String::new();
    let digest: [F; 8] = commit.into();
    record(&mut out, "commit", digest.map(word));
    for (i, row) in opened.iter().enumerate() { record(&mut out, &format!("opened_{i}"), row.iter().map(|&x| word(x))); }
    for (i, d) in proof.iter().enumerate() { record(&mut out, &format!("sibling_{i}"), d.map(word)); }
    out
}

/// The flat stream of INTEGRATION.md section 3 (what integration/zkm-hip/src/decode.rs reads back), chips by position.
fn stream(p: &ShardProof<KoalaBearPoseidon2>) -> (Vec<u32>, Vec<String>) {
    let mut w: Vec<u32> = vec![];
    let ext = |w: &mut Vec<u32>, e: &<KoalaBearPoseidon2 as StarkGenericConfig>::Challenge| {
        use p3_field::AbstractExtensionField;
        w.extend(<_ as AbstractExtensionField<F>>::as_base_slice(e).iter().map(|&x| word(x)));
    };
    let digest = |w: &mut Vec<u32>, d: [F; 8]| w.extend(d.map(word));
    digest(&mut w, p.commitment.main_commit.into());
    digest(&mut w, p.commitment.permutation_commit.into());
    digest(&mut w, p.commitment.quotient_commit.into());
    let mut names = vec![String::new(); p.opened_values.chips.len()];
    for (name, &pos) in p.chip_ordering.iter() { names[pos] = name.clone(); }                      // crates/stark/src/types.rs:83
    w.push(names.len() as u32);
    for (pos, c) in p.opened_values.chips.iter().enumerate() {
        w.push(pos as u32);
        w.push(c.log_degree as u32);
        let opened = |w: &mut Vec<u32>, o: &AirOpenedValues<_>| {
            w.push(o.local.len() as u32);
            for e in o.local.iter().chain(o.next.iter()) { ext(w, e); }
        };
        opened(&mut w, &c.preprocessed);
        opened(&mut w, &c.main);
        opened(&mut w, &c.permutation);
        w.push(c.quotient.len() as u32);
        for chunk in &c.quotient { for e in chunk { ext(&mut w, e); } }
        w.extend(c.global_cumulative_sum.0.x.0.map(word));
        w.extend(c.global_cumulative_sum.0.y.0.map(word));
        ext(&mut w, &c.local_cumulative_sum);
    }
    let f = &p.opening_proof;                                                                      // p3-fri FriProof
    w.push(f.commit_phase_commits.len() as u32);
    for c in &f.commit_phase_commits { digest(&mut w, (*c).into()); }
    w.push(f.query_proofs.len() as u32);
    for q in &f.query_proofs {
        w.push(q.input_proof.len() as u32);
        for b in &q.input_proof {
            w.push(b.opened_values.len() as u32);
            for row in &b.opened_values { w.push(row.len() as u32); w.extend(row.iter().map(|&x| word(x))); }
            w.push(b.opening_proof.len() as u32);
            for d in &b.opening_proof { w.extend(d.map(word)); }
        }
        w.push(q.commit_phase_openings.len() as u32);
        for s in &q.commit_phase_openings {
            ext(&mut w, &s.sibling_value);
            w.push(s.opening_proof.len() as u32);
            for d in &s.opening_proof { w.extend(d.map(word)); }
        }
    }
    ext(&mut w, &f.final_poly);
    w.push(word(f.pow_witness));
    w.push(p.public_values.len() as u32);
    w.extend(p.public_values.iter().map(|&x| word(x)));
    (w, names)
}

/// A symbolic constraint at the fixed point (p3-uni-stark `SymbolicExpression`, `Entry`: as read from the fork's symbolic_builder).
fn at_point(e: &p3_uni_stark::SymbolicExpression<F>) -> F {
    use p3_uni_stark::{Entry, SymbolicExpression as E};
    match e {
        E::Variable(v) => match v.entry {
            Entry::Main { offset } => value(7, offset as u32, v.index as u32),
            Entry::Preprocessed { offset } => value(11, offset as u32, v.index as u32),
            Entry::Public => value(13, 0, v.index as u32),
            _ => panic!("a main constraint reads the permutation trace or a challenge"),
        },
        E::IsFirstRow => value(17, 0, 0),
        E::IsLastRow => value(17, 0, 1),
        E::IsTransition => value(17, 0, 2),
        E::Constant(c) => *c,
        E::Add { x, y, .. } => at_point(x) + at_point(y),
        E::Sub { x, y, .. } => at_point(x) - at_point(y),
        E::Neg { x, .. } => -at_point(x),
        E::Mul { x, y, .. } => at_point(x) * at_point(y),
    }
}

fn airs<A>(out: &mut String, machine: &str, chips: &[zkm_stark::Chip<F, A>])
where A: zkm_stark::air::MachineAir<F> + for<'a> p3_air::Air<p3_uni_stark::SymbolicAirBuilder<F>> {
    for chip in chips {
        let cs = p3_uni_stark::get_symbolic_constraints(chip.air(), chip.preprocessed_width(), zkm_stark::PROOF_MAX_NUM_PVS);
        let kinds = |ls: &[zkm_stark::lookup::Lookup<F>]| { let mut k: Vec<String> = ls.iter().map(|l| format!("{:?}", l.kind)).collect(); k.sort(); k.join(",") };
        writeln!(out, "chip {machine} {} prep {} main {} lqd {} constraints {} sends [{}] receives [{}]", chip.name(), chip.preprocessed_width(), chip.width(),
                 chip.log_quotient_degree(), cs.len(), kinds(chip.sends()), kinds(chip.receives())).unwrap();
        record(out, &format!("values_{machine}_{}", chip.name()), cs.iter().map(|c| word(at_point(c))));
    }
}

fn main() {
    let dir = PathBuf::from(std::env::args().nth(1).expect("usage: dump_golden <out dir>"));
    fs::create_dir_all(&dir).unwrap();
    fs::write(dir.join("pcs_size_gaps.txt"), pcs_size_gaps()).unwrap();
    let mut a = String::new();
    airs(&mut a, "core", MipsAir::<F>::machine(KoalaBearPoseidon2::new()).chips());                                   // crates/core/machine/src/mips/mod.rs
    airs(&mut a, "compress", zkm_recursion_core::machine::RecursionAir::<F, 3>::compress_machine(KoalaBearPoseidon2::new()).chips());   // recursion/core/src/machine.rs:112-132
    fs::write(dir.join("airs.txt"), a).unwrap();
    let mut runtime = Executor::new(simple_program(), ZKMCoreOpts::default());
    runtime.run().unwrap();
    let proof = run_test_core::<CpuProver<KoalaBearPoseidon2, MipsAir<F>>>(runtime, ZKMStdin::new(), None).unwrap();   // proves and verifies
    for (k, shard) in proof.shard_proofs.iter().enumerate() {
        let (words, names) = stream(shard);
        let mut out = String::new();
        writeln!(out, "names {}", names.join(" ")).unwrap();
        record(&mut out, "stream", words);
        fs::write(dir.join(format!("simple_program_shard_{k}.txt")), out).unwrap();
    }
}
