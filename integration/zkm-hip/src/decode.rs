//! The flat ShardProof stream of `zkm_open` / `zkm_prove_shard` -> `ShardProof<KoalaBearPoseidon2>`
//! (crates/stark/src/types.rs:37-83; the opening proof's type is `InnerFriProof`, crates/stark/src/kb31_poseidon2.rs:37-44).
//!
//! Field elements arrive as the words Plonky3 keeps in memory (Montgomery u32), so a word *is* a `KoalaBear`
//! (`#[repr(transparent)]`), four words are a `BinomialExtensionField<KoalaBear, 4>`, eight words a digest.

use hashbrown::HashMap;
use p3_commit::BatchOpening;
use p3_field::{extension::BinomialExtensionField, FieldExtensionAlgebra};
use p3_fri::{CommitPhaseProofStep, FriProof, QueryProof};
use p3_koala_bear::KoalaBear;
use zkm_stark::{
    koala_bear_poseidon2::KoalaBearPoseidon2, septic_curve::SepticCurve, septic_digest::SepticDigest, septic_extension::SepticExtension,
    AirOpenedValues, ChipOpenedValues, ShardCommitment, ShardOpenedValues, ShardProof,
};

type F = KoalaBear;
type EF = BinomialExtensionField<F, 4>;

pub struct Reader<'a> { w: &'a [u32], pos: usize }

impl<'a> Reader<'a> {
    pub fn new(w: &'a [u32]) -> Self { Self { w, pos: 0 } }
    fn u(&mut self) -> usize { self.pos += 1; self.w[self.pos - 1] as usize }
    fn f(&mut self) -> F { self.pos += 1; unsafe { core::mem::transmute::<u32, F>(self.w[self.pos - 1]) } }
    fn ext(&mut self) -> EF { let c: [F; 4] = core::array::from_fn(|_| self.f()); EF::from_base_slice(&c) }
    fn digest(&mut self) -> [F; 8] { core::array::from_fn(|_| self.f()) }
    fn exts(&mut self, n: usize) -> Vec<EF> { (0..n).map(|_| self.ext()).collect() }
    fn path(&mut self) -> Vec<[F; 8]> { let n = self.u(); (0..n).map(|_| self.digest()).collect() }
    fn done(&self) -> bool { self.pos == self.w.len() }
}

/// `names[i]` = name of the chip at caller index i (the order `zkm_commit` received them in).
pub fn decode_shard_proof(words: &[u32], names: &[String]) -> ShardProof<KoalaBearPoseidon2> {
    let mut r = Reader::new(words);
    let commitment = ShardCommitment { main_commit: r.digest().into(), permutation_commit: r.digest().into(), quotient_commit: r.digest().into() };
    let n_chips = r.u();
    let mut chips = Vec::with_capacity(n_chips);
    let mut chip_ordering = HashMap::new();
    for position in 0..n_chips {
        let caller_index = r.u();
        chip_ordering.insert(names[caller_index].clone(), position);   // types.rs:83: name -> position in `opened_values.chips`
        let log_degree = r.u();
        let mut opened = |r: &mut Reader| { let w = r.u(); AirOpenedValues { local: r.exts(w), next: r.exts(w) } };
        let preprocessed = opened(&mut r);
        let main = opened(&mut r);
        // the stream carries the permutation trace's *base* columns evaluated at zeta; the reference keeps them the same way
        // (flattened ext columns, prover.rs:596-607)
        let permutation = opened(&mut r);
        let n_chunks = r.u();
        let quotient: Vec<Vec<EF>> = (0..n_chunks).map(|_| r.exts(4)).collect();
        let x: [F; 7] = core::array::from_fn(|_| r.f());
        let y: [F; 7] = core::array::from_fn(|_| r.f());
        let global_cumulative_sum = SepticDigest(SepticCurve { x: SepticExtension(x), y: SepticExtension(y) });
        let local_cumulative_sum = r.ext();
        chips.push(ChipOpenedValues { preprocessed, main, permutation, quotient, global_cumulative_sum, local_cumulative_sum, log_degree });
    }
    // FriProof (p3-fri; fields as the in-tree witness code names them, crates/recursion/circuit/src/witness/stark.rs:78-141)
    let n_commit = r.u();
    let commit_phase_commits = (0..n_commit).map(|_| r.digest().into()).collect();
    let n_queries = r.u();
    let mut query_proofs = Vec::with_capacity(n_queries);
    for _ in 0..n_queries {
        let n_rounds = r.u();
        let mut input_proof = Vec::with_capacity(n_rounds);
        for _ in 0..n_rounds {
            let n_mats = r.u();
            let opened_values: Vec<Vec<F>> = (0..n_mats).map(|_| { let w = r.u(); (0..w).map(|_| r.f()).collect() }).collect();
            let opening_proof = r.path();
            input_proof.push(BatchOpening { opened_values, opening_proof });
        }
        let n_steps = r.u();
        let commit_phase_openings = (0..n_steps).map(|_| { let sibling_value = r.ext(); CommitPhaseProofStep { sibling_value, opening_proof: r.path() } }).collect();
        query_proofs.push(QueryProof { input_proof, commit_phase_openings });
    }
    let final_poly = r.ext();
    let pow_witness = r.f();
    let opening_proof = FriProof { commit_phase_commits, query_proofs, final_poly, pow_witness };
    let n_pv = r.u();
    let public_values = (0..n_pv).map(|_| r.f()).collect();
    assert!(r.done(), "trailing words in the ShardProof stream");
    ShardProof { commitment, opened_values: ShardOpenedValues { chips }, opening_proof, chip_ordering, public_values }
}
