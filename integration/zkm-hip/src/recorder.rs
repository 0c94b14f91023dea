//! The symbolic constraint recorder: turns a chip's `Air::eval` and its lookups into the two blobs of
//! `zkm_chip_desc` — the straight-line constraint bytecode and the lookup table — that `zkm_open` evaluates on the GPU.
//!
//! Nothing here re-implements a chip: the main constraints come from the reference's own symbolic pass,
//! `p3_uni_stark::get_symbolic_constraints(&chip.air, ..)` — the call `StarkMachine::setup` already makes to count constraints
//! (crates/stark/src/machine.rs:377-382; every chip is `Air<SymbolicAirBuilder<F>>`, chip.rs:67, prover.rs:217) — and the lookups from
//! `chip.sends()` / `chip.receives()` (chip.rs:32-40). The permutation constraints are rebuilt from the lookups exactly as
//! `eval_permutation_constraints` states them (crates/stark/src/permutation.rs:205-347), so the constraint order — hence the powers
//! of alpha — is the reference's: main constraints first, then the permutation ones (chip.rs:257-276).
//!
//! The executable specification of this file is ziren_amd/air.py (`AirBuilder.assemble`, `eval_permutation_constraints`,
//! `encode_lookups`); tests/ pin that one against the oracle and the reference's per-chip costs.

use std::collections::HashMap;
use std::sync::Arc;

use p3_air::VirtualPairCol;
use p3_field::{FieldAlgebra, PrimeField32};
use p3_koala_bear::KoalaBear;
use p3_uni_stark::{get_symbolic_constraints, Entry, SymbolicExpression};
use zkm_stark::{air::{LookupScope, MachineAir}, lookup::Lookup, Chip, PROOF_MAX_NUM_PVS};

type F = KoalaBear;

// opcodes of include/zkm_hip.h ("Constraint bytecode")
const LD_MAIN: u32 = 1;
const LD_PREP: u32 = 2;
const LD_PERM: u32 = 3;
const LD_CONST: u32 = 4;
const LD_PV: u32 = 5;
const LD_CHALLENGE: u32 = 6;
const LD_LOCAL_SUM: u32 = 7;
const LD_GLOBAL_SUM: u32 = 8;
const LD_IS_FIRST: u32 = 9;
const LD_IS_LAST: u32 = 10;
const LD_IS_TRANS: u32 = 11;
const ADD_B: u32 = 16;
const SUB_B: u32 = 17;
const MUL_B: u32 = 18;
const NEG_B: u32 = 19;
const ADD_E: u32 = 20;
const SUB_E: u32 = 21;
const MUL_E: u32 = 22;
const NEG_E: u32 = 23;
const ADD_EB: u32 = 24;
const SUB_EB: u32 = 25;
const MUL_EB: u32 = 26;
const ASSERT_B: u32 = 32;
const ASSERT_E: u32 = 33;

/// Montgomery word of a field element: what crosses the ABI (KoalaBear is a repr(transparent) Montgomery u32 in Plonky3).
fn monty(x: F) -> u32 {
    // to_unique_u32 would be canonical; the in-memory word is the Montgomery form
    unsafe { core::mem::transmute::<F, u32>(x) }
}

type Id = usize;

#[derive(Clone)]
struct Node {
    op: u32,
    a: usize,        // leaf: row offset (0 local, 1 next); inner: left operand id
    c: Option<Id>,   // right operand
    imm: u32,
    ext: bool,
    uses: u32,
    reg: Option<u32>,
}

/// Expression DAG over base- and extension-field values (air.py: class Expr / AirBuilder).
pub struct Dag {
    nodes: Vec<Node>,
    leaves: HashMap<(u32, usize, u32), Id>,
    asserts: Vec<Id>,
}

impl Dag {
    fn new() -> Self { Self { nodes: Vec::new(), leaves: HashMap::new(), asserts: Vec::new() } }

    fn leaf(&mut self, op: u32, a: usize, imm: u32, ext: bool) -> Id {
        if let Some(&id) = self.leaves.get(&(op, a, imm)) { return id; }
        self.nodes.push(Node { op, a, c: None, imm, ext, uses: 0, reg: None });
        self.leaves.insert((op, a, imm), self.nodes.len() - 1);
        self.nodes.len() - 1
    }
    fn konst(&mut self, v: F) -> Id { self.leaf(LD_CONST, 0, monty(v), false) }
    fn bin(&mut self, op: u32, a: Id, c: Id, ext: bool) -> Id {
        self.nodes.push(Node { op, a, c: Some(c), imm: 0, ext, uses: 0, reg: None });
        self.nodes.len() - 1
    }
    fn ext(&self, x: Id) -> bool { self.nodes[x].ext }

    pub fn add(&mut self, x: Id, y: Id) -> Id {
        match (self.ext(x), self.ext(y)) {
            (true, true) => self.bin(ADD_E, x, y, true),
            (true, false) => self.bin(ADD_EB, x, y, true),
            (false, true) => self.bin(ADD_EB, y, x, true),
            (false, false) => self.bin(ADD_B, x, y, false),
        }
    }
    pub fn sub(&mut self, x: Id, y: Id) -> Id {
        match (self.ext(x), self.ext(y)) {
            (true, true) => self.bin(SUB_E, x, y, true),
            (true, false) => self.bin(SUB_EB, x, y, true),
            (false, true) => { let t = self.bin(SUB_EB, y, x, true); self.neg(t) } // base - ext = -(ext - base)
            (false, false) => self.bin(SUB_B, x, y, false),
        }
    }
    pub fn mul(&mut self, x: Id, y: Id) -> Id {
        match (self.ext(x), self.ext(y)) {
            (true, true) => self.bin(MUL_E, x, y, true),
            (true, false) => self.bin(MUL_EB, x, y, true),
            (false, true) => self.bin(MUL_EB, y, x, true),
            (false, false) => self.bin(MUL_B, x, y, false),
        }
    }
    pub fn neg(&mut self, x: Id) -> Id {
        let ext = self.ext(x);
        self.nodes.push(Node { op: if ext { NEG_E } else { NEG_B }, a: x, c: None, imm: 0, ext, uses: 0, reg: None });
        self.nodes.len() - 1
    }
    fn assert_zero(&mut self, x: Id) { self.asserts.push(x); }
    fn assert_eq(&mut self, x: Id, y: Id) { let d = self.sub(x, y); self.assert_zero(d); }

    /// A constraint of the chip's own `eval`, as the reference's symbolic builder recorded it.
    fn from_symbolic(&mut self, e: &SymbolicExpression<F>, memo: &mut HashMap<*const SymbolicExpression<F>, Id>) -> Id {
        let key = e as *const _;
        if let Some(&id) = memo.get(&key) { return id; }
        let id = match e {
            SymbolicExpression::Variable(v) => match v.entry {
                Entry::Main { offset } => self.leaf(LD_MAIN, offset, v.index as u32, false),
                Entry::Preprocessed { offset } => self.leaf(LD_PREP, offset, v.index as u32, false),
                Entry::Public => self.leaf(LD_PV, 0, v.index as u32, false),
                other => panic!("a chip's main constraints do not read {other:?}"),
            },
            SymbolicExpression::IsFirstRow => self.leaf(LD_IS_FIRST, 0, 0, false),
            SymbolicExpression::IsLastRow => self.leaf(LD_IS_LAST, 0, 0, false),
            SymbolicExpression::IsTransition => self.leaf(LD_IS_TRANS, 0, 0, false),
            SymbolicExpression::Constant(c) => self.konst(*c),
            SymbolicExpression::Add { x, y, .. } => { let (a, b) = (self.from_symbolic(x, memo), self.from_symbolic(y, memo)); self.add(a, b) }
            SymbolicExpression::Sub { x, y, .. } => { let (a, b) = (self.from_symbolic(x, memo), self.from_symbolic(y, memo)); self.sub(a, b) }
            SymbolicExpression::Mul { x, y, .. } => { let (a, b) = (self.from_symbolic(x, memo), self.from_symbolic(y, memo)); self.mul(a, b) }
            SymbolicExpression::Neg { x, .. } => { let a = self.from_symbolic(x, memo); self.neg(a) }
        };
        memo.insert(key, id);
        id
    }

    /// `VirtualPairCol::apply` over recorded row elements.
    fn apply(&mut self, col: &Affine) -> Id {
        let mut acc = self.konst(col.constant);
        for &(is_main, c, w) in &col.terms {
            let v = self.leaf(if is_main { LD_MAIN } else { LD_PREP }, 0, c as u32, false);
            let k = self.konst(w);
            let t = self.mul(v, k);
            acc = self.add(acc, t);
        }
        acc
    }

    /// Emit the `program` blob: header {n_instr, n_ext_regs, n_constraints, n_base_regs}, then two words per instruction
    /// (air.py `AirBuilder.assemble`: use counts over the DAG, two register files, lowest free register first).
    fn assemble(mut self) -> Vec<u32> {
        let mut seen = vec![false; self.nodes.len()];
        for &root in &self.asserts.clone() {
            let mut stack = vec![root];
            while let Some(x) = stack.pop() {
                self.nodes[x].uses += 1;
                if seen[x] { continue; }
                seen[x] = true;
                if self.nodes[x].op >= ADD_B {
                    stack.push(self.nodes[x].a);
                    if let Some(c) = self.nodes[x].c { stack.push(c); }
                }
            }
        }
        let mut instrs: Vec<(u32, u32)> = Vec::new();
        let mut free: [Vec<u32>; 2] = [Vec::new(), Vec::new()];
        let mut nregs = [0u32; 2];
        let mut alloc = |ext: bool, free: &mut [Vec<u32>; 2], nregs: &mut [u32; 2]| -> u32 {
            if let Some(r) = free[ext as usize].pop() { return r; }
            let r = nregs[ext as usize];
            nregs[ext as usize] += 1;
            assert!(r <= 255, "constraint program needs more than 256 registers");
            r
        };
        for &root in &self.asserts.clone() {
            // iterative post-order
            let mut stack = vec![(root, false)];
            while let Some((x, ready)) = stack.pop() {
                if self.nodes[x].reg.is_some() { continue; }
                let n = self.nodes[x].clone();
                if n.op < ADD_B {
                    let r = alloc(n.ext, &mut free, &mut nregs);
                    self.nodes[x].reg = Some(r);
                    instrs.push((n.op | r << 8 | ((n.a as u32) & 0xff) << 16, n.imm));
                    continue;
                }
                if !ready {
                    stack.push((x, true));
                    if let Some(c) = n.c { if self.nodes[c].reg.is_none() { stack.push((c, false)); } }
                    if self.nodes[n.a].reg.is_none() { stack.push((n.a, false)); }
                    continue;
                }
                let ra = self.nodes[n.a].reg.unwrap();
                let rc = n.c.map(|c| self.nodes[c].reg.unwrap()).unwrap_or(0);
                for operand in [Some(n.a), n.c].into_iter().flatten() {
                    self.nodes[operand].uses -= 1;
                    if self.nodes[operand].uses == 0 {
                        let (e, r) = (self.nodes[operand].ext, self.nodes[operand].reg.take().unwrap());
                        free[e as usize].push(r);
                    }
                }
                let r = alloc(n.ext, &mut free, &mut nregs);
                self.nodes[x].reg = Some(r);
                instrs.push((n.op | r << 8 | ra << 16 | rc << 24, 0));
            }
            let (e, r) = (self.nodes[root].ext, self.nodes[root].reg.unwrap());
            instrs.push(((if e { ASSERT_E } else { ASSERT_B }) | r << 16, 0));
            self.nodes[root].uses -= 1;
            if self.nodes[root].uses == 0 { self.nodes[root].reg = None; free[e as usize].push(r); }
        }
        let mut words = vec![instrs.len() as u32, nregs[1].max(1), self.asserts.len() as u32, nregs[0].max(1)];
        for (w0, w1) in instrs { words.push(w0); words.push(w1); }
        words
    }
}

/// A `VirtualPairCol` with its weights exposed: sum_i weight_i * (preprocessed | main)[col_i] + constant.
pub struct Affine { terms: Vec<(bool, usize, F)>, constant: F }

/// p3-air keeps `VirtualPairCol`'s fields private; it is affine, so probing it with unit rows recovers them exactly.
fn affine_of(col: &VirtualPairCol<F>, prep_width: usize, main_width: usize) -> Affine {
    let zero_p = vec![F::ZERO; prep_width.max(1)];
    let zero_m = vec![F::ZERO; main_width];
    let constant = col.apply::<F, F>(&zero_p, &zero_m);
    let mut terms = Vec::new();
    for c in 0..prep_width {
        let mut p = zero_p.clone();
        p[c] = F::ONE;
        let w = col.apply::<F, F>(&p, &zero_m) - constant;
        if w != F::ZERO { terms.push((false, c, w)); }
    }
    for c in 0..main_width {
        let mut m = zero_m.clone();
        m[c] = F::ONE;
        let w = col.apply::<F, F>(&zero_p, &m) - constant;
        if w != F::ZERO { terms.push((true, c, w)); }
    }
    Affine { terms, constant }
}

/// The `lookups` blob of zkm_chip_desc (include/zkm_hip.h; air.py `encode_lookups`): Local-scope sends, then receives.
fn encode_lookups(sends: &[(u32, Vec<Affine>, Affine)], receives: &[(u32, Vec<Affine>, Affine)]) -> Vec<u32> {
    let mut w = vec![sends.len() as u32, receives.len() as u32];
    for (kind, values, mult) in sends.iter().chain(receives.iter()) {
        w.push(*kind);
        w.push(values.len() as u32);
        for pc in values.iter().chain(std::iter::once(mult)) {
            w.push(pc.terms.len() as u32);
            w.push(monty(pc.constant));
            for &(is_main, col, weight) in &pc.terms {
                w.push(((is_main as u32) << 31) | col as u32);
                w.push(monty(weight));
            }
        }
    }
    w
}

/// What `HipProver::new` keeps per chip (the owner of the memory `ZkmChipDesc` points into).
pub struct RecordedChip {
    pub name: std::ffi::CString,
    pub main_width: u32,
    pub prep_width: u32,
    pub log_quotient_degree: u32,
    pub local_only: bool,
    pub commit_scope_global: bool,
    pub num_constraints: u32,
    pub lookups: Vec<u32>,
    pub program: Vec<u32>,
}

/// permutation.rs:18-23
fn local_permutation_trace_width(n_lookups: usize, batch: usize) -> usize {
    if n_lookups == 0 { 0 } else { n_lookups.div_ceil(batch) + 1 }
}

pub fn record_chip<A>(chip: &Chip<F, A>) -> RecordedChip
where
    A: MachineAir<F> + p3_air::Air<p3_uni_stark::SymbolicAirBuilder<F>>,
{
    let (prep_width, main_width) = (chip.preprocessed_width(), p3_air::BaseAir::<F>::width(chip));
    let batch = chip.logup_batch_size();
    let flatten = |ls: &[Lookup<F>]| -> Vec<(u32, Vec<Affine>, Affine)> {
        ls.iter().filter(|l| l.scope == LookupScope::Local)
            .map(|l| (l.kind as u32, l.values.iter().map(|v| affine_of(v, prep_width, main_width)).collect(), affine_of(&l.multiplicity, prep_width, main_width)))
            .collect()
    };
    let (sends, receives) = (flatten(chip.sends()), flatten(chip.receives()));
    let mut dag = Dag::new();
    // 1. the chip's own constraints, in eval order
    let mut memo = HashMap::new();
    let constraints: Vec<SymbolicExpression<F>> = get_symbolic_constraints(&chip.air, prep_width, PROOF_MAX_NUM_PVS);
    for c in &constraints {
        let id = dag.from_symbolic(c, &mut memo);
        dag.assert_zero(id);
    }
    // 2. eval_permutation_constraints (permutation.rs:205-347)
    let width = local_permutation_trace_width(sends.len() + receives.len(), batch);
    let (alpha, beta) = (dag.leaf(LD_CHALLENGE, 0, 0, true), dag.leaf(LD_CHALLENGE, 0, 1, true));
    if width > 0 {
        let lookups: Vec<(&(u32, Vec<Affine>, Affine), bool)> = sends.iter().map(|l| (l, true)).chain(receives.iter().map(|l| (l, false))).collect();
        for (k, chunk) in lookups.chunks(batch).enumerate() {
            let entry = dag.leaf(LD_PERM, 0, k as u32, true);
            let (mut rlcs, mut mults) = (Vec::new(), Vec::new());
            for ((kind, values, mult), is_send) in chunk {
                let kk = dag.konst(F::from_canonical_u32(*kind));
                let mut rlc = dag.add(alpha, kk);                 // beta^0 * argument_index
                let mut bp = beta;
                for (i, v) in values.iter().enumerate() {
                    let val = dag.apply(v);
                    let t = dag.mul(bp, val);
                    rlc = dag.add(rlc, t);
                    if i + 1 < values.len() { bp = dag.mul(bp, beta); }
                }
                rlcs.push(rlc);
                let m = dag.apply(mult);
                mults.push(if *is_send { m } else { dag.neg(m) });
            }
            let (mut product, mut numerator): (Option<Id>, Option<Id>) = (None, None);
            for i in 0..rlcs.len() {
                product = Some(match product { None => rlcs[i], Some(p) => dag.mul(p, rlcs[i]) });
                let mut all_but: Option<Id> = None;
                for (j, &other) in rlcs.iter().enumerate() {
                    if j != i { all_but = Some(match all_but { None => other, Some(a) => dag.mul(a, other) }); }
                }
                let term = match all_but { None => mults[i], Some(a) => dag.mul(a, mults[i]) };
                numerator = Some(match numerator { None => term, Some(n) => dag.add(n, term) });
            }
            let lhs = dag.mul(product.unwrap(), entry);
            dag.assert_eq(lhs, numerator.unwrap());
        }
        let (mut sum_local, mut sum_next): (Option<Id>, Option<Id>) = (None, None);
        for k in 0..width - 1 {
            let (x, y) = (dag.leaf(LD_PERM, 0, k as u32, true), dag.leaf(LD_PERM, 1, k as u32, true));
            sum_local = Some(match sum_local { None => x, Some(s) => dag.add(s, x) });
            sum_next = Some(match sum_next { None => y, Some(s) => dag.add(s, y) });
        }
        let (phi_local, phi_next) = (dag.leaf(LD_PERM, 0, (width - 1) as u32, true), dag.leaf(LD_PERM, 1, (width - 1) as u32, true));
        let (first, last, trans) = (dag.leaf(LD_IS_FIRST, 0, 0, false), dag.leaf(LD_IS_LAST, 0, 0, false), dag.leaf(LD_IS_TRANS, 0, 0, false));
        let d = dag.sub(phi_local, sum_local.unwrap());
        let c = dag.mul(d, first); dag.assert_zero(c);                                     // when_first_row
        let d = dag.sub(phi_next, phi_local); let d = dag.sub(d, sum_next.unwrap());
        let c = dag.mul(d, trans); dag.assert_zero(c);                                     // when_transition
        let ls = dag.leaf(LD_LOCAL_SUM, 0, 0, true);
        let d = dag.sub(phi_local, ls);
        let c = dag.mul(d, last); dag.assert_zero(c);                                      // when_last_row
    }
    let commit_scope_global = chip.commit_scope() == LookupScope::Global;
    if commit_scope_global {
        let last = dag.leaf(LD_IS_LAST, 0, 0, false);
        for i in 0..7u32 {
            for (col, word) in [(main_width as u32 - 14 + i, i), (main_width as u32 - 7 + i, 7 + i)] {
                let (cell, g) = (dag.leaf(LD_MAIN, 0, col, false), dag.leaf(LD_GLOBAL_SUM, 0, word, false));
                let d = dag.sub(cell, g);
                let c = dag.mul(d, last);
                dag.assert_zero(c);
            }
        }
    }
    let lookups = encode_lookups(&sends, &receives);
    let program = dag.assemble();
    RecordedChip {
        name: std::ffi::CString::new(chip.name()).unwrap(),
        main_width: main_width as u32,
        prep_width: prep_width as u32,
        log_quotient_degree: chip.log_quotient_degree() as u32,
        local_only: chip.local_only(),
        commit_scope_global,
        num_constraints: program[2],
        lookups,
        program,
    }
}

/// Canonical value of a Montgomery word (for diagnostics only).
#[allow(dead_code)]
fn canonical(x: F) -> u32 { x.as_canonical_u32() }

#[allow(dead_code)]
fn _assert_send<T: Send>() {}
#[allow(dead_code)]
fn _arc_unused(_: Arc<()>) {}
