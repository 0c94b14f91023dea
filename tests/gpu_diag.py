"""Staged GPU-vs-oracle diagnostics (run on the GPU box): prints where parity breaks."""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
from ziren_amd import prover, abi, synth, field as F

P = F.P
rng = np.random.default_rng(7)
ctx = prover.Context(0)
ok_all = True


def report(name, ok, extra=""):
    global ok_all
    ok_all &= bool(ok)
    print(("PASS " if ok else "FAIL ") + name + (" " + extra if extra else ""), flush=True)


def rand(shape):
    return rng.integers(0, P, shape, dtype=np.uint64).astype(np.uint32)


def stage(fn):
    try:
        fn()
    except Exception as e:
        global ok_all
        ok_all = False
        print("EXC  " + fn.__name__ + ": " + repr(e), flush=True)
        traceback.print_exc()


def s_poseidon2():
    st = rand((1000, 16))
    report("poseidon2_permute_batch", np.array_equal(prover.poseidon2_permute_batch(ctx, st), O.poseidon2_permute_batch(st)))


def s_transpose():
    for h, w in [(1, 1), (2, 3), (64, 5), (1024, 67), (4096, 33)]:
        m = rand((h, w))
        d = ctx.upload(m)
        report(f"upload/download {h}x{w}", np.array_equal(d.to_host(), m))


def s_lde():
    for k, w, bl in [(0, 1, 1), (1, 2, 1), (3, 3, 1), (5, 4, 2), (8, 5, 1), (10, 3, 3), (13, 2, 1), (14, 3, 1), (15, 2, 2), (17, 2, 1)]:
        m = rand((1 << k, w))
        shift = F.to_monty(3)
        t = time.time(); got = prover.coset_lde_batch(ctx, m, bl, shift); tg = time.time() - t
        exp = O.coset_lde_batch(m, bl, shift)
        eq = np.array_equal(got, exp)
        extra = ""
        if not eq:
            bad = np.argwhere(got != exp)
            extra = f"mismatches={len(bad)} first={bad[:5].tolist()} got={got[tuple(bad[0])]} exp={exp[tuple(bad[0])]}"
        report(f"lde k={k} w={w} bl={bl}", eq, extra)
    # quotient-chunk style shift
    k = 9
    m = rand((1 << k, 4))
    sh = F.to_monty(F.inv(F.two_adic_generator(k + 1)))
    report("lde shift=w^-1", np.array_equal(prover.coset_lde_batch(ctx, m, 1, sh), O.coset_lde_batch(m, 1, sh)))


def s_commit():
    cases = [[(8, 3)], [(64, 8)], [(1024, 67), (1024, 5), (512, 9), (64, 17), (8, 8)], [(1000 if False else 1024, 8)] * 2 + [(64, 8)] * 3]
    for shapes in cases:
        mats = [rand(s) for s in shapes]
        root_o, ldes_o, layers_o = O.pcs_commit(mats, 1)
        dm = [ctx.upload(m) for m in mats]
        d = prover.pcs_commit(ctx, dm, 1)
        eq = np.array_equal(d.root, root_o)
        report(f"pcs_commit root {shapes}", eq)
        idx = 5 % (shapes[0][0] * 2)
        v, pr = d.open_batch(idx)
        vo, po, okv = O.pcs_open_batch(mats, 1, idx)
        report(f"open_batch {shapes}", np.array_equal(v, vo) and np.array_equal(pr, po) and okv)
    # with domain shifts (quotient chunks)
    mats = [rand((256, 4)), rand((256, 4))]
    sh = [F.to_monty(3), F.to_monty(3 * F.two_adic_generator(9) % P)]
    root_o, _, _ = O.pcs_commit(mats, 1, domain_shifts=sh)
    d = prover.pcs_commit(ctx, [ctx.upload(m) for m in mats], 1, domain_shifts=sh)
    report("pcs_commit shifted domains", np.array_equal(d.root, root_o))


def shard_case(k, with_prep, queries=10, pow_bits=8):
    sh = synth.syn_shard(k, with_prep=with_prep)
    fri = abi.FriConfig(1, queries, pow_bits)
    prep_tr = [c.prep_trace for c in sh.chips if c.prep_width]
    opk = O.Pk(prep_tr, [0] * len(prep_tr), sh.pc_start, sh.initial_global_cumulative_sum, 1)
    och = O.new_challenger(); opk.observe_into(och); vch = och.copy()
    oproof, _ = O.prove_shard(opk, sh.chips, [c.trace for c in sh.chips], sh.public_values, fri, synth.NUM_PV_ELTS, och)
    hp = prover.HipProver(sh.chips, fri, synth.NUM_PV_ELTS, ctx=ctx)
    pk = hp.setup(prep_tr, [0] * len(prep_tr), sh.pc_start, sh.initial_global_cumulative_sum)
    report(f"SYN-{k} prep={with_prep} pk commit", np.array_equal(pk.commit, opk.commitment()))
    gch = prover.new_challenger(); pk.observe_into(gch)
    traces = hp.upload_traces([c.trace for c in sh.chips])
    data = hp.commit(sh.public_values, traces)
    report(f"SYN-{k} main commit", np.array_equal(data.main_commit, oproof[:8]))
    t = time.time()
    gproof = hp.open(pk, data, gch)
    print("   open wall", time.time() - t, ctx.last_timings(), flush=True)
    same = len(gproof) == len(oproof) and np.array_equal(gproof, oproof)
    extra = ""
    if not same:
        nmin = min(len(gproof), len(oproof))
        bad = np.nonzero(gproof[:nmin] != oproof[:nmin])[0]
        extra = f"len gpu={len(gproof)} oracle={len(oproof)} first_diff={bad[:8].tolist()}"
    report(f"SYN-{k} prep={with_prep} proof stream bit-exact", same, extra)
    report(f"SYN-{k} challenger state", gch.as_tuple() == och.as_tuple())
    verdict = O.verify_shard(opk, sh.chips, fri, synth.NUM_PV_ELTS, vch, gproof)
    report(f"SYN-{k} oracle verifier accepts GPU proof", verdict == 0, f"verdict={verdict}")


def s_shard_small():
    shard_case(6, False)
    shard_case(9, True)


def s_shard_mid():
    shard_case(13, True, queries=20, pow_bits=12)


def s_shard_16():
    # GPU-only prove at 2^16 (needs the multi-pass LDE), oracle only verifies
    k = 17
    sh = synth.syn_shard(k, with_prep=True)
    fri = abi.FriConfig(1, 84, 16)
    prep_tr = [c.prep_trace for c in sh.chips if c.prep_width]
    hp = prover.HipProver(sh.chips, fri, synth.NUM_PV_ELTS, ctx=ctx)
    pk = hp.setup(prep_tr, [0] * len(prep_tr), sh.pc_start, sh.initial_global_cumulative_sum)
    gch = prover.new_challenger(); pk.observe_into(gch); vch = gch.copy()
    traces = hp.upload_traces([c.trace for c in sh.chips])
    t = time.time()
    proof = hp.prove_shard(pk, sh.public_values, traces, gch).copy()
    print("   prove_shard wall", time.time() - t, ctx.last_timings(), flush=True)
    # the oracle verifier only needs the commitment of the preprocessed traces: recompute with the oracle
    opk = O.Pk(prep_tr, [0] * len(prep_tr), sh.pc_start, sh.initial_global_cumulative_sum, 1)
    report("SYN-17 pk commit", np.array_equal(pk.commit, opk.commitment()))
    verdict = O.verify_shard(opk, sh.chips, fri, synth.NUM_PV_ELTS, vch, proof)
    report("SYN-17 oracle verifier accepts GPU proof", verdict == 0, f"verdict={verdict}")


stages = [s_poseidon2, s_transpose, s_lde, s_commit, s_shard_small, s_shard_mid, s_shard_16]
sel = sys.argv[1:]
for s in stages:
    if sel and s.__name__ not in sel:
        continue
    print("== " + s.__name__, flush=True)
    stage(s)
print("ALL PASS" if ok_all else "SOME FAILED")
