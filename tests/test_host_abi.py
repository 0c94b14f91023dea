"""CPU-only: the C-ABI library loads, exports every symbol include/zkm_hip.h declares, fails loudly
without a GPU, and its host-side transcript arithmetic (Montgomery field, extension, Poseidon2,
duplex challenger) matches the canonical-arithmetic oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from ziren_amd import abi, air, lib, prover, synth, field as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "zkm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(zkm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(lib.EXPORTS)
    # ... and the library exports nothing beyond the header (dynamic symbol table of the built .so)
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("zkm_")}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    for name in declared:
        assert hasattr(L, name), name


def _addsub_chip():
    from ziren_amd import chips, events as E
    return chips.record_chip(E.CHIP_ADD_SUB, 6)


def _build_c_consumer(tmp_path, with_desc=False):
    import subprocess
    exe = str(tmp_path / "consumer")
    extra = []
    if with_desc:     # the AddSub chip's descriptor as the recorder emits it, as C arrays: what the Rust shim would hand to zkm_open
        c = _addsub_chip()
        arr = lambda name, a: f"static const uint32_t {name}[] = {{" + ", ".join(f"{int(x)}u" for x in a) + "};\n"   # noqa: E731
        (tmp_path / "addsub_desc.h").write_text(
            arr("ADDSUB_PROGRAM", c.program) + arr("ADDSUB_LOOKUPS", c.lookups_blob) +
            f"enum {{ ADDSUB_NUM_CONSTRAINTS = {c.num_constraints}, ADDSUB_LQD = {c.log_quotient_degree}, N_PUBLIC_VALUES = {synth.PROOF_MAX_NUM_PVS}, "
            f"NUM_PV_ELTS = {synth.NUM_PV_ELTS} }};\n")
        extra = ["-DZKM_HAVE_ADDSUB_DESC", "-I", str(tmp_path)]
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include")] + extra +
                          [os.path.join(ROOT, "tests", "c_abi", "consumer.c"), "-L", os.path.join(ROOT, "ziren_amd"), "-lzkm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "ziren_amd"), "-o", exe])
    return exe


def test_rust_ffi_is_generated_from_the_header():
    """integration/zkm-hip/src/ffi.rs (the Rust side of the boundary) is generated from include/zkm_hip.h: regenerating it gives the
    committed file, and it declares every entry point the library exports."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    text, names = g.gen(os.path.join(ROOT, "include", "zkm_hip.h"))
    assert text == open(os.path.join(ROOT, "integration", "zkm-hip", "src", "ffi.rs")).read(), "run python tools/gen_rust_ffi.py"
    assert set(names) == set(lib.EXPORTS)
    for f in ("lib.rs", "recorder.rs", "decode.rs"):
        src = open(os.path.join(ROOT, "integration", "zkm-hip", "src", f)).read()
        assert src.count("{") == src.count("}") and src.count("(") == src.count(")"), f      # not compiled here: at least balanced


def test_plain_c_consumer_links_and_is_refused_without_a_gpu(tmp_path):
    """include/zkm_hip.h is a C header a foreign-language binding can consume as is: a C11 program compiles against it with -Wall
    -Werror, links to libzkm_hip.so, and without a GPU gets a refusal (never a CPU fallback)."""
    import subprocess
    exe = _build_c_consumer(tmp_path)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and out.stdout.startswith("refused:") and "no CPU fallback" in out.stdout, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_gpu_plain_c_consumer(tmp_path, hip_ctx, oracle):
    """The same C program on a GPU: AluEvent -> device trace -> commitment, checked against the oracle's root."""
    import subprocess
    from ziren_amd import events as E
    exe = _build_c_consumer(tmp_path)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok "), (out.returncode, out.stdout, out.stderr)
    ev = E.make_alu_events([E.ADD], [5], [7], pc0=0x1000)
    want = oracle.pcs_commit([oracle.tracegen_alu(E.CHIP_ADD_SUB, ev)], 1)[0]
    assert int(out.stdout.split()[1]) == int(want[0])
    # ... and two GlobalLookupEvents -> the Global chip's trace: the digest the C program printed is the oracle's
    from ziren_amd import miniexec as M
    ge = np.zeros(2, dtype=M.GLOBAL_LOOKUP_EVENT)
    ge["message"][0, :3] = [0, 0, 8]
    ge["message"][1, :4] = [1, 4003, 8, 12]
    ge["is_receive"], ge["kind"] = [1, 0], 1
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("global ")][0]
    assert [int(w) for w in line.split()[1:]] == oracle.tracegen_global(ge, -1)[-1, 85:].tolist()


def test_no_gpu_means_loud_failure():
    import subprocess, sys
    # in a process that cannot see a GPU the context must refuse, not fall back
    code = ("import ctypes as C; from ziren_amd import lib; L = lib.load(); h = C.c_void_p();"
            "rc = L.zkm_ctx_create(0, C.byref(h)); print(rc, L.zkm_last_error().decode())")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT).stdout
    assert out.startswith("-1 ") and "no CPU fallback" in out, out


@pytest.mark.gpu
def test_gpu_plain_c_consumer_proves_a_shard(tmp_path, oracle):
    """A C program with no Python in the process drives the whole hot path through the ABI — descriptor, device trace generation,
    zkm_pk_setup, transcript, zkm_commit, zkm_open (with the too-small-buffer retry) — and its proof stream is the oracle's, word for
    word (compared through length, FNV-1a hash, the main commitment and the next transcript sample)."""
    import subprocess
    from ziren_amd import events as E
    exe = _build_c_consumer(tmp_path, with_desc=True)
    out = subprocess.run([exe, "prove"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("proof "), (out.returncode, out.stdout, out.stderr)
    n_words, fnv, commit0, sample = (int(x) for x in out.stdout.split()[1:5])
    i = np.arange(40, dtype=np.uint64)
    b = (0x01020304 * (i + 1)) & 0xffffffff
    c = (0xfffefdfc - 77 * i) & 0xffffffff
    ev = E.make_alu_events([E.ADD if k % 2 == 0 else E.SUB for k in range(40)], [int(x) for x in b], [int(x) for x in c], pc0=0x1000)
    chip = _addsub_chip()
    chip.trace = oracle.tracegen_alu(E.CHIP_ADD_SUB, ev)
    fri = abi.FriConfig(1, 84, 16)
    opk = oracle.Pk([], [], 0, np.zeros(14, dtype=np.uint32), 1)
    ch = oracle.new_challenger()
    opk.observe_into(ch)
    pv = np.zeros(synth.PROOF_MAX_NUM_PVS, dtype=np.uint32)
    proof, _ = oracle.prove_shard(opk, [chip], [chip.trace], pv, fri, synth.NUM_PV_ELTS, ch)
    h = 2166136261
    for w in proof.tolist():
        h = ((h ^ w) * 16777619) & 0xffffffff
    assert (n_words, fnv, commit0) == (len(proof), h, int(proof[0]))
    assert sample == oracle.lib().orc_challenger_sample(C.byref(ch))


@pytest.mark.gpu
def test_gpu_plain_c_consumer_survives_failures(tmp_path):
    """SURVEY 8(b) "Errors": a freed handle reused, a chip list that disagrees with the commit, an allocation a capped pool refuses — each a
    non-zero status with its message, none an unwind or a crash — and the same context then produces the very proof `consumer prove`
    prints (which test_gpu_plain_c_consumer_proves_a_shard holds against the oracle)."""
    import subprocess
    exe = _build_c_consumer(tmp_path, with_desc=True)
    good = subprocess.run([exe, "prove"], capture_output=True, text=True)
    out = subprocess.run([exe, "fail"], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    lines = out.stdout.strip().splitlines()
    refused = [l for l in lines if l.startswith("refused: ")]
    assert len(refused) == 3 and "not a live handle" in refused[0] and "out of device memory" in refused[1] and "does not match" in refused[2], lines
    assert lines[-1].startswith("proof ") and lines[-1] == good.stdout.strip().splitlines()[-1]


@pytest.mark.gpu
def test_gpu_out_of_memory_in_the_middle_of_a_proof_leaves_the_context_usable(oracle):
    """A pool capped at half of what a shard proof needs: zkm_prove_shard fails somewhere inside commit / open with "out of device memory",
    the caller's transcript has not moved, the failing call has given back what it took (the pool is no larger than the cap), and with the
    cap lifted the same context makes the very proof it made before."""
    from ziren_amd import prover
    ctx = prover.Context(0)
    sh = synth.syn_shard(14)
    fri = abi.FriConfig(1, 84, 16)
    hp = prover.HipProver(sh.chips, fri, synth.NUM_PV_ELTS, ctx=ctx)
    pk = hp.setup([], [], sh.pc_start, sh.initial_global_cumulative_sum)
    ch0 = prover.new_challenger()
    pk.observe_into(ch0)
    traces = hp.upload_traces([c.trace for c in sh.chips])
    ctx.trim()
    held0 = ctx.memory_held()
    good = hp.prove_shard(pk, sh.public_values, traces, ch0.copy()).copy()
    need = ctx.memory_held() - held0
    assert need > (8 << 20)
    ctx.trim()
    for frac in (0.5, 0.15, 0.3):
        ctx.set_memory_limit(held0 + int(frac * need))
        ch = ch0.copy()
        with pytest.raises(lib.ZkmError, match="out of device memory"):
            hp.prove_shard(pk, sh.public_values, traces, ch)
        assert bytes(ch) == bytes(ch0)                                  # the transcript only advances with a delivered proof
        assert ctx.memory_held() <= held0 + int(frac * need)
    ctx.set_memory_limit(0)
    again = hp.prove_shard(pk, sh.public_values, traces, ch0.copy()).copy()
    assert np.array_equal(again, good)
    for t in traces:
        t.free()
    pk.free()
    ctx.close()


def test_api_never_unwinds_whatever_is_thrown():
    """API_END catches everything (csrc/zkm_hip.hip): the macro has a catch (...) arm behind the std::exception one."""
    src = open(os.path.join(ROOT, "ziren_amd", "csrc", "zkm_hip.hip")).read()
    macro = src[src.index("#define API_END"):src.index("return 0;", src.index("#define API_END"))]
    assert "catch (const std::exception& e)" in macro and "catch (...)" in macro


def test_host_field_ext_poseidon2_match_oracle(oracle):
    L = lib.load()
    rng = np.random.default_rng(1)
    a = rng.integers(1, F.P, 64, dtype=np.uint64)
    b = rng.integers(0, F.P, 64, dtype=np.uint64)
    am, bm = F.to_monty(a), F.to_monty(b)
    for x, y, xm, ym in zip(a, b, am, bm):
        assert F.from_monty(L.zkm_host_field_mul(int(xm), int(ym))) == int(x) * int(y) % F.P
        assert F.from_monty(L.zkm_host_field_inv(int(xm))) == F.inv(int(x))
    for k in range(25):
        assert L.zkm_host_two_adic_generator(k) == F.to_monty(F.two_adic_generator(k))
    A = rng.integers(0, F.P, (40, 4), dtype=np.uint64).astype(np.uint32)
    B = rng.integers(0, F.P, (40, 4), dtype=np.uint64).astype(np.uint32)
    om, oi = np.zeros_like(A), np.zeros_like(A)
    oracle.lib().orc_ext_ops(abi.as_u32p(A), abi.as_u32p(B), C.c_size_t(40), abi.as_u32p(om), abi.as_u32p(oi))
    for i in range(40):
        o = (C.c_uint32 * 4)()
        L.zkm_host_ext_mul((C.c_uint32 * 4)(*map(int, A[i])), (C.c_uint32 * 4)(*map(int, B[i])), o)
        assert list(o) == list(map(int, om[i]))
        L.zkm_host_ext_inv((C.c_uint32 * 4)(*map(int, A[i])), o)
        assert list(o) == list(map(int, oi[i]))
    st = rng.integers(0, F.P, (8, 16), dtype=np.uint64).astype(np.uint32)
    exp = oracle.poseidon2_permute_batch(st)
    for i in range(8):
        s = (C.c_uint32 * 16)(*map(int, st[i]))
        L.zkm_host_poseidon2_permute(s)
        assert list(s) == list(map(int, exp[i]))
    # states that push the signed partial rounds (unreduced int32 lanes, 64-bit lane sum) towards their bounds
    P = F.P
    ext = [np.full(16, P - 1), np.tile([0, P - 1], 8), np.full(16, (P - 1) // 2)] + \
          [np.where(np.arange(16) == k, 0, P - 1) for k in range(16)] + [np.where(np.arange(16) == k, P - 1, 0) for k in range(16)]
    ext = F.to_monty(np.array(ext, dtype=np.uint64))
    exp = oracle.poseidon2_permute_batch(ext)
    for i in range(len(ext)):
        s = (C.c_uint32 * 16)(*map(int, ext[i]))
        L.zkm_host_poseidon2_permute(s)
        assert list(s) == list(map(int, exp[i]))


def test_fp64_poseidon2_formulation_is_exact(oracle):
    """The hashing kernels run Poseidon2 on the FP64 pipe (csrc/poseidon2_f64.cuh): exact integers in doubles, two-product +
    Barrett quotient. This is the host build of the same code (same IEEE operations: fma, round-to-nearest-even), against
    the oracle's canonical `% p` permutation: random states, states made of values next to the rounding boundaries of the
    products, and long chains (the output of one permutation feeding the next)."""
    L = lib.load()
    P = F.P
    rng = np.random.default_rng(5)
    edge = [0, 1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, (P - 3) // 2] + [(1 << k) + d for k in range(10, 31) for d in (-1, 0, 1)]
    edge = np.array(edge, dtype=np.uint64) % P
    canon = np.concatenate([rng.integers(0, P, (3000, 16), dtype=np.uint64), edge[rng.integers(0, len(edge), (3000, 16))],
                            np.full((1, 16), P - 1, dtype=np.uint64), np.zeros((1, 16), dtype=np.uint64)])
    st = F.to_monty(canon)
    exp = oracle.poseidon2_permute_batch(st)
    got = st.copy()
    for i in range(len(got)):
        L.zkm_host_poseidon2_permute_f64(got[i].ctypes.data_as(C.POINTER(C.c_uint32)))
    assert np.array_equal(got, exp)
    chain_g, chain_o = st[:64].copy(), st[:64].copy()
    for _ in range(20):
        for i in range(64):
            L.zkm_host_poseidon2_permute_f64(chain_g[i].ctypes.data_as(C.POINTER(C.c_uint32)))
        chain_o = oracle.poseidon2_permute_batch(chain_o)
    assert np.array_equal(chain_g, chain_o)


def test_fp64_poseidon2_unreduced_flows_are_exact_and_within_their_bounds(oracle):
    """What the hashing kernels actually do with the FP64 permutation, on the host build of the same code: a sponge whose capacity is
    carried from permutation to permutation as unreduced doubles (hash_leaves), and compress(node, hash(row)) with both halves handed
    over unreduced (compress_layer) — on random and on adversarial words (values next to every rounding boundary, all p - 1, all zero),
    every digest equal to the oracle's `% p` hashing. The host build records the largest magnitude it meets at each point the exactness
    argument rests on; they must stay inside the bounds documented in csrc/poseidon2_f64.cuh."""
    L = lib.load()
    P = F.P
    rng = np.random.default_rng(11)
    edge = np.array([0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2] + [(1 << k) + d for k in range(20, 31) for d in (-1, 0, 1)], dtype=np.uint64) % P
    audit = (C.c_double * 7)()
    L.zkm_host_poseidon2_f64_audit(audit, 1)

    def words(n, kind):
        if kind == 0:
            return rng.integers(0, P, n, dtype=np.uint64)
        if kind == 1:
            return edge[rng.integers(0, len(edge), n)]
        return np.full(n, P - 1 if kind == 2 else 0, dtype=np.uint64)

    u32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    for trial in range(160):
        kind = trial % 4
        n = int(rng.integers(1, 200)) if trial % 8 else [8, 16, 64, 7, 9, 1, 333, 128][trial // 8 % 8]
        row = F.to_monty(words(n, kind))
        got = np.zeros(8, dtype=np.uint32)
        L.zkm_host_poseidon2_f64_sponge(u32p(row), C.c_size_t(n), u32p(got))
        assert np.array_equal(got, oracle.hash_slice(row)), (trial, n, kind)
        left, right = F.to_monty(words(8, (kind + 1) % 4)), F.to_monty(words(8, kind))
        for inj in (0, n):
            out = np.zeros(8, dtype=np.uint32)
            L.zkm_host_poseidon2_f64_compress_inject(u32p(left), u32p(right), u32p(row), C.c_size_t(inj), u32p(out))
            want = oracle.compress(left, right)
            if inj:
                want = oracle.compress(want, oracle.hash_slice(row))
            assert np.array_equal(out, want), (trial, n, kind, inj)
    L.zkm_host_poseidon2_f64_audit(audit, 1)
    perm_in, lane_sum, sbox_in, lane, frac_sum, inexact, sbox_fast_in = [float(x) for x in audit]
    assert 2.0 ** 30 < perm_in <= 2.0 ** 35.3          # unreduced capacity / node halves did flow in, and within the input bound
    assert lane_sum < 2.0 ** 50.4                       # reduce() takes |x| < 2^52
    assert sbox_in < 2.0 ** 40.6                        # sbox_wide() (the first sixteen S-boxes): |y| < 2^41
    assert 2.0 ** 31 < sbox_fast_in < 2.0 ** 37.3       # sbox() (nine instructions): |y| < 2^38.5
    assert lane < 2.0 ** 49.3
    assert 2.0 ** 31 < frac_sum < 2.0 ** 35.8           # the lanes kept as dyadic rationals between rounds: their sum is exact below 2^36
    assert inexact == 0                                 # no fractional-lane operation lost a bit; every four-instruction product exact


def test_challenger_matches_oracle(oracle):
    # the reference's own challenger test observes 1,2,2,2 then samples (recursion/circuit/src/challenger.rs:462-500)
    L = lib.load()
    c1, c2 = prover.new_challenger(), oracle.new_challenger()
    seq = F.to_monty(np.array([1, 2, 2, 2], dtype=np.uint64))
    L.zkm_challenger_observe(C.byref(c1), abi.as_u32p(seq), C.c_size_t(4))
    oracle.challenger_observe(c2, seq)
    assert L.zkm_challenger_sample(C.byref(c1)) == oracle.lib().orc_challenger_sample(C.byref(c2))
    rng = np.random.default_rng(2)
    for n in (1, 7, 8, 9, 23):
        v = rng.integers(0, F.P, n, dtype=np.uint64).astype(np.uint32)
        L.zkm_challenger_observe(C.byref(c1), abi.as_u32p(v), C.c_size_t(n))
        oracle.challenger_observe(c2, v)
        for _ in range(5):
            assert L.zkm_challenger_sample(C.byref(c1)) == oracle.lib().orc_challenger_sample(C.byref(c2))
        assert L.zkm_challenger_sample_bits(C.byref(c1), 11) == oracle.lib().orc_challenger_sample_bits(C.byref(c2), 11)
        assert c1.as_tuple() == c2.as_tuple()


def test_recorder_counts_and_widths():
    # count_permutation_constraints / local_permutation_trace_width (permutation.rs:18-23,355-389)
    sh = synth.syn_shard(6, with_prep=True, with_trace=False)
    for c in sh.chips:
        n_lk = len(c.sends) + len(c.receives)
        assert c.perm_ext_width == air.local_permutation_trace_width(n_lk, 2)
        perm_c = air.count_permutation_constraints(n_lk, 2, c.commit_scope_global)
        assert c.num_constraints >= perm_c
        assert int(c.program[2]) == c.num_constraints
        assert len(c.program) == 4 + 2 * int(c.program[0])
    # SYN column totals equal the reference's per-row chip costs (mips_costs.json)
    costs = {"Cpu": 119, "AddSub": 47, "MemoryInstrs": 115, "Branch": 90, "Lt": 52, "DivRem": 162, "MemoryLocal": 100,
             "Global": 115}
    for c in sh.chips:
        if c.name in costs:
            assert c.main_width + 4 * c.perm_ext_width + 8 == costs[c.name]


def test_bounded_reduction_of_linear_forms_is_exact():
    """csrc/kb31.cuh reduce96_bounded — what the generated quotient / permutation kernels finish a linear form with — on the host build,
    against Python integers: (hi 2^64 + lo) / 2^32 mod p for sums up to the documented bound 127 * 2^63, at its edges (the largest sum,
    sums whose Montgomery step comes out negative, quotients -1, 0 and 126) and on what the forms are made of (products of reduced
    words, a constant in the middle word)."""
    L = lib.load()
    P = F.P
    rinv = pow(1 << 32, -1, P)
    rng = np.random.default_rng(17)
    top = 127 << 63
    xs = [0, 1, P - 1, P, (1 << 32) - 1, 1 << 32, (P - 1) << 32, ((P - 1) << 32) + (1 << 32) - 1, (1 << 64) - 1, 1 << 64, top - 1, top - P, top - (1 << 32)]
    xs += [126 * (P - 1) ** 2, 126 * (P - 1) ** 2 + ((P - 1) << 32), 4 * (P - 1) ** 2, 2 * (P - 1) ** 2 + ((P - 1) << 32)]
    xs += [int(rng.integers(0, 1 << 32)) for _ in range(50)]                                                # a low word only: the step is negative
    xs += [(int(rng.integers(0, 127)) << 63) + int(rng.integers(0, 1 << 63)) for _ in range(2000)]          # anywhere below the bound
    xs += [sum(int(a) * int(b) for a, b in zip(rng.integers(0, P, t), rng.integers(0, P, t))) + (int(rng.integers(0, P)) << 32)
           for t in list(range(1, 126)) * 4]                                                                # forms as the generator builds them
    xs += [k * (1 << 63) + d for k in (1, 2, 63, 125, 126) for d in (-1, 0, 1)]
    for x in xs:
        assert 0 <= x < top
        got = L.zkm_host_reduce96_bounded(x >> 64, x & ((1 << 64) - 1))
        assert got == x * rinv % P, hex(x)
