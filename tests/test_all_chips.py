"""Every chip of the core machine in one run: a generated program that, besides ordinary instructions of every kind, calls each of the twenty-seven
precompiles at least once. The run is cut into CPU shards, one precompile shard per syscall kind and a memory shard; together their tables are the
fifty `MipsAir` variants of crates/core/machine/src/mips/mod.rs:73-178 — the names the reference's cost table lists
(tests/golden/mips_costs.json). Every constraint holds on the oracle's rows, every shard's lookups cancel, the global digests sum to zero; on the GPU
all shards are proven from device-generated traces and the proofs pass the restated StarkMachine::verify."""
import json
import os

import pytest

from ziren_amd import abi, events as E, miniexec as M, synth

import machine_lib as ML
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine


def everything_machine():
    return M.run_machine(
        6000, seed=21, shard_cycles=4096, poseidon2_calls=2, keccak_calls=1, sha_calls=1, ed_calls=2,
        curve_calls={"Secp256k1": 1, "Secp256r1": 1, "Bn254": 1, "Bls12381": 1}, fp_calls={"Bn254": 6, "Bls12381": 6},
        decompress_calls={"Secp256k1": 2, "Secp256r1": 2, "Bls12381": 2}, uint256_calls=2, u2048_calls=1, garble_calls=(3, -2),
        linux_calls=((E.SYS_BRK, 0x5000, 0), (E.SYS_MMAP, 0, 0x1234), (E.SYS_FCNTL, 1, 3), (E.SYS_WRITE_LINUX, 1, 0x100, 5), (E.SYS_OPEN, 0, 0)))


def all_chip_names():
    return set(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"])


def test_one_run_uses_every_chip_of_the_machine(oracle):
    m = everything_machine()
    assert len(all_chip_names()) == 50
    shards = check_machine_airs(oracle, m)
    assert {c.name for cs in shards for c in cs} == all_chip_names()
    assert sum(1 for s in m.shards if s.kind == "precompile") == 27
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


@pytest.mark.gpu
def test_gpu_proves_a_run_that_uses_every_chip(hip_ctx, oracle):
    m = everything_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert {c.name for cs in oshards for c in cs} == all_chip_names()
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
