#!/usr/bin/env python3
"""Copy the per-row chip costs the reference pins (crates/core/executor/src/artifacts/mips_costs.json, checked by its own
`core_air_cost_consistency` test, crates/core/machine/src/mips/mod.rs:748-756) for the chips that are recorded here into
tests/golden/mips_costs.json. Data only. Run in the build container, where /root/reference exists."""
import json
import os

REF = os.environ.get("ZKM_REFERENCE", "/root/reference")
CHIPS = ["Cpu", "Program", "AddSub", "Bitwise", "Lt", "ShiftLeft", "ShiftRight", "CloClz", "Mul", "DivRem", "Branch", "Jump", "MovCond",
         "MemoryInstrs", "MemoryLocal", "SyscallInstrs", "MiscInstrs", "Byte", "Global", "MemoryGlobalInit", "MemoryGlobalFinalize", "SyscallCore",
         "SyscallPrecompile", "Poseidon2Permute", "KeccakSponge", "ShaExtend", "ShaCompress", "EdAddAssign", "EdDecompress", "Secp256k1AddAssign", "Secp256k1DoubleAssign", "Secp256r1AddAssign",
         "Secp256r1DoubleAssign", "Bn254AddAssign", "Bn254DoubleAssign", "Bls12381AddAssign", "Bls12381DoubleAssign", "Bn254FpOpAssign", "Bn254Fp2AddSubAssign",
         "Bn254Fp2MulAssign", "Bls12381FpOpAssign", "Bls12831Fp2AddSubAssign", "Bls12831Fp2MulAssign", "Secp256k1Decompress", "Secp256r1Decompress",
         "Bls12381Decompress", "Uint256MulMod", "U256XU2048Mul", "BooleanCircuitGarble", "SysLinux"]
src = json.load(open(os.path.join(REF, "crates/core/executor/src/artifacts/mips_costs.json")))
out = {"generated_by": "tests/golden/gen_mips_costs.py", "costs": {c: src[c] for c in CHIPS}}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mips_costs.json"), "w"), indent=0)
print(out["costs"])
