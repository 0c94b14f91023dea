#!/usr/bin/env python3
"""Extract the compress machine's allowed shapes as data.

Source: `RecursionShapeConfig::default()` (crates/recursion/core/src/shape.rs:116-171): three maps chip name -> log2 height that every
compress-machine shard is padded to (the smallest that holds the program's event counts), so that the reduce programs only ever have to
know three verifying-key shapes. The names are the chips' `name()`s (`Poseidon2WideDeg3`: chips/poseidon2_wide/trace.rs:53-55);
`PublicValues` is always 2^PUB_VALUES_LOG_HEIGHT (chips/public_values.rs:31). Output: ziren_amd/data/recursion_shapes.json — numbers and
names only. Run in the build container (needs /root/reference); the output is committed.
"""
import json
import os
import re

REF = "/root/reference/crates/recursion/core/src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "..", "ziren_amd", "data", "recursion_shapes.json")

# the variable a shape entry uses -> the name() of the chip it was taken from (shape.rs:120-131)
VARS = {"mem_const": "MemoryConst", "mem_var": "MemoryVar", "base_alu": "BaseAlu", "ext_alu": "ExtAlu", "poseidon2_wide": "Poseidon2WideDeg3",
        "batch_fri": "BatchFRI", "select": "Select", "exp_reverse_bits_len": "ExpReverseBitsLen", "public_values": "PublicValues"}


def main():
    src = open(os.path.join(REF, "shape.rs")).read()
    pv = int(re.search(r"PUB_VALUES_LOG_HEIGHT: usize = (\d+)", open(os.path.join(REF, "chips/public_values.rs")).read()).group(1))
    body = src[src.index("let allowed_shapes = ["):src.index(".map(HashMap::from)")]
    shapes = []
    for block in re.findall(r"\[\s*((?:\(\w+\.clone\(\), \w+\),\s*)+)\]", body):
        shape = {}
        for var, h in re.findall(r"\((\w+)\.clone\(\), (\w+)\)", block):
            shape[VARS[var]] = pv if h == "PUB_VALUES_LOG_HEIGHT" else int(h)
        assert set(shape) == set(VARS.values()), shape
        shapes.append(shape)
    assert len(shapes) == 3
    out = {"source": "crates/recursion/core/src/shape.rs:116-171 (RecursionShapeConfig::default), chips/public_values.rs:31",
           "reduce_batch_size": int(re.search(r"REDUCE_BATCH_SIZE: usize = (\d+)", open("/root/reference/crates/prover/src/lib.rs").read()).group(1)),
           "shapes": shapes}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
