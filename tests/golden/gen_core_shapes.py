#!/usr/bin/env python3
"""Extract the reference's table of maximal core shapes as data.

Source: crates/core/machine/src/shape/maximal_shapes.json (the file CoreShapeConfig::default embeds, crates/core/machine/src/shape/mod.rs:31,452-461)
and the per-chip costs crates/core/executor/src/artifacts/mips_costs.json (mod.rs:517-520). Output: ziren_amd/data/core_shapes.json —
{"airs": [chip names, fixed order], "shapes": {"<log2 shard size>": [[log2 height per chip of `airs`, -1 where the shape omits the chip], ...]},
"costs": {chip: cells per row}} — numbers only. Run in the build container (needs /root/reference); the output is committed.
"""
import json
import os

REF = "/root/reference/crates"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "..", "ziren_amd", "data", "core_shapes.json")

# every chip the maximal shapes name (the 18 `MipsAir::core_heights` entries, crates/core/machine/src/mips/mod.rs:456-487), in a fixed order
AIRS = ["Cpu", "AddSub", "Bitwise", "Mul", "DivRem", "Lt", "ShiftLeft", "ShiftRight", "CloClz", "Branch", "Jump", "MovCond", "MemoryInstrs",
        "MiscInstrs", "SyscallInstrs", "SyscallCore", "MemoryLocal", "Global"]


def main():
    src = json.load(open(os.path.join(REF, "core/machine/src/shape/maximal_shapes.json")))
    costs = json.load(open(os.path.join(REF, "core/executor/src/artifacts/mips_costs.json")))
    shapes = {}
    for key, lst in sorted(src.items(), key=lambda kv: int(kv[0])):
        rows = []
        for s in lst:
            inner = s["inner"]
            assert set(inner) <= set(AIRS), set(inner) - set(AIRS)
            rows.append([int(inner.get(a, -1)) for a in AIRS])
        shapes[key] = rows
    out = {"source": "crates/core/machine/src/shape/maximal_shapes.json + crates/core/executor/src/artifacts/mips_costs.json",
           "airs": AIRS, "shapes": shapes, "costs": {k: int(v) for k, v in sorted(costs.items())}}
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print({k: len(v) for k, v in shapes.items()}, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
