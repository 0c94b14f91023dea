#!/usr/bin/env python3
"""Collect the ALU instruction vectors the reference's own chip tests prove and verify
(crates/core/machine/src/alu/{add_sub,bitwise,lt,sll,sr,clo_clz,mul,divrem}/mod.rs, `#[cfg(test)]` modules) into
tests/golden/alu_events.json. Each record is data only: (chip, opcode, a, b, c) with a = b op c as the reference
states it. Run in the build container, where /root/reference exists; the JSON is what travels.

  python tests/golden/gen_alu_events.py
"""
import json
import os
import re
import sys

REF = os.environ.get("ZKM_REFERENCE", "/root/reference")
ALU = os.path.join(REF, "crates/core/machine/src/alu")
OPC = {"ADD": 0, "SUB": 1, "MUL": 2, "MULT": 3, "MULTU": 4, "DIV": 5, "DIVU": 6, "MOD": 7, "MODU": 8, "SLL": 9, "SRL": 10, "SRA": 11, "ROR": 12, "SLT": 13, "SLTU": 14, "AND": 15, "OR": 16, "XOR": 17, "NOR": 18, "CLZ": 19, "CLO": 20}
FILES = {"AddSub": "add_sub/mod.rs", "Bitwise": "bitwise/mod.rs", "Lt": "lt/mod.rs", "ShiftLeft": "sll/mod.rs", "ShiftRight": "sr/mod.rs", "CloClz": "clo_clz/mod.rs", "Mul": "mul/mod.rs", "DivRem": "divrem/mod.rs"}


def num(tok, consts):
    tok = tok.strip()
    if tok in consts:
        return consts[tok]
    tok = tok.replace("_", "")
    return int(tok, 0)


def main():
    out = []
    for chip, rel in FILES.items():
        src = open(os.path.join(ALU, rel)).read()
        tests = src[src.index("#[cfg(test)]"):]
        consts = {m.group(1): int(m.group(2).replace("_", ""), 0)
                  for m in re.finditer(r"const (\w+): u32 = (0[bx][0-9a-fA-F_]+|\d+);", tests)}
        seen = set()
        pats = [r"(?<!Comp)AluEvent::new\(\s*\w+,\s*Opcode::(\w+),\s*([\w]+),\s*([\w]+),\s*([\w]+)\s*\)",
                r"CompAluEvent::new\(\s*\w+,\s*Opcode::(\w+),\s*([\w]+),\s*([\w]+),\s*([\w]+)\s*\)",
                r"\(Opcode::(\w+),\s*([\w]+),\s*([\w]+),\s*([\w]+)\)"]
        for pat in pats:
            for m in re.finditer(pat, tests):
                op, a, b, c = m.groups()
                if op not in OPC:
                    continue
                try:
                    rec = (OPC[op], num(a, consts), num(b, consts), num(c, consts))
                except ValueError:
                    continue  # operands that are variables of a randomised test
                if rec in seen:
                    continue
                seen.add(rec)
                out.append({"chip": chip, "opcode": rec[0], "a": rec[1], "b": rec[2], "c": rec[3], "source": rel})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "alu_events.json")
    with open(path, "w") as f:
        json.dump({"generated_by": "tests/golden/gen_alu_events.py", "events": out}, f, indent=0)
    print(f"{len(out)} events -> {path}", file=sys.stderr)
    for chip in FILES:
        print(chip, sum(1 for e in out if e["chip"] == chip), file=sys.stderr)


if __name__ == "__main__":
    main()
