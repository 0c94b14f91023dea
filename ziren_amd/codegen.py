"""Per-chip quotient kernels: constraint bytecode -> straight-line HIP -> gfx950 code object.

The bytecode interpreter (`stark::quotient_kernel`) keeps its register files in LDS and pays a
dispatch, two LDS reads and an exposed global-load latency per instruction. A chip's AIR is fixed
for the life of a machine (Ziren has 50 `MipsAir` variants, crates/core/machine/src/mips/mod.rs:73-190),
so the same bytecode can be turned once into a specialised kernel whose values live in VGPRs and whose
loads the compiler hoists and batches. `specialize()` emits that kernel, compiles it with hipcc
(`--genco`, gfx950) into an in-tree cache, and `HipProver` registers the code object with the library
(`zkm_ctx_register_quotient_kernel`); `zkm_open` then picks it by the hash of the chip's program words
and falls back to the interpreter for unregistered programs. Both paths compute the same field values.

In the Rust integration this runs in the shim's build script (or `HipProver::new`) — INTEGRATION.md.
"""
import hashlib
import os
import subprocess
import tempfile
from typing import Optional

import numpy as np

from . import air

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
CACHE = os.path.join(HERE, "_jit")
KERNEL_NAME = "zkm_quotient_specialized"
BLOCK = 256


TEMPLATE_VERSION = b"5"  # bump when emit_source changes


def _template_key() -> bytes:
    """Cached code objects are keyed on the program *and* on everything the generated source pulls in, so an edit to
    the shared prologue (quotient_args.cuh) or the field arithmetic (kb31.cuh) can never leave a stale kernel behind."""
    h = hashlib.sha256(TEMPLATE_VERSION)
    for name in ("quotient_args.cuh", "kb31.cuh"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.digest()


def program_hash(program: np.ndarray) -> str:
    return hashlib.sha256(_template_key() + np.ascontiguousarray(program, dtype=np.uint32).tobytes()).hexdigest()[:24]


def emit_source(program: np.ndarray) -> str:
    """Straight-line HIP for one chip. Every register write becomes a fresh SSA value."""
    prog = np.asarray(program, dtype=np.uint32)
    n_instr = int(prog[0])
    cur_b, cur_e = {}, {}   # register -> current variable name
    lines = []
    nv = [0]

    def fresh(prefix):
        nv[0] += 1
        return f"{prefix}{nv[0]}"

    cidx = 0
    for k in range(n_instr):
        w0, imm = int(prog[4 + 2 * k]), int(prog[5 + 2 * k])
        op, dst, ra, rb = w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, w0 >> 24
        row = "q.pn" if ra else "q.p"
        if op == air.LD_MAIN:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.main_lde[(size_t){imm} * a.main_stride + {row}];")
        elif op == air.LD_PREP:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.prep_lde[(size_t){imm} * a.prep_stride + {row}];")
        elif op == air.LD_PERM:
            v = fresh("e"); cur_e[dst] = v
            base = f"a.perm_lde + (size_t){4 * imm} * a.perm_stride + {row}"
            lines.append(f"const kb::E4 {v} = kb::E4{{{{({base})[0], ({base})[a.perm_stride], ({base})[2 * a.perm_stride], "
                         f"({base})[3 * a.perm_stride]}}}};")
        elif op == air.LD_CONST:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = {imm}u;")
        elif op == air.LD_PV:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.public_values[{imm}];")
        elif op == air.LD_CHALLENGE:
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = {'a.perm_beta' if imm else 'a.perm_alpha'};")
        elif op == air.LD_LOCAL_SUM:
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = a.local_sum;")
        elif op == air.LD_GLOBAL_SUM:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.consts[{imm}];")
        elif op in (air.LD_IS_FIRST, air.LD_IS_LAST, air.LD_IS_TRANS):
            v = fresh("b"); cur_b[dst] = v
            sel = {air.LD_IS_FIRST: "q.is_first", air.LD_IS_LAST: "q.is_last", air.LD_IS_TRANS: "q.is_trans"}[op]
            lines.append(f"const uint32_t {v} = {sel};")
        elif op in (air.ADD_B, air.SUB_B, air.MUL_B):
            fn = {air.ADD_B: "add", air.SUB_B: "sub", air.MUL_B: "mul"}[op]
            x, y = cur_b[ra], cur_b[rb]
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = kb::{fn}({x}, {y});")
        elif op == air.NEG_B:
            x = cur_b[ra]
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = kb::neg({x});")
        elif op in (air.ADD_E, air.SUB_E, air.MUL_E):
            fn = {air.ADD_E: "eadd", air.SUB_E: "esub", air.MUL_E: "emul"}[op]
            x, y = cur_e[ra], cur_e[rb]
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = kb::{fn}({x}, {y});")
        elif op == air.NEG_E:
            x = cur_e[ra]
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = kb::eneg({x});")
        elif op in (air.ADD_EB, air.SUB_EB, air.MUL_EB):
            fn = {air.ADD_EB: "eadd_base", air.SUB_EB: "esub_base", air.MUL_EB: "escale"}[op]
            x, y = cur_e[ra], cur_b[rb]
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = kb::{fn}({x}, {y});")
        elif op == air.ASSERT_B:
            lines.append(f"acc = kb::eadd(acc, kb::escale(a.alpha_pows[{cidx}], {cur_b[ra]}));")
            cidx += 1
        elif op == air.ASSERT_E:
            lines.append(f"acc = kb::eadd(acc, kb::emul(a.alpha_pows[{cidx}], {cur_e[ra]}));")
            cidx += 1
        else:
            raise ValueError(f"bad opcode {op}")
    assert cidx == int(prog[2])
    body = "\n  ".join(lines)
    return f"""// GENERATED by ziren_amd/codegen.py from a chip's constraint bytecode ({n_instr} instructions,
// {cidx} constraints). Same arithmetic as stark::quotient_kernel (the interpreter), values in VGPRs.
#include "quotient_args.cuh"

extern "C" __global__ __launch_bounds__({BLOCK}) void {KERNEL_NAME}(stark::QuotientArgs a) {{
  stark::QuotientPoint q;
  if (!stark::quotient_point(a, stark::quotient_row(a), q)) return;
  kb::E4 acc = kb::ezero();
  {body}
  stark::quotient_store(a, q, acc);
}}
"""


MAX_SPECIALIZED_INSTRS = 40000   # straight-line code beyond this takes hipcc tens of minutes (KeccakSponge: 114 324 instructions)


def specialize(program: np.ndarray, force: bool = False, verbose: bool = False) -> Optional[bytes]:
    """Return the gfx950 code object for `program`, compiling it on first use (cached in-tree); None for a program too long to be worth a
    straight-line kernel — the library's bytecode interpreter evaluates it."""
    if int(np.asarray(program)[0]) > MAX_SPECIALIZED_INSTRS:
        return None
    os.makedirs(CACHE, exist_ok=True)
    h = program_hash(program)
    out = os.path.join(CACHE, f"q_{h}.hsaco")
    if force or not os.path.exists(out):
        src = emit_source(program)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, f"q_{h}.hip")
            with open(path, "w") as f:
                f.write(src)
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", "-I", CSRC, path, "-o", out + ".tmp"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        os.replace(out + ".tmp", out)
    _note_in_manifest(program, os.path.basename(out))
    with open(out, "rb") as f:
        return f.read()


def _note_in_manifest(program: np.ndarray, filename: str):
    """Ahead-of-time story for a host with no compiler at run time (the Rust shim): `_jit/manifest.json` maps the sha256
    of a chip's program words (little-endian bytes — a key any host language can compute) to its code object. `__graft_entry__.build()`
    specialises every recorded chip, so the manifest and the code objects ship with the build; an unknown program falls back to the
    bytecode interpreter inside the library."""
    import json
    key = hashlib.sha256(np.ascontiguousarray(program, dtype="<u4").tobytes()).hexdigest()
    path = os.path.join(CACHE, "manifest.json")
    try:
        with open(path) as f:
            m = json.load(f)
    except (OSError, ValueError):
        m = {}
    if m.get(key) != filename:
        m[key] = filename
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(m, f, indent=0, sort_keys=True)
        os.replace(tmp, path)
