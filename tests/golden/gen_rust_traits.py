#!/usr/bin/env python3
"""Extract from the reference the trait definitions the Rust shim must satisfy, as data: for each trait its associated types and its
methods (name, number of non-self arguments, whether the trait supplies a default body). Written to tests/golden/rust_traits.json;
tests/test_rust_shim_conformance.py holds integration/zkm-hip and integration/sdk-hip against it (there is no cargo here to do it).

    MachineProver, MachineProvingKey   crates/stark/src/prover.rs:30-199
    ZKMProverComponents                crates/prover/src/components.rs:6-26
    Prover                             crates/sdk/src/provers/mod.rs:66-

Run in the build container, where /root/reference exists. Names and counts only — no source text is kept."""
import json
import os
import re

REF = os.environ.get("ZKM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def matching(src, i, open_c, close_c):
    """index of the bracket that closes the one at src[i]"""
    depth = 0
    for j in range(i, len(src)):
        if src[j] == open_c:
            depth += 1
        elif src[j] == close_c:
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced")


def split_args(sig):
    """top-level comma split of the text between a fn's parentheses"""
    out, depth, cur = [], 0, ""
    for k, c in enumerate(sig):
        if c in "<([{":
            depth += 1
        elif c in ")]}" or (c == ">" and sig[k - 1] != "-"):
            depth -= 1
        if c == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_fns(body):
    fns = []
    depth_at = []
    d = 0
    for c in body:           # brace depth before each character: only items at depth 0 belong to the trait / impl itself
        depth_at.append(d)
        if c == "{":
            d += 1
        elif c == "}":
            d -= 1
    for m in re.finditer(r"\bfn\s+(\w+)", body):
        if depth_at[m.start()] != 0:
            continue
        k = body.index("(", m.end())
        j = matching(body, k, "(", ")")
        args = [a for a in split_args(body[k + 1:j]) if not re.match(r"^&?\s*('\w+\s+)?(mut\s+)?self\b", a)]
        t = j + 1
        while body[t] not in ";{":     # return type and where-clause hold no braces or semicolons in these traits
            t += 1
        fns.append({"name": m.group(1), "args": len(args), "provided": body[t] == "{"})
    return fns


def parse_trait(path, name):
    src = open(os.path.join(REF, path)).read()
    src = re.sub(r"//[^\n]*", "", src)   # comments out (line structure kept)
    m = re.search(r"pub trait %s\b" % name, src)
    assert m, name
    i = src.index("{", m.end())
    j = matching(src, i, "{", "}")
    body = src[i + 1:j]
    types = [t for t in re.findall(r"^    type\s+(\w+)", body, flags=re.M)]
    return {"file": path, "lines": [src[:m.start()].count("\n") + 1, src[:j].count("\n") + 1], "types": types, "fns": parse_fns(body)}


if __name__ == "__main__":
    out = {
        "MachineProver": parse_trait("crates/stark/src/prover.rs", "MachineProver"),
        "MachineProvingKey": parse_trait("crates/stark/src/prover.rs", "MachineProvingKey"),
        "ZKMProverComponents": parse_trait("crates/prover/src/components.rs", "ZKMProverComponents"),
        "Prover": parse_trait("crates/sdk/src/provers/mod.rs", "Prover"),
    }
    path = os.path.join(HERE, "rust_traits.json")
    json.dump(out, open(path, "w"), indent=1)
    for k, v in out.items():
        print(k, v["file"], v["lines"], "types", v["types"], "required", [(f["name"], f["args"]) for f in v["fns"] if not f["provided"]])
