#!/usr/bin/env python3
"""Known answers for the septic extension's Frobenius map: the reference keeps z^(p*i) and z^(p^2*i), i = 1..6, as numeric tables
(crates/stark/src/septic_extension.rs, `z_pow_p` and `z_pow_p2`). This script copies the numbers (data only) into
tests/golden/septic_frobenius.json; the oracle and the device compute the same elements by exponentiation, and
tests/test_cpu_shard.py::test_septic_arithmetic_matches_the_reference_tables compares. Run in the build container, where /root/reference exists."""
import json
import os
import re

REF = os.environ.get("ZKM_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "crates/stark/src/septic_extension.rs")).read()


def table(fn):
    body = src[src.index(f"fn {fn}(index: u32)"):]
    body = body[:body.index("\n    }\n")]
    out = {}
    for m in re.finditer(r"index == (\d+) \{(.*?)\}", body, re.S):
        nums = [int(x) for x in re.findall(r"from_canonical_u32\((\d+)\)", m.group(2))]
        if len(nums) == 7:
            out[int(m.group(1))] = nums
    return [out[i] for i in range(1, 7)]


out = {"generated_by": "tests/golden/gen_septic_frobenius.py", "modulus": "z^7 + 2z - 8 over KoalaBear",
       "z_pow_p": table("z_pow_p"), "z_pow_p2": table("z_pow_p2")}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "septic_frobenius.json"), "w"), indent=0)
print(out["z_pow_p"][0], out["z_pow_p2"][5])
