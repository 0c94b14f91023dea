#!/usr/bin/env python3
"""Check / refresh vector 1 of tests/golden/poseidon2_kat.json against the known-answer test the reference itself holds:
crates/recursion/gnark-ffi/go/main.go:326-365 (TestPoseidonKoalaBear2 - the zero state and its sixteen expected outputs).
Data only (numbers). Run in the build container, where /root/reference exists."""
import json
import os
import re

REF = os.environ.get("ZKM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(REF, "crates/recursion/gnark-ffi/go/main.go")).read()
body = src[src.index("func TestPoseidonKoalaBear2"):]
body = body[:body.index("circuit :=")]
nums = [int(x) for x in re.findall(r'koalabear\.NewF\("(\d+)"\)', body)]
assert len(nums) == 32, len(nums)
path = os.path.join(HERE, "poseidon2_kat.json")
d = json.load(open(path))
assert d["vectors"][0]["input"] == nums[:16] and d["vectors"][0]["output"] == nums[16:], "vector 1 differs from the reference's Go test"
print("vector 1 == crates/recursion/gnark-ffi/go/main.go TestPoseidonKoalaBear2:", nums[16:20], "...")
