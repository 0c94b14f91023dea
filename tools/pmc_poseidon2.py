#!/usr/bin/env python3
"""Dynamic VALU instructions per Poseidon2 permutation, as the hardware counted them: SQ_INSTS_VALU over the kernels of tools/ubench_p2
(bench_int: integer-pipe formulation, bench_f64_pure: FP64-pipe formulation, bench_f64: FP64 + the 16-in / 8-out conversions of a
compression). A wave executes 64 permutations per REPS iteration, so instructions per permutation = SQ_INSTS_VALU / (waves x REPS).

  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d out/p2 -o p2 -- tools/ubench_p2 gpu
  python tools/pmc_poseidon2.py out/p2/.../p2_results.db profiles/r02_poseidon2_isa.json"""
import json
import sqlite3
import sys

REPS = 64   # tools/ubench_p2.hip


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    agg = {}
    for k, c, v, n in rows:
        agg.setdefault(k.split("(")[0].replace("void ", ""), {})[c] = (v, n)
    out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES over tools/ubench_p2 gpu (tools/pmc_poseidon2.py)", "reps_per_wave": REPS}
    for name, key in (("bench_int", "int"), ("bench_f64_pure", "fp64"), ("bench_f64", "fp64_with_io")):
        d = agg.get(name)
        if not d:
            continue
        insts, waves = d["SQ_INSTS_VALU"][0], d["SQ_WAVES"][0]
        out[key] = {"valu_instr_per_permutation": round(insts / (waves * REPS)), "sq_insts_valu": insts, "sq_waves": waves, "launches": d["SQ_WAVES"][1]}
    json.dump(out, open(out_path, "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
