#!/bin/bash
# A/B timing on ONE box (boxes differ by ~1 %): bench.py with another build of the library (ZKM_HIP_LIB) and with the tree's own, alternating.
#   gpurun -- 'bash tools/ab_bench.sh ziren_amd/libzkm_hip_r02.so [rounds]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$R/$1; N=${2:-2}
mkdir -p $R/gpurun_out/ab
for i in $(seq 1 $N); do
  ZKM_HIP_LIB=$A python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-inflight2 --fib= > $R/gpurun_out/ab/a$i.json 2>/dev/null
  python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-inflight2 --fib= > $R/gpurun_out/ab/b$i.json 2>/dev/null
done
python - <<PY
import json, glob
for tag in "ab":
    for f in sorted(glob.glob("$R/gpurun_out/ab/%s*.json" % tag)):
        d = json.load(open(f))
        print(tag, d["ms_per_step"], {k: v["ms"] for k, v in d["kernels_ms"].items() if v["ms"] > 1.0})
PY
