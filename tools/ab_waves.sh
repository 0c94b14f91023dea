#!/bin/bash
# A/B on ONE box (round 5): the generated quotient kernels capped at four waves per SIMD (ZKM_Q_WAVES=4: 128 registers, a few words
# spilled) against the compiler's choice (134 registers, three waves): the resident leg's quotient time, then the default line.
#   gpurun --timeout 1800 -- 'bash tools/ab_waves.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_codegen.py -m gpu -x -q 2>&1 | tail -2
run() {  # label, env...
  local L=$1; shift
  env "$@" python bench.py --resident --no-extra --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L: step %.3f ms  quotient %.3f ms  perm_rows %.3f  verified %s' % (l['ms_per_step'], k['quotient']['ms'], k['perm_rows']['ms'], l['verified']))"
}
for rep in 1 2; do
  run waves_free ZKM_Q_WAVES=0
  run waves4 ZKM_Q_WAVES=4
done
bash tools/ab_default_line.sh ZKM_Q_WAVES 0 4 2 40
