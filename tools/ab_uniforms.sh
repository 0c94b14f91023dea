#!/bin/bash
# A/B on ONE box (round 5): the generated quotient kernels' wave-uniform values computed by every wavefront (ZKM_Q_UNITABLE=0) or once per
# launch by a one-wavefront kernel into a table (the default, codegen.split_uniform); parity first (generated == interpreter == oracle).
# The experiment that led there — ZKM_Q_FAKEUNI=1, the uniform values read from an unrelated table: wrong values, the kernel's duration
# only (tools/time_quotient.py does not verify) — measured 3.92 -> 2.84 ms per proof.
#   gpurun --timeout 1800 -- 'bash tools/ab_uniforms.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_codegen.py tests/test_gpu_parity.py -m gpu -x -q -k "specialized or specialised or generated or quotient or permutation" 2>&1 | tail -3
run() {  # label, env...
  local L=$1; shift
  env "$@" python bench.py --resident --no-extra --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L: step %.3f ms  quotient %.3f ms  perm_rows %.3f  eval_columns %.3f  verified %s' % (l['ms_per_step'], k['quotient']['ms'], k['perm_rows']['ms'], k['eval_columns']['ms'], l['verified']))"
}
for rep in 1 2; do
  run every_wavefront ZKM_Q_UNITABLE=0
  run table ZKM_Q_UNITABLE=1
done
for i in 1 2; do python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('default line (table): %.3f ms/shard = %.3f shard-proofs/s, resident one lane %.3f ms' % (l['ms_per_step'], l['value'], l['resident_one_lane']['ms_per_step']))"; done
