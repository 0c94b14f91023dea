#!/bin/bash
# A/B on ONE box (round 5): base-field sums of products reduced once (ZKM_Q_SUMS, codegen.emit_sum) and the number of fraction columns
# whose inversions share one base-field inversion in the generated permutation kernels (ZKM_P_GROUP); parity first.
#   gpurun --timeout 1800 -- 'bash tools/ab_sums.sh'
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
  run sums0_group2 ZKM_Q_SUMS=0
  run sums1_group2 ZKM_Q_SUMS=1
  run sums1_group4 ZKM_Q_SUMS=1 ZKM_P_GROUP=4
  run sums1_group8 ZKM_Q_SUMS=1 ZKM_P_GROUP=8
done
