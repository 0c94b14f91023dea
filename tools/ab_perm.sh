#!/bin/bash
# A/B of the generated permutation-trace kernels against the generic stark::perm_rows on ONE box.  gpurun --timeout 1500 -- 'bash tools/ab_perm.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L rep $rep: step %.3f ms  perm_rows %.3f ms (%d launches)  quotient %.3f  verified %s' % (l['ms_per_step'], k['perm_rows']['ms'], k['perm_rows']['launches'], k['quotient']['ms'], l['verified']))"
  done
}
run warm ZKM_X=0 > /dev/null
run generic ZKM_NO_PERM_KERNELS=1
run generated ZKM_X=0
run generic_again ZKM_NO_PERM_KERNELS=1
run generated_again ZKM_X=0
