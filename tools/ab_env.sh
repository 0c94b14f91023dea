#!/bin/bash
# A/B of runtime environment knobs on ONE box: step time of the default workload.  gpurun --timeout 900 -- 'bash tools/ab_env.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 2 --kernel-timing 0 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('$L rep $rep: step %.3f ms verified %s' % (l['ms_per_step'], l['verified']))"
  done
}
run default ZKM_X=0
run no_interrupt HSA_ENABLE_INTERRUPT=0
run default_again ZKM_X=0
run no_interrupt_again HSA_ENABLE_INTERRUPT=0
