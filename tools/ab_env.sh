#!/bin/bash
# A/B of runtime knobs on ONE box: step time and the commit phases of the default workload.  gpurun --timeout 900 -- 'bash tools/ab_env.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); p=l['phases_ms']
print('$L rep $rep: step %.3f ms  commit main %.3f  commit perm %.3f  commit quotient %.3f  FRI commit phase %.3f  verified %s' % (l['ms_per_step'], p['commit main'], p['commit permutation'], p['commit quotient'], p['open: FRI commit phase'], l['verified']))"
  done
}
run poll_off ZKM_ROOT_POLL=0
run poll_on ZKM_ROOT_POLL=1
run poll_off_again ZKM_ROOT_POLL=0
run poll_on_again ZKM_ROOT_POLL=1
# measured before with the same script: ZKM_LDE_OVERLAP=0 / 1 (the side stream at all: 55.0 -> 54.3 ms); the side stream at the greatest priority (no difference)
# measured, no difference: HSA_ENABLE_INTERRUPT=0
