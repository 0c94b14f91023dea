#!/bin/bash
# A/B of generated-quotient-kernel variants on ONE box (ziren_amd/codegen.py's experiment knobs): for each variant the resident leg's
# quotient time and step time (bench.py --resident).   gpurun --timeout 1500 -- 'bash tools/ab_quotient.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {  # label, env...
  local L=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --resident --no-extra --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L rep $rep: step %.3f ms  quotient %.3f ms  perm_rows %.3f  eval_columns %.3f  verified %s' % (l['ms_per_step'], k['quotient']['ms'], k['perm_rows']['ms'], k['eval_columns']['ms'], l['verified']))"
  done
}
run tile1 ZKM_Q_TILE=1
run tile0 ZKM_Q_TILE=0
run tile1_again ZKM_Q_TILE=1
# round 4, measured and not kept (EXPERIMENTS.md): ZKM_Q_PREFETCH=2 / 6, ZKM_Q_AHEAD=2, ZKM_Q_WAVES=5 / 6, ZKM_Q_SINGLE=700 ZKM_Q_PART=480
