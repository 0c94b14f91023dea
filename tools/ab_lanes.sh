#!/bin/bash
# A/B on ONE box of the default line (claim queue from events, two lanes) across library builds and generator knobs, alternating:
#   gpurun --timeout 2400 -- 'bash tools/ab_lanes.sh'
# r04 = round 4's library (ziren_amd/libzkm_hip_r04.so, built from `git archive b96931a`) with round 4's kernel mapping (ZKM_Q_TILE=0 ZKM_Q_PAIR=0).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L: two lanes from events %.3f ms/shard   resident one lane %.3f ms   quotient %.3f  eval_columns %.3f  cpu s/shard %.3f  verified %s' % (l['ms_per_shard'], l['resident_one_lane']['ms_per_step'], k['quotient']['ms'], k['eval_columns']['ms'], l['host_cpu_s_per_shard']['rank0'], l['verified']))"
}
for rep in 1 2 3; do
  [ -f ziren_amd/libzkm_hip_r04.so ] && run r04_all ZKM_HIP_LIB=$R/ziren_amd/libzkm_hip_r04.so ZKM_Q_TILE=0 ZKM_Q_PAIR=0
  run tree_oldkernels ZKM_Q_TILE=0 ZKM_Q_PAIR=0
  run tree_tile ZKM_Q_TILE=1 ZKM_Q_PAIR=0
  run tree_tile_pair ZKM_Q_TILE=1 ZKM_Q_PAIR=1
done
