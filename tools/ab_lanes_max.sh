#!/bin/bash
# The largest tree layer / FRI layer handed to the lane-parallel kernels (ZKM_LANES_MAX, ZKM_FRI_LANES_MAX), alternating on one box:
# the resident one-lane core shard and a shape-0 recursion shard.   gpurun --timeout 900 -- 'bash tools/ab_lanes_max.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --resident --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('$L: core step %.3f ms  FRI commit phase %.3f' % (l['ms_per_step'], l['phases_ms']['open: FRI commit phase']))"
  env "$@" python tools/prof_recursion_shard.py --shape 0 --steps 30 2>/dev/null | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().split('phases ',1)[1]); print('$L: shape-0 recursion shard: phases sum %.3f ms  FRI commit phase %.3f' % (sum(d.values()), d['open: FRI commit phase']))"
}
for rep in 1 2; do
  run 4096/4096 ZKM_X=0
  run 8192/4096 ZKM_LANES_MAX=8192
  run 4096/16384 ZKM_FRI_LANES_MAX=16384
  run 2048/2048 ZKM_LANES_MAX=2048 ZKM_FRI_LANES_MAX=2048
done
