#!/bin/bash
# The default line (claim queue from events, two lanes) under two settings of one environment knob, alternating on ONE box:
#   gpurun --timeout 1800 -- 'bash tools/ab_default_line.sh ZKM_Q_UNITABLE 0 1 [reps] [steps]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VAR=$1; A=$2; B=$3; REPS=${4:-3}; STEPS=${5:-40}
for rep in $(seq $REPS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-extra --no-cpu-baseline --steps $STEPS --warmup 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$VAR=$v: %.3f ms/shard = %.3f shard-proofs/s; resident one lane %.3f ms; quotient %.3f ms' % (l['ms_per_step'], l['value'], l['resident_one_lane']['ms_per_step'], l['kernels_ms']['quotient']['ms']))"
  done
done
