#!/bin/bash
# The opening kernels through column tables (a constant column read as one word) against the library before them (tools/_ab/libzkm_hip_prev.so),
# alternating on one box: the default line, the resident leg and the two kernels.   gpurun -- 'bash tools/ab_col_tables.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']; p=l['phases_ms']
print('%-8s default line %.3f ms/shard | resident %.3f | reduce_openings %.3f ms (%s GB/s)  eval_columns %.3f ms (%s GB/s) | open: evaluations %.3f  reduced openings %.3f | verified %s' % ('$L', l['ms_per_step'], l['resident_one_lane']['ms_per_step'], k['reduce_openings']['ms'], k['reduce_openings']['GBps'], k['eval_columns']['ms'], k['eval_columns']['GBps'], p['open: evaluations'], p['open: reduced openings'], l['verified']))"
}
for rep in 1 2 3; do
  run before ZKM_HIP_LIB=$R/tools/_ab/libzkm_hip_prev.so
  run tables ZKM_X=0
done
