#!/bin/bash
# A/B on ONE box of build_tree's two ways to hash the rows of a commit's shorter heights: inside compress_layer (ZKM_ROWS_UP_FRONT=0, rounds
# 1-5) or by one hash_rows launch in front of the tree levels (default). Alternating; per run the default line (two lanes, from events) and
# its resident one-lane leg.   gpurun --timeout 1200 -- 'bash tools/ab_rows_up_front.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps ${STEPS:-20} --warmup 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); p=l['phases_ms']; k=l['kernels_ms']; r=l['resident_one_lane']
h=sum(v['ms'] for n,v in k.items() if n in ('compress_layer','compress_layer_rowdig','hash_leaves','hash_rows'))
print('   ', {n: v['ms'] for n, v in k.items() if n in ('compress_layer','compress_layer_rowdig','hash_leaves','hash_rows')})
print('$L: default line %.3f ms/shard | resident one lane %.3f ms | commit main %.3f perm %.3f quotient %.3f | hashing kernels (serialised) %.3f ms | verified %s' % (l['ms_per_step'], r['ms_per_step'], p['commit main'], p['commit permutation'], p['commit quotient'], h, l['verified']))"
}
for rep in $(seq 1 ${REPS:-3}); do
  run "inside compress_layer (rep $rep)" ZKM_ROWS_UP_FRONT=0
  run "hash_rows up front    (rep $rep)" ZKM_ROWS_UP_FRONT=1
done
