#!/bin/bash
# A/B of other builds of the library against the tree's own on ONE box, alternating: step time, the three commit phases, the leaf hashing.
#   gpurun --timeout 900 -- 'bash tools/ab_commit.sh ziren_amd/libzkm_hip_x.so [...]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); p=l['phases_ms']; k=l['kernels_ms']
print('$L: step %.3f ms  commit main %.3f  perm %.3f  quotient %.3f  hash_leaves %.3f  verified %s' % (l['ms_per_step'], p['commit main'], p['commit permutation'], p['commit quotient'], k['hash_leaves']['ms'], l['verified']))"
}
for rep in 1 2; do
  run tree ZKM_X=0
  for lib in "$@"; do run $(basename $lib) ZKM_HIP_LIB=$R/$lib; done
done
