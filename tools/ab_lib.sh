#!/bin/bash
# A/B of other builds of the library (ZKM_HIP_LIB) against the tree's own on ONE box, alternating: step time and the hashing kernels.
#   gpurun --timeout 900 -- 'bash tools/ab_lib.sh ziren_amd/libzkm_hip_x.so [ziren_amd/libzkm_hip_y.so ...]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L: step %.3f ms  compress_layer %.3f  hash_leaves %.3f  verified %s' % (l['ms_per_step'], k['compress_layer']['ms'], k['hash_leaves']['ms'], l['verified']))"
}
for rep in 1 2; do
  run tree ZKM_X=0
  for lib in "$@"; do run $(basename $lib) ZKM_HIP_LIB=$R/$lib; done
done
