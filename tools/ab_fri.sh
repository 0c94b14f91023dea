#!/bin/bash
# A/B of builds of the library with other FRI-tree fusing parameters (-DZKM_FRI_FUSE_LEAVES / -DZKM_FRI_FUSE_LEVELS) against the tree's own on ONE box,
# alternating: step time, the FRI commit phase, its fused leaf + tree kernel.
#   gpurun --timeout 900 -- 'bash tools/ab_fri.sh ziren_amd/libzkm_hip_x.so [...]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L: step %.3f ms  FRI commit phase %.3f  hash_fri_leaves_tree %.3f (%d launches)  compress_layer %.3f (%d)  verified %s' % (l['ms_per_step'], l['phases_ms']['open: FRI commit phase'], k['hash_fri_leaves_tree']['ms'], k['hash_fri_leaves_tree']['launches'], k['compress_layer']['ms'], k['compress_layer']['launches'], l['verified']))"
}
for rep in 1 2; do
  run tree ZKM_X=0
  for lib in "$@"; do run $(basename $lib) ZKM_HIP_LIB=$R/$lib; done
done
