#!/bin/bash
# Round 6's two kernel-path changes against the library of the round's first commit, alternating on ONE box (resident one-lane leg):
#   * the FRI commit phase's transcript on the device (no host round trip per layer)      -> phases_ms "open: FRI commit phase", step
#   * the trace generators' "no events" mark (the LDE's discovery read skipped)             -> kernels_ms lde_cols_inverse
#   gpurun --timeout 900 -- 'bash tools/ab_r06.sh ziren_amd/libzkm_hip_old.so'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python bench.py --resident --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L: step %.3f ms  FRI commit phase %.3f  lde_cols_inverse %.3f  kernels_ms_sum %.3f  verified %s' % (l['ms_per_step'], l['phases_ms']['open: FRI commit phase'], k['lde_cols_inverse']['ms'], l['kernels_ms_source']['kernels_ms_sum'], l['verified']))"
}
for rep in 1 2 3; do
  run tree ZKM_X=0
  for lib in "$@"; do run $(basename $lib) ZKM_HIP_LIB=$R/$lib; done
done
