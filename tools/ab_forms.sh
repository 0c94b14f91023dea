#!/bin/bash
# A/B of the generated kernels' linear forms on ONE box (codegen.emit_form, round 5): rounds 3-4's shape (ZKM_Q_FORMS=0: fold_zero /
# acc96_reduce / modular additions) against bounded accumulators (the default), with and without a four-wave register limit; for each
# variant the resident leg's quotient / perm_rows time and step time. Parity first (generated == interpreter == oracle).
#   gpurun --timeout 1800 -- 'bash tools/ab_forms.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_codegen.py tests/test_gpu_parity.py -m gpu -x -q -k "specialized or specialised or generated or quotient or permutation" 2>&1 | tail -3
run() {  # label, env...
  local L=$1; shift
  env "$@" python bench.py --resident --no-extra --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('$L: step %.3f ms  quotient %.3f ms  perm_rows %.3f  eval_columns %.3f  verified %s' % (l['ms_per_step'], k['quotient']['ms'], k['perm_rows']['ms'], k['eval_columns']['ms'], l['verified']))"
}
for rep in 1 2; do
  run forms0 ZKM_Q_FORMS=0
  run forms1 ZKM_Q_FORMS=1
  run forms1_waves4 ZKM_Q_FORMS=1 ZKM_Q_WAVES=4
done
