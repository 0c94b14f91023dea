#!/bin/bash
# The Keccak precompile shard's quotient time (a program cut into parts: tools/bench_keccak_shard.py) under the generator's round-5 knobs.
#   gpurun --timeout 2400 -- 'bash tools/ab_keccak_quotient.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  local L=$1; shift
  env "$@" python tools/bench_keccak_shard.py --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$L: prove %.3f ms  quotient %.3f ms (%d launches)  perm_rows %.3f' % (d['prove_ms'], k['quotient'][0], k['quotient'][1], k['perm_rows'][0]))"
}
run rounds_3_4 ZKM_Q_FORMS=0 ZKM_Q_SUMS=0 ZKM_Q_UNITABLE=0
run forms ZKM_Q_FORMS=1 ZKM_Q_SUMS=0 ZKM_Q_UNITABLE=0
run forms_sums ZKM_Q_FORMS=1 ZKM_Q_SUMS=1 ZKM_Q_UNITABLE=0
run forms_sums_table ZKM_Q_FORMS=1 ZKM_Q_SUMS=1 ZKM_Q_UNITABLE=1
run forms_table ZKM_Q_FORMS=1 ZKM_Q_SUMS=0 ZKM_Q_UNITABLE=1
