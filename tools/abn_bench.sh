#!/bin/bash
# N-way timing of several builds of the library on ONE box, alternating:  gpurun -- 'bash tools/abn_bench.sh rounds lib1.so lib2.so ...'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=$1; shift
for i in $(seq 1 $N); do for L in "$@"; do
  ZKM_HIP_LIB=$R/$L python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-inflight2 --fib= 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels_ms']; print('$L', d['ms_per_step'], k['compress_layer']['ms'], k['hash_leaves']['ms'], k['lde_rows']['ms'])"
done; done
