#!/bin/bash
# SQ counters + kernel stats of the fibonacci-guest workload:  gpurun -- 'bash tools/profile_fib.sh 21'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; K=${1:-21}
OUT=$R/gpurun_out/r03fibprof; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload fib --log-rows $K --steps 2"
db() { find "$1" -name '*_results.db' | head -1; }
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $B > $OUT/r03_fib${K}_bench_profiled.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq -o sq -- $B > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $OUT/sq2 -o sq2 -- $B > /dev/null 2> $OUT/sq2.err
cd $R
python tools/rocprof_summary.py "$(db $OUT/stats)" $OUT/r03_fib${K}_kernel_stats.csv
python tools/pmc_sq_summary.py "$(db $OUT/sq)" $OUT/r03_fib${K}_sq_counters.csv "$(db $OUT/sq2)"
find $OUT -name '*.db' -delete; rm -rf $OUT/stats $OUT/sq $OUT/sq2
