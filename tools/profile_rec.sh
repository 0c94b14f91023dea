#!/bin/bash
# Counters of ONE compress-machine shard at an allowed shape (stand-in program, traces resident; tools/prof_recursion_shard.py):
# kernel stats, SQ counters (two passes), FETCH_SIZE / WRITE_SIZE in separate passes (the guide's gfx950 correction in tools/pmc_traffic.py).
#   gpurun --timeout 900 -- 'bash tools/profile_rec.sh [shape]'      -> gpurun_out/r06rec/r06_rec_shape<k>_{kernel_stats.csv,sq_counters.csv,hbm_traffic.json}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-0}
OUT=$R/gpurun_out/r06rec
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*_results.db' | head -1; }
B="python $R/tools/prof_recursion_shard.py --shape $K"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $B --steps 6 > $OUT/stats.out 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq -o sq -- $B --steps 2 > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $OUT/sq2 -o sq2 -- $B --steps 2 > /dev/null 2> $OUT/sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f -- $B --steps 2 > /dev/null 2> $OUT/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o w -- $B --steps 2 > /dev/null 2> $OUT/w.err
cd $R
python tools/rocprof_summary.py "$(db $OUT/stats)" $OUT/r06_rec_shape${K}_kernel_stats.csv
python tools/pmc_sq_summary.py "$(db $OUT/sq)" $OUT/r06_rec_shape${K}_sq_counters.csv "$(db $OUT/sq2)"
python tools/pmc_traffic.py "$(db $OUT/f)" "$(db $OUT/w)" $OUT/r06_rec_shape${K}_hbm_traffic.json "rec_shape$K (setup + proofs of tools/prof_recursion_shard.py)" 2
find $OUT -name '*.db' -delete; rm -rf $OUT/stats $OUT/sq $OUT/sq2 $OUT/f $OUT/w
ls -la $OUT
