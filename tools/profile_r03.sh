#!/bin/bash
# Round-3 profiles of the benchmark command, to be run on the GPU box from the repository root:
#   gpurun --timeout 1500 -- 'bash tools/profile_r03.sh [quick]'
# 1. rocprofv3 --kernel-trace --stats of `python bench.py` (per-kernel average durations)          -> r03_syn22_kernel_stats.csv, r03_syn22_bench.json
#    and from the same trace the gaps between dispatches                                            -> r03_syn22_idle_gaps.json
# 2. PMC passes (separate runs, counters only): FETCH_SIZE, WRITE_SIZE -> HBM bytes per launch      -> r03_syn22_hbm_traffic.json
#    SQ wave-cycle breakdown + VALU busy (two passes: the counters do not fit one)                  -> r03_syn22_sq_counters.csv
# 3. SQ_INSTS_VALU over the Poseidon2 microbenchmark                                                -> r03_poseidon2_isa.json
# `quick` stops after step 1 and the SQ pass.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --log-rows 22 --no-cpu-baseline --no-pcie --no-inflight2 --fib="
db() { find "$1" -name '*_results.db' | head -1; }
python $R/bench.py --log-rows 22 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r03_syn22_bench_unprofiled.json 2> $OUT/bench_unprofiled.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $B --steps 5 --warmup 1 > $OUT/r03_syn22_bench.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq -o sq -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $OUT/sq2 -o sq2 -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/sq2.err
cd $R
python tools/rocprof_summary.py "$(db $OUT/stats)" $OUT/r03_syn22_kernel_stats.csv
python tools/rocprof_gaps.py "$(db $OUT/stats)" $OUT/r03_syn22_idle_gaps.json > $OUT/gaps.txt
python tools/pmc_sq_summary.py "$(db $OUT/sq)" $OUT/r03_syn22_sq_counters.csv "$(db $OUT/sq2)"
if [ "${1:-}" != "quick" ]; then
  cd /tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/f.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o w -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/w.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/p2 -o p2 -- $R/tools/ubench_p2 gpu > $OUT/r03_ubench_poseidon2_int_vs_f64.txt 2> $OUT/p2.err
  cd $R
  python tools/pmc_traffic.py "$(db $OUT/f)" "$(db $OUT/w)" $OUT/r03_syn22_hbm_traffic.json
  python tools/pmc_poseidon2.py "$(db $OUT/p2)" $OUT/r03_poseidon2_isa.json
fi
find $OUT -name '*.db' -delete     # the databases are large; the summaries are what is kept
rm -rf $OUT/stats $OUT/sq $OUT/sq2 $OUT/f $OUT/w $OUT/p2
ls -la $OUT
