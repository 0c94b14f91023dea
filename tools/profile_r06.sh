#!/bin/bash
# Round-6 profiles of the benchmark command (default workload: the shaped fibonacci shard, tag fibs21), run on the GPU box from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/profile_r06.sh [quick] [syn]'
# 1. (last) the default line unprofiled, as the driver runs it (--steps 20 --warmup 5)              -> r06_fibs21_bench.json
# 2. rocprofv3 --kernel-trace --stats of the resident leg (bench.py --resident), overlap on and off (per-kernel average durations, gaps)      -> r06_fibs21_kernel_stats.csv, r06_fibs21_idle_gaps.json
# 3. PMC passes (separate runs, counters only): SQ wave-cycle breakdown + VALU busy (two passes)    -> r06_fibs21_sq_counters.csv
#    FETCH_SIZE, WRITE_SIZE -> HBM bytes per launch                                                 -> r06_fibs21_hbm_traffic.json
#    SQ_INSTS_VALU over the Poseidon2 microbenchmark                                                -> r06_poseidon2_isa.json
# 4. the recursion-tree reduce: per-shape legs + trees of 8 / 16 / 32 leaves (tools/bench_reduce_tree.py)  -> r06_reduce_tree.json
#    and the dispatch timeline of one shape-0 recursion shard (rocprofv3 --kernel-trace)                 -> r06_rec_shape0_timeline.json, r06_rec_shape0_dispatches.csv
# 5. from the default line: r06_micro.json (its `micro`), r06_cpu_baseline_full.json (its `cpu_baseline`, measured at full size)
# `quick` stops after the SQ passes; `syn` repeats 2-3 for --workload syn (tag syn22). ZKM_COMMIT (set by the caller: the GPU box has no .git)
# is recorded in the traffic profile.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*_results.db' | head -1; }
QUICK=0; SYN=0
for a in "$@"; do [ "$a" = quick ] && QUICK=1; [ "$a" = syn ] && SYN=1; done
profile() {   # tag, extra bench args
  local T=$1; shift
  local B="python $R/bench.py --resident $* --no-cpu-baseline --no-extra"      # the resident one-lane leg: what the per-kernel figures of the line are quoted on
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$T -o stats -- $B --steps 5 --warmup 1 > $OUT/r06_${T}_bench_profiled.json 2> $OUT/stats_$T.err
  # the same with the side-stream LDE overlap off: every kernel alone on the main stream (the mode of the line's kernels_ms table and of the PMC passes, which serialise)
  ZKM_LDE_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $OUT/statsser_$T -o stats -- $B --steps 5 --warmup 1 > $OUT/r06_${T}_bench_profiled_serialised.json 2> $OUT/statsser_$T.err
  # counter passes: 1 warm-up + 1 timed proof, no per-kernel timing pass (2 proofs per run)
  local P="$B --steps 1 --warmup 1 --kernel-timing 0"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq_$T -o sq -- $P > /dev/null 2> $OUT/sq_$T.err
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $OUT/sq2_$T -o sq2 -- $P > /dev/null 2> $OUT/sq2_$T.err
  cd $R
  python tools/rocprof_summary.py "$(db $OUT/stats_$T)" $OUT/r06_${T}_kernel_stats.csv
  python tools/rocprof_summary.py "$(db $OUT/statsser_$T)" $OUT/r06_${T}_kernel_stats_serialised.csv
  python tools/rocprof_gaps.py "$(db $OUT/stats_$T)" $OUT/r06_${T}_idle_gaps.json > $OUT/gaps_$T.txt
  python tools/pmc_sq_summary.py "$(db $OUT/sq_$T)" $OUT/r06_${T}_sq_counters.csv "$(db $OUT/sq2_$T)"
  if [ $QUICK = 0 ]; then
    cd /tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f_$T -o f -- $P > /dev/null 2> $OUT/f_$T.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w_$T -o w -- $P > /dev/null 2> $OUT/w_$T.err
    cd $R
    python tools/pmc_traffic.py "$(db $OUT/f_$T)" "$(db $OUT/w_$T)" $OUT/r06_${T}_hbm_traffic.json "$T" 2
  fi
  find $OUT -name '*.db' -delete
  rm -rf $OUT/stats_$T $OUT/statsser_$T $OUT/sq_$T $OUT/sq2_$T $OUT/f_$T $OUT/w_$T
}
profile fibs21
[ $SYN = 1 ] && profile syn22 --workload syn
if [ $QUICK = 0 ] && [ -x $R/tools/ubench_p2 ]; then      # a build product (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ziren_amd/csrc tools/ubench_p2.hip -o tools/ubench_p2)
  cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/p2 -o p2 -- $R/tools/ubench_p2 gpu > $OUT/r06_ubench_poseidon2_int_vs_f64.txt 2> $OUT/p2.err
  cd $R
  python tools/pmc_poseidon2.py "$(db $OUT/p2)" $OUT/r06_poseidon2_isa.json
  find $OUT -name '*.db' -delete; rm -rf $OUT/p2
fi
# the default line LAST (the claim queue from events, two lanes + the resident leg), with this run's counter traffic in place: bench.py quotes
# roofline.traffic only from a profile whose csrc digest is the tree's
cp $OUT/r06_fibs21_hbm_traffic.json $R/profiles/ 2>/dev/null
cd $R
python $R/bench.py --steps 20 --warmup 5 > $OUT/r06_fibs21_bench.json 2> $OUT/bench.err
python - <<PY
import json
l = json.loads(open("$OUT/r06_fibs21_bench.json").read().strip().splitlines()[-1])
json.dump(l["micro"], open("$OUT/r06_micro.json", "w"), indent=1)
json.dump(l["cpu_baseline"], open("$OUT/r06_cpu_baseline_full.json", "w"), indent=1)
open("$OUT/core_ms.txt", "w").write(str(l["ms_per_shard"]))
PY
python $R/tools/bench_reduce_tree.py --leaves 8,16,32 --steps 5 --core-ms $(cat $OUT/core_ms.txt) --out $OUT/r06_reduce_tree.json > /dev/null 2> $OUT/reduce.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/tl -o t -- python $R/tools/prof_recursion_shard.py --shape 0 --steps 6 > $OUT/tl.out 2> $OUT/tl.err
cd $R
python tools/rocprof_timeline.py "$(db $OUT/tl)" $OUT/r06_rec_shape0_timeline.json $OUT/r06_rec_shape0_dispatches.csv > /dev/null
find $OUT -name '*.db' -delete; rm -rf $OUT/tl
ls -la $OUT
