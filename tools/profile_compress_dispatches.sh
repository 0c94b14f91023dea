#!/bin/bash
# Per-dispatch counters of merkle::compress_layer over two proofs of the resident leg (gpurun -- 'bash tools/profile_compress_dispatches.sh'):
#   -> gpurun_out/r05_compress_layer_dispatches.csv
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cdisp; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/c -o c -- \
  python $R/bench.py --resident --no-cpu-baseline --no-extra --steps 1 --warmup 1 --kernel-timing 0 > /dev/null 2> $OUT/c.err
cd $R
DB=$(find $OUT/c -name '*_results.db' | head -1)
python tools/pmc_per_dispatch.py "$DB" "" $R/gpurun_out/r05_all_dispatches.csv
python tools/pmc_per_dispatch.py "$DB" compress_layer $R/gpurun_out/r05_compress_layer_dispatches.csv
python tools/pmc_per_dispatch.py "$DB" hash_leaves $R/gpurun_out/r05_hash_leaves_dispatches.csv
python tools/pmc_per_dispatch.py "$DB" hash_rows $R/gpurun_out/r05_hash_rows_dispatches.csv
python tools/pmc_per_dispatch.py "$DB" lde_cols $R/gpurun_out/r05_lde_cols_dispatches.csv
rm -rf $OUT/c
