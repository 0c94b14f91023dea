#!/bin/bash
# rocprofv3 traces (kernels + memory copies; no counters) of the pipelined events -> proof loop and of the same proofs on resident traces:
#   gpurun --timeout 900 -- 'bash tools/profile_e2p.sh'      -> gpurun_out/${ROUND}e2p/${ROUND}_e2p_trace.json, ${ROUND}_resident_trace.json (ROUND: r05 unless set)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r05}
OUT=$R/gpurun_out/${ROUND}e2p; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*_results.db' | head -1; }
python $R/tools/bench_e2p.py 8 > $OUT/e2p_unprofiled.json 2> $OUT/e2p.err
python $R/tools/bench_e2p.py 8 21 resident > $OUT/resident_unprofiled.json 2>> $OUT/e2p.err
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t1 -o t -- python $R/tools/bench_e2p.py 8 > $OUT/e2p_profiled.json 2> $OUT/t1.err
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t2 -o t -- python $R/tools/bench_e2p.py 8 21 resident > $OUT/resident_profiled.json 2> $OUT/t2.err
cd $R
python tools/rocprof_e2p.py "$(db $OUT/t1)" $OUT/${ROUND}_e2p_trace.json 6 > $OUT/e2p_summary.txt 2>&1
python tools/rocprof_e2p.py "$(db $OUT/t2)" $OUT/${ROUND}_resident_trace.json 6 > $OUT/resident_summary.txt 2>&1
find $OUT -name '*.db' -delete; rm -rf $OUT/t1 $OUT/t2
cat $OUT/e2p_unprofiled.json $OUT/resident_unprofiled.json; tail -40 $OUT/e2p_summary.txt; tail -30 $OUT/resident_summary.txt
