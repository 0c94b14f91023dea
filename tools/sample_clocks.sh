#!/bin/bash
# Engine clock and socket power sampled every 0.25 s while the default line proves 400 shards (about 18 s of sustained load), and while the
# Poseidon2 micro-benchmark's 24 ms launches run: is the hashing rate set by the nominal 2.4 GHz or by what the power limit sustains?
#   gpurun --timeout 600 -- 'bash tools/sample_clocks.sh'   -> gpurun_out/r05_clocks_under_load.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r05_clocks_under_load.txt
sample() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|Average Graphics Package Power" | tr -s ' \t' ' ' | tr '\n' '|'; echo; }
{ echo "# idle"; sample; } > $OUT
python bench.py --no-extra --no-cpu-baseline --steps 400 --warmup 5 > $R/gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
echo "# under the default line (400 shards, two lanes)" >> $OUT
while kill -0 $BP 2>/dev/null; do sample >> $OUT; sleep 0.25; done
python -c "
import json; l=json.load(open('$R/gpurun_out/clock_bench.json')); print('# line: %.3f ms/shard over %d shards' % (l['ms_per_step'], l['shards']))" >> $OUT
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -E "Max Graphics Package Power|Performance Level" | tr -s ' \t' ' ' >> $OUT
