#!/bin/bash
# A/B on ONE box: how a farm lane's host thread waits (zkm_ctx_set_host_wait): ms per shard of the default line and host cores busy.
#   gpurun --timeout 1200 -- 'bash tools/ab_host_wait.sh'
run() { local L=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); h=l['host_cpu_s_per_shard']; print('$L: %.3f ms/shard  resident one lane (spinning) %.3f  cores busy %.2f lanes %s other %.2f' % (l['ms_per_shard'], l['resident_one_lane']['ms_per_step'], h['cores_busy_rank0'], h['lane_threads_cores_rank0'], h['other_threads_cores_rank0']))"; }
for rep in 1 2 3; do
run lanes_blocking ZKM_BENCH_LANE_WAIT=blocking
run lanes_spinning ZKM_BENCH_LANE_WAIT=spin
run lanes_blocking_confined_to_2_cores ZKM_BENCH_LANE_WAIT=blocking taskset -c 0-1
run lanes_blocking_confined_to_1_core ZKM_BENCH_LANE_WAIT=blocking taskset -c 0
done
