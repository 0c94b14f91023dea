#!/bin/bash
# Three (four) lanes were measured "no better than two" with HIP's default four hardware queues, where a third lane's streams share queues
# with the first two's. With a queue for every stream?   gpurun -- 'bash tools/ab_lanes_queues.sh'
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('%-34s %.3f ms/shard  host %.2f ms/proof  cores %.2f' % ('$1', l['ms_per_shard'], l['host_ms_per_shard']['rank0_mean'], l['host_cpu_s_per_shard']['cores_busy_rank0']))"; }
for rep in 1 2; do
  python bench.py --steps 48 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "2 lanes, default queues"
  for q in 6 8 12; do
    GPU_MAX_HW_QUEUES=$q python bench.py --inflight 3 --steps 48 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "3 lanes, GPU_MAX_HW_QUEUES=$q"
  done
  python bench.py --inflight 3 --steps 48 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "3 lanes, default queues"
  GPU_MAX_HW_QUEUES=12 python bench.py --inflight 4 --steps 48 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "4 lanes, GPU_MAX_HW_QUEUES=12"
done
