#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('%-30s %.3f ms/shard  host %.2f ms/proof' % ('$1', l['ms_per_shard'], l['host_ms_per_shard']['rank0_mean']))"; }
for rep in 1 2; do
  for q in 4 3 5 6; do
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "plain GPU_MAX_HW_QUEUES=$q"
  done
  for q in 4 5 6 8; do
    GPU_MAX_HW_QUEUES=$q ZKM_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$q python bench.py --steps 40 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "rccl  GPU_MAX_HW_QUEUES=$q"
  done
done
