#!/bin/bash
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two lanes own ten streams. Does the default line care?
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('%-22s %.3f ms/shard  host %.2f ms/proof' % ('$1', l['ms_per_shard'], l['host_ms_per_shard']['rank0_mean']))"; }
for rep in 1 2 3; do
  for q in 4 8 16 2; do
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "GPU_MAX_HW_QUEUES=$q"
  done
done
