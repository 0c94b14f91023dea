#!/bin/bash
# The side-stream LDE overlap was tuned with one lane. With two lanes?  (GPU_MAX_HW_QUEUES 16: every stream its own hardware queue; 4: default)
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('%-34s %.3f ms/shard  host %.2f ms/proof' % ('$1', l['ms_per_shard'], l['host_ms_per_shard']['rank0_mean']))"; }
for rep in 1 2 3; do
  for q in 16 4; do
    for ov in 1 0; do
      GPU_MAX_HW_QUEUES=$q ZKM_LDE_OVERLAP=$ov python bench.py --steps 40 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show "queues=$q side-stream overlap $ov"
    done
  done
done
