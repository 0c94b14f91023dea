#!/bin/bash
# The default line plain, with torch.cuda initialised before the lanes, and on the N > 1 code path (RCCL, world size 1): ms per shard, gather s,
# closing barrier s, host ms per proof.   gpurun -- 'bash tools/diag_rccl_overhead.sh'   (profiles/r05_rccl_world1_overhead.txt)
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('$1', l['ms_per_shard'], l['wall_s']['gather_inside_it'], l['wall_s']['closing_barrier_inside_it'], l['host_ms_per_shard']['rank0_mean'])"; }
for rep in 1 2; do
python bench.py --steps 40 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show plain
python -c "
import torch, sys, runpy
torch.zeros(4, device='cuda').sum().item()
sys.argv=['bench.py','--steps','40','--warmup','3','--no-extra','--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | show torch_cuda_only
ZKM_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2961$rep timeout 300 python bench.py --steps 40 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show rccl
done
