#!/bin/bash
# What in "torch.cuda initialised beside the prover" costs the default line 0.4 ms per shard?  gpurun -- 'bash tools/diag_torch_overhead.sh'
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('%-28s %.3f ms/shard  host %.2f ms/proof  cores %.2f' % ('$1', l['ms_per_shard'], l['host_ms_per_shard']['rank0_mean'], l['host_cpu_s_per_shard']['cores_busy_rank0']))"; }
variant() {   # label, python prelude
  python -c "
$2
import sys, runpy
sys.argv=['bench.py','--steps','40','--warmup','3','--no-extra','--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | show "$1"
}
for rep in 1 2 3; do
  variant plain "pass"
  variant import_torch "import torch"
  variant cuda_init_no_kernel "import torch; torch.cuda.init(); torch.cuda.current_device()"
  variant cuda_init_and_kernel "import torch; torch.zeros(4, device='cuda').sum().item()"
done
