show() { python -c "
import json,sys
l=json.loads(sys.stdin.read()); h=l['host_cpu_s_per_shard']; print('$1: %.3f ms/shard cores busy %.2f lanes %s other %.2f backend %s' % (l['ms_per_shard'], h['cores_busy_rank0'], h['lane_threads_cores_rank0'], h['other_threads_cores_rank0'], l['config']['backend']))"; }
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show plain
ZKM_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 timeout 300 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show rccl_world1
ZKM_BENCH_ONE_DEVICE=1 ZKM_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 timeout 300 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | show gloo_world1
ZKM_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 2>/dev/null | show gloo_two_ranks_one_device
