#!/bin/bash
# shake out the N = 8 path and the host-CPU budget of a rank on the ONE GPU a box has.
#   gpurun --timeout 2400 -- 'bash tools/shakeout_n8.sh'
# (a) eight real processes sharing device 0 (ZKM_BENCH_ONE_DEVICE=1: process group over gloo), one lane each, shards cut at SHARD_SIZE 2^18:
#     start-up, page-locked pools, the claim counter in the store, the gather, the bounded post-run verification; wall time per phase.
# (b) the one-GPU two-lane line at full size with the process confined to TWO cores (16 CPUs / 8 ranks) against unconfined: ms per shard and
#     process CPU seconds per shard — what a rank of an 8-GPU node needs from the host.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/shakeout; mkdir -p $OUT
t0=$(date +%s.%N)
ZKM_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --inflight 1 --shard-size-log 18 --steps 4 --warmup 1 > $OUT/eight_ranks.json 2> $OUT/eight_ranks.err
rc=$?; t1=$(date +%s.%N)
echo "eight ranks on one device: rc $rc, wall $(python3 -c "print(round($t1 - $t0, 1))") s"
t0=$(date +%s.%N)
ZKM_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --inflight 2 --shard-size-log 18 --steps 4 --warmup 1 > $OUT/eight_ranks_two_lanes.json 2> $OUT/eight_ranks_two_lanes.err
rc2=$?; t2=$(date +%s.%N)
echo "eight ranks x two lanes on one device: rc $rc2, wall $(python3 -c "print(round($t2 - $t0, 1))") s"
for rep in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $OUT/free_$rep.json 2> $OUT/free_$rep.err
  taskset -c 0-1 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $OUT/two_cores_$rep.json 2> $OUT/two_cores_$rep.err
  taskset -c 0 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $OUT/one_core_$rep.json 2> $OUT/one_core_$rep.err
done
python - "$OUT" "$(python3 -c "print(round($t1 - $t0, 1))")" <<'PY'
import json, sys, os
out = sys.argv[1]
def line(name):
    try:
        return json.loads(open(os.path.join(out, name)).read().strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e), "stderr_tail": open(os.path.join(out, name.replace(".json", ".err"))).read()[-1500:]}
res = {"eight_ranks_one_lane": {}, "eight_ranks_two_lanes": {}, "host_budget": {}}
for key, f in (("eight_ranks_one_lane", "eight_ranks.json"), ("eight_ranks_two_lanes", "eight_ranks_two_lanes.json")):
    l = line(f)
    res[key] = l if "error" in l else {k: l[k] for k in ("n_gpus", "value", "ms_per_shard", "shards", "shards_proved", "fewest_shards_on_a_rank", "verified_proofs",
                                                            "host_ms_per_shard", "host_cpu_s_per_shard", "wall_s")} | {"parallelism": l["config"]["parallelism"], "backend": l["config"]["backend"]}
for name in ("free", "two_cores", "one_core"):
    runs = [line(f"{name}_{r}.json") for r in (1, 2)]
    res["host_budget"][name] = [r if "error" in r else {"ms_per_shard": r["ms_per_shard"], "value": r["value"], "host_cpu_s_per_shard": r["host_cpu_s_per_shard"]["rank0"],
                                                         "resident_one_lane_ms": r["resident_one_lane"]["ms_per_step"], "wall_s": r["wall_s"]} for r in runs]
json.dump(res, open(os.path.join(out, "r05_eight_ranks_one_device.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
