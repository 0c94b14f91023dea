#!/bin/bash
# How long may a generated quotient kernel be? (round 5: the instruction cache two CUs share holds 64 KiB; KeccakSponge's parts are
# 160-210 KiB each, DivRem's / Global's single kernels about 100.) ZKM_Q_PART = statements per part, ZKM_Q_SINGLE = bytecode
# instructions above which a program is cut at all.   gpurun --timeout 2400 -- 'bash tools/ab_part_size.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
keccak() {
  local L=$1; shift
  env "$@" python tools/bench_keccak_shard.py --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('keccak $L: prove %.3f ms  quotient %.3f ms (%d launches)' % (d['prove_ms'], k['quotient'][0], k['quotient'][1]))"
}
fib() {
  local L=$1; shift
  env "$@" python bench.py --resident --no-extra --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels_ms']
print('fib $L: step %.3f ms  quotient %.3f ms (%d launches)  verified %s' % (l['ms_per_step'], k['quotient']['ms'], k['quotient']['launches'], l['verified']))"
}
keccak part6000 ZKM_Q_PART=6000
keccak part3000 ZKM_Q_PART=3000
keccak part1500 ZKM_Q_PART=1500
keccak part800 ZKM_Q_PART=800
fib single ZKM_Q_SINGLE=12000
fib cut_above_1200_parts_of_600 ZKM_Q_SINGLE=1200 ZKM_Q_PART=600
fib cut_above_1000_parts_of_500 ZKM_Q_SINGLE=1000 ZKM_Q_PART=500
fib single_again ZKM_Q_SINGLE=12000
