import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as O
from ziren_amd import synth, abi
sh = synth.syn_shard(14)
fri = abi.FriConfig(1, 84, 16)
for nt in (8, 16, 32, 64, 128):
    O.lib().orc_set_num_threads(nt)
    pk = O.Pk([], [], sh.pc_start, sh.initial_global_cumulative_sum, 1)
    ch = O.new_challenger(); pk.observe_into(ch)
    t = time.time()
    _, tm = O.prove_shard(pk, sh.chips, [c.trace for c in sh.chips], sh.public_values, fri, synth.NUM_PV_ELTS, ch)
    print(nt, "threads:", round(time.time() - t, 2), "s", tm, flush=True)
