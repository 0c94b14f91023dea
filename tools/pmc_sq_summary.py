#!/usr/bin/env python3
"""Per-kernel SQ counter summary from rocprofv3 --pmc results: where wave cycles go, and how busy the vector ALU is.
Usage: pmc_sq_summary.py sq_results.db out.csv [sq2_results.db]

Definitions (raw counters; gfx950 has 256 CUs x 4 SIMDs, quad-cycle = 4 clocks):
  SQ_WAVE_CYCLES        sum over waves of the quad-cycles each wave is resident                     (a wave's lifetime)
  SQ_WAIT_ANY           ... of those, waiting for anything a s_waitcnt / barrier names              -> WaitAnyPct
  SQ_WAIT_INST_ANY      ... waiting for an instruction to be issued (the issue slot is taken)       -> WaitInstAnyPct
  SQ_ACTIVE_INST_VALU   ... with a VALU instruction of that wave executing                          -> ValuActivePct = share of a WAVE's lifetime
  GRBM_GUI_ACTIVE       clocks the GPU is busy during the dispatch (second pass)
  ValuPipeBusyPct = 100 * SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE * n_SIMD)   with n_SIMD = 1024: share of the SIMDs' time a VALU
                    instruction is in flight — rocprofv3's own derived metric VALUBusy (100 * ACTIVE_INST_VALU / CU_NUM / GUI_ACTIVE, both
                    sides in quad-cycles per CU) is the same number.
  The two percentages answer different questions. With W waves resident per SIMD and the vector pipe never idle, each wave executes 1/W of
  the time: ValuActivePct = ValuPipeBusyPct / W. hash_leaves: 14.6 % of a wave's lifetime at ~6.5 resident waves = ~95 % pipe busy; the
  82 % "issue stall" of the same kernel is the other waves holding the pipe, not lost time.
  bench.py's valu.frac is a third thing: permutations/s against (FP64 lane-rate / instructions per permutation), i.e. pipe-busy share
  times the share of issued instructions that are the permutation's own (the rest: address arithmetic, conversions, loop control)."""
import collections
import csv
import sqlite3
import sys

N_SIMD = 1024


def load(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration), max(vgpr_count), "
                      "max(lds_block_size) from counters_collection group by kernel_name, counter_name").fetchall()
    agg, meta = collections.defaultdict(dict), {}
    for k, c, v, n, dur, vg, lds in rows:
        short = k.split("(")[0].replace("void ", "")
        agg[short][c] = v
        meta[short] = (n, dur, vg, lds)
    return agg, meta


def main(db_path, out_path, db2_path=None):
    agg, meta = load(db_path)
    agg2 = load(db2_path)[0] if db2_path else {}
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Dispatches", "TotalDurationUs", "VGPRs", "LDSBytes", "WaitAnyPct(memory/barrier)",
                    "WaitInstAnyPct(issue stall)", "ValuActivePct(of wave lifetime)", "LdsActivePct", "ValuInstsPerWave",
                    "ValuPipeBusyPct(of SIMD time)", "MeanWavesPerSimd", "LdsInstsPerWave", "VmemInstsPerWave", "SaluInstsPerWave",
                    "LdsBankConflictPctOfGpuTime"])
        for k, d in sorted(agg.items(), key=lambda kv: -meta[kv[0]][1]):
            n, dur, vg, lds = meta[k]
            wc = d.get("SQ_WAVE_CYCLES", 0) or 1
            waves = max(d.get("SQ_WAVES", 1), 1)
            pct = lambda c: round(100 * d.get(c, 0) / wc, 1)
            d2 = agg2.get(k, {})
            gui = d2.get("GRBM_GUI_ACTIVE", 0)
            # GRBM_GUI_ACTIVE is reported per XCC instance: the sum over the 8 XCCs / 8 is the dispatch's busy clocks
            busy = round(100 * d2.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (gui / 8 * N_SIMD), 1) if gui else ""
            resident = round(wc * 4 / (gui / 8 * N_SIMD), 2) if gui else ""
            per_wave = lambda c: round(d2.get(c, 0) / waves, 1) if d2 else ""
            conflict = round(100 * d2.get("SQ_LDS_BANK_CONFLICT", 0) / (gui / 8 * 256), 1) if gui else ""
            w.writerow([k, n, round(dur / 1e3, 1), vg, lds, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"),
                        pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_LDS"), round(d.get("SQ_INSTS_VALU", 0) / waves),
                        busy, resident, per_wave("SQ_INSTS_LDS"), per_wave("SQ_INSTS_VMEM"), per_wave("SQ_INSTS_SALU"), conflict])
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
