"""The reference's shape step for core shards: where the executor closes a shard, and the padded shape `fix_shape` gives its record.

Reference (restated, nothing copied): `CoreShapeConfig` — crates/core/machine/src/shape/mod.rs:40-47 (fields), :71-191 (`fix_shape`, the core
branch :139-191), :370-402 (`maximal_core_shapes`), :448-451 (`estimate_lde_size`), :452-531 (`default`), :533-610
(`derive_cluster_from_maximal_shape`); `ShapeCluster::find_shape` — crates/stark/src/shape/cluster.rs:24-48; the executor's shape check that
ends a shard early — crates/core/executor/src/executor.rs:2407-2516 with `estimate_mips_event_counts`, crates/core/executor/src/cost.rs:96-195.
The prover applies it by default (crates/prover/src/lib.rs:210-213), and `generate_trace` of every chip pads to `fixed_log2_rows`
(the record's shape) instead of the next power of two.

Data: ziren_amd/data/core_shapes.json — the numbers of shape/maximal_shapes.json and mips_costs.json, extracted by
tests/golden/gen_core_shapes.py. `small_shapes.json` (packed records that hold memory init/finalize events beside the cpu events) is not
part of the reference checkout, and a middle shard never takes that branch."""
import json
import os
from functools import lru_cache

HERE = os.path.dirname(os.path.abspath(__file__))

# MipsAirId::core(), crates/core/executor/src/air.rs:126-145: the chips the executor's check looks at (Lt and CloClz are not among them)
EXECUTOR_CHECKED = ("Cpu", "AddSub", "Mul", "Bitwise", "ShiftLeft", "ShiftRight", "DivRem", "MemoryLocal", "Branch", "Jump", "MemoryInstrs",
                    "SyscallInstrs", "MovCond", "MiscInstrs", "SyscallCore", "Global")
# offsets of derive_cluster_from_maximal_shape (mod.rs:552-606): how far below the maximal height a chip may be fixed
_MIN_OFFSET = {"DivRem": 1, "Bitwise": 1, "Mul": 1, "ShiftRight": 1, "ShiftLeft": 1, "Global": 1}
PREPROCESSED_ALLOWED = {"Program": [19, 20, 21, 22], "Byte": [16]}      # mod.rs:465-468
SHAPE_CHECK_FREQUENCY = 16                                             # crates/stark/src/opts.rs:163


class ShapeError(ValueError):
    """CoreShapeError::ShapeError: no allowed shape covers the record."""


@lru_cache(maxsize=1)
def _data():
    return json.load(open(os.path.join(HERE, "data", "core_shapes.json")))


def airs():
    return list(_data()["airs"])


def costs():
    return dict(_data()["costs"])


def maximal_shapes(log_shard_size):
    """The maximal shapes registered under one log2 shard size, as {chip: log2 height} (chips a shape omits are absent)."""
    d = _data()
    return [{a: h for a, h in zip(d["airs"], row) if h >= 0} for row in d["shapes"][str(log_shard_size)]]


def derive_cluster(shape):
    """derive_cluster_from_maximal_shape (mod.rs:533-610): per chip the allowed log2 heights, ascending; None = the chip may be left out."""
    gap = 22 - shape["Cpu"]
    threshold = 18 - gap
    cluster = {}
    for air in _data()["airs"]:
        m = shape.get(air)
        if m is None:
            cluster[air] = [None, 10]
        else:
            tallest = max(m, threshold)
            cluster[air] = list(range(max(tallest - _MIN_OFFSET.get(air, 0), 0), tallest + 1))
    return cluster


@lru_cache(maxsize=None)
def clusters(log_shard_size):
    return [derive_cluster(s) for s in maximal_shapes(log_shard_size)]


def registered_sizes():
    return sorted(int(k) for k in _data()["shapes"])


def find_shape(cluster, heights):
    """ShapeCluster::find_shape (cluster.rs:24-48): per chip the first allowed height that holds its rows; None if some chip fits nowhere.
    Chips whose choice is "left out" (zero rows and None allowed) are dropped from the result."""
    out = {}
    for air, rows in heights.items():
        for lh in cluster.get(air, []):
            if rows <= (0 if lh is None else 1 << lh):
                if lh is not None:
                    out[air] = lh
                break
        else:
            return None
    return out


def lde_size(shape):
    """estimate_lde_size (mod.rs:448-450): cells of the padded traces."""
    c = _data()["costs"]
    return sum(c[a] << h for a, h in shape.items())


def fix_shape(heights):
    """`fix_shape` of a normal core record (mod.rs:139-191): among the clusters registered for shard sizes >= the record's, the covering
    shape of least area. `heights` = MipsAir::core_heights (rows per chip, all 18 entries). Returns ({chip: log2 height}, log2 shard size key,
    cluster index); raises ShapeError when nothing covers the record (the reference returns CoreShapeError::ShapeError and proving stops)."""
    cpu = heights["Cpu"]
    log2_shard = max(cpu - 1, 0).bit_length()              # next_power_of_two().ilog2()
    best = None
    for key in registered_sizes():
        if key < log2_shard:
            continue
        for i, cl in enumerate(clusters(key)):
            s = find_shape(cl, heights)
            if s is not None and (best is None or lde_size(s) < best[0]):
                best = (lde_size(s), s, key, i)
    if best is None:
        raise ShapeError("no shape found for " + str({a: max(h - 1, 0).bit_length() for a, h in heights.items() if h}))
    return best[1], best[2], best[3]


def maximal_core_shapes(log_shard_size):
    """maximal_core_shapes (mod.rs:370-402) as the executor receives them (utils/prove.rs:146-148): per cluster of that shard size the tallest
    allowed height of every chip (10 for a chip the maximal shape omits)."""
    key = max(log_shard_size, min(registered_sizes()))
    return [{a: v[-1] for a, v in cl.items()} for cl in clusters(key)]


def executor_fits(shapes, clk, counts):
    """One shape check of inc_shard_if_need (executor.rs:2450-2497): is there a maximal shape that still holds the estimated counts with a
    margin of 32 checks' worth of events on every counted chip? `clk` is the shard's clock (5 per cycle), `counts` the estimated events."""
    for shape in shapes:
        if clk > ((1 << shape["Cpu"]) << 2):
            continue
        linf, too_small = None, False
        for air in EXECUTOR_CHECKED:
            if air == "Cpu":
                continue
            threshold, count = 1 << shape[air], counts.get(air, 0)
            if count > threshold:
                too_small = True
                break
            if count != 0 and (linf is None or threshold - count < linf):
                linf = threshold - count
        if too_small:
            continue
        if linf is None or linf >= 32 * SHAPE_CHECK_FREQUENCY:
            return True
    return False


# the largest `num_extra_cycles` of the default syscall map (SHA_EXTEND's 48: crates/core/executor/src/syscalls/precompiles/sha256/extend.rs:10-12;
# every other syscall 0 or 1), executor.rs:303-305
MAX_SYSCALL_CYCLES = 48


def executor_shard_cycles(shard_size, counts_at, first_check=0):
    """Cycles the executor runs before it closes a shard whose estimated event counts after c cycles are `counts_at(c)`
    (estimate_mips_event_counts: opcode counts, Mul and Lt raised by the DivRem count, no other dependency — cost.rs:96-195): the smaller of
    the clock limit (`max_syscall_cycles + clk >= shard_size * 4`, executor.rs:325,2423; clk advances 5 per cycle) and the first shape check
    — made when the global clock is a multiple of 16, `first_check` being the shard's first such cycle — at which no maximal shape fits."""
    shapes = maximal_core_shapes(shard_size.bit_length() - 1)          # opts.shard_size.ilog2(), utils/prove.rs:147
    limit = -(-(4 * shard_size - MAX_SYSCALL_CYCLES) // 5)                       # first c with max_syscall_cycles + 5 c >= 4 shard_size
    lo, hi = 0, (limit - first_check) // SHAPE_CHECK_FREQUENCY + 1
    if executor_fits(shapes, 5 * limit, counts_at(limit)):
        return limit, "clock"
    # the counts only grow, so "some shape fits" is monotone: bisect over the check points
    while lo < hi:
        mid = (lo + hi) // 2
        c = first_check + mid * SHAPE_CHECK_FREQUENCY
        if executor_fits(shapes, 5 * c, counts_at(c)):
            lo = mid + 1
        else:
            hi = mid
    return first_check + lo * SHAPE_CHECK_FREQUENCY, "shape"
