"""ziren_amd/shape.py — the reference's shape step (crates/core/machine/src/shape/mod.rs, crates/stark/src/shape/cluster.rs, the executor's
shape check crates/core/executor/src/executor.rs:2429-2516) — against the reference's own table and hand-computed cases."""
import json
import os

import pytest

from ziren_amd import fibfast, shape as SH

REF = "/root/reference/crates/core/machine/src/shape/maximal_shapes.json"


def test_the_committed_table_is_the_reference_s():
    """Every maximal shape of the reference's JSON, chip for chip (runs where /root/reference exists; the GPU box has only the fixture)."""
    if not os.path.exists(REF):
        pytest.skip("no /root/reference here")
    src = json.load(open(REF))
    assert sorted(int(k) for k in src) == SH.registered_sizes()
    for key, lst in src.items():
        mine = SH.maximal_shapes(int(key))
        assert len(mine) == len(lst)
        for a, b in zip(mine, lst):
            assert a == b["inner"]
    costs = json.load(open("/root/reference/crates/core/executor/src/artifacts/mips_costs.json"))
    assert SH.costs() == costs


def test_table_is_sane():
    assert SH.registered_sizes() == [17, 18, 19, 20, 21, 22]
    assert len(SH.maximal_shapes(22)) == 364 and len(SH.maximal_shapes(17)) == 1389
    for k in SH.registered_sizes():
        for s in SH.maximal_shapes(k):
            assert "Cpu" in s and s["Cpu"] <= 22 and all(0 <= h <= 22 for h in s.values())
    assert SH.costs()["Cpu"] == 119 and SH.costs()["AddSub"] == 47 and SH.costs()["Program"] == 31


def test_derive_cluster_by_hand():
    """mod.rs:533-610 on a shape written out by hand: Cpu 21 -> gap 1, threshold 17; a chip at or above the threshold keeps its height
    (and, for DivRem / Bitwise / Mul / ShiftLeft / ShiftRight / Global, the height below it as well); a shorter chip is raised to the
    threshold; a chip the maximal shape omits may be left out or sit at 2^10."""
    shape = {"Cpu": 21, "AddSub": 20, "Lt": 12, "DivRem": 4, "Mul": 19, "Global": 17, "MemoryLocal": 15, "Branch": 18}
    cl = SH.derive_cluster(shape)
    assert cl["Cpu"] == [21] and cl["AddSub"] == [20] and cl["Lt"] == [17] and cl["DivRem"] == [16, 17] and cl["Mul"] == [18, 19]
    assert cl["Global"] == [16, 17] and cl["MemoryLocal"] == [17] and cl["Branch"] == [18]
    assert cl["Bitwise"] == [None, 10] and cl["Jump"] == [None, 10] and cl["SyscallCore"] == [None, 10]
    assert set(cl) == set(SH.airs())


def test_find_shape_takes_the_first_height_that_holds_the_rows():
    cl = {"Cpu": [21], "AddSub": [19, 20], "Bitwise": [None, 10], "Mul": [None, 10]}
    assert SH.find_shape(cl, {"Cpu": 1 << 21, "AddSub": (1 << 19) + 1, "Bitwise": 0, "Mul": 3}) == {"Cpu": 21, "AddSub": 20, "Mul": 10}
    assert SH.find_shape(cl, {"Cpu": (1 << 21) + 1, "AddSub": 5, "Bitwise": 0, "Mul": 0}) is None
    assert SH.find_shape(cl, {"Cpu": 7, "AddSub": 5, "Bitwise": 1025, "Mul": 0}) is None
    assert SH.find_shape(cl, {"Cpu": 7, "AddSub": 5, "Jump": 1}) is None            # a chip the cluster does not know


def test_fix_shape_covers_the_record_with_least_area():
    heights = dict.fromkeys(SH.airs(), 0)
    heights.update({"Cpu": 1_000_000, "AddSub": 400_000, "Lt": 100_000, "Branch": 90_000, "MemoryInstrs": 200_000, "MemoryLocal": 5000, "Global": 40_000,
                    "Bitwise": 30_000, "ShiftLeft": 20_000, "ShiftRight": 20_000, "Jump": 10_000})
    shape, key, idx = SH.fix_shape(heights)
    assert key >= 20
    for a, rows in heights.items():
        assert (a in shape and rows <= (1 << shape[a])) or rows == 0
        if a in shape:
            assert shape[a] in SH.clusters(key)[idx][a]
    area = SH.lde_size(shape)
    for k in SH.registered_sizes():                    # nothing registered for this or a larger shard size covers it with fewer cells
        if k >= 20:
            for cl in SH.clusters(k):
                s = SH.find_shape(cl, heights)
                assert s is None or SH.lde_size(s) >= area
    with pytest.raises(SH.ShapeError):
        SH.fix_shape(dict(heights, AddSub=(1 << 22) + 1))


def test_the_fibonacci_loop_under_the_reference_s_defaults():
    """What BASELINE.json's "fibonacci 2^22-row trace" is for this guest: at SHARD_SIZE = 2^21 (MAX_SHARD_SIZE, crates/stark/src/opts.rs:6) the
    executor's shape check closes the shard after 1 569 808 cycles (DivRem / Mul reach 2^18, the tallest any maximal shape of that shard
    size allows), and fix_shape pads the record to a shape registered under 2^22 with Cpu at 2^22 rows. At SHARD_SIZE = 2^22 the shard the
    executor would produce has no covering shape (AddSub with its dependency events needs 2^22 rows): the reference's fix_shape fails too."""
    cycles, why = SH.executor_shard_cycles(1 << 21, fibfast.loop_event_estimate)
    assert (cycles, why) == (1_569_808, "shape")
    m, c2, _ = fibfast.shaped_shard(1 << 21)
    ds = fibfast.DeviceShard(m, shape="fix")
    assert c2 == cycles and ds.shape_key[0] == 22
    assert ds.shape["Cpu"] == 22 and ds.shape["AddSub"] == 21 and ds.shape["Lt"] == 21 and ds.shape["DivRem"] == 18 and ds.shape["Mul"] == 18
    assert ds.shape["Program"] == 19 and "MovCond" not in ds.shape and "MiscInstrs" not in ds.shape
    assert len(ds.chips) == 18
    cycles22, why22 = SH.executor_shard_cycles(1 << 22, fibfast.loop_event_estimate)
    assert why22 == "shape" and 3_100_000 < cycles22 < 3_200_000
    it = cycles22 // 6
    heights = dict.fromkeys(SH.airs(), 0)
    heights.update({"Cpu": cycles22, "AddSub": 5 * it, "Lt": 3 * it, "Mul": it, "DivRem": it, "Branch": it, "MemoryLocal": 2, "Global": 12})
    with pytest.raises(SH.ShapeError):
        SH.fix_shape(heights)


def test_executor_check_margins():
    """executor.rs:2463-2497 by hand: a shape is skipped once the clock passes four times its Cpu height, rejected when a counted chip
    exceeds its height, and only counts as fitting with a margin of 512 events on every chip that has events."""
    shapes = [{a: 10 for a in SH.airs()} | {"Cpu": 12, "AddSub": 11}]
    assert SH.executor_fits(shapes, 100, {"AddSub": 1000})
    assert SH.executor_fits(shapes, 100, {"AddSub": 2048 - 512})
    assert not SH.executor_fits(shapes, 100, {"AddSub": 2048 - 511})
    assert not SH.executor_fits(shapes, 100, {"AddSub": 10, "Mul": 1025})
    assert not SH.executor_fits(shapes, (1 << 14) + 1, {"AddSub": 10})
    assert SH.executor_fits(shapes, 100, {"AddSub": 10, "Lt": 5000})      # Lt is not among the counted chips (MipsAirId::core)
