"""Parity of the HIP bitmask-claimed table with SimpleNonOwningHashTable (common/dpcpp/hashtable.hpp:5-93):
the reference's own known answers (tests/hash_table_tests.cpp) at slot level, then the oracle's table on
random keys, through the C ABI."""
import json

import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).cuda()


@pytest.fixture(scope="module")
def kats(golden_dir):
    return json.loads((golden_dir / "reference_kats.json").read_text())


def _table(k, inserts):
    from dwarf_bench_amd import ops
    t = ops.BitmaskTable(k["size"], hash_kind=0)
    ins = np.array(inserts, dtype=np.uint32)
    t.insert(_dev(ins[:, 0]), _dev(ins[:, 1]), serial=True)  # one at a time, like the single_task in the test
    return t


def test_reference_build_kat(kats):
    k = kats["bitmask_table_build"]
    t = _table(k, k["inserts"])
    data = t.slot_values().cpu().numpy().view(np.uint32)
    for slot, val in k["expected_data"].items():
        assert data[int(slot)] == val
    # hash_table_tests.cpp:38-54: the same key twice claims two neighbouring slots
    from dwarf_bench_amd import ops
    t2 = ops.BitmaskTable(k["size"], hash_kind=0)
    key = k["double_insert_key"]
    t2.insert(_dev([key, key]), _dev(k["double_insert_vals"]), serial=False)  # both inserts in flight at once
    d = t2.slot_values().cpu().numpy().view(np.uint32)
    assert int(d[10]) + int(d[11]) == k["expected_sum_slots_10_11"]


def test_reference_has_kat(kats):
    k = kats["bitmask_table_has"]
    t = _table(k, k["inserts"])
    _, found = t.lookup(_dev(k["queries"]))
    assert found.cpu().tolist() == k["expected"]


def test_reference_probe_kat(kats):
    k = kats["bitmask_table_probe"]
    t = _table(k, k["inserts"])
    vals, found = t.lookup(_dev(k["queries"]))
    assert found.cpu().tolist() == [1] * len(k["queries"])
    assert vals.cpu().tolist() == k["expected_vals"]


@pytest.mark.parametrize("hash_kind,seed", [(0, 0), (1, 421), (1, 7)])
@pytest.mark.parametrize("size,n", [(64, 10), (1000, 500), (1 << 16, 30000)])
def test_serial_insert_matches_oracle_slots(size, n, hash_kind, seed):
    """inserted in input order the table must be slot-for-slot the oracle's"""
    from dwarf_bench_amd import ops
    keys = po.gen_uniform_u32(n, 11, 0, 2**32 - 2)
    vals = po.gen_uniform_u32(n, 12, 0, 2**32 - 1)
    want = po.BitmaskTable(size, hash_kind, seed)
    for kk, vv in zip(keys.tolist(), vals.tolist()):
        want.insert(kk, vv)
    t = ops.BitmaskTable(size, hash_kind, seed)
    t.insert(_dev(keys), _dev(vals), serial=True)
    assert np.array_equal(t.slot_values().cpu().numpy().view(np.uint32), want.data())


@pytest.mark.parametrize("hash_kind,seed", [(0, 0), (1, 421)])
@pytest.mark.parametrize("size,n", [(64, 64), (1000, 500), (1 << 16, 30000), (1 << 21, 1 << 20), (3000001, 1500000)])
def test_parallel_insert_then_lookup(size, n, hash_kind, seed):
    """unique keys inserted concurrently: slot order is free, the key->payload map is not
    (the contract hash_build.cpp:60-83 and join.cpp:96-131 rely on)"""
    from dwarf_bench_amd import ops
    keys = po.gen_unique_sorted_u32(n, 3)
    rng = np.random.default_rng(1)
    rng.shuffle(keys)
    vals = po.gen_uniform_u32(n, 12, 0, 2**32 - 1)
    t = ops.BitmaskTable(size, hash_kind, seed)
    t.insert(_dev(keys), _dev(vals))
    got_vals, found = t.lookup(_dev(keys))
    assert bool((found == 1).all())
    assert np.array_equal(got_vals.cpu().numpy().view(np.uint32), vals)
    # keys that were never inserted are reported absent while the table has free slots
    if n < size:
        absent = np.setdiff1d(po.gen_uniform_u32(1000, 99, 0, 2**32 - 2), keys)
        _, f2 = t.lookup(_dev(absent))
        assert not bool(f2.any())


def test_reset_empties_the_table():
    from dwarf_bench_amd import ops
    t = ops.BitmaskTable(4096, 1, 5)
    keys = _dev(np.arange(1, 2001))
    t.insert(keys, keys)
    assert bool((t.lookup(keys)[1] == 1).all())
    t.reset()
    assert not bool(t.lookup(keys)[1].any())


def test_duplicate_keys_all_found():
    """hash_build.cpp feeds keys from [1, 10000] with duplicates: each occupies its own slot, every key is found
    (BigBuild in tests/hash_table_tests.cpp:183-236 does the same with 500 keys)"""
    from dwarf_bench_amd import ops
    n = 100000
    keys = po.gen_uniform_u32(n, 42, 1, 10000)
    t = ops.BitmaskTable(2 * n, 1, 421)
    t.insert(_dev(keys), _dev(keys))
    vals, found = t.lookup(_dev(keys))
    assert bool((found == 1).all())
    assert np.array_equal(vals.cpu().numpy().view(np.uint32), keys)


def test_ragged_last_word_wraps_instead_of_leaving_the_table():
    """size % 32 != 0: a run reaching the end continues at slot 0 (hashtable.hpp:83-85); the reference's
    `minor += occupied` could step past `size` there, oracle and kernel wrap instead"""
    from dwarf_bench_amd import ops
    size = 70
    ins = [(69, 1), (69, 2), (69, 3), (5, 4), (139, 5), (68, 6), (68, 7)]
    want = po.BitmaskTable(size, 0, 0)
    slots = [want.insert(k, v) for k, v in ins]
    assert slots == [69, 0, 1, 5, 2, 68, 3] and max(slots) < size
    t = ops.BitmaskTable(size, hash_kind=0)
    arr = np.array(ins, dtype=np.uint32)
    t.insert(_dev(arr[:, 0]), _dev(arr[:, 1]), serial=True)
    t.check()
    assert np.array_equal(t.slot_values().cpu().numpy().view(np.uint32), want.data())
    vals, found = t.lookup(_dev([69, 139, 68, 5, 70]))
    assert found.cpu().tolist() == [1, 1, 1, 1, 0] and vals.cpu().tolist()[:4] == [1, 5, 6, 4]
    # the same rows inserted concurrently: every slot stays inside the table, every key is found
    t2 = ops.BitmaskTable(size, hash_kind=0)
    t2.insert(_dev(arr[:, 0]), _dev(arr[:, 1]))
    t2.check()
    assert bool((t2.lookup(_dev(arr[:, 0]))[1] == 1).all())


@pytest.mark.parametrize("size", [64, 70, 1000])
def test_fill_to_capacity_then_overflow_is_reported(size):
    from dwarf_bench_amd import _capi, ops
    t = ops.BitmaskTable(size, 1, 3)
    keys = np.arange(1, size + 1, dtype=np.uint32)
    t.insert(_dev(keys), _dev(keys))
    t.check()
    vals, found = t.lookup(_dev(keys))
    assert bool((found == 1).all()) and np.array_equal(vals.cpu().numpy().view(np.uint32), keys)
    t.insert(_dev([size + 1]), _dev([0]))  # no free slot: the reference would spin forever
    with pytest.raises(_capi.DbhipError):
        t.check()
