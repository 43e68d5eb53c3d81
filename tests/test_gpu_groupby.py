"""Parity of the HIP group-by with expected_GroupBy (groupby/groupby.cpp:8-19) through the C ABI."""
import json

import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _run(keys_h, vals_h, groups):
    from dwarf_bench_amd import ops
    k = torch.from_numpy(np.ascontiguousarray(keys_h, dtype=np.uint32).view(np.int32)).cuda()
    v = torch.from_numpy(np.ascontiguousarray(vals_h, dtype=np.uint32).view(np.int32)).cuda()
    return ops.groupby_sum(k, v, groups).cpu().numpy().view(np.uint32)


def test_reference_fixture(golden_dir):
    k = json.loads((golden_dir / "reference_kats.json").read_text())["groupby_fixture"]
    assert _run(k["keys"], k["vals"], k["groups"]).tolist() == k["expected"]


@pytest.mark.parametrize("groups", [1, 2, 20, 64, 257, 4096, 32768, 32769, 65536, 100000])
@pytest.mark.parametrize("n", [0, 1, 5, 128, 4096, 100003, 1 << 20])
def test_groupby_matches_oracle(n, groups):
    keys = po.gen_uniform_u32(n, 42, 0, groups - 1)
    vals = po.gen_uniform_u32(n, 43, 1, 10000)
    assert np.array_equal(_run(keys, vals, groups), po.groupby_sum(keys, vals, groups))


def test_wraparound_sums():
    n, groups = 50000, 7
    keys = po.gen_uniform_u32(n, 1, 0, groups - 1)
    vals = po.gen_uniform_u32(n, 2, 2**31, 2**32 - 1)
    assert np.array_equal(_run(keys, vals, groups), po.groupby_sum(keys, vals, groups))


@pytest.mark.parametrize("groups", [32769, 65535, 65536, 65537, 131072 + 5, 300001])
@pytest.mark.parametrize("vals_kind", ["reference", "carry_often", "full_range", "all_ones_16", "skew_one_pair"])
def test_more_than_32768_groups_whatever_the_values(groups, vals_kind):
    """More than 32768 groups: the kernel chooses between two 16-bit partial sums per LDS word (every row read once) and
    key ranges of 32-bit sums (every row read per range) from a sample of the columns.  Value ranges that never, sometimes
    and always carry, wide values, every row on one word's two groups; group counts at the table boundaries, odd, and
    over several key ranges; uint32 wrap-around exact against expected_GroupBy restated."""
    n = 400003
    keys, vals = _big_group_case(n, groups, vals_kind)
    assert np.array_equal(_run(keys, vals, groups), po.groupby_sum(keys, vals, groups))


def _big_group_case(n, groups, vals_kind):
    keys = po.gen_uniform_u32(n, 21, 0, groups - 1)
    if vals_kind == "reference":
        vals = po.gen_uniform_u32(n, 22, 1, 10000)
    elif vals_kind == "carry_often":
        vals = po.gen_uniform_u32(n, 22, 30000, 65535)
    elif vals_kind == "full_range":
        vals = po.gen_uniform_u32(n, 22, 0, 2**32 - 1)
    elif vals_kind == "all_ones_16":
        vals = np.full(n, 0xFFFF, np.uint32)
    else:  # every row on one word's two groups: both halves carry all the time, and into each other
        keys = (np.arange(n, dtype=np.uint32) & 1) + np.uint32(groups - 2 - (groups & 1))
        vals = po.gen_uniform_u32(n, 22, 0xFF00, 0x1FFFF)
    return keys, vals


@pytest.mark.parametrize("mode", ["force", "0"])
def test_both_large_table_modes_on_every_input(mode):
    """DBHIP_GB_PACKED=force / 0 pins the packed / the wide mode (a fresh process: the library reads the variable once):
    the packed table with its LDS carry counters (two 4-bit counters per word, saturating) and its global spill path is
    exact on inputs its own sample would never have chosen it for — wide values, values that carry on most rows, every
    row on one word, ONE private table for all rows (executors = 1: dozens of carries per group, saturated counters) —
    and the wide mode on the ones that would have been packed."""
    import os, subprocess, sys
    prog = (
        "import numpy as np, torch\n"
        "from dwarf_bench_amd import ops\n"
        "from oracle import pyoracle as po\n"
        "from tests.test_gpu_groupby import _big_group_case\n"
        "def dev(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).cuda()\n"
        "n = 400003\n"
        "for groups in (32776, 65536, 65537, 200001):\n"
        "    for kind in ('reference', 'carry_often', 'full_range', 'all_ones_16', 'skew_one_pair'):\n"
        "        k, v = _big_group_case(n, groups, kind)\n"
        "        for executors in (0, 1, 3):\n"
        "            plan = ops.GroupBySum(n, groups)\n"
        "            plan.partial(dev(k), dev(v), executors); plan.merge(executors)\n"
        "            got = plan.result().cpu().numpy().view(np.uint32)\n"
        "            assert np.array_equal(got, po.groupby_sum(k, v, groups)), (groups, kind, executors)\n"
        "print('both modes: ok')\n")
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "DBHIP_GB_PACKED": mode}, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert r.returncode == 0 and "both modes: ok" in r.stdout, (mode, r.stdout[-2000:], r.stderr[-3000:])


def test_the_mode_choice_avoids_the_packed_tables_cliffs():
    """What the sample-based choice is for: the packed table is the faster one at BASELINE's configuration (2^26 rows,
    2^16 groups, values in [1, 10000]) and would be 3.5x - 36x slower than the wide mode on full-range values, values up
    to 60000 at 2^26 rows without its LDS carry counters, or keys clustered in the input.  Timing-free check of the
    decision itself through the header word the kernel writes: mode 1 = packed, 2 = wide."""
    from dwarf_bench_amd import ops
    n, groups = 1 << 25, 65536
    keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)

    def mode_of(k, v):
        plan = ops.GroupBySum(n, groups)
        plan.launch(k, v)
        plan.result()
        return int(plan.ws[4:8].view(torch.int32).item())

    assert mode_of(keys, ops.gen_uniform_u32(n, 43, 1, 10000)) == 1
    assert mode_of(keys, ops.gen_uniform_u32(n, 43, 0, 2**32 - 1)) == 2          # wide values
    assert mode_of(torch.sort(keys).values, ops.gen_uniform_u32(n, 43, 1, 10000)) == 2  # clustered keys: hot in every window
    hot = keys.clone()
    hot[::50] = 7                                                                  # one key with 2 % of the rows
    assert mode_of(hot, ops.gen_uniform_u32(n, 43, 1, 10000)) == 2
    assert mode_of(keys, ops.gen_uniform_u32(n, 43, 1, 60000)) == 2                # sums that would carry on most groups
    small = ops.gen_uniform_u32(1 << 22, 42, 0, groups - 1)                        # too short for the packed table to pay
    plan = ops.GroupBySum(1 << 22, groups)
    plan.launch(small, ops.gen_uniform_u32(1 << 22, 43, 1, 10000))
    plan.result()
    assert int(plan.ws[4:8].view(torch.int32).item()) == 2


@pytest.mark.parametrize("groups", [8192, 16384, 32768, 40000, 65536, 100000])
@pytest.mark.parametrize("kind", ["one group", "90 % one group", "sorted keys", "runs of 24 and of 40 rows", "full-range values"])
def test_rows_of_one_group_in_many_lanes(kind, groups):
    """Tables with fewer than four lane copies sum a wave's rows of one group before the LDS add (groupby.hip
    crowd_guard): crowds of every size, mixed with spread keys, in ranges the workgroup does not own, ragged ends."""
    n = (1 << 20) + 3
    rng = np.random.default_rng(17)
    keys = po.gen_uniform_u32(n, 42, 0, groups - 1)
    vals = po.gen_uniform_u32(n, 43, 1, 10000)
    if kind == "one group":
        keys[:] = groups - 1
    elif kind == "90 % one group":
        keys[rng.random(n) < 0.9] = groups // 3
    elif kind == "sorted keys":
        keys = np.sort(keys)
    elif kind == "runs of 24 and of 40 rows":  # around the crowd threshold, four keys per lane apart
        keys[: n // 256 * 256].reshape(-1, 256)[:, 0:96:4] = 5
        keys[: n // 256 * 256].reshape(-1, 256)[:, 97:256:4] = groups - 2
    else:
        keys[:] = groups // 2
        vals = po.gen_uniform_u32(n, 43, 0, 2**32 - 1)
    assert np.array_equal(_run(keys, vals, groups), po.groupby_sum(keys, vals, groups))


def test_one_hot_group_costs_no_multiple_of_uniform_keys():
    """Round 3: 2^26 rows of one key took 251 us against 106 for uniform keys at 32768 groups (471 against 111 at 65536):
    64 lanes queueing on one LDS word.  Now 114 / 200 us; the bounds leave room for the box."""
    from dwarf_bench_amd import ops
    n = 1 << 26
    vals = ops.gen_uniform_u32(n, 43, 1, 10000)

    def median_us(keys, groups):
        plan = ops.GroupBySum(n, groups)
        out = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            plan.launch(keys, vals)
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) * 1e3)
        return sorted(out[1:])[2]

    # (a timing ratio inside a functional suite: the bound has to be missed three times in a row before the run fails;
    #  DBENCH_SKIP_PERF_TESTS=1 skips it; the correctness of these inputs is test_rows_of_one_group_in_many_lanes' business)
    import os
    if os.environ.get("DBENCH_SKIP_PERF_TESTS"):
        pytest.skip("DBENCH_SKIP_PERF_TESTS is set")
    for groups, bound in ((32768, 1.6), (65536, 2.6)):
        uniform_keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
        hot_keys = torch.full((n,), groups // 3, dtype=torch.int32, device="cuda")
        seen = []
        for _attempt in range(3):
            uniform, hot = median_us(uniform_keys, groups), median_us(hot_keys, groups)
            seen.append((hot, uniform))
            if hot < bound * uniform:
                break
        else:
            raise AssertionError((groups, seen))


def test_skewed_keys():
    n, groups = 1 << 18, 65536
    keys = np.zeros(n, np.uint32)
    keys[::3] = 65535
    keys[1::1000] = 32768
    vals = po.gen_uniform_u32(n, 5, 1, 10000)
    assert np.array_equal(_run(keys, vals, groups), po.groupby_sum(keys, vals, groups))


def test_out_of_range_key_is_flagged():
    from dwarf_bench_amd import ops, _capi
    k = torch.tensor([0, 1, 9, 2], dtype=torch.int32).cuda()
    v = torch.ones(4, dtype=torch.int32).cuda()
    plan = ops.GroupBySum(4, 4)
    plan.launch(k, v)
    assert ops.workspace_status(plan.ws) == ops.DEV_KEY_RANGE
    with pytest.raises(_capi.DbhipError):
        plan.result()


def test_baseline_config_2_26_rows_2_16_groups():
    """BASELINE configs[2].  Cross-checked with torch (independent) and with the oracle, both at full size."""
    from dwarf_bench_amd import ops
    n, groups = 1 << 26, 1 << 16
    keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
    vals = ops.gen_uniform_u32(n, 43, 1, 10000)
    got = ops.groupby_sum(keys, vals, groups)
    exp = torch.zeros(groups, dtype=torch.int64, device="cuda").index_add_(0, keys.to(torch.int64), vals.to(torch.int64))
    assert torch.equal(got.to(torch.int64) & 0xFFFFFFFF, exp & 0xFFFFFFFF)
    assert int(got.to(torch.int64).sum().item()) == int(vals.to(torch.int64).sum().item())  # checksum of checksums
    # the oracle (expected_GroupBy restated, groupby/groupby.cpp:8-19) on the WHOLE columns
    kh, vh = keys.cpu().numpy().view(np.uint32), vals.cpu().numpy().view(np.uint32)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), po.groupby_sum(kh, vh, groups))


# ---- two-phase entry points (GroupByLocal, groupby/groupby_local.cpp:52-112) --------------------------
@pytest.mark.parametrize("executors", [0, 1, 2, 7, 64, 1024, 1 << 20])
@pytest.mark.parametrize("n,groups", [(0, 3), (5, 1), (4096, 64), (100003, 257), (1 << 20, 64), (1 << 20, 40000),
                                      (300000, 100000)])
def test_partial_then_merge(n, groups, executors):
    from dwarf_bench_amd import ops
    keys = po.gen_uniform_u32(n, 42, 0, groups - 1)
    vals = po.gen_uniform_u32(n, 43, 1, 10000)
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int32)).cuda()
    plan = ops.GroupBySum(n, groups)
    plan.partial(k, v, executors)
    plan.merge(executors)
    got = plan.result().cpu().numpy().view(np.uint32)
    assert np.array_equal(got, po.groupby_sum(keys, vals, groups))
    if n and executors:  # the oracle's own privatised form agrees too (threads_count = executors)
        assert np.array_equal(got, po.groupby_local(keys, vals, groups, min(executors, 1024)))


def test_partial_merge_is_repeatable():
    from dwarf_bench_amd import ops
    n, groups = 200000, 64
    keys = po.gen_uniform_u32(n, 5, 0, groups - 1)
    vals = po.gen_uniform_u32(n, 6, 1, 10000)
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int32)).cuda()
    plan = ops.GroupBySum(n, groups)
    want = po.groupby_sum(keys, vals, groups)
    for executors in (4, 0, 4):
        plan.partial(k, v, executors)
        plan.merge(executors)
        assert np.array_equal(plan.result().cpu().numpy().view(np.uint32), want)


def test_zero_groups_and_unaligned_columns():
    from dwarf_bench_amd import ops
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    plan = ops.GroupBySum(0, 0)
    plan.ws.fill_(0xAB)
    plan.launch(empty, empty)
    assert plan.result().numel() == 0  # clean status word although nothing ran
    k = ops.gen_uniform_u32(1001, 1, 0, 9)
    v = ops.gen_uniform_u32(1001, 2, 1, 5)
    with pytest.raises(ValueError, match="16-byte"):
        ops.groupby_sum(k[1:], v[1:], 10)
