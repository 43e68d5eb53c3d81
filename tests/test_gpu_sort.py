"""Parity of the HIP radix sort with std::sort semantics (sort/radix.cpp:8-12) through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 100003, 1 << 20])
def test_sort_u32_full_range(n, bits):
    from dwarf_bench_amd import ops
    keys = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    host = keys.cpu().numpy().view(np.uint32)
    ops.radix_sort_(keys, signed=False, radix_bits=bits)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32), po.sort_u32(host))


@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("n", [128, 256, 512, 1024, 2048, 4096, 50000])
def test_sort_reference_distribution_signed(n, bits):
    """the reference sorts `int` drawn from [1,10000] (sort/radix.cpp:19): upper digits are constant -> skipped passes"""
    from dwarf_bench_amd import ops
    keys = ops.gen_uniform_u32(n, 7, 1, 10000)
    host = keys.cpu().numpy()
    ops.radix_sort_(keys, signed=True, radix_bits=bits)
    assert np.array_equal(keys.cpu().numpy(), po.sort_i32(host))


@pytest.mark.parametrize("bits", [8, 4])
def test_sort_signed_negative_values(bits):
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(3)
    host = rng.integers(-2**31, 2**31 - 1, 77777, dtype=np.int64).astype(np.int32)
    host[:5] = [np.iinfo(np.int32).min, np.iinfo(np.int32).max, 0, -1, 1]
    keys = torch.from_numpy(host.copy()).cuda()
    ops.radix_sort_(keys, signed=True, radix_bits=bits)
    assert np.array_equal(keys.cpu().numpy(), np.sort(host))


@pytest.mark.parametrize("bits", [8, 4])
def test_sort_degenerate_inputs(bits):
    from dwarf_bench_amd import ops
    for host in (np.zeros(10000, np.int32), np.full(9999, -1, np.int32), np.arange(20000, dtype=np.int32),
                 np.arange(20000, dtype=np.int32)[::-1].copy(), (np.arange(30000) % 2).astype(np.int32)):
        keys = torch.from_numpy(host.copy()).cuda()
        ops.radix_sort_(keys, signed=False, radix_bits=bits)
        assert np.array_equal(keys.cpu().numpy().view(np.uint32), np.sort(host.view(np.uint32)))


def _crowded(kind, n, rng):
    """keys that crowd into few digits: the lanes of a wave then meet on the same LDS counters (histograms in several
    copies, ranking by ballots where a wave sees a crowd: radix.hip rs_hist_copies / rs_rank_rows)"""
    full = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if kind == "two values":
        return np.where(rng.integers(0, 2, n) == 1, np.uint32(0xFFFFFFFF), np.uint32(0))
    if kind == "16 values":
        return (rng.integers(0, 16, n, dtype=np.uint64) * 0x11111111).astype(np.uint32)
    if kind == "90 % one value":
        return np.where(rng.random(n) < 0.9, np.uint32(0x9E3779B9), full)
    if kind == "geometric":
        return (rng.random(n) ** 8 * 4294967295.0).astype(np.uint64).astype(np.uint32)
    if kind == "sorted":
        return np.sort(full)
    if kind == "crowded and spread waves in one tile":  # every other run of 1024 keys (one wave's share of a tile) is one value
        k = full.copy()
        k.reshape(-1)[: n // 2048 * 2048].reshape(-1, 2, 1024)[:, 1, :] = 0x01020304
        return k
    raise ValueError(kind)


@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("kind", ["two values", "16 values", "90 % one value", "geometric", "sorted",
                                  "crowded and spread waves in one tile"])
def test_sort_keys_that_crowd_into_few_digits(kind, bits):
    from dwarf_bench_amd import ops
    for n in ((1 << 20) + 777, 5000):  # many chunks with a ragged last tile; the one-workgroup sort
        host = _crowded(kind, n, np.random.default_rng(11))
        keys = torch.from_numpy(host.view(np.int32).copy()).cuda()
        plan = ops.RadixSort(n, bits)
        plan.launch(keys)
        assert ops.workspace_status(plan.ws) == 0
        assert np.array_equal(keys.cpu().numpy().view(np.uint32), po.sort_u32(host))


def test_crowded_keys_cost_no_multiple_of_spread_keys():
    """Round 3 found the sort 3 - 3.5 x slower on two distinct values or 90 % one value than on uniform keys (2^24 keys,
    8-bit: 600-700 us against 200): same-word LDS atomics in the histograms and in the ranking.  Now 1.45 x; the bound
    leaves room for the box."""
    from dwarf_bench_amd import ops
    n = 1 << 24
    rng = np.random.default_rng(5)
    plan = ops.RadixSort(n, 8)

    def median_us(host):
        src = torch.from_numpy(host.view(np.int32).copy()).cuda()
        keys = src.clone()
        out = []
        for _ in range(6):
            keys.copy_(src)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            plan.launch(keys)
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) * 1e3)
        assert ops.workspace_status(plan.ws) == 0
        return sorted(out[1:])[2]

    # a timing ratio inside a functional suite: one noisy median on a shared box must not fail the run — the bound has to
    # be missed three times in a row (the correctness of these inputs is test_sort_keys_that_crowd_into_few_digits'
    # business); DBENCH_SKIP_PERF_TESTS=1 skips it altogether
    import os
    if os.environ.get("DBENCH_SKIP_PERF_TESTS"):
        pytest.skip("DBENCH_SKIP_PERF_TESTS is set")
    spread_keys = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    for kind in ("two values", "90 % one value"):
        crowded_keys = _crowded(kind, n, rng)
        seen = []
        for _attempt in range(3):
            spread, crowded = median_us(spread_keys), median_us(crowded_keys)
            seen.append((crowded, spread))
            if crowded < 2.2 * spread:
                break
        else:
            raise AssertionError((kind, seen))


@pytest.mark.parametrize("bits", [8, 4])
def test_sort_2_24_baseline_config(bits):
    """BASELINE configs[1]: 2^24 uint32 keys.  Checked against torch.sort of the same bits and by properties."""
    from dwarf_bench_amd import ops
    n = 1 << 24
    keys = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    as_i64 = keys.to(torch.int64) & 0xFFFFFFFF
    exp = torch.sort(as_i64).values
    plan = ops.RadixSort(n, bits)
    plan.launch(keys)
    assert ops.workspace_status(plan.ws) == 0
    got = keys.to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got, exp)
    # idempotence: sorting sorted data changes nothing
    plan.launch(keys)
    assert torch.equal(keys.to(torch.int64) & 0xFFFFFFFF, exp)
    # the oracle (std::sort restated, sort/radix.cpp:8-12) on the WHOLE column
    host = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1).cpu().numpy().view(np.uint32)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32), po.sort_u32(host))


@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("n", [(1 << 24) + 1, 2896 * 8192 - 3, 2897 * 8192 + 5, (1 << 25) + 8193, 1025 * 8192 + 1])
def test_sort_sizes_around_the_chunk_geometry_steps(n, bits):
    """the number of tiles per chunk steps where tiles / target passes sqrt(q (q + 1)) (8-bit) or an integer (4-bit):
    radix.hip rs_geometry; the count matrix's rows are padded to four chunks.  Against torch.sort of the same bits."""
    from dwarf_bench_amd import ops
    keys = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    exp = torch.sort(keys.to(torch.int64) & 0xFFFFFFFF).values
    plan = ops.RadixSort(n, bits)
    plan.launch(keys)
    assert ops.workspace_status(plan.ws) == 0
    assert torch.equal(keys.to(torch.int64) & 0xFFFFFFFF, exp)


def test_large_input_properties_2_30():
    """2^30 + 3 keys (4 GiB): sorted, same multiset (order-independent checksums), 64 tiles per chunk"""
    from dwarf_bench_amd import ops
    n = (1 << 30) + 3
    keys = ops.gen_uniform_u32(n, 5, 0, 2**32 - 1)
    s0 = int(keys.to(torch.int64).sum().item())
    plan = ops.RadixSort(n, 8)
    plan.launch(keys)
    torch.cuda.synchronize()
    assert ops.workspace_status(plan.ws) == 0
    step = 1 << 28
    for lo in range(0, n - 1, step):  # unsigned order: compare in int64 with the sign bit folded
        a = keys[lo: min(lo + step + 1, n)].to(torch.int64) & 0xFFFFFFFF
        assert bool((a[1:] >= a[:-1]).all()), lo
    assert int(keys.to(torch.int64).sum().item()) == s0
    # the histogram of the top byte is preserved and matches a uniform draw
    top = (keys.to(torch.int64) & 0xFFFFFFFF) >> 24
    c = torch.bincount(top, minlength=256)
    assert int(c.sum()) == n and float(c.max()) / float(c.min()) < 1.01


def test_empty_input_and_unaligned_column():
    """n == 0 leaves a clean status word in whatever workspace is passed; a column that does not start on a
    16-byte boundary is refused with a clear message (include/dbhip.h)"""
    from dwarf_bench_amd import ops
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    assert ops.radix_sort_(empty).numel() == 0
    plan = ops.RadixSort(0)
    plan.ws.fill_(0xAB)
    plan.launch(empty)
    assert ops.workspace_status(plan.ws) == 0
    keys = ops.gen_uniform_u32(1001, 1, 0, 2**32 - 1)
    with pytest.raises(ValueError, match="16-byte"):
        ops.radix_sort_(keys[1:])


def test_rank_by_lds_atomics_is_self_checked_and_both_rankings_agree():
    """The scatter ranks with one returning LDS atomic per key on gfx950 (allow-list); dbhip_radix_sort_prepare runs the
    device-side lane-order self-test and pins the mode to what it saw (gfx950 passes); DBHIP_RS_RANK=ballot keeps the
    ballot ranking: both give the oracle's order, at sizes on every path (single tile, fused scan, chunked) and both
    digit widths."""
    import os, subprocess, sys
    from dwarf_bench_amd import _capi, ops
    assert _capi.lib().dbhip_radix_sort_rank_mode() == 1  # before any sort and without a self-test: the allow-list
    assert ops.radix_sort_prepare() == 1                  # the self-test agrees
    keys = ops.gen_uniform_u32(1 << 20, 3, 0, 2**32 - 1)
    ops.radix_sort_(keys, signed=False, radix_bits=8)
    assert _capi.lib().dbhip_radix_sort_rank_mode() == 1
    prog = (
        "import numpy as np, torch\n"
        "from dwarf_bench_amd import _capi, ops\n"
        "from oracle import pyoracle as po\n"
        "for n in (100, 8192, 8193, 200000, (1 << 22) + 77):\n"
        "    for bits in (8, 4):\n"
        "        k = ops.gen_uniform_u32(n, 11, 0, 2**32 - 1); h = k.cpu().numpy().view(np.uint32)\n"
        "        ops.radix_sort_(k, signed=False, radix_bits=bits)\n"
        "        assert np.array_equal(k.cpu().numpy().view(np.uint32), po.sort_u32(h)), (n, bits)\n"
        "print('mode', _capi.lib().dbhip_radix_sort_rank_mode())\n")
    for mode, want in (("ballot", "mode 0"), ("atomic", "mode 1")):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300,
                           env={**os.environ, "DBHIP_RS_RANK": mode}, cwd=os.path.dirname(os.path.dirname(__file__)))
        assert r.returncode == 0 and want in r.stdout, (mode, r.stdout, r.stderr)


def test_unstable_ranking_raises_the_rank_order_status():
    """Every tile of every pass checks the invariant the stable ranking exists for (a tile re-ordered by the pass's digit
    is sorted by its lower digits).  DBHIP_RS_INJECT_UNSTABLE=1 swaps one pair of neighbours inside a digit run of the
    first tile (of the one-workgroup kernel for inputs of at most 8192 keys too), after the ranking and before the write-out
    — what an unstable rank would produce: the status word must
    carry DBHIP_DEV_RANK_ORDER (and the result really is mis-sorted), in both rank modes and both digit widths; the same
    program without the injection is clean."""
    import os, subprocess, sys
    prog = (
        "import numpy as np, torch\n"
        "from dwarf_bench_amd import ops\n"
        "bad = 0\n"
        "for n in (5000, 100003, (1 << 21) + 5):\n"
        "    for bits in (8, 4):\n"
        "        k = ops.gen_uniform_u32(n, 11, 0, 2**32 - 1); h = np.sort(k.cpu().numpy().view(np.uint32))\n"
        "        plan = ops.RadixSort(n, bits); plan.launch(k); torch.cuda.synchronize()\n"
        "        st = ops.workspace_status(plan.ws)\n"
        "        same = bool(np.array_equal(k.cpu().numpy().view(np.uint32), h))\n"
        "        assert (st == 0) == same, (n, bits, st, same)\n"
        "        assert st in (0, ops.DEV_RANK_ORDER), st\n"
        "        bad += st != 0\n"
        "print('flagged', bad)\n")
    for mode in ("atomic", "ballot"):
        for inject, want in (("1", "flagged 6"), ("0", "flagged 0")):
            r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300,
                               env={**os.environ, "DBHIP_RS_RANK": mode, "DBHIP_RS_INJECT_UNSTABLE": inject},
                               cwd=os.path.dirname(os.path.dirname(__file__)))
            assert r.returncode == 0 and want in r.stdout, (mode, inject, r.stdout, r.stderr)
