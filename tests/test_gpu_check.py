"""The exclusive-scan entry point and the device-side validators the `...Hip` dwarfs use at
BASELINE sizes, each against the oracle / a host restatement, through the C ABI.

  exclusive scan   tests/scan_tests.cpp:46-51 KAT ({0,1,1,0,0,1,1} -> {0,0,1,2,2,2,3}), prefix_sum_scalar :14-21
  validators       they replace the dwarfs' own host checks (scan/scan.cpp:157-164, sort/radix.cpp:46-52,
                   groupby/groupby.cpp:95-103, join/join_omnisci.cpp:31-45): each must accept what the oracle
                   produces and reject a corrupted copy of it.
"""
import json

import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
M64 = (1 << 64) - 1
FP_MUL = 0x9E3779B97F4A7C15


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


# ---------------------------------------------------------------------------------------------------------------
def test_exclusive_scan_reference_kat(golden_dir):
    from dwarf_bench_amd import ops
    kat = json.loads((golden_dir / "reference_kats.json").read_text())["scan_prefix_sum"]
    got = ops.exclusive_scan(_dev(np.array(kat["in"], dtype=np.uint32))).cpu().numpy()
    assert got.tolist() == kat["expected"] == [0, 0, 1, 2, 2, 2, 3]


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 32767, 32768, 32769, 65536, 100003,
                               (1 << 22) + 5, (1 << 24) + 1027])
def test_exclusive_scan_matches_oracle(n):
    from dwarf_bench_amd import ops
    src = ops.gen_uniform_u32(n, 3, 0, 0xFFFFFFFF if n % 2 else 10000)  # wrap-around sums on the odd sizes
    host = src.cpu().numpy().view(np.uint32)
    exp = np.zeros(n, dtype=np.uint32)
    if n:
        np.cumsum(host[:-1], dtype=np.uint32, out=exp[1:])
    got = ops.exclusive_scan(src).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, exp)
    if n <= 4096:  # the oracle's restatement of prefix_sum_scalar (int arithmetic; same bits)
        assert np.array_equal(got.view(np.int32), po.prefix_sum_exclusive(host.view(np.int32)))


def test_exclusive_scan_init_in_place_and_unaligned():
    from dwarf_bench_amd import ops
    n = 70001
    base = ops.gen_uniform_u32(n + 3, 5, 0, 1000)
    src = base[3:]  # 12 bytes off a 16-byte boundary: the scalar path
    host = src.cpu().numpy().view(np.uint32)
    exp = (np.concatenate([[0], np.cumsum(host[:-1], dtype=np.uint64)]) + 77).astype(np.uint32)
    assert np.array_equal(ops.exclusive_scan(src, init=77).cpu().numpy().view(np.uint32), exp)
    ops.exclusive_scan(src, init=77, out=src)
    assert np.array_equal(src.cpu().numpy().view(np.uint32), exp)


def test_exclusive_scan_aligned_in_place_init_and_streams():
    """the single-launch path (16-byte aligned columns): init, in place, and four scans in flight on four streams
    (chunks go by ticket: a chunk's predecessors always belong to running workgroups)"""
    from dwarf_bench_amd import ops
    n = (1 << 23) + 77
    src = ops.gen_uniform_u32(n, 9, 0, 0xFFFFFFFF)
    host = src.cpu().numpy().view(np.uint32)
    exp = np.zeros(n, dtype=np.uint32)
    np.cumsum(host[:-1], dtype=np.uint32, out=exp[1:])
    exp += np.uint32(123)
    outs = [torch.empty_like(src) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    plans = [ops.ExclusiveScan(n) for _ in range(4)]  # asynchronous launches, one workspace each
    torch.cuda.synchronize()
    for _ in range(3):
        for o, st, plan in zip(outs, streams, plans):
            with torch.cuda.stream(st):
                plan.launch(src, init=123, out=o)
    torch.cuda.synchronize()
    for o, plan in zip(outs, plans):
        assert ops.workspace_status(plan.ws) == 0  # no bounded wait of the single-launch path ran out
        assert np.array_equal(plan.result().cpu().numpy().view(np.uint32), exp)
    work = src.clone()
    ops.exclusive_scan(work, init=123, out=work)
    assert np.array_equal(work.cpu().numpy().view(np.uint32), exp)


def test_exclusive_scan_beyond_32_bit_byte_offsets():
    from dwarf_bench_amd import ops
    n = (1 << 30) + 5
    src = ops.gen_uniform_u32(n, 4, 0, 3)
    got = ops.exclusive_scan(src)
    step, carry = 1 << 27, 0
    for lo in range(0, n, step):
        part = src[lo: lo + step].to(torch.int64)
        incl = torch.cumsum(part, 0) + carry
        assert torch.equal(got[lo: lo + step].to(torch.int64) & 0xFFFFFFFF, (incl - part) & 0xFFFFFFFF), lo
        carry = int(incl[-1])


def test_exclusive_scan_feeds_the_omnisci_positions():
    """count -> position step of common/dpcpp/omnisci_hashtable.hpp:252-254 on the reference's join fixture"""
    from dwarf_bench_amd import ops
    cnt = np.array([0, 2, 0, 1, 1, 0, 3, 0], dtype=np.uint32)
    assert ops.exclusive_scan(_dev(cnt)).cpu().numpy().tolist() == [0, 0, 2, 2, 3, 4, 4, 7]


# ---------------------------------------------------------------------------------------------------------------
def _host_fingerprint(seq):
    h = 0
    for x in seq:
        h = (h * FP_MUL + (int(np.uint32(x)) + 1)) & M64
    return h, len(seq)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 16385, 300007])
@pytest.mark.parametrize("filt", [5, 5001, 20000])
def test_fingerprint_matches_host_restatement(n, filt):
    from dwarf_bench_amd import ops
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    exp = _host_fingerprint(po.copy_if_lt(src.cpu().numpy(), filt))
    assert ops.check_fingerprint_lt(src, filt) == exp


def test_fingerprint_validates_the_scan_and_rejects_faults():
    from dwarf_bench_amd import ops
    n = (1 << 22) + 77
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    for filt in (5, 1001):
        out = ops.copy_if_lt(src, filt)
        want = ops.check_fingerprint_lt(src, filt)
        assert want[1] == out.numel() and ops.check_fingerprint_lt(out, filt) == want
        if out.numel() >= 3:
            bad = out.clone()
            bad[[1, 2]] = bad[[2, 1]].clone() if int(bad[1]) != int(bad[2]) else torch.tensor([0, 1], dtype=torch.int32, device="cuda")
            assert ops.check_fingerprint_lt(bad, filt) != want  # order matters
            assert ops.check_fingerprint_lt(out[:-1], filt) != want  # a lost element


def test_fingerprint_at_the_headline_size():
    """2^28 rows: the property RadixHip/TwoPassScanHip rely on where a host copy_if is out of reach"""
    from dwarf_bench_amd import ops
    n = 1 << 28
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    out = ops.copy_if_lt(src, 5)
    want = ops.check_fingerprint_lt(src, 5)
    assert want[1] == out.numel() > 0
    assert ops.check_fingerprint_lt(out, 5) == want


# ---------------------------------------------------------------------------------------------------------------
def test_sorted_check():
    from dwarf_bench_amd import ops
    n = 1 << 20
    keys = ops.gen_uniform_u32(n, 7, 0, 0xFFFFFFFF)
    before = ops.check_sorted(keys)
    assert before[0] > 0
    ops.radix_sort_(keys)
    after = ops.check_sorted(keys)
    assert after[0] == 0 and after[1:] == before[1:]
    host = keys.cpu().numpy().view(np.uint32)
    assert after[2] == int(host.astype(np.uint64).sum()) & M64
    # signed order: the same array is NOT sorted as int32 (negative keys come last in unsigned order)
    assert ops.check_sorted(keys, signed=True)[0] == 1
    keys[12345] = keys[12345] + 1  # multiset changes (and possibly the order)
    assert ops.check_sorted(keys)[1:] != before[1:]


def test_weighted_sum_check():
    from dwarf_bench_amd import ops
    n, groups = 1 << 20, 1000
    k = ops.gen_uniform_u32(n, 1, 0, groups - 1)
    v = ops.gen_uniform_u32(n, 2, 0, 0xFFFFFFFF)  # wrap-around group sums
    out = ops.groupby_sum(k, v, groups)
    assert ops.check_weighted_sum(None, out) == ops.check_weighted_sum(k, v)
    bad = out.clone()
    bad[3], bad[4] = out[4], out[3]  # sums delivered to the wrong group
    assert int(out[3]) == int(out[4]) or ops.check_weighted_sum(None, bad) != ops.check_weighted_sum(k, v)


def test_permutation_check():
    from dwarf_bench_amd import ops
    n = 100003
    perm = torch.randperm(n, device="cuda").to(torch.int32)
    assert ops.check_permutation(perm) == 0
    perm[5] = perm[6]
    assert ops.check_permutation(perm) == 1
    perm[7] = n
    assert ops.check_permutation(perm) == 2
    assert ops.check_permutation(perm[:0]) == 0


@pytest.mark.parametrize("n,hi", [(4096, 10000), (1 << 17, (1 << 17) - 1), (300007, 5000)])
def test_join_check_accepts_the_join_and_rejects_faults(n, hi):
    from dwarf_bench_amd import ops
    a = ops.gen_uniform_u32(n, 42, 1, hi)
    b = ops.gen_uniform_u32(n, 43, 1, hi)
    pos, cnt, ids = ops.hash_join(a, b)
    srt = a.clone()
    ops.radix_sort_(srt)
    bad, total = ops.check_join(srt, b, pos, cnt, ids, build_keys=a)
    exp = po.join_counts_fast(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))
    assert bad == 0 and total == int(exp.astype(np.uint64).sum())
    assert ops.check_permutation(ids) == 0
    hit = int(torch.nonzero(cnt > 0)[0])
    c2 = cnt.clone()
    c2[hit] += 1
    assert ops.check_join(srt, b, pos, c2, ids, build_keys=a)[0] == 1
    i2 = ids.clone()
    victim = int(pos[hit])
    other = int(torch.nonzero(a != a[int(ids[victim])])[0])
    i2[victim] = other  # an id of a row that carries another key
    assert ops.check_join(srt, b, pos, cnt, i2, build_keys=a)[0] >= 1


def test_join_check_with_generated_global_ids():
    """the partitioned join's id buffer holds GLOBAL row ids: the key of an id is regenerated, not looked up"""
    from dwarf_bench_amd import ops
    n, first = 1 << 16, 1 << 20
    a = ops.gen_uniform_u32(n, 42, 0, n - 1, first_index=first)
    b = ops.gen_uniform_u32(n, 43, 0, n - 1, first_index=first)
    rid = torch.arange(first, first + n, device="cuda", dtype=torch.int64).to(torch.int32)
    plan = ops.HashJoin(n, n)
    plan.build(a, rid)
    plan.probe(b)
    pos, cnt, ids = plan.result()
    srt = a.clone()
    ops.radix_sort_(srt)
    assert ops.check_join(srt, b, pos, cnt, ids, build_keys=None, gen=(42, 0, n - 1))[0] == 0
    assert ops.check_join(srt, b, pos, cnt, ids, build_keys=None, gen=(41, 0, n - 1))[0] > 0  # wrong generator
    assert ops.check_gen_uniform(a, 42, 0, n - 1, first_index=first) == 0
    assert ops.check_gen_uniform(a, 42, 0, n - 1, indices=rid) == 0
    assert ops.check_gen_uniform(b, 42, 0, n - 1, first_index=first) > 0


def test_ujoin_check():
    from dwarf_bench_amd import ops
    n = 1 << 17
    ak, av = ops.gen_unique_sorted_u32(n, 11), ops.gen_unique_sorted_u32(n, 12)
    bk, bv = ops.gen_unique_sorted_u32(n, 13), ops.gen_unique_sorted_u32(n, 14)
    plan = ops.UniqueJoin(n, n)
    plan.build(ak, av)
    plan.probe(bk, bv)
    ok, o1, o2 = plan.result()
    bad, hits = ops.check_ujoin(ak, av, bk, bv, ok, o1, o2)
    assert bad == 0 and hits == int((ok != -1).sum())
    o1b = o1.clone()
    o1b[int(torch.nonzero(ok != -1)[0])] ^= 1
    assert ops.check_ujoin(ak, av, bk, bv, ok, o1b, o2)[0] == 1
