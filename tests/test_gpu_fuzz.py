"""Seeded randomised parity sweep: many (size, parameter, distribution) combinations per dwarf against the oracle,
to catch tail / alignment / boundary cases the hand-picked sizes miss.  Deterministic (fixed seeds)."""
import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def _sizes(rng, count, hi_log2):
    """log-uniform sizes plus neighbours of powers of two and of the kernels' tile sizes"""
    out = [int(2 ** rng.uniform(0, hi_log2)) for _ in range(count)]
    for base in (64, 256, 1024, 2048, 4096, 8192, 32768, 1 << 16, 1 << 17, 1 << 20):
        out += [base + int(d) for d in rng.integers(-3, 4, size=2)]
    return [max(1, s) for s in out]


def _keys(rng, n, kind):
    if kind == 0:
        return po.gen_uniform_u32(n, int(rng.integers(1, 1 << 30)), 1, 10000)           # the reference's distribution
    if kind == 1:
        return po.gen_uniform_u32(n, int(rng.integers(1, 1 << 30)), 0, 2**32 - 1)       # full range
    if kind == 2:
        return np.full(n, int(rng.integers(0, 2**32 - 1)), dtype=np.uint32)             # all equal
    if kind == 3:
        return rng.integers(0, 4, size=n, dtype=np.uint32) * np.uint32(0x01000000)      # only one byte varies
    if kind == 5:                                                                       # 90 % one value, the rest anything
        k = po.gen_uniform_u32(n, int(rng.integers(1, 1 << 30)), 0, 2**32 - 1)
        k[rng.random(n) < 0.9] = np.uint32(rng.integers(0, 2**32 - 1))
        return k
    if kind == 6:                                                                       # two distinct values in every byte
        return np.where(rng.integers(0, 2, size=n) == 1, np.uint32(0xFFFFFFFF), np.uint32(0))
    return np.sort(po.gen_uniform_u32(n, int(rng.integers(1, 1 << 30)), 0, max(n, 1)))  # sorted, many duplicates


def test_fuzz_scan():
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(101)
    for n in _sizes(rng, 40, 23):
        src = _keys(rng, n, int(rng.integers(0, 2))).view(np.int32)
        filt = int(rng.choice([-5, 0, 1, 5, 101, 5001, 10001, 2**31 - 1]))
        off = int(rng.integers(0, 4))  # 4-byte-granular misalignment of the source pointer
        buf = _dev(np.concatenate([np.zeros(off, dtype=np.int32), src]))
        got = ops.copy_if_lt(buf[off:], filt).cpu().numpy()
        assert np.array_equal(got, po.copy_if_lt(src, filt)), (n, filt, off)
        got = ops.copy_if_lt(buf[off:], filt, dense=True).cpu().numpy()  # the single-launch entry point
        assert np.array_equal(got, po.copy_if_lt(src, filt)), ("dense", n, filt, off)


def test_fuzz_sort():
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(102)
    for n in _sizes(rng, 40, 21):
        keys = _keys(rng, n, int(rng.integers(0, 8)))
        bits = int(rng.choice([4, 8]))
        signed = bool(rng.integers(0, 2))
        t = _dev(keys)
        ops.radix_sort_(t, signed=signed, radix_bits=bits)
        got = t.cpu().numpy()
        want = np.sort(keys.view(np.int32)) if signed else np.sort(keys).view(np.int32)
        assert np.array_equal(got, want), (n, bits, signed)


def test_fuzz_groupby():
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(103)
    for n in _sizes(rng, 30, 21):
        groups = int(rng.choice([1, 2, 3, 20, 64, 1000, 4096, 32768, 32769, 65536, 70001, 200000]))
        keys = po.gen_uniform_u32(n, int(rng.integers(1, 1 << 30)), 0, groups - 1)
        if rng.integers(0, 3) == 0:
            keys[:] = keys[0]  # one hot group
        vals = po.gen_uniform_u32(n, int(rng.integers(1, 1 << 30)), 0, int(rng.choice([1, 10000, 2**32 - 1])))
        got = ops.groupby_sum(_dev(keys), _dev(vals), groups).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, po.groupby_sum(keys, vals, groups)), (n, groups)


def test_fuzz_join():
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(104)
    for nb in _sizes(rng, 24, 20):
        npr = max(1, int(nb * rng.uniform(0.3, 2.0)))
        hi = int(rng.choice([10, 10000, max(nb - 1, 1), 2**32 - 2]))
        build = po.gen_uniform_u32(nb, int(rng.integers(1, 1 << 30)), 0, hi)
        probe = po.gen_uniform_u32(npr, int(rng.integers(1, 1 << 30)), 0, hi)
        if hi == 10 and nb > 300000:
            continue  # millions of duplicates per key: covered by the dedicated skew tests, slow to verify here
        pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in ops.hash_join(_dev(build), _dev(probe)))
        assert np.array_equal(cnt, po.join_counts_fast(build, probe).astype(np.uint32)), (nb, npr, hi)
        assert np.array_equal(np.sort(ids), np.arange(nb, dtype=np.uint32)), (nb, npr, hi)
        hit = np.flatnonzero(cnt > 0)
        if hit.size:
            for which in (0, -1):  # first and last id of every bucket carry the probe key
                at = pos[hit] + (cnt[hit] - 1 if which else 0)
                assert np.array_equal(build[ids[at]], probe[hit]), (nb, npr, hi)


def test_fuzz_joins_with_hot_keys():
    """both join paths on 2^18 .. 2^21 build rows with zero to three hot keys of 2 .. 60 % of a side, narrow and wide key
    ranges (50 distinct keys: every partition that holds rows is a giant one): counts per probe row, ids a permutation in
    which every key is one run, ranges that start and end in their key's run — against numpy"""
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(2026)
    for _ in range(10):
        nb = int(rng.integers(1 << 18, 1 << 21))
        npr = int(rng.integers(1000, 1 << 20))
        hi = int(rng.choice([nb, 10000, 2**32 - 2, 50]))
        hb = rng.integers(0, hi, nb, dtype=np.uint64).astype(np.uint32)
        hp = rng.integers(0, hi, npr, dtype=np.uint64).astype(np.uint32)
        for _k in range(int(rng.integers(0, 4))):
            key = np.uint32(rng.integers(0, hi))
            frac = rng.choice([0.02, 0.1, 0.3, 0.6])
            side = rng.integers(0, 3)
            if side in (0, 2):
                hb[rng.random(nb) < frac] = key
            if side in (1, 2):
                hp[rng.random(npr) < frac] = key
        want = po.join_counts_fast(hb, hp).astype(np.uint32)
        runs = np.unique(hb).size
        pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in ops.hash_join(_dev(hb), _dev(hp)))
        assert np.array_equal(cnt, want) and np.array_equal(np.sort(ids), np.arange(nb, dtype=np.uint32))
        in_order, hit = hb[ids], cnt > 0
        assert np.count_nonzero(in_order[1:] != in_order[:-1]) + 1 == runs
        assert np.array_equal(in_order[pos[hit]], hp[hit]) and np.array_equal(in_order[pos[hit] + cnt[hit] - 1], hp[hit])
        rid, pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in ops.radix_join(_dev(hb), _dev(hp)))
        assert np.array_equal(np.sort(rid), np.arange(npr, dtype=np.uint32)) and np.array_equal(cnt, want[rid])
        assert np.array_equal(np.sort(ids), np.arange(nb, dtype=np.uint32))
        in_order, hit = hb[ids], cnt > 0
        assert np.count_nonzero(in_order[1:] != in_order[:-1]) + 1 == runs
        assert np.array_equal(in_order[pos[hit]], hp[rid[hit]]) and np.array_equal(in_order[pos[hit] + cnt[hit] - 1], hp[rid[hit]])


def test_fuzz_ujoin():
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(105)
    for nb in _sizes(rng, 20, 20):
        npr = max(1, int(nb * rng.uniform(0.3, 2.0)))
        ak = po.gen_unique_sorted_u32(nb, int(rng.integers(1, 1 << 20)))
        bk = po.gen_unique_sorted_u32(npr, int(rng.integers(1, 1 << 20)))
        if rng.integers(0, 2):
            rng.shuffle(ak)
        av = po.gen_uniform_u32(nb, 7, 0, 2**32 - 2)
        bv = po.gen_uniform_u32(npr, 8, 0, 2**32 - 2)
        plan = ops.UniqueJoin(nb, npr)
        plan.build(_dev(ak), _dev(av))
        plan.probe(_dev(bk), _dev(bv))
        ok, o1, o2 = (t.cpu().numpy().view(np.uint32) for t in plan.result())
        ek, e1, e2 = po.ujoin(ak, av, bk, bv)
        assert np.array_equal(ok, ek) and np.array_equal(o1, e1) and np.array_equal(o2, e2), (nb, npr)


def test_fuzz_partition_and_reduce():
    from dwarf_bench_amd import ops
    from tests.pjoin_testlib import dest_of
    rng = np.random.default_rng(106)
    for n in _sizes(rng, 20, 21):
        parts = int(rng.choice([1, 2, 3, 5, 8, 16, 100, 256]))
        keys = _keys(rng, n, int(rng.integers(0, 2)))
        first = int(rng.integers(0, 2**31))
        ok, orid, cnt = ops.partition_by_hash(_dev(keys), first, parts)
        d = dest_of(keys, parts)
        c = cnt.cpu().numpy()
        assert np.array_equal(c, np.bincount(d, minlength=parts)), (n, parts)
        r2 = orid.cpu().numpy().view(np.uint32).astype(np.int64) - first
        assert np.array_equal(np.sort(r2), np.arange(n)) and np.array_equal(keys[r2], ok.cpu().numpy().view(np.uint32))
        assert np.all(np.diff(d[r2]) >= 0)  # bucket-major
        src = keys.view(np.int32)
        assert int(ops.reduce_sum(_dev(src)).cpu()[0]) == po.reduce_sum(src), n
