"""Parity of the HIP hash joins with the oracle restatements (omnisci_hashtable.hpp / join.cpp) through the C ABI.
Canonical forms per SURVEY 8(a): per-probe-row count identical, id lists identical after sorting inside a bucket;
payload join compared as a multiset of rows."""
import json

import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).cuda()


def _join(build, probe):
    from dwarf_bench_amd import ops
    pos, cnt, ids = ops.hash_join(_dev(build), _dev(probe))
    return (pos.cpu().numpy().view(np.uint32), cnt.cpu().numpy().view(np.uint32), ids.cpu().numpy().view(np.uint32))


def _check_against_bruteforce(build, probe):
    pos, cnt, ids = _join(build, probe)
    bc, off, bids = po.join_bruteforce(build, probe)
    assert np.array_equal(cnt.astype(np.uint64), bc)
    for i in range(len(probe)):
        got = np.sort(ids[pos[i]: pos[i] + cnt[i]])
        assert np.array_equal(got.astype(np.uint64), bids[int(off[i]): int(off[i + 1])]), i
    # ids is a permutation of the build rows, grouped by key
    assert np.array_equal(np.sort(ids), np.arange(len(build), dtype=np.uint32))


@pytest.mark.parametrize("n", [1, 2, 128, 256, 512, 1024, 2048, 4096])
def test_join_reference_distribution(n):
    """join/join_omnisci.cpp:53-58: both sides uniform in [1,10000] (heavy duplicates)."""
    _check_against_bruteforce(po.gen_uniform_u32(n, 42, 1, 10000), po.gen_uniform_u32(n, 43, 1, 10000))


def test_join_ragged_sizes_and_misses():
    _check_against_bruteforce(po.gen_uniform_u32(3000, 1, 1, 500), po.gen_uniform_u32(777, 2, 400, 900))
    _check_against_bruteforce(np.array([5, 5, 5, 5], np.uint32), np.array([5, 6], np.uint32))
    _check_against_bruteforce(np.array([0, 4294967294], np.uint32), np.array([4294967294, 0, 1], np.uint32))


def test_join_empty_sides():
    pos, cnt, ids = _join(np.array([], np.uint32), np.array([1, 2, 3], np.uint32))
    assert cnt.tolist() == [0, 0, 0] and pos.tolist() == [0, 0, 0]
    pos, cnt, ids = _join(np.array([1, 2, 3], np.uint32), np.array([], np.uint32))
    assert len(cnt) == 0 and sorted(ids.tolist()) == [0, 1, 2]


def test_join_vs_oracle_table_large():
    """2^20 x 2^20 with the OmniSci restatement (multi-threaded oracle) — counts + sorted buckets on a sample."""
    n = 1 << 20
    build = po.gen_uniform_u32(n, 42, 0, n - 1)
    probe = po.gen_uniform_u32(n, 43, 0, n - 1)
    pos, cnt, ids = _join(build, probe)
    opos, ocnt, oids = po.join_omnisci(build, probe, threads=8)
    assert np.array_equal(cnt.astype(np.uint64), ocnt)
    assert np.array_equal(cnt.astype(np.uint64), po.join_counts_fast(build, probe))
    for i in range(0, n, 4099):
        a = np.sort(ids[pos[i]: pos[i] + cnt[i]]).astype(np.uint64)
        b = np.sort(oids[int(opos[i]): int(opos[i] + ocnt[i])])
        assert np.array_equal(a, b)
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))
    assert np.all(build[ids[pos[cnt > 0]]] == probe[cnt > 0])


def test_join_baseline_config_2_26_properties():
    """BASELINE configs[3]: 2^26 x 2^26.  Size-independent properties, checked on the device, then the oracle at full size."""
    from dwarf_bench_amd import ops
    n = 1 << 26
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    pos, cnt, ids = ops.hash_join(build, probe)
    # ids is a permutation of the build rows
    assert int(ids.to(torch.int64).sum().item()) == n * (n - 1) // 2
    assert torch.equal(torch.sort(ids).values, torch.arange(n, dtype=torch.int32, device="cuda"))
    # every reported match really matches, on first and last id of each bucket
    hit = cnt > 0
    p = pos[hit].to(torch.int64)
    c = cnt[hit].to(torch.int64)
    assert torch.equal(build[ids[p].to(torch.int64)], probe[hit])
    assert torch.equal(build[ids[p + c - 1].to(torch.int64)], probe[hit])
    # total matches == sum over distinct keys of cnt_build * cnt_probe (independent torch computation)
    ub, cb = torch.unique(build, return_counts=True)
    up, cp = torch.unique(probe, return_counts=True)
    idx = torch.searchsorted(ub, up).clamp_(max=ub.numel() - 1)
    m = ub[idx] == up
    assert int(cnt.to(torch.int64).sum().item()) == int((cb[idx][m] * cp[m]).sum().item())
    # the oracle on the WHOLE columns: the OmniSci table restated (omnisci_hashtable.hpp:80-192, 223-261) on all host
    # cores gives every probe row's count and id list; counts compared row by row, id lists on a stride of rows
    # (sorted inside the bucket: the reference compares size + membership, join/join_omnisci.cpp:31-45)
    import os
    hb, hp = build.cpu().numpy().view(np.uint32), probe.cpu().numpy().view(np.uint32)
    opos, ocnt, oids = po.join_omnisci(hb, hp, threads=os.cpu_count() or 8)
    hcnt, hpos, hids = (t.cpu().numpy().view(np.uint32) for t in (cnt, pos, ids))
    assert np.array_equal(hcnt.astype(np.uint64), ocnt)
    for i in range(0, n, 65537):
        a = np.sort(hids[hpos[i]: hpos[i] + hcnt[i]]).astype(np.uint64)
        assert np.array_equal(a, np.sort(oids[int(opos[i]): int(opos[i] + ocnt[i])])), i


def test_join_2_27_rows_skewed_keys_through_the_packed_histogram():
    """2^26 < n <= 2^27 build rows (the shard size of every rank of the 8-GPU join): both partition levels' histograms come
    from one read of the keys with two 16-bit counters per LDS word.  Half of the build rows carry ONE key, so a
    workgroup's share of that key's partition (2^18 rows) wraps its 16-bit counter several times: the carries are
    settled in the global accumulators.  Counts per probe row against numpy, ids a permutation that really matches."""
    from dwarf_bench_amd import ops
    n, m = (1 << 26) + 12345, 1 << 20
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    build[::2] = 123456789  # every other row
    probe = ops.gen_uniform_u32(m, 43, 0, n - 1)
    probe[::1000] = 123456789
    pos, cnt, ids = ops.hash_join(build, probe)
    hb, hp = build.cpu().numpy().view(np.uint32), probe.cpu().numpy().view(np.uint32)
    assert np.array_equal(cnt.cpu().numpy().view(np.uint32).astype(np.uint64), po.join_counts_fast(hb, hp))
    assert int(ids.to(torch.int64).sum().item()) == n * (n - 1) // 2  # a permutation of the build rows (sum of 0..n-1)
    hit = cnt > 0
    p, c = pos[hit].to(torch.int64), cnt[hit].to(torch.int64)
    assert torch.equal(build[ids[p].to(torch.int64)], probe[hit])
    assert torch.equal(build[ids[p + c - 1].to(torch.int64)], probe[hit])


def _fmix32(h):
    h = h.astype(np.uint64)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def _keys_of_partition(n_build, partition, how_many):
    """distinct keys that the build of n_build rows puts into one partition (join_lds.hip jl_pid: the high bits of
    fmix32(key) * parts; parts as join_common.hpp jl_layout has them)"""
    want = min(max(1, -(-n_build // 2048)), 1 << 20)
    lg = want.bit_length() - 1  # floor(log2(want))
    k2 = 1 if want <= 1024 else 1 << (lg // 2)
    parts = -(-want // k2) * k2
    cand = np.arange(1, 1 + how_many * parts * 2, dtype=np.uint64)
    mine = cand[(_fmix32(cand) * parts) >> 32 == partition][:how_many]
    assert mine.size == how_many
    return mine.astype(np.uint32)


def _check_grouped_join(build, probe):
    """counts per probe row against numpy; ids a permutation of the build rows in which every key's rows are ONE run;
    every hit's range starts and ends inside its key's run (with the count right, the range IS the run)"""
    from dwarf_bench_amd import ops
    plan = ops.HashJoin(len(build), len(probe))
    plan.build(_dev(build))
    plan.probe(_dev(probe))
    pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in plan.result())
    assert np.array_equal(cnt.astype(np.uint64), po.join_counts_fast(build, probe))
    assert np.array_equal(np.sort(ids), np.arange(len(build), dtype=np.uint32))
    in_order = build[ids]
    assert np.count_nonzero(in_order[1:] != in_order[:-1]) + 1 == np.unique(build).size
    hit = cnt > 0
    assert np.array_equal(in_order[pos[hit]], probe[hit])
    assert np.array_equal(in_order[pos[hit] + cnt[hit] - 1], probe[hit])


@pytest.mark.parametrize("n", [(1 << 18) + 5, 1 << 22])  # one and two scatter levels
@pytest.mark.parametrize("kind", ["every row one key", "every other row one key", "three hot keys",
                                  "1500 keys of one partition, 40 rows each", "two giants and 1200 keys of a third"])
def test_join_build_with_giant_partitions(kind, n):
    """A partition far above its expected 2048 rows (hot keys) is counted and filled by all workgroups together
    (join_lds.hip jl_giant_count / jl_giant_ids; from 2^18 build rows, giants above max(32768, n / 1024) rows)."""
    rng = np.random.default_rng(23)
    build = po.gen_uniform_u32(n, 42, 0, n - 1)
    probe = po.gen_uniform_u32(1 << 18, 43, 0, n - 1)
    if kind == "every row one key":
        build[:] = 777
    elif kind == "every other row one key":
        build[::2] = 123456789
    elif kind == "three hot keys":
        r = rng.random(n)
        build[r < 0.3] = 5
        build[(r >= 0.3) & (r < 0.6)] = 4000000000
        build[(r >= 0.6) & (r < 0.9)] = 99
    elif kind == "1500 keys of one partition, 40 rows each":  # (the partition's own ~1300 distinct keys come on top)
        build[: 1500 * 40] = np.repeat(_keys_of_partition(n, 3, 1500), 40)
        build = rng.permutation(build)
    else:
        build[: 1200 * 30] = np.repeat(_keys_of_partition(n, 0, 1200), 30)
        build[100000:150000] = 31
        build[150000:230000:2] = 32
        build = rng.permutation(build)
    probe[::7] = build[rng.integers(0, n, probe[::7].size)]  # probe rows that hit, hot keys among them
    _check_grouped_join(build, probe)


def test_many_giant_partitions_of_many_slices_each_count_exactly():
    """Sixty hot keys of 40000..90000 rows each in a 2^22-row build: dozens of giant partitions, five to eleven 8192-row
    slices each, whose per-key counts are added slice by slice with relaxed memory-side atomics and turned into positions
    by whichever workgroup finishes a giant's last slice (jl_giant_count).  Every hot key's count and id range is checked
    exactly, so a lost or doubled slice shows as a wrong count, not as a rare mis-join."""
    rng = np.random.default_rng(41)
    n = 1 << 22
    build = po.gen_uniform_u32(n, 42, 0, n - 1)
    hot = rng.choice(np.arange(1, 1 << 30, dtype=np.uint32), 60, replace=False)
    sizes = rng.integers(40000, 90000, hot.size)
    at = 0
    for k, c in zip(hot, sizes):
        build[at: at + c] = k
        at += c
    build = rng.permutation(build)
    probe = np.concatenate([hot, po.gen_uniform_u32(1 << 16, 43, 0, n - 1)]).astype(np.uint32)
    _check_grouped_join(build, probe)
    # and through the radix join (the giants' scratch sub-tables)
    from dwarf_bench_amd import ops
    rid, pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in ops.radix_join(_dev(build), _dev(probe)))
    assert np.array_equal(cnt, po.join_counts_fast(build, probe).astype(np.uint32)[rid])
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))
    in_order = build[ids]
    hit = cnt > 0
    assert np.array_equal(in_order[pos[hit]], probe[rid[hit]]) and np.array_equal(in_order[pos[hit] + cnt[hit] - 1], probe[rid[hit]])


def test_the_no_giants_escape_hatch_still_joins_everything():
    """DBHIP_JL_NO_GIANTS=1 (every partition through the per-partition kernel whatever its size; read once
    per process, hence the child process): hot keys are then walked by one workgroup each — slowly, correctly — and a
    partition with more distinct keys than slots still reaches the spill path, whose launch no longer has a giants' list
    to look at."""
    import os, subprocess, sys
    prog = (
        "import numpy as np, torch\n"
        "import tests.test_gpu_join as t\n"
        "from oracle import pyoracle as po\n"
        "rng = np.random.default_rng(3)\n"
        "n = 1 << 18\n"
        "b = po.gen_uniform_u32(n, 42, 0, n - 1); b[::2] = 777\n"
        "p = po.gen_uniform_u32(1 << 16, 43, 0, n - 1); p[::5] = 777\n"
        "t._check_grouped_join(b, p)\n"
        "b = po.gen_uniform_u32(n, 42, 0, n - 1); mine = t._keys_of_partition(n, 1, 3500)\n"
        "b[: 3500 * 20] = np.repeat(mine, 20); b = rng.permutation(b); p[::3] = mine[rng.integers(0, 3500, p[::3].size)]\n"
        "t._check_grouped_join(b, p)\n"
        "print('no giants ok')\n")
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "DBHIP_JL_NO_GIANTS": "1"}, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert r.returncode == 0 and "no giants ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def _keys_of_partition_of(parts, partition, how_many):
    cand = np.arange(1, 1 + how_many * parts * 2, dtype=np.uint64)
    mine = cand[(_fmix32(cand) * parts) >> 32 == partition][:how_many]
    assert mine.size == how_many
    return mine.astype(np.uint32)


def _radix_parts(n_build):
    """partitions of the radix join (join_common.hpp jl_layout with kJrRowsPerPart = 1792 rows per partition)"""
    want = min(max(1, -(-n_build // 1792)), 1 << 20)
    lg = (want - 1).bit_length()
    if want <= 1024:
        k2 = 1
    else:
        lgs = lg - 1 if (1 << lg) != want else lg
        k2 = 1 << (lgs // 2)
    k1 = -(-want // k2)
    while k1 > 1024:
        k2 *= 2
        k1 = -(-want // k2)
    return k1 * k2


SPILL_SHAPES = [  # (build rows, distinct keys put into ONE partition, rows per key): more keys than the 3072 slots of a sub-table
    (1 << 16, 3500, 1),    # below 2^18 rows the build kernel's workgroup builds the spill table itself
    (1 << 16, 4000, 9),
    (1 << 18, 3500, 1),    # listed by the build kernel, built at the end of jl_giant_ids
    (1 << 18, 3500, 20),   # 70000 rows: a giant partition — jl_giant_count finds its sub-table full and lists it
    (1 << 20, 9000, 5),    # two scatter levels; a giant whose single slices overflow the LDS sub-table
    (1 << 18, 40000, 1),   # a partition of 40000 distinct keys
]


@pytest.mark.parametrize("n,keys,per_key", SPILL_SHAPES)
def test_a_partition_with_more_distinct_keys_than_slots_joins_like_any_other(n, keys, per_key):
    """The reference's table takes any keys (ht_size = 2 * distinct, join/join_omnisci.cpp:69-70; linear probing until a
    slot is found, omnisci_hashtable.hpp:80-108).  A partition of the LDS join holds at most 3072 DISTINCT keys in its
    sub-table; keys constructed against the partition hash go beyond that: such a partition is built in an
    open-addressing table of its own in HBM (jl_spill_partition) and the join's results are what they are for any other
    input — counts per probe row, ids grouped by key, status 0 (until round 4 this raised DBHIP_DEV_TABLE_FULL)."""
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(31)
    build = po.gen_uniform_u32(n, 42, 0, n - 1)
    mine = _keys_of_partition(n, 1, keys)
    build[: keys * per_key] = np.repeat(mine, per_key)
    build = rng.permutation(build)
    probe = po.gen_uniform_u32(1 << 16, 43, 0, n - 1)
    probe[::3] = mine[rng.integers(0, keys, probe[::3].size)]          # hits in the spilled partition
    probe[1::11] = _keys_of_partition(n, 1, keys + 500)[keys:][rng.integers(0, 500, probe[1::11].size)]  # and misses there
    _check_grouped_join(build, probe)
    plan = ops.HashJoin(n, 16)
    plan.build(_dev(build))
    plan.probe(_dev(build[:16]))
    assert ops.workspace_status(plan.ws) == 0
    plan.build(_dev(po.gen_uniform_u32(n, 44, 0, n - 1)))  # the same plan on ordinary keys afterwards: nothing left behind
    plan.probe(_dev(build[:16]))
    assert ops.workspace_status(plan.ws) == 0


@pytest.mark.parametrize("n,keys,per_key", SPILL_SHAPES)
def test_radix_join_of_a_partition_with_more_distinct_keys_than_slots(n, keys, per_key):
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(37)
    parts = _radix_parts(n)
    ha = po.gen_uniform_u32(n, 42, 0, n - 1)
    mine = _keys_of_partition_of(parts, parts // 3, keys + 300)
    ha[: keys * per_key] = np.repeat(mine[:keys], per_key)
    ha = rng.permutation(ha)
    npr = (1 << 17) + 77
    hb = po.gen_uniform_u32(npr, 43, 0, n - 1)
    hb[::3] = mine[rng.integers(0, keys + 300, hb[::3].size)]  # hits and misses inside the spilled partition
    rid, pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in ops.radix_join(_dev(ha), _dev(hb)))
    assert np.array_equal(np.sort(rid), np.arange(npr, dtype=np.uint32))
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))
    assert np.array_equal(cnt, po.join_counts_fast(ha, hb).astype(np.uint32)[rid])
    in_order = ha[ids]
    assert np.count_nonzero(in_order[1:] != in_order[:-1]) + 1 == np.unique(ha).size  # every key's ids are one run
    hit = cnt > 0
    assert np.array_equal(in_order[pos[hit]], hb[rid[hit]])
    assert np.array_equal(in_order[pos[hit] + cnt[hit] - 1], hb[rid[hit]])


def test_join_a_few_rows_above_2_27_through_the_packed_histogram():
    """what a rank of the 8-GPU join of 2^30 rows receives: 2^27 rows and a few thousand — 65792 partitions, still one
    read of the keys for both histograms (two 16-bit counters per LDS word, up to 81920 partitions).  Counts per probe
    row against a binary search in the sorted build column, ids a permutation whose ranges hold the probe's key."""
    from dwarf_bench_amd import ops
    n, m = (1 << 27) + 4097, 1 << 20
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(m, 43, 0, n - 1)
    pos, cnt, ids = ops.hash_join(build, probe)
    sb = torch.sort(build.to(torch.int64) & 0xFFFFFFFF).values
    pk = probe.to(torch.int64) & 0xFFFFFFFF
    want = torch.searchsorted(sb, pk, right=True) - torch.searchsorted(sb, pk, right=False)
    del sb
    assert torch.equal(cnt.to(torch.int64), want)
    assert int(ids.to(torch.int64).sum().item()) == n * (n - 1) // 2
    hit = cnt > 0
    p, c = pos[hit].to(torch.int64), cnt[hit].to(torch.int64)
    assert torch.equal(build[ids[p].to(torch.int64)], probe[hit])
    assert torch.equal(build[ids[p + c - 1].to(torch.int64)], probe[hit])


def test_ujoin_reference_fixture_shape(golden_dir):
    """unique-key payload join vs seq_join (join_helpers.hpp:86-104) as a multiset of rows (join.cpp:133)."""
    from dwarf_bench_amd import ops
    for n in (1, 128, 1024, 4096, 50000):
        ak, bk = po.gen_unique_sorted_u32(n, 11), po.gen_unique_sorted_u32(n, 12)
        av, bv = po.gen_uniform_u32(n, 13, 0, 10**6), po.gen_uniform_u32(n, 14, 0, 10**6)
        plan = ops.UniqueJoin(n, n)
        plan.build(_dev(ak), _dev(av))
        plan.probe(_dev(bk), _dev(bv))
        ok, o1, o2 = (t.cpu().numpy().view(np.uint32) for t in plan.result())
        ek, e1, e2 = po.ujoin(ak, av, bk, bv)
        assert np.array_equal(ok, ek) and np.array_equal(o1, e1) and np.array_equal(o2, e2)
        if n <= 4096:
            hit = ok != 0xFFFFFFFF
            sk, s1, s2 = po.seq_join(ak, av, bk, bv)
            assert sorted(zip(ok[hit].tolist(), o1[hit].tolist(), o2[hit].tolist())) == \
                sorted(zip(sk.tolist(), s1.tolist(), s2.tolist()))


@pytest.mark.parametrize("n", [1 << 16, 100003, 1 << 20, (1 << 22) + 77])
def test_ujoin_partitioned_path_matches_oracle(n):
    """n >= 2^16: radix-partitioned build with LDS sub-tables, one 8-byte gather per probe row (join_lds.hip);
    one and two partition levels, shuffled (unsorted) build keys, a probe side of a different size"""
    from dwarf_bench_amd import ops
    ak = po.gen_unique_sorted_u32(n, 11)
    np.random.default_rng(3).shuffle(ak)
    m = n // 2 + 13
    bk = po.gen_unique_sorted_u32(m, 12)
    av, bv = po.gen_uniform_u32(n, 13, 0, 2**32 - 2), po.gen_uniform_u32(m, 14, 0, 2**32 - 2)
    plan = ops.UniqueJoin(n, m)
    plan.build(_dev(ak), _dev(av))
    plan.probe(_dev(bk), _dev(bv))
    ok, o1, o2 = (t.cpu().numpy().view(np.uint32) for t in plan.result())
    ek, e1, e2 = po.ujoin(ak, av, bk, bv)
    assert np.array_equal(ok, ek) and np.array_equal(o1, e1) and np.array_equal(o2, e2)
    assert 0 < int((ok != 0xFFFFFFFF).sum()) < m  # hits and misses both present


@pytest.mark.parametrize("n,crowd", [(1 << 16, 3300), (1 << 18, 9000), (1 << 22, 12000)])
def test_ujoin_of_a_partition_with_more_keys_than_slots(n, crowd):
    """Unique build keys of which `crowd` are constructed to fall into ONE partition of the LDS-partitioned build (more than
    the 3072 slots of its sub-table): that partition is built in the spill pool and probed through the directory — the
    join's rows are what seq_join gives for any other keys (until round 4: DBHIP_DEV_TABLE_FULL).  Probe: hits and
    misses inside the spilled partition and elsewhere."""
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(43)
    mine = _keys_of_partition(n, 2, crowd + 400)
    rest = np.setdiff1d(po.gen_unique_sorted_u32(n, 11), mine)[: n - crowd]
    ak = np.concatenate([mine[:crowd], rest]).astype(np.uint32)
    assert np.unique(ak).size == n
    rng.shuffle(ak)
    m = n // 2 + 13
    bk = np.unique(np.concatenate([mine[rng.integers(0, crowd + 400, m // 3)], po.gen_unique_sorted_u32(m, 12)]))[:m].astype(np.uint32)
    rng.shuffle(bk)
    av, bv = po.gen_uniform_u32(n, 13, 0, 2**32 - 2), po.gen_uniform_u32(bk.size, 14, 0, 2**32 - 2)
    plan = ops.UniqueJoin(n, bk.size)
    plan.build(_dev(ak), _dev(av))
    plan.probe(_dev(bk), _dev(bv))
    ok, o1, o2 = (t.cpu().numpy().view(np.uint32) for t in plan.result())  # raises on a non-zero status
    ek, e1, e2 = po.ujoin(ak, av, bk, bv)
    assert np.array_equal(ok, ek) and np.array_equal(o1, e1) and np.array_equal(o2, e2)
    in_crowd = np.isin(bk, mine[:crowd])
    assert int((ok[in_crowd] != 0xFFFFFFFF).sum()) == int(in_crowd.sum()) > 0  # every probe of a spilled key hits
    assert int((ok[np.isin(bk, mine[crowd:])] != 0xFFFFFFFF).sum()) == 0          # its neighbours in the partition miss


def test_ujoin_baseline_size_properties():
    """2^26 x 2^26 unique keys in [0, 10n): every hit row carries the build payload of ITS key (payload = f(key)),
    the number of hits equals the size of the key-set intersection (torch as an independent cross-check)"""
    from dwarf_bench_amd import ops
    n = 1 << 26
    ak = ops.gen_unique_sorted_u32(n, 11)
    bk = ops.gen_unique_sorted_u32(n, 12)
    av = ak ^ 0x5A5A5A5A  # payload determined by the key
    plan = ops.UniqueJoin(n, n)
    plan.build(ak, av)
    plan.probe(bk, bk)
    ok, o1, o2 = plan.result()
    hit = ok != -1
    assert torch.equal(ok[hit], bk[hit]) and torch.equal(o1[hit], bk[hit] ^ 0x5A5A5A5A) and torch.equal(o2[hit], bk[hit])
    assert bool((o1[~hit] == -1).all()) and bool((o2[~hit] == -1).all())
    pos = torch.searchsorted(ak.to(torch.int64) & 0xFFFFFFFF, bk.to(torch.int64) & 0xFFFFFFFF).clamp_(max=n - 1)
    assert int(hit.sum()) == int((ak[pos] == bk).sum())


def test_sentinel_key_is_flagged_not_joined():
    """0xFFFFFFFF is the empty-slot marker (join/join_omnisci.cpp:52): as a build key it raises DEV_KEY_RANGE and
    is dropped (the status word marks the result as invalid), as a probe key it finds nothing and nothing is flagged"""
    from dwarf_bench_amd import _capi, ops
    n = 1 << 17
    a = ops.gen_uniform_u32(n, 5, 0, n - 1)
    b = ops.gen_uniform_u32(n, 6, 0, n - 1)
    b[7] = -1
    b[n - 3] = -1
    pos, cnt, ids = ops.hash_join(a, b)  # sentinel on the probe side only: a clean status
    exp = po.join_counts_fast(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32))
    assert int(cnt[7]) == 0 and int(cnt[n - 3]) == 0 and np.array_equal(cnt.cpu().numpy().view(np.uint32), exp.astype(np.uint32))
    a2 = a.clone()
    a2[11] = -1
    plan = ops.HashJoin(n, n)
    plan.build(a2)
    plan.probe(b)
    assert ops.workspace_status(plan.ws) == ops.DEV_KEY_RANGE
    with pytest.raises(_capi.DbhipError):
        plan.result()
    for nn in (1000, 1 << 17):  # unique-key join: small CAS table and the partitioned path
        ak = ops.gen_unique_sorted_u32(nn, 11)
        ak[5] = -1
        uj = ops.UniqueJoin(nn, nn)
        uj.build(ak, ak)
        assert ops.workspace_status(uj.ws) == ops.DEV_KEY_RANGE


def test_join_answers_records():
    """JoinOneToMany {pointer into the id buffer, size} (common/dpcpp/omnisci_hashtable.hpp:12-17)"""
    from dwarf_bench_amd import ops
    n = 5000
    a = ops.gen_uniform_u32(n, 3, 1, 1000)
    b = ops.gen_uniform_u32(n, 4, 1, 1200)
    pos, cnt, ids = ops.hash_join(a, b)
    ans = ops.join_answers(ids, pos, cnt).cpu().numpy()
    assert np.array_equal(ans[:, 1], cnt.cpu().numpy().astype(np.int64))
    assert np.array_equal(ans[:, 0], ids.data_ptr() + 4 * pos.cpu().numpy().astype(np.int64))


# ---- radix join (dbhip_join_radix_*): same semantics, results in the probe side's partition order -------------------
@pytest.mark.parametrize("nb,npr,hi", [(0, 10, 5), (10, 0, 5), (1, 1, 1), (1000, 777, 300), (4096, 4096, 10000),
                                       (100003, 65537, 5000), (1 << 17, 1 << 17, (1 << 17) - 1), (300007, 1 << 18, 2**32 - 2),
                                       (1 << 20, 1 << 19, 10000)])
def test_radix_join_matches_oracle(nb, npr, hi):
    """per probe row (found through its row id): count == the key's multiplicity, ids = exactly the build rows with
    the key (join/join_omnisci.cpp:31-45); every probe row appears once; ids are a permutation of the build rows"""
    from dwarf_bench_amd import ops
    a = ops.gen_uniform_u32(nb, 42, 0 if hi == 2**32 - 2 else 1, hi)
    b = ops.gen_uniform_u32(npr, 43, 0 if hi == 2**32 - 2 else 1, hi)
    rid, pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in ops.radix_join(a, b))
    ha, hb = a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)
    assert np.array_equal(np.sort(rid), np.arange(npr, dtype=np.uint32))
    assert np.array_equal(np.sort(ids), np.arange(nb, dtype=np.uint32))
    want = po.join_counts_fast(ha, hb).astype(np.uint32) if nb and npr else np.zeros(npr, dtype=np.uint32)
    assert np.array_equal(cnt, want[rid])
    hit = cnt > 0
    assert np.all(ha[ids[pos[hit]]] == hb[rid[hit]]) and np.all(ha[ids[pos[hit] + cnt[hit] - 1]] == hb[rid[hit]])
    if nb <= 4096 and nb and npr:  # every id of every row
        bc, off, bids = po.join_bruteforce(ha, hb)
        for i in range(0, npr, 7):
            r = int(rid[i])
            assert np.array_equal(np.sort(ids[pos[i]: pos[i] + cnt[i]]).astype(np.uint64), bids[int(off[r]): int(off[r + 1])])


@pytest.mark.parametrize("nb,npr", [(1 << 18, 1 << 18), (1 << 22, (1 << 21) + 9), (1 << 16, 1 << 20), (1 << 20, 3000)])
@pytest.mark.parametrize("kind", ["hot build key", "hot probe key", "the same key hot on both sides", "hot keys that differ",
                                  "every row of both sides one key"])
def test_radix_join_with_giant_partitions(kind, nb, npr):
    """partitions whose build OR probe side is far above the expected size are left out by the fused kernel and done by
    all workgroups together: scratch sub-tables (jl_giant_count / jl_giant_ids), their probe rows in jl_giant_ids' second half"""
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(29)
    ha = po.gen_uniform_u32(nb, 42, 0, nb - 1)
    hb = po.gen_uniform_u32(npr, 43, 0, nb - 1)
    if kind in ("hot build key", "the same key hot on both sides", "hot keys that differ"):
        ha[rng.random(nb) < 0.5] = 4242
    if kind in ("hot probe key", "the same key hot on both sides"):
        hb[rng.random(npr) < 0.5] = 4242
    if kind == "hot keys that differ":
        hb[rng.random(npr) < 0.5] = 17
    if kind == "every row of both sides one key":
        if nb * npr > 1 << 41:
            pytest.skip("2^42 matches and more: nothing to learn")
        ha[:] = 9
        hb[:] = 9
    rid, pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in ops.radix_join(_dev(ha), _dev(hb)))
    assert np.array_equal(np.sort(rid), np.arange(npr, dtype=np.uint32))
    assert np.array_equal(np.sort(ids), np.arange(nb, dtype=np.uint32))
    assert np.array_equal(cnt, po.join_counts_fast(ha, hb).astype(np.uint32)[rid])
    in_order = ha[ids]
    assert np.count_nonzero(in_order[1:] != in_order[:-1]) + 1 == np.unique(ha).size  # every key's ids are one run
    hit = cnt > 0
    assert np.array_equal(in_order[pos[hit]], hb[rid[hit]])
    assert np.array_equal(in_order[pos[hit] + cnt[hit] - 1], hb[rid[hit]])


def test_radix_join_carries_caller_row_ids_and_agrees_with_the_probe_path():
    from dwarf_bench_amd import ops
    n, first = 1 << 18, 3 << 20
    a = ops.gen_uniform_u32(n, 42, 0, n - 1, first_index=first)
    b = ops.gen_uniform_u32(n + 5, 43, 0, n - 1, first_index=first)
    ar = torch.arange(first, first + n, device="cuda", dtype=torch.int64).to(torch.int32)
    br = torch.arange(first, first + n + 5, device="cuda", dtype=torch.int64).to(torch.int32)
    rid, pos, cnt, ids = ops.radix_join(a, b, ar, br)
    srt = a.clone()
    ops.radix_sort_(srt)
    # the device-side validator regenerates the key of every global id
    assert ops.check_gen_uniform(b[(rid.to(torch.int64) - first)], 43, 0, n - 1, indices=rid) == 0
    assert ops.check_join(srt, b[(rid.to(torch.int64) - first)].contiguous(), pos, cnt, ids, build_keys=None, gen=(42, 0, n - 1))[0] == 0
    p2, c2, _ = ops.hash_join(a, b)
    assert torch.equal(c2[(rid.to(torch.int64) - first)], cnt)


@pytest.mark.parametrize("n", [160_000_001, (1 << 28) + 12345])
def test_radix_join_whose_level_1_histogram_reads_the_digit_column(n):
    """a build side of more than 1.47e8 rows has more partitions than the fused histograms count (81920): the level-0
    scatter then writes every row's level-1 bucket as a 16-bit column and the level-1 histogram reads that instead of
    the pairs (jl_hist1d_kernel) — with 4096-row tiles (349 x 256 buckets) at the first size, 16384-row tiles
    (586 x 256) at the second.  The probe side is partitioned by the same geometry whatever its size — a few thousand
    rows per level-0 bucket here, so the ragged ends of every bucket's range count.  Checked against torch: the count
    of every probe row, the id buffer as a permutation, first and last id of every hit."""
    from dwarf_bench_amd import ops
    m = (1 << 22) + 77
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(m, 43, 0, n - 1)
    rid, pos, cnt, ids = ops.radix_join(build, probe)
    per_key = torch.bincount(build.to(torch.int64), minlength=n)
    rid64, pos64, cnt64 = rid.to(torch.int64), pos.to(torch.int64) & 0xFFFFFFFF, cnt.to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(torch.sort(rid64).values, torch.arange(m, device=rid.device))
    keys_of_rows = probe.to(torch.int64)[rid64]
    assert torch.equal(cnt64, per_key[keys_of_rows])
    del per_key
    ids64 = ids.to(torch.int64) & 0xFFFFFFFF
    assert int(ids64.sum()) == n * (n - 1) // 2 and int(ids64.max()) == n - 1
    hit = cnt64 > 0
    b64 = build.to(torch.int64)
    assert torch.equal(b64[ids64[pos64[hit]]], keys_of_rows[hit])
    assert torch.equal(b64[ids64[pos64[hit] + cnt64[hit] - 1]], keys_of_rows[hit])
    del rid, pos, cnt, ids, rid64, pos64, cnt64, ids64, keys_of_rows, hit
    torch.cuda.empty_cache()
    # the row-ordered join partitions its build side through the same code (2048 rows per partition)
    pos, cnt, ids = ops.hash_join(build, probe)
    per_key = torch.bincount(b64, minlength=n)
    p64 = probe.to(torch.int64)
    cnt64, pos64, ids64 = cnt.to(torch.int64) & 0xFFFFFFFF, pos.to(torch.int64) & 0xFFFFFFFF, ids.to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(cnt64, per_key[p64])
    assert int(ids64.sum()) == n * (n - 1) // 2
    hit = cnt64 > 0
    assert torch.equal(b64[ids64[pos64[hit]]], p64[hit])
    assert torch.equal(b64[ids64[pos64[hit] + cnt64[hit] - 1]], p64[hit])
