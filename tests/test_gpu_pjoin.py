"""Device pieces of the partitioned join through the C ABI, and the whole path with 2 ranks sharing the one GPU
of the test box (gloo moves the buckets through the host there; on a multi-GPU node the same code runs over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parts", [1, 2, 3, 8, 256])
@pytest.mark.parametrize("n", [0, 1, 5, 4095, 4096, 4097, 100003, 1 << 20])
def test_partition_matches_hash_restatement(n, parts):
    from dwarf_bench_amd import ops
    from tests.pjoin_testlib import dest_of
    keys = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    first = 1000003
    ok, orid, cnt = ops.partition_by_hash(keys, first, parts)
    h = keys.cpu().numpy().view(np.uint32)
    d = dest_of(h, parts)
    counts = cnt.cpu().numpy()
    assert np.array_equal(counts, np.bincount(d, minlength=parts))
    k2 = ok.cpu().numpy().view(np.uint32)
    r2 = orid.cpu().numpy().view(np.uint32)
    # every pair is (key, global row id of that key), buckets are contiguous and hold exactly their rows
    assert np.array_equal(h[r2.astype(np.int64) - first], k2)
    bounds = np.concatenate([[0], np.cumsum(counts)])
    for b in range(parts):
        seg = r2[bounds[b]: bounds[b + 1]].astype(np.int64) - first
        assert np.all(d[seg] == b)
    assert np.array_equal(np.sort(r2), np.arange(first, first + n, dtype=np.uint32))


def test_gather():
    from dwarf_bench_amd import ops
    table = ops.gen_uniform_u32(100000, 1, 0, 2**32 - 1)
    idx = ops.gen_uniform_u32(333333, 2, 0, 99999)
    got = ops.gather_u32(table, idx).cpu().numpy()
    assert np.array_equal(got, table.cpu().numpy()[idx.cpu().numpy()])


def test_bucket_balance_on_uniform_keys():
    from dwarf_bench_amd import ops
    n = 1 << 24
    keys = ops.gen_uniform_u32(n, 42, 0, n - 1)
    _, _, cnt = ops.partition_by_hash(keys, 0, 8)
    c = cnt.cpu().numpy()
    assert c.sum() == n and c.max() / c.mean() < 1.01


@pytest.mark.parametrize("parts", [2, 8])
def test_rank_hash_is_independent_of_the_local_partition_hash(parts):
    """A rank joins only keys of ONE rank bucket.  If the rank hash were the same high hash bits the local build
    partitions by, those keys would fill 1/parts of its LDS sub-tables parts-fold (TABLE_FULL at 8 ranks)."""
    from dwarf_bench_amd import ops
    n = 1 << 22
    keys = ops.gen_uniform_u32(n, 42, 0, n - 1)
    ok, orid, cnt = ops.partition_by_hash(keys, 0, parts)
    c = cnt.cpu().numpy()
    for b in (0, parts - 1):
        lo = int(c[:b].sum())
        mine = ok[lo: lo + int(c[b])].contiguous()
        rids = orid[lo: lo + int(c[b])].contiguous()
        assert mine.numel() >= 1 << 16  # the LDS-partitioned build path
        plan = ops.HashJoin(mine.numel(), mine.numel())
        plan.build(mine, rids)
        plan.probe(mine)
        pos, cnt_out, ids = plan.result()  # raises on TABLE_FULL
        h = mine.cpu().numpy().view(np.uint32)
        assert np.array_equal(cnt_out.cpu().numpy().view(np.uint32), po.join_counts_fast(h, h).astype(np.uint32))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, key_hi, q, hot=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dwarf_bench_amd import ops, pjoin
        per = n // world
        lo, hi = rank * per, (n if rank == world - 1 else (rank + 1) * per)
        build = ops.gen_uniform_u32(hi - lo, 42, 1, key_hi, first_index=lo)
        probe = ops.gen_uniform_u32(hi - lo, 43, 1, key_hi, first_index=lo)
        if hot:  # (shards start at even rows: the same rows as [::2] / [::128] of the whole columns)
            build[::2] = 77
            probe[::128] = 77
        res = pjoin.partitioned_join(build, probe, lo, lo)  # HipBackend
        u = lambda t: t.cpu().numpy().view(np.uint32).copy()
        q.put((rank, u(res.probe_row_ids), u(res.pos), u(res.cnt), u(res.build_row_ids)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,key_hi", [(2, 3000, 2000), (2, 50001, 10000), (3, 400003, 2**32 - 2), (4, 1 << 20, (1 << 20) - 1)])
def test_ranks_sharing_one_gpu(world, n, key_hi):
    """the whole partitioned join (HIP backend) with several ranks on the box's one GPU; 3 and 4 ranks, ragged shards,
    and shards large enough for the LDS-partitioned local join"""
    import torch.multiprocessing as mp
    from tests.pjoin_testlib import check_global
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, key_hi, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    build_all = po.gen_uniform_u32(n, 42, 1, key_hi)
    probe_all = po.gen_uniform_u32(n, 43, 1, key_hi)
    if n <= 5000:
        check_global([(o[1], o[2], o[3], o[4]) for o in outs], build_all, probe_all)
    else:  # counts by global row id + total
        want = po.join_counts_fast(build_all, probe_all)
        got = np.zeros(n, dtype=np.uint64)
        for _, rid, pos, cnt, ids in outs:
            got[rid] = cnt
            hit = cnt > 0
            assert np.all(build_all[ids[pos[hit]]] == probe_all[rid[hit]])
        assert np.array_equal(got, want)


def test_ranks_sharing_one_gpu_with_a_hot_key():
    """every other build row carries one key: all of them travel to ONE rank, whose local radix join finds them in one
    giant partition (join_lds.hip jl_giant_*); counts per global probe row and the ids' ends against numpy"""
    import torch.multiprocessing as mp
    world, n, key_hi = 2, 1 << 21, (1 << 21) - 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, key_hi, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    build_all = po.gen_uniform_u32(n, 42, 1, key_hi)
    probe_all = po.gen_uniform_u32(n, 43, 1, key_hi)
    build_all[::2] = 77
    probe_all[::128] = 77
    want = po.join_counts_fast(build_all, probe_all)
    got = np.zeros(n, dtype=np.uint64)
    rows = 0
    for _, rid, pos, cnt, ids in outs:
        got[rid] = cnt
        rows += rid.size
        hit = cnt > 0
        assert np.all(build_all[ids[pos[hit]]] == probe_all[rid[hit]])
        assert np.all(build_all[ids[pos[hit] + cnt[hit] - 1]] == probe_all[rid[hit]])
    assert rows == n and np.array_equal(got, want)
    assert max(o[4].size for o in outs) > n // 2  # the hot key's rows all went to one rank


@pytest.mark.parametrize("n,direct", [(1000, False), (1000, True), (300007, False), (1 << 22, True), (1 << 22, False)])
def test_native_engine_single_rank(n, direct):
    """the C++ engine (pjoin_engine.cpp) through its C entry points, one rank: the pipelined path with a self
    exchange, and the direct local join; its device-side checks and the match count against the oracle"""
    from dwarf_bench_amd import pjoin_native
    eng = pjoin_native.NativePartitionedJoin(n, rank=0, world=1, device=0, direct_single=direct)
    for _ in range(2):
        t = eng.step()
        assert t["total_us"] > 0 and t["build_us"] > 0 and t["probe_us"] > 0
    chk = eng.check()
    eng.close()
    build = po.gen_uniform_u32(n, 42, 0, n - 1)
    probe = po.gen_uniform_u32(n, 43, 0, n - 1)
    assert chk["bad_pairs"] == chk["bad_route"] == chk["bad_rows"] == 0 and chk["conserved"]
    assert chk["recv_build"] == n and chk["recv_probe"] == n
    assert chk["matches"] == int(po.join_counts_fast(build, probe).astype(np.uint64).sum())


def test_native_engine_multi_process_path_with_one_rank():
    """ncclCommInitRank + ncclAllGather of the counts + the send/recv group + the ncclAllReduce of the conservation
    sums: the code a rank of `bench.py --gpus N` runs, rehearsed with a communicator of one rank"""
    from dwarf_bench_amd import pjoin_native
    n = 1 << 20
    eng = pjoin_native.NativePartitionedJoin(n, rank=0, world=1, device=0, nccl_id=pjoin_native.unique_id())
    info = eng.info()  # what the bench line reports as rccl_ranks_seen / rank_devices: ncclCommCount says one rank here
    assert info["rccl_ranks_seen"] == 1 and info["world"] == 1 and info["device"] == 0 and info["device_name"]
    for _ in range(2):
        t = eng.step()
    assert t["exchange_us"] > 0 and t["partition_us"] > 0 and t["exchange_r_us"] > 0 and t["exchange_s_us"] > 0
    chk = eng.check()
    eng.close()
    build = po.gen_uniform_u32(n, 42, 0, n - 1)
    probe = po.gen_uniform_u32(n, 43, 0, n - 1)
    assert chk["bad_pairs"] == chk["bad_route"] == chk["bad_rows"] == 0 and chk["conserved"]
    assert chk["matches"] == int(po.join_counts_fast(build, probe).astype(np.uint64).sum())


def test_native_engine_step_keeps_its_six_value_contract():
    """dbench_pjoin_step (the first form of the call) writes six doubles — a caller built against `double t[6]` must
    not be overrun by the eight values dbench_pjoin_step_n can deliver"""
    import ctypes as C
    from dwarf_bench_amd import pjoin_native
    lib = pjoin_native.lib()
    lib.dbench_pjoin_step.restype = C.c_int
    lib.dbench_pjoin_step.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    eng = pjoin_native.NativePartitionedJoin(1 << 18, rank=0, world=1, device=0, direct_single=False)
    t = (C.c_double * 8)(*([-1.0] * 8))
    assert lib.dbench_pjoin_step(eng._h, t) == 0
    assert all(x >= 0 for x in t[:6]) and t[0] > 0 and t[6] == -1.0 and t[7] == -1.0
    t3 = (C.c_double * 3)()
    assert lib.dbench_pjoin_step_n(eng._h, t3, 3) == 3 and t3[0] > 0
    assert eng.info()["rccl_ranks_seen"] == 0  # one rank without an id: no communicator
    eng.close()
