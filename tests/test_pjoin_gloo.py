"""The N>1 path of the partitioned join on CPU: world_size 2 and 3 over gloo (127.0.0.1), exercising the real
orchestration (dwarf_bench_amd/pjoin.py: counts exchange, split sizes, all_to_all, row-id bookkeeping) with a
test-only numpy/oracle backend in place of the HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as po


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_build, n_probe, key_hi, q, max_elems=None, lossy=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dwarf_bench_amd import pjoin
        from tests.pjoin_testlib import LossyBackend, OracleBackend
        # contiguous shards of the global columns (the last rank takes the remainder: ragged shards)
        def shard(n):
            per = n // world
            lo = rank * per
            return lo, (n if rank == world - 1 else lo + per)
        blo, bhi = shard(n_build)
        plo, phi = shard(n_probe)
        build = po.gen_uniform_u32(bhi - blo, 42, 1, key_hi, first_index=blo)
        probe = po.gen_uniform_u32(phi - plo, 43, 1, key_hi, first_index=plo)
        args = (torch.from_numpy(build.view(np.int32).copy()), torch.from_numpy(probe.view(np.int32).copy()), blo, plo)
        if lossy:
            try:
                pjoin.partitioned_join(*args, backend=LossyBackend(rank))
                q.put((rank, "no error"))
            except pjoin.ExchangeError as e:
                q.put((rank, str(e)))
            dist.barrier()
            return
        res = pjoin.partitioned_join(*args, backend=OracleBackend(), max_message_elems=max_elems)
        u = lambda t: t.numpy().view(np.uint32).copy()
        q.put((rank, u(res.probe_row_ids), u(res.pos), u(res.cnt), u(res.build_row_ids), res.sent_rows,
               res.recv_build_rows, res.recv_probe_rows))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_build,n_probe,key_hi,max_elems", [
    (2, 600, 500, 300, None), (3, 1001, 777, 10000, None), (2, 64, 0, 10, None),
    # messages cut into rounds (the product's limit is 2^28 elements; forced small here): segments of ~300 rows go out
    # in pieces of 64 / 97 (ragged last round, a different number of rounds per peer) / 1 element
    (2, 600, 500, 300, 64), (3, 1001, 777, 10000, 97), (2, 40, 30, 10, 1)])
def test_partitioned_join_over_gloo(world, n_build, n_probe, key_hi, max_elems):
    from tests.pjoin_testlib import check_global, dest_of
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_build, n_probe, key_hi, q, max_elems)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs.sort(key=lambda t: t[0])
    build_all = po.gen_uniform_u32(n_build, 42, 1, key_hi)
    probe_all = po.gen_uniform_u32(n_probe, 43, 1, key_hi)
    check_global([(o[1], o[2], o[3], o[4]) for o in outs], build_all, probe_all)
    # every rank received exactly the rows whose key hashes to it, and the totals add up
    for rank, rid, pos, cnt, ids, sent, rb, rp in outs:
        assert np.all(dest_of(probe_all[rid], world) == rank)
        assert rb == int(np.sum(dest_of(build_all, world) == rank))
    assert sum(o[6] for o in outs) == n_build and sum(o[7] for o in outs) == n_probe


def test_world_size_one_is_a_plain_join():
    from dwarf_bench_amd import pjoin
    from tests.pjoin_testlib import OracleBackend, check_global
    b = po.gen_uniform_u32(300, 1, 1, 100)
    p = po.gen_uniform_u32(200, 2, 1, 100)
    res = pjoin.partitioned_join(torch.from_numpy(b.view(np.int32).copy()), torch.from_numpy(p.view(np.int32).copy()), 0, 0,
                                 backend=OracleBackend())
    u = lambda t: t.numpy().view(np.uint32)
    check_global([(u(res.probe_row_ids), u(res.pos), u(res.cnt), u(res.build_row_ids))], b, p)


def test_conservation_check_raises_on_every_rank():
    """a column whose received sum differs from the sent one (here: a backend that misreports one sum on rank 1)
    makes partitioned_join raise ExchangeError on EVERY rank — the check is collective"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 600, 500, 300, q, None, True)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("did not conserve column(s) [1]" in outs[r] for r in range(world)), outs
