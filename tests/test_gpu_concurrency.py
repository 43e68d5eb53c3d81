"""Several dwarfs in flight at once on different streams (separate workspaces).  One kernel of the library waits on
other workgroups — the dense scan's chunk hand-off: chunks are taken by ticket, so a chunk's predecessors always belong
to workgroups that are already running, and the wait is time-bounded.  Concurrent launches must neither deadlock nor
disturb each other's results."""
import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def test_dwarfs_on_concurrent_streams():
    from dwarf_bench_amd import ops
    rounds = 6
    n_scan, n_sort, n_gb, groups, n_join = 3_000_000, 1 << 20, 1 << 21, 4096, 40_000
    src = po.gen_uniform_u32(n_scan, 1, 1, 10000).view(np.int32)
    keys = po.gen_uniform_u32(n_sort, 2, 0, 2**32 - 1)
    gk, gv = po.gen_uniform_u32(n_gb, 3, 0, groups - 1), po.gen_uniform_u32(n_gb, 4, 1, 10000)
    jb, jp = po.gen_uniform_u32(n_join, 5, 1, 10000), po.gen_uniform_u32(n_join, 6, 1, 10000)
    want_scan, want_sort = po.copy_if_lt(src, 2000), np.sort(keys)
    want_dense = po.copy_if_lt(src, 6000)
    want_gb, want_cnt = po.groupby_sum(gk, gv, groups), po.join_counts_fast(jb, jp).astype(np.uint32)

    d_src, d_keys0, d_gk, d_gv, d_jb, d_jp = _dev(src), _dev(keys), _dev(gk), _dev(gv), _dev(jb), _dev(jp)
    streams = [torch.cuda.Stream() for _ in range(5)]
    scans = [ops.CopyIfLt(n_scan) for _ in range(rounds)]
    dense = [ops.CopyIfLt(n_scan) for _ in range(rounds)]
    sorts = [(ops.RadixSort(n_sort, 8), d_keys0.clone()) for _ in range(rounds)]
    gbs = [ops.GroupBySum(n_gb, groups) for _ in range(rounds)]
    joins = [ops.HashJoin(n_join, n_join) for _ in range(rounds)]
    torch.cuda.synchronize()
    for r in range(rounds):  # everything is enqueued before anything is waited for
        with torch.cuda.stream(streams[0]):
            scans[r].launch(d_src, 2000)
        with torch.cuda.stream(streams[1]):
            sorts[r][0].launch(sorts[r][1])
        with torch.cuda.stream(streams[2]):
            gbs[r].launch(d_gk, d_gv)
        with torch.cuda.stream(streams[3]):
            joins[r].build(d_jb)
            joins[r].probe(d_jp)
        with torch.cuda.stream(streams[4]):
            dense[r].launch(d_src, 6000, dense=True)
    torch.cuda.synchronize()
    for r in range(rounds):
        assert np.array_equal(scans[r].result().cpu().numpy(), want_scan), r
        assert np.array_equal(dense[r].result().cpu().numpy(), want_dense), r
        assert ops.workspace_status(sorts[r][0].ws) == 0
        assert np.array_equal(sorts[r][1].cpu().numpy().view(np.uint32), want_sort), r
        assert np.array_equal(gbs[r].result().cpu().numpy().view(np.uint32), want_gb), r
        pos, cnt, ids = joins[r].result()
        assert np.array_equal(cnt.cpu().numpy().view(np.uint32), want_cnt), r


def test_two_large_scans_share_the_gpu():
    """two 2^26 scans running together still produce exact results"""
    from dwarf_bench_amd import ops
    n = 1 << 26
    a, b = ops.gen_uniform_u32(n, 11, 1, 10000), ops.gen_uniform_u32(n, 12, 1, 10000)
    pa, pb = ops.CopyIfLt(n), ops.CopyIfLt(n)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(s1):
            pa.launch(a, 7)
        with torch.cuda.stream(s2):
            pb.launch(b, 9)
    torch.cuda.synchronize()
    assert torch.equal(pa.result(), a[a < 7]) and torch.equal(pb.result(), b[b < 9])
