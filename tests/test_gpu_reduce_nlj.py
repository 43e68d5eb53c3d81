"""Parity of the reduce and nested-loop-join kernels with the oracle (reduce/reduce.cpp:10-22,
join/nested_join.cpp:52-90) through the C ABI."""
import json

import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _dev(a, dt=np.uint32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt).view(np.int32)).cuda()


@pytest.mark.parametrize("n", [0, 1, 63, 64, 511, 512, 4096, 16383, 16384, 16385, 100003, 1 << 20, (1 << 24) + 77])
def test_reduce_reference_data(n):
    from dwarf_bench_amd import ops
    src = po.gen_uniform_u32(n, 42, 1, 10000).view(np.int32)
    got = int(ops.reduce_sum(_dev(src, np.int32)).cpu()[0])
    assert got == po.reduce_sum(src)
    if n < (1 << 17):  # no overflow possible: the plain integer sum (std::accumulate on reference data)
        assert got == int(src.astype(np.int64).sum())


def test_reduce_wraps_like_int32_and_handles_negatives():
    from dwarf_bench_amd import ops
    n = 3_000_001
    src = po.gen_uniform_u32(n, 9, 0, 2**32 - 1).view(np.int32)
    got = int(ops.reduce_sum(_dev(src, np.int32)).cpu()[0])
    low = int(src.astype(np.int64).sum()) & 0xFFFFFFFF  # exact sum folded to 32 bits, then read as int32
    assert got == po.reduce_sum(src) == (low - (1 << 32) if low >= 2**31 else low)


def test_reduce_unaligned_view_and_repeat():
    from dwarf_bench_amd import ops
    base = _dev(po.gen_uniform_u32(100000, 3, 1, 10000))
    whole = po.gen_uniform_u32(100000, 3, 1, 10000).view(np.int32)
    for off in (4, 8, 64):  # 16-byte aligned sub-views (the C ABI asks for 16-B alignment of columns)
        v = base[off:]
        assert int(ops.reduce_sum(v).cpu()[0]) == po.reduce_sum(whole[off:])
        assert int(ops.reduce_sum(v).cpu()[0]) == po.reduce_sum(whole[off:])


def _compact(cells):
    """nested_join.cpp:81-90: keep the cells whose key is not 0, in cell order"""
    k, v1, v2 = (c.reshape(-1) for c in cells)
    keep = k != 0
    return k[keep], v1[keep], v2[keep]


def test_nested_join_reference_fixture(golden_dir):
    from dwarf_bench_amd import ops
    f = json.loads((golden_dir / "reference_kats.json").read_text())["seq_join_fixture"]
    cells = ops.nested_join(_dev(f["keys_a"]), _dev(f["vals_a"]), _dev(f["keys_b"]), _dev(f["vals_b"]))
    k, v1, v2 = _compact([c.cpu().numpy().view(np.uint32) for c in cells])
    assert [list(map(int, r)) for r in zip(k, v1, v2)] == f["expected_rows"]  # a-major, b-minor: seq_join's order


@pytest.mark.parametrize("na,nb", [(1, 1), (7, 7), (16, 256), (17, 257), (128, 128), (1000, 333), (333, 1000),
                                   (2048, 2048), (4096, 4096)])
def test_nested_join_matches_oracle(na, nb):
    from dwarf_bench_amd import ops
    ak = po.gen_uniform_u32(na, 1, 1, 10000)
    av = po.gen_uniform_u32(na, 2, 1, 10000)
    bk = po.gen_uniform_u32(nb, 3, 1, 10000)
    bv = po.gen_uniform_u32(nb, 4, 1, 10000)
    got = [c.cpu().numpy().view(np.uint32) for c in ops.nested_join(_dev(ak), _dev(av), _dev(bk), _dev(bv))]
    want = po.nested_join(ak, av, bk, bv)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # compacted in cell order it is exactly seq_join (join_helpers.hpp:86-104), no sorting needed
    for g, w in zip(_compact(got), po.seq_join(ak, av, bk, bv)):
        assert np.array_equal(g, w)


def test_nested_join_agrees_with_hash_join_counts():
    """the nested-loop join as a device-side oracle for the hash join: matches per probe row"""
    from dwarf_bench_amd import ops
    n = 3000
    build = po.gen_uniform_u32(n, 42, 1, 2000)
    probe = po.gen_uniform_u32(n, 43, 1, 2000)
    ones = np.ones(n, dtype=np.uint32)
    cells = ops.nested_join(_dev(probe), _dev(ones), _dev(build), _dev(ones))
    per_probe = (cells[0] != 0).sum(dim=1).cpu().numpy()
    j = ops.HashJoin(n, n)
    j.build(_dev(build))
    j.probe(_dev(probe))
    assert np.array_equal(j.result()[1].cpu().numpy().view(np.uint32)[:n], per_probe.astype(np.uint32))


def test_reduce_unaligned_column():
    """any 4-byte aligned column: the elements in front of the first 16-byte boundary are added one by one"""
    from dwarf_bench_amd import ops
    base = ops.gen_uniform_u32(300011, 8, 0, 2**32 - 1)
    for off in (1, 2, 3):
        src = base[off:]
        assert int(ops.reduce_sum(src).cpu()[0]) == po.reduce_sum(src.cpu().numpy())
    assert int(ops.reduce_sum(base[1:3]).cpu()[0]) == po.reduce_sum(base[1:3].cpu().numpy())
