"""Parity of the HIP scan/compaction kernel (through the C ABI) with the oracle (scan/scan.cpp:12-17)."""
import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _src(n, seed=42, lo=1, hi=10000):
    from dwarf_bench_amd import ops
    return ops.gen_uniform_u32(n, seed, lo, hi)


@pytest.mark.parametrize("n", [0, 1, 3, 63, 64, 65, 1024, 4095, 8191, 8192, 8193, 100003, 1 << 20, (1 << 22) + 12345])
@pytest.mark.parametrize("filt", [5, 1001, 5001, 20000, -3])
def test_copy_if_matches_oracle(n, filt):
    from dwarf_bench_amd import ops
    src = _src(n)
    host = src.cpu().numpy()
    got = ops.copy_if_lt(src, filt).cpu().numpy()
    exp = po.copy_if_lt(host, filt)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)


def test_generator_twin_is_bit_identical():
    from dwarf_bench_amd import ops
    for n, first in ((1000, 0), (4097, 123456789012)):
        d = ops.gen_uniform_u32(n, 7, 1, 10000, first_index=first).cpu().numpy().astype(np.uint32)
        assert np.array_equal(d, po.gen_uniform_u32(n, 7, 1, 10000, first_index=first))
        u = ops.gen_unique_sorted_u32(n, 9, first_index=first % 1000).cpu().numpy().astype(np.uint32)
        assert np.array_equal(u, po.gen_unique_sorted_u32(n, 9, first_index=first % 1000))


def test_reference_plumbing_config():
    """BASELINE configs[0]: TwoPassScan --input_size=1024 --iterations=9, filter 5, data in [1,10000]."""
    from dwarf_bench_amd import ops
    src = _src(1024)
    plan = ops.CopyIfLt(1024)
    exp = po.copy_if_lt(src.cpu().numpy(), 5)
    for _ in range(9):
        plan.launch(src, 5)
        assert np.array_equal(plan.result().cpu().numpy(), exp)


def test_negative_and_extreme_values():
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(1)
    host = rng.integers(-2**31, 2**31 - 1, 50000, dtype=np.int64).astype(np.int32)
    host[::97] = np.iinfo(np.int32).min
    host[1::89] = np.iinfo(np.int32).max
    src = torch.from_numpy(host).cuda()
    for filt in (np.iinfo(np.int32).min, 0, np.iinfo(np.int32).max):
        got = ops.copy_if_lt(src, int(filt)).cpu().numpy()
        assert np.array_equal(got, po.copy_if_lt(host, int(filt)))


def test_unaligned_source_pointer():
    from dwarf_bench_amd import ops
    base = _src(100000 + 3)
    for off in (1, 2, 3):
        src = base[off: off + 100000]
        got = ops.copy_if_lt(src, 777)  # a view starting 4/8/12 bytes past a 16-byte boundary
        assert np.array_equal(got.cpu().numpy(), po.copy_if_lt(src.cpu().numpy(), 777))


def test_full_size_properties_2_28():
    """BASELINE headline size: 2^28 int32.  Size-independent checks: count == sum(src < f), output sorted
    positions preserved (stable): out equals torch's boolean-mask compaction; idempotence."""
    from dwarf_bench_amd import ops
    n = 1 << 28
    src = _src(n)
    for filt in (5, 1001):
        plan = ops.CopyIfLt(n)
        plan.launch(src, filt)
        out = plan.result()
        assert out.numel() == int((src < filt).sum().item())
        assert torch.equal(out, src[src < filt])
        again = ops.copy_if_lt(out.clone(), filt)  # idempotent
        assert torch.equal(again, out)
        del plan, out, again
    # the oracle (std::copy_if restated, scan/scan.cpp:12-17) on the WHOLE column, both filters
    host = src.cpu().numpy()
    for filt in (5, 1001):
        assert np.array_equal(ops.copy_if_lt(src, filt).cpu().numpy(), po.copy_if_lt(host, filt)), filt


@pytest.mark.parametrize("n", [(1 << 24) + 777, 250007, 77777])  # several chunks per workgroup, a few chunks, a single workgroup
def test_stress_under_uneven_load(n):
    """Hand-offs under uneven load: varying selectivity per region + a second stream hammering HBM."""
    from dwarf_bench_amd import ops
    host = po.gen_uniform_u32(n, 3, 1, 10000).astype(np.int32)
    host[: n // 3] = 1           # dense matches up front
    host[n // 3: n // 2] = 9999  # none
    src = torch.from_numpy(host).cuda()
    exp = po.copy_if_lt(host, 50)
    noise = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    for it in range(5):
        with torch.cuda.stream(side):
            noise.add_(1)
        got = ops.copy_if_lt(src, 50)
        assert np.array_equal(got.cpu().numpy(), exp), it
    torch.cuda.synchronize()


@pytest.mark.parametrize("log2n,extra,filt,dense", [(31, 5, 5, False), (30, 12345, 5001, False), (31, 5, 5001, True),
                                                    (30, 12345, 10001, True)])
def test_beyond_32_bit_byte_offsets(log2n, extra, filt, dense):
    """n * 4 bytes > 4 GiB: 64-bit indexing everywhere, both entry points (torch's boolean indexing as the independent
    check)"""
    from dwarf_bench_amd import ops
    n = (1 << log2n) + extra
    src = ops.gen_uniform_u32(n, 77, 1, 10000)
    plan = ops.CopyIfLt(n)
    plan.launch(src, filt, dense=dense)
    got = plan.result()
    # compare piecewise to bound the temporary memory of the torch reference
    step, off = 1 << 28, 0
    for lo in range(0, n, step):
        part = src[lo: lo + step]
        want = part[part < filt]
        assert torch.equal(got[off: off + want.numel()], want), lo
        off += want.numel()
    assert off == got.numel()


# ---- dense variant (dbhip_copy_if_lt_dense_i32): one launch, chunk-granular hand-off, final positions at first write;
# a wave owns 4096 elements of a 32768-element chunk: sizes around both ----
@pytest.mark.parametrize("n", [0, 1, 3, 64, 4095, 4096, 4097, 8191, 32768, 32769, 65535, 65536, 65537, 131072, 131073,
                               524288, 524289, 1 << 20, (1 << 22) + 12345, (1 << 25) + 7])
@pytest.mark.parametrize("filt", [5, 5001, 20000, -3])
def test_dense_copy_if_matches_oracle(n, filt):
    from dwarf_bench_amd import ops
    src = _src(n)
    got = ops.copy_if_lt(src, filt, dense=True).cpu().numpy()
    exp = po.copy_if_lt(src.cpu().numpy(), filt)
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_dense_variant_unaligned_extremes_and_plan_switch():
    from dwarf_bench_amd import ops
    rng = np.random.default_rng(2)
    host = rng.integers(-2**31, 2**31 - 1, 700001, dtype=np.int64).astype(np.int32)
    host[::97] = np.iinfo(np.int32).min
    host[1::89] = np.iinfo(np.int32).max
    base = torch.from_numpy(host).cuda()
    for off in (0, 1, 3):
        src = base[off:]
        for filt in (np.iinfo(np.int32).min, 0, np.iinfo(np.int32).max):
            got = ops.copy_if_lt(src, int(filt), dense=True).cpu().numpy()
            assert np.array_equal(got, po.copy_if_lt(host[off:], int(filt)))
    # a plan picks the variant from the selectivity it saw last for that filter value; results never differ
    n = 1 << 22
    src = _src(n)
    plan = ops.CopyIfLt(n)
    exp_dense, exp_sparse = po.copy_if_lt(src.cpu().numpy(), 5001), po.copy_if_lt(src.cpu().numpy(), 5)
    for _ in range(3):
        plan.launch(src, 5001)  # first call: two-launch path; then the dense one (selectivity 0.5)
        assert np.array_equal(plan.result().cpu().numpy(), exp_dense)
        plan.launch(src, 5)     # stays on the two-launch path (selectivity 4e-4)
        assert np.array_equal(plan.result().cpu().numpy(), exp_sparse)
    assert plan._seen[5001] > plan.DENSE_ABOVE > plan._seen[5]


def test_dense_variant_at_full_size_and_on_concurrent_streams():
    """2^28 rows at 50 % selectivity against the two-launch path (fingerprint + length), and four dense scans in flight
    on four streams: chunks are taken by ticket, so a chunk's predecessors always belong to running workgroups"""
    from dwarf_bench_amd import ops
    n = 1 << 28
    src = _src(n)
    a = ops.CopyIfLt(n)
    a.launch(src, 5001, dense=True)
    out = a.result()
    want = ops.check_fingerprint_lt(src, 5001)
    assert want[1] == out.numel() and ops.check_fingerprint_lt(out, 5001) == want
    del a, out
    m = 1 << 24
    small = src[:m]
    plans = [ops.CopyIfLt(m) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    for _ in range(5):
        for p, st in zip(plans, streams):
            with torch.cuda.stream(st):
                p.launch(small, 5001, dense=True)
    torch.cuda.synchronize()
    exp = ops.copy_if_lt(small, 5001)
    for p in plans:
        assert torch.equal(p.result(), exp)
