"""The C ABI promises asynchronous entry points that neither allocate nor synchronise, so a dwarf can be
captured in a hipGraph and replayed (launch-bound inner loops, the reference's 9-iteration sweeps at small sizes).
Each test captures one call sequence on torch's capture stream, refills the SAME input buffers with new data,
replays, and checks the replayed result against the oracle."""
import numpy as np
import pytest
import torch

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _fill(t, host):
    t.copy_(torch.from_numpy(np.ascontiguousarray(host).view(np.int32)))


def _capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()  # warm-up outside capture (lazy module loads, attribute calls)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


@pytest.mark.parametrize("n", [5000, 1 << 20, (1 << 23) + 5])
def test_scan_graph_replay(n):
    from dwarf_bench_amd import ops
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    plan = ops.CopyIfLt(n)
    _fill(src, po.gen_uniform_u32(n, 1, 1, 10000))
    g = _capture(lambda: plan.launch(src, 50))
    for seed in (7, 8):
        host = po.gen_uniform_u32(n, seed, 1, 10000)
        _fill(src, host)
        g.replay()
        assert np.array_equal(plan.result().cpu().numpy(), po.copy_if_lt(host.view(np.int32), 50))


@pytest.mark.parametrize("n", [5000, (1 << 22) + 5])
def test_dense_scan_and_exclusive_scan_graph_replay(n):
    """the two single-launch prefix kernels (ticketed chunks, granules cleared by a fill inside the captured sequence)"""
    from dwarf_bench_amd import ops
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    plan = ops.CopyIfLt(n)
    _fill(src, po.gen_uniform_u32(n, 1, 1, 10000))
    lib = __import__("dwarf_bench_amd._capi", fromlist=["lib"]).lib()
    ws_bytes = lib.dbhip_exclusive_scan_u32_workspace_bytes(n)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")

    def both():
        plan.launch(src, 6000, dense=True)
        rc = lib.dbhip_exclusive_scan_u32(src.data_ptr(), n, 5, out.data_ptr(), ws.data_ptr(), ws_bytes,
                                          torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    g = _capture(both)
    for seed in (7, 8):
        host = po.gen_uniform_u32(n, seed, 1, 10000)
        _fill(src, host)
        g.replay()
        assert np.array_equal(plan.result().cpu().numpy(), po.copy_if_lt(host.view(np.int32), 6000))
        exp = np.zeros(n, dtype=np.uint32)
        np.cumsum(host[:-1], dtype=np.uint32, out=exp[1:])
        assert np.array_equal(out.cpu().numpy().view(np.uint32), exp + np.uint32(5))


@pytest.mark.parametrize("n,bits", [(4096, 8), (1 << 20, 8), (300007, 4)])
def test_sort_graph_replay(n, bits):
    from dwarf_bench_amd import ops
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    plan = ops.RadixSort(n, bits)
    _fill(keys, po.gen_uniform_u32(n, 1, 0, 2**32 - 1))
    g = _capture(lambda: plan.launch(keys))
    for seed, hi in ((7, 2**32 - 1), (8, 10000)):  # the second replay skips passes on a device-side flag
        host = po.gen_uniform_u32(n, seed, 0, hi)
        _fill(keys, host)
        g.replay()
        torch.cuda.synchronize()
        assert ops.workspace_status(plan.ws) == 0
        assert np.array_equal(keys.cpu().numpy().view(np.uint32), np.sort(host))


def test_sort_captured_as_the_first_sort_of_a_process():
    """The sort entry points never synchronise, and a captured sort ranks exactly like an eager one: in a fresh process
    the very first sort is captured (no eager call before it — the module-load warm-up a torch capture needs is done
    with another dwarf), the rank mode it reports is the LDS-atomic one, and replays give the oracle's order."""
    import os, subprocess, sys
    prog = (
        "import numpy as np, torch\n"
        "from dwarf_bench_amd import _capi, ops\n"
        "from oracle import pyoracle as po\n"
        "n = (1 << 20) + 3\n"
        "ops.reduce_sum(ops.gen_uniform_u32(1024, 1, 0, 9)); torch.cuda.synchronize()  # loads the code object\n"
        "keys = torch.empty(n, dtype=torch.int32, device='cuda')\n"
        "plans = {b: ops.RadixSort(n, b) for b in (8, 4)}\n"
        "for bits, plan in plans.items():\n"
        "    g = torch.cuda.CUDAGraph()\n"
        "    with torch.cuda.graph(g):\n"
        "        plan.launch(keys)\n"
        "    assert _capi.lib().dbhip_radix_sort_rank_mode() == 1\n"
        "    for seed in (3, 4):\n"
        "        host = po.gen_uniform_u32(n, seed, 0, 2**32 - 1)\n"
        "        keys.copy_(torch.from_numpy(host.view(np.int32)))\n"
        "        g.replay(); torch.cuda.synchronize()\n"
        "        assert ops.workspace_status(plan.ws) == 0\n"
        "        assert np.array_equal(keys.cpu().numpy().view(np.uint32), po.sort_u32(host)), (bits, seed)\n"
        "print('captured first: ok')\n")
    env = {k: v for k, v in os.environ.items() if k != "DBHIP_RS_RANK"}
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(__file__)))
    assert r.returncode == 0 and "captured first: ok" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.parametrize("n,groups", [(100003, 64), (1 << 21, 65536)])
def test_groupby_graph_replay(n, groups):
    from dwarf_bench_amd import ops
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    vals = torch.empty(n, dtype=torch.int32, device="cuda")
    plan = ops.GroupBySum(n, groups)
    _fill(keys, po.gen_uniform_u32(n, 1, 0, groups - 1))
    _fill(vals, po.gen_uniform_u32(n, 2, 1, 10000))
    g = _capture(lambda: plan.launch(keys, vals))
    for seed in (7, 8):
        hk, hv = po.gen_uniform_u32(n, seed, 0, groups - 1), po.gen_uniform_u32(n, seed + 10, 1, 10000)
        _fill(keys, hk)
        _fill(vals, hv)
        g.replay()
        assert np.array_equal(plan.result().cpu().numpy().view(np.uint32), po.groupby_sum(hk, hv, groups))


@pytest.mark.parametrize("n", [3000, 1 << 18])  # HBM-table path and LDS-partitioned path
def test_join_graph_replay(n):
    from dwarf_bench_amd import ops
    build = torch.empty(n, dtype=torch.int32, device="cuda")
    probe = torch.empty(n, dtype=torch.int32, device="cuda")
    plan = ops.HashJoin(n, n)
    _fill(build, po.gen_uniform_u32(n, 1, 0, n - 1))
    _fill(probe, po.gen_uniform_u32(n, 2, 0, n - 1))

    def both():
        plan.build(build)
        plan.probe(probe)
    g = _capture(both)
    for seed in (7, 8):
        hb, hp = po.gen_uniform_u32(n, seed, 0, n - 1), po.gen_uniform_u32(n, seed + 10, 0, n - 1)
        _fill(build, hb)
        _fill(probe, hp)
        g.replay()
        pos, cnt, ids = (t.cpu().numpy().view(np.uint32) for t in plan.result())
        assert np.array_equal(cnt, po.join_counts_fast(hb, hp).astype(np.uint32))
        hit = cnt > 0
        assert np.all(hb[ids[pos[hit]]] == hp[hit])
        assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))
