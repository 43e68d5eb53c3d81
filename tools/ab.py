"""development aid: `python tools/ab.py <dwarf> [log2 size]` — HIP-event timings of one dwarf through the C ABI, one
line per run, for A/B comparisons of library builds (DBHIP_LIB=<path to a libdbhip_*.so built by tools/build_variant.sh>
selects the build; the line starts with its tag).  Every mode also checks its result against torch, so a faster
variant that computes something else shows up as WRONG.  This is how the per-kernel figures in DESIGN.md were taken;
the contract numbers come from bench.py.

  scan            two-launch vs dense copy_if at 2^28 rows over a selectivity sweep (median of 9)
  sort [lg]       2^lg-key sort, 8- and 4-bit digits (drop-max-mean of 9, refresh copy subtracted); default lg 24
  sort-only       three 2^24 sorts and nothing else (counter collection; SORT_BITS=4|8, SORT_SHAPE=<index into sort-shapes' list>)
  groupby         2^26 rows at 2^16 / 2^15 / 2^10 / 64 groups (drop-max-mean of 9)
  groupby-shapes  more than 32768 groups over row counts, group counts and value ranges (median of 5): run it with
                  DBHIP_GB_PACKED=0 beside the default to see what the kernel's choice between its two large-table modes
                  buys and that it costs nothing where the packed table would be slow
  groupby-skew    2^26 rows whose keys crowd into one or a few groups, at 64 .. 2^16 groups (median of 5)
  sort-shapes [lg] 2^lg keys of eight distributions (few distinct values, sorted, reversed, skewed ...; median of 5)
  join [lg]       build / probe / radix join of 2^lg x 2^lg (drop-max-mean of 7); default lg 26
  radix [lg]      the radix join alone at 2^lg x 2^lg (default 30: the P = 1 point of the partitioned join)
  radix-sizes     the radix join at 1, 1.25, 1.5, 1.75 x 2^25 .. 2^29 rows (geometry steps)
  join-skew [lg]  the same over key shapes (hot keys, strided keys, sorted, few distinct keys; median of 3)
  size-sweep      every dwarf at sizes next to and between powers of two, 2^13 .. 2^26 (geometry cliffs)
  partition       rank-level partition (dbhip_pjoin_partition_u32) of 2^27 rows into P buckets
  reduce          2^28-row reduce (median of 15; DBHIP_RED_WGS)
  xscan [lg]      exclusive scan of 2^lg uint32, aligned (one launch) and offset by one element (three launches)
  graph           direct launches vs hipGraph replay of sort and join at 2^14..2^20 rows (host wall clock)
  launch-join [lg] / launch-sort [lg] / launch-all
                  a few untimed calls and nothing else: the program to put behind `rocprofv3 --kernel-trace --stats` or
                  `--pmc` (tools/pmc_kernel_counters.sh); tools/prof_show.py prints the tables
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dwarf_bench_amd import ops  # noqa: E402

TAG = os.environ.get("DBHIP_LIB", "default").split("libdbhip_")[-1]


def times(fn, k, warm=2):
    """device time of k calls, us each (events on torch's current stream = the launch stream), sorted"""
    for _ in range(warm):
        fn()
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return sorted(out)


def dropmax(ts):
    return sum(ts[:-1]) / (len(ts) - 1)


def median(ts):
    return ts[len(ts) // 2]


def u64(t):
    return t.to(torch.int64) & 0xFFFFFFFF


def bits32(t):
    """int64 values in [0, 2^32) as the int32 tensor holding the same bits (how the C ABI takes uint32 keys)"""
    return ((t + 2**31) % 2**32 - 2**31).to(torch.int32)


def scan(_):
    n = 1 << 28
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    plan = ops.CopyIfLt(n)
    for filt in (5, 101, 251, 501, 751, 1001, 1501, 2001, 2501, 5001, 7501, 10001):
        t0 = median(times(lambda: plan.launch(src, filt, dense=False), 9))
        m = plan.result().numel()
        t1 = median(times(lambda: plan.launch(src, filt, dense=True), 9))
        m1 = plan.result().numel()
        byts = 4 * n + 4 * m
        print(f"{TAG:16s} s={m / n:6.4f}: two-launch {t0:7.1f} us ({byts / t0 / 8e6 * 100:4.1f} %)   dense {t1:7.1f} us "
              f"({byts / t1 / 8e6 * 100:4.1f} %)  {'same count' if m == m1 == int((src < filt).sum()) else 'COUNT DIFFERS'}", flush=True)


def sort(lg):
    n = 1 << (lg or 24)
    keys0 = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    keys = keys0.clone()
    ref = torch.sort(u64(keys0)).values
    res = []
    for bits in (8, 4):
        plan = ops.RadixSort(n, bits)

        def run():
            keys.copy_(keys0)
            plan.launch(keys)

        t = dropmax(times(run, 9)) - dropmax(times(lambda: keys.copy_(keys0), 9))
        run()
        ok = bool(torch.equal(u64(keys), ref)) and ops.workspace_status(plan.ws) == 0
        res.append(f"{bits}-bit {t:7.1f} us {'ok' if ok else 'WRONG'}")
    print(f"{TAG:24s} 2^{lg or 24}: " + "   ".join(res), flush=True)


def sort_only(_):
    n = 1 << 24
    keys0 = _shape_keys(n, SHAPES[int(os.environ.get("SORT_SHAPE", "0"))])
    keys = keys0.clone()
    plan = ops.RadixSort(n, int(os.environ.get("SORT_BITS", "8")))
    for _i in range(3):
        keys.copy_(keys0)
        plan.launch(keys)
    torch.cuda.synchronize()
    print("ok")


def groupby(_):
    n = 1 << 26
    res = []
    for groups in (1 << 16, 1 << 15, 1 << 10, 64):
        keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
        vals = ops.gen_uniform_u32(n, 43, 1, 10000)
        plan = ops.GroupBySum(n, groups)
        t = dropmax(times(lambda: plan.launch(keys, vals), 9))
        ref = torch.zeros(groups, dtype=torch.int64, device="cuda").index_add_(0, keys.to(torch.int64), vals.to(torch.int64))
        ok = bool(torch.equal(u64(plan.result()), ref & 0xFFFFFFFF))
        res.append(f"G={groups}: {t:6.1f} us {'ok' if ok else 'WRONG'}")
    print(f"{TAG:16s} " + "  ".join(res), flush=True)


def groupby_shapes(_):
    tag = "packed=" + os.environ.get("DBHIP_GB_PACKED", "auto")
    for lg, groups, hi in ((26, 65536, 10000), (27, 65536, 10000), (28, 65536, 10000), (26, 65536, 60000),
                           (26, 65536, 2**32 - 1), (26, 1 << 18, 10000), (27, 1 << 17, 10000), (24, 65536, 10000)):
        n = 1 << lg
        keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
        vals = ops.gen_uniform_u32(n, 43, 1, hi)
        plan = ops.GroupBySum(n, groups)
        t = median(times(lambda: plan.launch(keys, vals), 5))
        ref = torch.zeros(groups, dtype=torch.int64, device="cuda").index_add_(0, keys.to(torch.int64), vals.to(torch.int64))
        ok = bool(torch.equal(u64(plan.result()), ref & 0xFFFFFFFF))
        mode = int(plan.ws[4:8].view(torch.int32).item())
        print(f"{tag:12s} n=2^{lg} G={groups} vals<={hi}: {t:8.1f} us  mode {('-', 'packed', 'wide')[mode]}  {'ok' if ok else 'WRONG'}", flush=True)
        del keys, vals, plan


def _shape_keys(n, kind):
    g = torch.Generator(device="cuda").manual_seed(7)
    if kind == "uniform 32-bit":
        return ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    if kind == "[1, 10000]":
        return ops.gen_uniform_u32(n, 42, 1, 10000)
    if kind == "two values":
        return bits32(torch.randint(0, 2, (n,), device="cuda", generator=g, dtype=torch.int64) * 0xFFFFFFFF)
    if kind == "16 values":
        return bits32(torch.randint(0, 16, (n,), device="cuda", generator=g, dtype=torch.int64) * 0x11111111)
    if kind == "90 % one value":
        k = torch.randint(0, 2**32, (n,), device="cuda", generator=g, dtype=torch.int64)
        hot = torch.rand(n, device="cuda", generator=g) < 0.9
        return bits32(torch.where(hot, torch.full_like(k, 0x9E3779B9), k))
    if kind == "sorted":
        return bits32(torch.sort(u64(ops.gen_uniform_u32(n, 42, 0, 2**32 - 1))).values)
    if kind == "reversed":
        return bits32(torch.sort(u64(ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)), descending=True).values)
    if kind == "geometric":  # key = 2^32 * u^8: most keys small, every byte still varies
        u = torch.rand(n, device="cuda", generator=g, dtype=torch.float64)
        return bits32((u ** 8 * 4294967295.0).to(torch.int64))
    raise ValueError(kind)


SHAPES = ("uniform 32-bit", "[1, 10000]", "two values", "16 values", "90 % one value", "sorted", "reversed", "geometric")


def sort_shapes(lg):
    n = 1 << (lg or 24)
    for kind in SHAPES:
        keys0 = _shape_keys(n, kind)
        keys = keys0.clone()
        ref = torch.sort(u64(keys0)).values
        res = []
        for bits in (8, 4):
            plan = ops.RadixSort(n, bits)

            def run():
                keys.copy_(keys0)
                plan.launch(keys)

            t = median(times(run, 5)) - median(times(lambda: keys.copy_(keys0), 5))
            run()
            ok = bool(torch.equal(u64(keys), ref)) and ops.workspace_status(plan.ws) == 0
            res.append(f"{bits}-bit {t:8.1f} us {'ok' if ok else 'WRONG'}")
        print(f"{TAG:16s} 2^{lg or 24} {kind:16s}: " + "   ".join(res), flush=True)


def groupby_skew(_):
    n = 1 << 26
    for groups in (64, 1 << 10, 1 << 12, 1 << 14, 1 << 15, 1 << 16):
        for kind in ("uniform", "one group", "90 % one group", "16 groups", "sorted keys"):
            g = torch.Generator(device="cuda").manual_seed(7)
            keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
            if kind == "one group":
                keys = torch.full_like(keys, groups // 3)
            elif kind == "90 % one group":
                hot = torch.rand(n, device="cuda", generator=g) < 0.9
                keys = torch.where(hot, torch.full_like(keys, groups // 3), keys)
            elif kind == "16 groups":
                keys = bits32(u64(keys) % 16 * (groups // 16))
            elif kind == "sorted keys":
                keys = bits32(torch.sort(u64(keys)).values)
            vals = ops.gen_uniform_u32(n, 43, 1, 10000)
            plan = ops.GroupBySum(n, groups)
            t = median(times(lambda: plan.launch(keys, vals), 5))
            ref = torch.zeros(groups, dtype=torch.int64, device="cuda").index_add_(0, u64(keys), u64(vals))
            ok = bool(torch.equal(u64(plan.result()), ref & 0xFFFFFFFF))
            print(f"{TAG:16s} G={groups:6d} {kind:16s}: {t:9.1f} us {'ok' if ok else 'WRONG'}", flush=True)
            del keys, vals, plan


def join(lg):
    lg = lg or 26
    n = 1 << lg
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    plan = ops.HashJoin(n, n)
    plan.build(build)
    plan.probe(probe)
    b = dropmax(times(lambda: plan.build(build), 7, warm=0))
    p = dropmax(times(lambda: plan.probe(probe), 7, warm=0))
    plan.result()
    total = int(plan.cnt.to(torch.int64).sum())
    del plan
    rj = ops.RadixJoin(n, n)

    def radix():
        rj.partition_build(build)
        rj.partition_probe(probe)
        rj.match()

    r = dropmax(times(radix, 7, warm=1))
    pb = dropmax(times(lambda: rj.partition_build(build), 7, warm=0))
    m = dropmax(times(rj.match, 7, warm=0))
    rj.result()
    ok = int(rj.cnt.to(torch.int64).sum()) == total
    print(f"{TAG:20s} 2^{lg}: build {b:8.1f} probe {p:8.1f} total {b + p:8.1f} us | radix join {r:8.1f} us "
          f"(partition one side {pb:7.1f}, match {m:7.1f}) matches {'equal' if ok else 'DIFFER'}", flush=True)


def radix(lg):
    """the radix join alone (no table in HBM): sizes up to 2^30 x 2^30 fit; the match count is printed so that variants
    can be compared with each other (at 2^26 it is checked against torch by `join`)"""
    lg = lg or 30
    n = 1 << lg
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    rj = ops.RadixJoin(n, n)

    def run():
        rj.partition_build(build)
        rj.partition_probe(probe)
        rj.match()

    k = 5 if lg >= 29 else 7
    r = dropmax(times(run, k, warm=1))
    pb = dropmax(times(lambda: rj.partition_build(build), k, warm=0))
    m = dropmax(times(rj.match, k, warm=0))
    rj.result()
    total = int(rj.cnt.to(torch.int64).sum())
    ids_ok = int(rj.ids.to(torch.int64).sum()) == n * (n - 1) // 2  # the id buffer is a permutation of the build rows
    print(f"{TAG:20s} 2^{lg}: radix join {r:9.1f} us (partition one side {pb:8.1f}, match {m:8.1f}) matches {total} "
          f"ids {'a permutation sum' if ids_ok else 'WRONG'}", flush=True)


def radix_offsets(lg):
    """does the partition's time depend on WHERE its buffers lie?  One side of the radix join partitioned with the key
    column and the workspace at different byte offsets inside two larger allocations (the same kernels, the same data)"""
    lg = lg or 30
    n = 1 << lg
    rj = ops.RadixJoin(n, n)
    slack = 1 << 27
    big_ws = torch.empty(rj.ws_bytes + slack, dtype=torch.uint8, device="cuda")
    big_in = torch.empty(n + slack // 4, dtype=torch.int32, device="cuda")
    src = ops.gen_uniform_u32(n, 42, 0, n - 1)
    print(f"{TAG:12s} 2^{lg}: workspace {rj.ws_bytes / 2**30:.1f} GiB at {big_ws.data_ptr():#x}, keys at {big_in.data_ptr():#x}", flush=True)
    for in_off in (0, 4096, 1 << 20, (1 << 21) + (1 << 16), (1 << 26) + (1 << 13)):
        keys = big_in[in_off // 4: in_off // 4 + n]
        keys.copy_(src)
        row = []
        for ws_off in (0, 256, 4096, 1 << 16, 1 << 20, 1 << 21, (1 << 21) + 4096, 3 << 21, (1 << 25) + (1 << 12), (1 << 26) + (1 << 18)):
            rj.ws = big_ws[ws_off: ws_off + rj.ws_bytes]
            t = dropmax(times(lambda: rj.partition_build(keys), 4, warm=1))
            row.append(f"{t:8.1f}")
        print(f"{TAG:12s} keys +{in_off:9d} B | workspace +0, +256, +4K, +64K, +1M, +2M, +2M4K, +6M, +32M4K, +64M256K: {' '.join(row)}", flush=True)


def radix_alloc(lg):
    """the same radix join on buffers from torch's allocator and on buffers from plain hipMalloc calls in two orders (the
    C++ engine allocates that way): does the allocation decide the partition's time?"""
    import ctypes as C
    from dwarf_bench_amd import _capi
    lg = lg or 30
    n = 1 << lg
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    L = _capi.lib()
    ws_bytes = L.dbhip_join_radix_workspace_bytes(n, n)

    def raw(size):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), size) == 0
        return p.value

    def run(label, build, probe, ws, ids, rid, pos, cnt):
        st = ops._stream()
        assert L.dbhip_gen_uniform_u32(build, n, 42, 0, 0, n - 1, st) == 0
        assert L.dbhip_gen_uniform_u32(probe, n, 43, 0, 0, n - 1, st) == 0
        pb = lambda: _capi.check(L.dbhip_join_radix_partition_u32(0, build, None, n, n, n, ws, ws_bytes, st), "p")
        pp = lambda: _capi.check(L.dbhip_join_radix_partition_u32(1, probe, None, n, n, n, ws, ws_bytes, st), "p")
        mm = lambda: _capi.check(L.dbhip_join_radix_match_u32(n, n, ids, rid, pos, cnt, ws, ws_bytes, st), "m")

        def whole():
            pb(); pp(); mm()

        r = dropmax(times(whole, 4, warm=1))
        a = dropmax(times(pb, 4, warm=0))
        b = dropmax(times(pp, 4, warm=0))
        m = dropmax(times(mm, 4, warm=0))
        print(f"{TAG:12s} 2^{lg} {label:44s}: radix join {r:9.1f} us (build side {a:8.1f}, probe side {b:8.1f}, match {m:8.1f}) "
              f"ws at {ws:#x} keys at {build:#x} / {probe:#x}", flush=True)

    t = [ops.gen_uniform_u32(n, 42, 0, n - 1), ops.gen_uniform_u32(n, 43, 0, n - 1), torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")]
    t += [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4)]
    run("torch tensors", *[x.data_ptr() for x in t])
    del t
    torch.cuda.empty_cache()
    for label, order in (("hipMalloc: keys, keys, 512 B, workspace, out", "bpcwo"), ("hipMalloc: workspace first", "wbpco"),
                         ("hipMalloc: keys, keys, workspace, out (no small)", "bpwo")):
        got, ptrs = {}, []
        for ch in order:
            if ch == "o":
                for k in ("ids", "rid", "pos", "cnt"):
                    got[k] = raw(n * 4)
            else:
                got[ch] = raw({"b": n * 4, "p": n * 4, "c": 512, "w": ws_bytes}[ch])
        run(label, got["b"], got["p"], got["w"], got["ids"], got["rid"], got["pos"], got["cnt"])
        torch.cuda.synchronize()
        for v in got.values():
            hip.hipFree(v)


def radix_stream(lg):
    """the radix join on the null stream and on a created (non-blocking) stream, the kind the C++ engine launches on"""
    lg = lg or 30
    n = 1 << lg
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    rj = ops.RadixJoin(n, n)

    def run():
        rj.partition_build(build)
        rj.partition_probe(probe)
        rj.match()

    torch.cuda.synchronize()
    for label, stream in (("null stream", None), ("created stream", torch.cuda.Stream()), ("null stream again", None),
                          ("created stream again", torch.cuda.Stream())):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            r = dropmax(times(run, 5, warm=1))
            pb = dropmax(times(lambda: rj.partition_build(build), 5, warm=0))
            pp = dropmax(times(lambda: rj.partition_probe(probe), 5, warm=0))
            m = dropmax(times(rj.match, 5, warm=0))
        torch.cuda.synchronize()
        print(f"{TAG:12s} 2^{lg} {label:22s}: radix join {r:9.1f} us (build side {pb:8.1f}, probe side {pp:8.1f}, match {m:8.1f})", flush=True)


def radix_idle(lg):
    """what an idle GPU costs the next join: the radix join timed after the device sat idle for a given time (events
    around each phase of ONE join; 4 joins per gap, the mean)"""
    import time
    lg = lg or 30
    n = 1 << lg
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    rj = ops.RadixJoin(n, n)
    rj.partition_build(build); rj.partition_probe(probe); rj.match()
    torch.cuda.synchronize()
    for gap in (0.0, 0.0005, 0.002, 0.01, 0.05, 0.25, 1.0, 0.0):
        acc = [0.0, 0.0, 0.0]
        for _ in range(4):
            torch.cuda.synchronize()
            time.sleep(gap)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(); rj.partition_build(build); ev[1].record(); rj.partition_probe(probe); ev[2].record(); rj.match(); ev[3].record()
            torch.cuda.synchronize()
            for i in range(3):
                acc[i] += ev[i].elapsed_time(ev[i + 1]) * 1e3 / 4
        print(f"{TAG:12s} 2^{lg} idle {gap * 1e3:7.1f} ms before: radix join {sum(acc):9.1f} us (build side {acc[0]:8.1f}, probe side {acc[1]:8.1f}, match {acc[2]:8.1f})", flush=True)


def ramp(_):
    """how long after an idle period do the dwarfs run slower?  Each dwarf: 100 ms idle, then back-to-back steps with an
    event between every two; the mean step time over windows of the sequence"""
    import time
    n = 1 << 28
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    scan_plan = ops.CopyIfLt(n)
    gk = ops.gen_uniform_u32(1 << 26, 42, 0, 65535)
    gv = ops.gen_uniform_u32(1 << 26, 43, 1, 10000)
    gb = ops.GroupBySum(1 << 26, 1 << 16)
    k0 = ops.gen_uniform_u32(1 << 24, 42, 0, 2**32 - 1)
    keys = k0.clone()
    sp = ops.RadixSort(1 << 24, 8)

    def sort_step():
        keys.copy_(k0)
        sp.launch(keys)

    for name, fn, steps in (("scan 2^28", lambda: scan_plan.launch(src, 5), 600), ("group-by 2^26", lambda: gb.launch(gk, gv), 600),
                            ("sort 2^24 (+ copy)", sort_step, 400)):
        fn()
        for idle in (0.1, 0.0):
            torch.cuda.synchronize()
            time.sleep(idle)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            ev[0].record()
            for i in range(steps):
                fn()
                ev[i + 1].record()
            torch.cuda.synchronize()
            t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps)]
            cum = [ev[0].elapsed_time(ev[i + 1]) for i in range(steps)]
            wins = [(0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 400), (400, steps)]
            row = "  ".join(f"[{a}:{b}) {sum(t[a:b]) / (b - a):6.1f}" for a, b in wins if b <= steps and a < b)
            print(f"{TAG:12s} {name:20s} after {idle * 1e3:5.0f} ms idle, us per step: {row}   (step 25 ends at {cum[24]:.1f} ms, step 100 at {cum[99]:.1f} ms)", flush=True)


def radix_sizes(_):
    """the radix join at sizes between the powers of two (2^25 .. 2^30): ns per row should move smoothly — a row that
    costs much more than its neighbours is a geometry step (level fan-outs and tile shapes follow the partition count)"""
    for lg in (25, 26, 27, 28, 29):
        for num in (4, 5, 6, 7):
            n = (num << lg) // 4
            build = ops.gen_uniform_u32(n, 42, 0, n - 1)
            probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
            rj = ops.RadixJoin(n, n)

            def run():
                rj.partition_build(build)
                rj.partition_probe(probe)
                rj.match()

            r = dropmax(times(run, 5, warm=1))
            pb = dropmax(times(lambda: rj.partition_build(build), 5, warm=0))
            rj.result()
            ok = int(rj.ids.to(torch.int64).sum()) == n * (n - 1) // 2
            print(f"{TAG:12s} n = {num}/4 * 2^{lg} = {n:11d}: radix join {r:9.1f} us  {r * 1e3 / n:6.3f} ns/row  (partition one side "
                  f"{pb:8.1f} us) {'ok' if ok else 'WRONG'}", flush=True)
            del build, probe, rj
            torch.cuda.empty_cache()


def _matches(build, probe):
    """number of (build row, probe row) pairs with equal keys, by torch"""
    bk, bc = torch.unique(u64(build), return_counts=True)
    pk, pc = torch.unique(u64(probe), return_counts=True)
    at = torch.searchsorted(bk, pk).clamp(max=bk.numel() - 1)
    hit = bk[at] == pk
    return int((bc[at][hit] * pc[hit]).sum())


def join_skew(lg):
    lg = lg or 26
    n = 1 << lg
    g = torch.Generator(device="cuda").manual_seed(7)
    uni_b, uni_p = ops.gen_uniform_u32(n, 42, 0, n - 1), ops.gen_uniform_u32(n, 43, 0, n - 1)
    hot = torch.rand(n, device="cuda", generator=g) < 0.5

    def shapes():
        yield "uniform", uni_b, uni_p
        yield "half the build rows one key", torch.where(hot, torch.full_like(uni_b, 12345), uni_b), uni_p
        yield "half the probe rows one key", uni_b, torch.where(hot, torch.full_like(uni_p, 12345), uni_p)
        yield "keys are multiples of 65536", bits32(u64(uni_b) % 1024 * 65536), bits32(u64(uni_p) % 2048 * 65536)
        yield "keys are multiples of 1024", bits32(u64(uni_b) % 65536 * 1024), bits32(u64(uni_p) % 65536 * 1024)
        yield "sorted", bits32(torch.sort(u64(uni_b)).values), bits32(torch.sort(u64(uni_p)).values)
        yield "1024 distinct keys on the build side", bits32(u64(uni_b) % 1024), uni_p
        yield "both sides in [1, 10000] (reference)", ops.gen_uniform_u32(n, 42, 1, 10000), ops.gen_uniform_u32(n, 43, 1, 10000)
        yield "both sides in [0, 2^15)", bits32(u64(uni_b) % 32768), bits32(u64(uni_p) % 32768)

    for kind, build, probe in shapes():
        plan = ops.HashJoin(n, n)
        plan.build(build)
        plan.probe(probe)
        b = median(times(lambda: plan.build(build), 3, warm=0))
        p = median(times(lambda: plan.probe(probe), 3, warm=0))
        plan.result()
        total = int(plan.cnt.to(torch.int64).sum())
        del plan
        rj = ops.RadixJoin(n, n)

        def radix():
            rj.partition_build(build)
            rj.partition_probe(probe)
            rj.match()

        r = median(times(radix, 3, warm=1))
        rj.result()
        ok = int(rj.cnt.to(torch.int64).sum()) == total == _matches(build, probe)
        del rj
        print(f"{TAG:16s} 2^{lg} {kind:38s}: build {b:9.1f} probe {p:9.1f} | radix join {r:9.1f} us  matches {'equal' if ok else 'DIFFER'}", flush=True)


def size_sweep(_):
    """every dwarf over sizes next to and between powers of two (median of 5): a row that costs much more per element
    than its neighbours is a geometry cliff"""
    def line(name, n, us):
        print(f"{TAG:12s} {name:28s} n={n:10d}: {us:9.1f} us  {us * 1e3 / max(n, 1):8.3f} ns/row", flush=True)

    for lg in (13, 14, 16, 18, 20, 22, 24, 26):
        for n in ((1 << lg) - 1, 1 << lg, (1 << lg) + 1, 3 << (lg - 1)):
            src = ops.gen_uniform_u32(n, 42, 1, 10000)
            plan = ops.CopyIfLt(n)
            line("scan x<5", n, median(times(lambda: plan.launch(src, 5), 5)))
            line("scan x<5001 dense", n, median(times(lambda: plan.launch(src, 5001, dense=True), 5)))
            keys0 = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
            keys = keys0.clone()
            cp = median(times(lambda: keys.copy_(keys0), 5))
            for bits in (8, 4):
                sp = ops.RadixSort(n, bits)

                def run():
                    keys.copy_(keys0)
                    sp.launch(keys)

                line(f"sort {bits}-bit", n, median(times(run, 5)) - cp)
            for groups in (1000, 32768, 32769, 65536, 65537):
                gk = ops.gen_uniform_u32(n, 42, 0, groups - 1)
                gp = ops.GroupBySum(n, groups)
                line(f"groupby G={groups}", n, median(times(lambda: gp.launch(gk, src), 5)))
                del gk, gp
            b = ops.gen_uniform_u32(n, 42, 0, n - 1)
            jp = ops.HashJoin(n, n)
            line("join build", n, median(times(lambda: jp.build(b), 3)))
            line("join probe", n, median(times(lambda: jp.probe(keys0), 3)))
            del jp, b, src, plan, keys0, keys
            torch.cuda.empty_cache()


def partition(_):
    n = 1 << 27
    keys = ops.gen_uniform_u32(n, 42, 0, (1 << 30) - 1)
    for parts in (2, 4, 8, 16, 64, 256):
        t = median(times(lambda: ops.partition_by_hash(keys, 0, parts), 5, warm=1))
        print(f"{TAG:12s} 2^27 rows -> {parts:4d} buckets: {t:8.1f} us (incl. output allocation)", flush=True)


def reduce(_):
    n = 1 << 28
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    want = int(src.sum(dtype=torch.int64)) & 0xFFFFFFFF
    t = median(times(lambda: ops.reduce_sum(src), 15, warm=3))
    got = int(ops.reduce_sum(src)) & 0xFFFFFFFF
    print(f"{TAG:12s} wgs={os.environ.get('DBHIP_RED_WGS', '-')}: {t:7.1f} us ({4 * n / t / 8e6 * 100:4.1f} %) "
          f"{'ok' if got == want else 'WRONG'}", flush=True)


def xscan(lg):
    lg = lg or 28
    n = 1 << lg
    base = ops.gen_uniform_u32(n + 4, 42, 0, 1000)
    outb = torch.empty(n + 4, dtype=torch.int32, device="cuda")
    for off in (0, 1):
        src, out = base[off: off + n], outb[off: off + n]
        plan = ops.ExclusiveScan(n)
        t = median(times(lambda: plan.launch(src, out=out), 9))
        plan.result()
        m = min(n - 1, 1 << 20)
        ref = torch.cumsum(src[:m].to(torch.int64), 0)
        ok = bool(torch.equal(u64(out[1:m + 1]), ref & 0xFFFFFFFF)) and int(out[0]) == 0
        print(f"{TAG:12s} 2^{lg} offset {off}: {t:8.1f} us  ({8 * n / t / 8e6 * 100:4.1f} % of 8 TB/s on 8n bytes)  "
              f"{'ok' if ok else 'WRONG'}", flush=True)


def graph(_):
    import time

    def wall(fn, iters=200):
        for _i in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _i in range(iters):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

    def captured(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    for lg in (14, 16, 18, 20):
        n = 1 << lg
        keys = ops.gen_uniform_u32(n, 1, 0, 2**32 - 1)
        plan = ops.RadixSort(n, 8)
        direct = wall(lambda: plan.launch(keys))
        replay = wall(captured(lambda: plan.launch(keys)).replay)
        b, p = ops.gen_uniform_u32(n, 2, 0, n - 1), ops.gen_uniform_u32(n, 3, 0, n - 1)
        j = ops.HashJoin(n, n)

        def both():
            j.build(b)
            j.probe(p)

        jd = wall(both)
        jg = wall(captured(both).replay)
        print(f"n=2^{lg}: sort direct {direct:.1f} us  graph {replay:.1f} us | join direct {jd:.1f} us  graph {jg:.1f} us", flush=True)


def launch_join(lg):
    n = 1 << (lg or 26)
    build, probe = ops.gen_uniform_u32(n, 42, 0, n - 1), ops.gen_uniform_u32(n, 43, 0, n - 1)
    hot = os.environ.get("JOIN_HOT", "")  # build | probe: every other row of that side carries one key; few: 1024 distinct build keys
    if hot == "build":
        build[::2] = 12345
    if hot == "probe":
        probe[::2] = 12345
    if hot == "few":  # 1024 distinct build keys
        build = bits32(u64(build) % 1024)
    plan = ops.HashJoin(n, n)
    for _i in range(5):
        plan.build(build)
        plan.probe(probe)
    torch.cuda.synchronize()
    plan.result()
    if os.environ.get("JOIN_RADIX"):
        rj = ops.RadixJoin(n, n)
        for _i in range(3):
            rj.partition_build(build)
            rj.partition_probe(probe)
            rj.match()
        rj.result()
    print("ok")


def launch_sort(lg):
    n = 1 << (lg or 24)
    keys0 = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    for bits in (8, 4):
        plan = ops.RadixSort(n, bits)
        for _i in range(5):
            k = keys0.clone()
            plan.launch(k)
    torch.cuda.synchronize()
    print("ok")


def launch_all(_):
    """a few calls of every BASELINE configuration (SQ counters per kernel)"""
    n = 1 << 28
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    sc = ops.CopyIfLt(n)
    for filt in (5, 5, 5001, 5001):
        sc.launch(src, filt, dense=filt > 1000)
    del src, sc
    launch_sort(24)
    n = 1 << 26
    keys, vals = ops.gen_uniform_u32(n, 42, 0, 65535), ops.gen_uniform_u32(n, 43, 1, 10000)
    gb = ops.GroupBySum(n, 1 << 16)
    for _i in range(3):
        gb.launch(keys, vals)
    torch.cuda.synchronize()
    print("ok")


MODES = {"radix": radix, "ramp": ramp, "radix-idle": radix_idle, "radix-stream": radix_stream, "radix-alloc": radix_alloc, "radix-offsets": radix_offsets, "radix-sizes": radix_sizes, "graph": graph, "launch-join": launch_join, "launch-sort": launch_sort, "launch-all": launch_all, "scan": scan, "sort": sort, "sort-only": sort_only, "groupby": groupby, "groupby-shapes": groupby_shapes, "groupby-skew": groupby_skew, "sort-shapes": sort_shapes, "join": join, "join-skew": join_skew, "size-sweep": size_sweep, "partition": partition,
         "reduce": reduce, "xscan": xscan}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in MODES:
        raise SystemExit(__doc__)
    MODES[sys.argv[1]](int(sys.argv[2]) if len(sys.argv) > 2 else None)
