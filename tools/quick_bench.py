"""Quick per-dwarf timing (torch events on the current stream) — development aid, not the contract bench."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops


def timeit(fn, iters=9, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def main():
    which = sys.argv[1:] or ["scan", "sort", "groupby", "join"]
    print(ops.device_info())
    if "scan" in which:
        for lg in (20, 24, 28):
            n = 1 << lg
            src = ops.gen_uniform_u32(n, 42, 1, 10000)
            plan = ops.CopyIfLt(n)
            for filt in (5, 1001, 5001):
                mn, med = timeit(lambda: plan.launch(src, filt))
                cnt = plan.result().numel()
                byts = 4 * n + 4 * cnt
                print(f"scan n=2^{lg} filter={filt} sel={cnt/n:.4f}: min {mn:.1f} us med {med:.1f} us  {n/med:.0f} Mrows/s  {byts/med/1e6:.3f} TB/s ({byts/med/1e6/8*100:.1f}% of 8TB/s)")
            del plan, src
    if "sort" in which:
        for lg in (20, 24):
            n = 1 << lg
            for bits in (8, 4):
                for (lo, hi, name) in ((0, 2**32 - 1, "full"), (1, 10000, "ref")):
                    keys0 = ops.gen_uniform_u32(n, 42, lo, hi)
                    keys = keys0.clone()
                    plan = ops.RadixSort(n, bits)
                    def run():
                        keys.copy_(keys0)
                        plan.launch(keys)
                    mn, med = timeit(run)
                    cmn, cmed = timeit(lambda: keys.copy_(keys0))
                    assert ops.workspace_status(plan.ws) == 0
                    t = med - cmed
                    print(f"sort n=2^{lg} bits={bits} {name}: med {t:.1f} us (copy {cmed:.1f})  {n/t:.0f} Mkeys/s  8N-roofline {8*n/t/1e6/8*100:.1f}%")
    if "groupby" in which:
        for (lg, groups) in ((20, 64), (26, 20), (26, 64), (26, 1 << 12), (26, 1 << 15), (26, 1 << 16), (26, 1 << 18)):
            n = 1 << lg
            keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
            vals = ops.gen_uniform_u32(n, 43, 1, 10000)
            plan = ops.GroupBySum(n, groups)
            mn, med = timeit(lambda: plan.launch(keys, vals))
            plan.result()
            byts = 8 * n + 4 * groups
            print(f"groupby n=2^{lg} groups={groups}: min {mn:.1f} med {med:.1f} us  {n/med:.0f} Mrows/s  {byts/med/1e6:.3f} TB/s ({byts/med/1e6/8*100:.1f}%)")
    if "join" in which:
        for lg in (20, 24, 26):
            n = 1 << lg
            build = ops.gen_uniform_u32(n, 42, 0, n - 1)
            probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
            plan = ops.HashJoin(n, n)
            bmn, bmed = timeit(lambda: plan.build(build), iters=5, warm=1)
            pmn, pmed = timeit(lambda: plan.probe(probe), iters=5, warm=1)
            plan.result()
            t = bmed + pmed
            print(f"join n=2^{lg}: build {bmed:.0f} us probe {pmed:.0f} us total {t:.0f} us  {2*n/t:.0f} Mrows/s  20N-roofline {20*n/t/1e6/8*100:.2f}%")
            del plan
    if "ujoin" in which:
        for lg in (20, 24, 26):
            n = 1 << lg
            ak, bk = ops.gen_unique_sorted_u32(n, 11), ops.gen_unique_sorted_u32(n, 12)
            plan = ops.UniqueJoin(n, n)
            bmn, bmed = timeit(lambda: plan.build(ak, ak), iters=5, warm=1)
            pmn, pmed = timeit(lambda: plan.probe(bk, bk), iters=5, warm=1)
            plan.result()
            t = bmed + pmed
            print(f"ujoin n=2^{lg}: build {bmed:.0f} us probe {pmed:.0f} us total {t:.0f} us  {2*n/t:.0f} Mrows/s")
            del plan
    if "reduce" in which:
        for lg in (20, 24, 28):
            n = 1 << lg
            src = ops.gen_uniform_u32(n, 42, 1, 10000)
            mn, med = timeit(lambda: ops.reduce_sum(src))
            print(f"reduce n=2^{lg}: min {mn:.1f} med {med:.1f} us  {n/med:.0f} Mrows/s  {4*n/med/1e6:.3f} TB/s ({4*n/med/1e6/8*100:.1f}%)")
    if "hashbuild" in which:
        for lg in (20, 24, 26):
            n = 1 << lg
            keys = ops.gen_uniform_u32(n, 42, 1, 10000)
            t = ops.BitmaskTable(2 * n, 1, 421)
            def run():
                t.reset()
                t.insert(keys, keys)
            mn, med = timeit(run, iters=5, warm=1)
            rmn, rmed = timeit(t.reset, iters=5, warm=1)
            print(f"hashbuild(bitmask, dup keys) n=2^{lg}: med {med - rmed:.0f} us (+reset {rmed:.0f})  {n/(med-rmed):.0f} Mrows/s")
            ukeys = ops.gen_unique_sorted_u32(n, 42) if n <= (1 << 28) else keys
            def run2():
                t.reset()
                t.insert(ukeys, ukeys)
            mn, med = timeit(run2, iters=5, warm=1)
            print(f"hashbuild(bitmask, unique keys) n=2^{lg}: med {med - rmed:.0f} us  {n/(med-rmed):.0f} Mrows/s")
            uj = ops.UniqueJoin(n, n)
            mn, med = timeit(lambda: uj.build(ukeys, ukeys), iters=5, warm=1)
            print(f"hashbuild(CAS, unique keys) n=2^{lg}: med {med:.0f} us  {n/med:.0f} Mrows/s")
            del t, uj
    if "nlj" in which:
        for n in (1024, 4096, 16384):
            a = ops.gen_uniform_u32(n, 42, 1, 10000)
            b = ops.gen_uniform_u32(n, 44, 1, 10000)
            mn, med = timeit(lambda: ops.nested_join(a, a, b, b), iters=5, warm=1)
            print(f"nested join n={n}: med {med:.1f} us  {12*n*n/med/1e6:.3f} TB/s written")


if __name__ == "__main__":
    main()
