"""development aid: two-launch vs dense copy_if at 2^28 rows over a selectivity sweep (HIP events, median of 9)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 28
src = ops.gen_uniform_u32(n, 42, 1, 10000)
plan = ops.CopyIfLt(n)
def med(fn, k=9):
    ts = []
    fn(); fn()
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for filt in (5, 101, 1001, 2501, 5001, 7501, 10001):
    t0 = med(lambda: plan.launch(src, filt, dense=False)); m = plan.result().numel()
    t1 = med(lambda: plan.launch(src, filt, dense=True)); m1 = plan.result().numel()
    byts = 4 * n + 4 * m
    print(f"s={m / n:6.4f}: two-launch {t0:7.1f} us ({byts / t0 / 8e6 * 100:4.1f} %)   dense {t1:7.1f} us ({byts / t1 / 8e6 * 100:4.1f} %)  {'same count' if m == m1 else 'COUNT DIFFERS'}", flush=True)
