"""development aid: a few calls of every BASELINE configuration, for rocprofv3 --pmc passes (SQ counters per kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 28
src = ops.gen_uniform_u32(n, 42, 1, 10000)
scan = ops.CopyIfLt(n)
for filt in (5, 5, 5001, 5001):
    scan.launch(src, filt, dense=filt > 1000)
del src, scan
n = 1 << 24
keys0 = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
for bits in (8, 4):
    plan = ops.RadixSort(n, bits)
    for _ in range(3):
        k = keys0.clone(); plan.launch(k)
n = 1 << 26
keys = ops.gen_uniform_u32(n, 42, 0, 65535); vals = ops.gen_uniform_u32(n, 43, 1, 10000)
gb = ops.GroupBySum(n, 1 << 16)
for _ in range(3):
    gb.launch(keys, vals)
torch.cuda.synchronize()
print("ok")
