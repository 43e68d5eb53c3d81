"""development aid: a few 2^24 sorts (8-bit digits) for counter collection"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 24
keys0 = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
keys = keys0.clone()
plan = ops.RadixSort(n, int(os.environ.get("SORT_BITS", "8")))
for _ in range(3):
    keys.copy_(keys0)
    plan.launch(keys)
torch.cuda.synchronize()
print("ok")
