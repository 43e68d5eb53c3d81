"""development aid: a few 2^24-key sorts (8- and 4-bit) for rocprofv3 --kernel-trace --stats"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
keys0 = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
for bits in (8, 4):
    plan = ops.RadixSort(n, bits)
    for _ in range(5):
        k = keys0.clone(); plan.launch(k)
torch.cuda.synchronize()
print("ok")
