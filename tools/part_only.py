"""development aid: rank-level partition (8 buckets) of 2^27 rows, a few times"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 27
keys = ops.gen_uniform_u32(n, 42, 0, (1 << 30) - 1)
for parts in (8, 2):
    for _ in range(3):
        ops.partition_by_hash(keys, 0, parts)
torch.cuda.synchronize()
print("ok")
