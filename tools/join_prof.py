"""development aid: a few 2^LG x 2^LG joins for rocprofv3 --kernel-trace --stats (per-kernel times of build and probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
build = ops.gen_uniform_u32(n, 42, 0, n - 1)
probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
plan = ops.HashJoin(n, n)
for _ in range(5):
    plan.build(build)
    plan.probe(probe)
torch.cuda.synchronize()
plan.result()
print("ok")
