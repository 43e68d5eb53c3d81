"""development aid: a few 2^26 x 2^26 joins and 2^26-row group-bys for counter collection"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 26
build = ops.gen_uniform_u32(n, 42, 0, n - 1)
probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
plan = ops.HashJoin(n, n)
for _ in range(3):
    plan.build(build)
    plan.probe(probe)
torch.cuda.synchronize()
del plan, build, probe
groups = 1 << 16
keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
vals = ops.gen_uniform_u32(n, 43, 1, 10000)
gb = ops.GroupBySum(n, groups)
for _ in range(3):
    gb.launch(keys, vals)
torch.cuda.synchronize()
print("ok")
