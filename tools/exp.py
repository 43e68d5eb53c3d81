import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 26
build = ops.gen_uniform_u32(n, 42, 0, n - 1)
probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
plan = ops.HashJoin(n, n)
plan.build(build)
torch.cuda.synchronize()
print("header words 28..40 after build:", plan.ws[28 * 4: 40 * 4].view(torch.int32).tolist() if plan.ws.dtype == torch.uint8 else plan.ws.view(torch.int32)[28:40].tolist())
plan.probe(probe)
torch.cuda.synchronize()
print("after probe:", plan.ws.view(torch.int32)[28:40].tolist() if plan.ws.dtype != torch.uint8 else plan.ws[28 * 4: 40 * 4].view(torch.int32).tolist())
