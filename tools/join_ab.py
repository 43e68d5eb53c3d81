"""development aid: build / probe time of the 2^LG x 2^LG join (HIP events, drop-max-mean of 7) for the library named by DBHIP_LIB"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
build = ops.gen_uniform_u32(n, 42, 0, n - 1)
probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
plan = ops.HashJoin(n, n)
def ev(fn, k=7):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    out.sort()
    return sum(out[:-1]) / (len(out) - 1)
plan.build(build); plan.probe(probe)
b = ev(lambda: plan.build(build)); p = ev(lambda: plan.probe(probe))
plan.result()
print(f"{os.environ.get('DBHIP_LIB', 'default'):70s} 2^{lg}: build {b:8.1f} us  probe {p:8.1f} us  total {b + p:8.1f}", flush=True)
