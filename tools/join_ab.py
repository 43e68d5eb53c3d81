"""development aid: build / probe / radix-join time of the 2^LG x 2^LG join (HIP events, drop-max-mean of 7) for DBHIP_LIB"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
build = ops.gen_uniform_u32(n, 42, 0, n - 1)
probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
def ev(fn, k=7):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    out.sort()
    return sum(out[:-1]) / (len(out) - 1)
plan = ops.HashJoin(n, n)
plan.build(build); plan.probe(probe)
b = ev(lambda: plan.build(build)); p = ev(lambda: plan.probe(probe))
plan.result()
total = int(plan.cnt.to(torch.int64).sum())
del plan
rj = ops.RadixJoin(n, n)
def radix():
    rj.partition_build(build); rj.partition_probe(probe); rj.match()
radix()
r = ev(radix)
pb = ev(lambda: rj.partition_build(build)); m = ev(rj.match)
rj.result()
ok = int(rj.cnt.to(torch.int64).sum()) == total
print(f"{os.environ.get('DBHIP_LIB', 'default').split('libdbhip_')[-1]:20s} 2^{lg}: build {b:8.1f} probe {p:8.1f} total {b + p:8.1f} us | radix join {r:8.1f} us "
      f"(partition one side {pb:7.1f}, match {m:7.1f}) matches {'equal' if ok else 'DIFFER'}", flush=True)
