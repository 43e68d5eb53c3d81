import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
from tools.quick_bench import timeit
for lg in (8, 10, 12, 14, 15, 16):
    n = 1 << lg
    for hi, name in ((n - 1, "uniform"), (10000, "ref")):
        b = ops.gen_uniform_u32(n, 42, 0, hi); p = ops.gen_uniform_u32(n, 43, 0, hi)
        plan = ops.HashJoin(n, n)
        def both(): plan.build(b); plan.probe(p)
        mn, med = timeit(both, iters=21, warm=3)
        plan.result()
        print(f"path={os.environ.get('DBHIP_JOIN_PATH','auto')} n=2^{lg} {name}: min {mn:.1f} med {med:.1f} us", flush=True)
