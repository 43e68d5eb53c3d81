import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
from tools.quick_bench import timeit
for lg in (8, 10, 12, 14, 15, 16):
    n = 1 << lg
    ak, bk = ops.gen_unique_sorted_u32(n, 11), ops.gen_unique_sorted_u32(n, 12)
    plan = ops.UniqueJoin(n, n)
    def both(): plan.build(ak, ak); plan.probe(bk, bk)
    mn, med = timeit(both, iters=21, warm=3)
    plan.result()
    print(f"path={os.environ.get('DBHIP_JOIN_PATH','auto')} ujoin n=2^{lg}: min {mn:.1f} med {med:.1f} us", flush=True)
