import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
from tools.quick_bench import timeit
for lg in (18, 19, 20, 21, 22, 23):
    n = 1 << lg
    src = ops.gen_uniform_u32(n, 42, 1, 10000)
    plan = ops.CopyIfLt(n)
    mn, med = timeit(lambda: plan.launch(src, 5), iters=21, warm=3)
    print(f"BIG_LOG2={os.environ.get('DBHIP_SCAN_BIG_LOG2','22')} n=2^{lg}: min {mn:.1f} med {med:.1f} us", flush=True)
