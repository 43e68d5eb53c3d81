import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
from tools.quick_bench import timeit
n = 1 << 28
src = ops.gen_uniform_u32(n, 42, 1, 10000)
plan = ops.CopyIfLt(n)
for filt in (5, 5001):
    mn, med = timeit(lambda: plan.launch(src, filt))
    print(f"{os.environ.get('TAG','')} filter={filt}: min {mn:.1f} med {med:.1f} us  {4*n/med/1e6:.3f} TB/s")
