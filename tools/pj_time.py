import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops, pjoin
for lg in (28, 30):
    n = 1 << lg
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    plan = ops.HashJoin(n, n)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.build(build); torch.cuda.synchronize(); t1 = time.perf_counter()
        plan.probe(probe); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"plan reuse 2^{lg} it{it}: build {1e3*(t1-t0):.1f} ms probe {1e3*(t2-t1):.1f} ms", flush=True)
    plan.result()
    del plan
    torch.cuda.empty_cache()
    res = None
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = pjoin.partitioned_join(build, probe, 0, 0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"partitioned_join(world=1) 2^{lg} it{it}: {1e3*(t1-t0):.1f} ms  reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
    del res, build, probe
    torch.cuda.empty_cache()
