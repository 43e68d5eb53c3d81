import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
for lg in (28, 29, 30):
    n = 1 << lg
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    rids = torch.arange(n, dtype=torch.int32, device="cuda")
    plan = ops.HashJoin(n, n)
    for name, r in (("keys only", None), ("pairs", rids), ("pairs after partition(1)", "part")):
        if r == "part":
            pk, pr, _ = ops.partition_by_hash(build, 0, 1)
            args = (pk, pr)
        else:
            args = (build, r)
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            plan.build(*args); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"2^{lg} {name}: build {1e3*(t1-t0):.1f} ms  status {ops.workspace_status(plan.ws)}", flush=True)
    del plan, build, rids
    torch.cuda.empty_cache()
