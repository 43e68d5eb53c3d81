"""development aid: group-by time (HIP events, drop-max-mean of 9) for DBHIP_LIB"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 26
res = []
for groups in (1 << 16, 1 << 15, 1 << 10, 64):
    keys = ops.gen_uniform_u32(n, 42, 0, groups - 1); vals = ops.gen_uniform_u32(n, 43, 1, 10000)
    plan = ops.GroupBySum(n, groups)
    plan.launch(keys, vals); plan.launch(keys, vals)
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.launch(keys, vals); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    ref = torch.zeros(groups, dtype=torch.int64, device="cuda").index_add_(0, keys.to(torch.int64), vals.to(torch.int64))
    ok = bool(torch.equal(plan.result().to(torch.int64) & 0xFFFFFFFF, ref & 0xFFFFFFFF))
    res.append(f"G={groups}: {sum(ts[:-1]) / 8:6.1f} us {'ok' if ok else 'WRONG'}")
print(f"{os.environ.get('DBHIP_LIB', 'default').split('libdbhip_')[-1]:16s} " + "  ".join(res), flush=True)
