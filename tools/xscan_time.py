"""development aid: exclusive scan of 2^LG uint32 (HIP events, median of 9), aligned (single launch) and offset by one
element (three launches)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << lg
base = ops.gen_uniform_u32(n + 4, 42, 0, 1000)
outb = torch.empty(n + 4, dtype=torch.int32, device="cuda")
for off in (0, 1):
    src, out = base[off: off + n], outb[off: off + n]
    ts = []
    for _ in range(11):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.exclusive_scan(src, out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts = sorted(ts[2:])
    m = min(n - 1, 1 << 20)
    ref = torch.cumsum(src[:m].to(torch.int64), 0)
    ok = bool(torch.equal(out[1:m + 1].to(torch.int64) & 0xFFFFFFFF, ref & 0xFFFFFFFF)) and int(out[0]) == 0
    print(f"2^{lg} offset {off}: {ts[len(ts) // 2]:8.1f} us  ({8 * n / ts[len(ts) // 2] / 8e6 * 100:4.1f} % of 8 TB/s on 8n bytes)  {'ok' if ok else 'WRONG'}", flush=True)
