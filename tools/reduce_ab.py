"""development aid: 2^28-row reduce time (HIP events, median of 15) for DBHIP_LIB / DBHIP_RED_WGS"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 28
src = ops.gen_uniform_u32(n, 42, 1, 10000)
want = int(src.sum(dtype=torch.int64)) & 0xFFFFFFFF
ts = []
for _ in range(18):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r = ops.reduce_sum(src); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
got = int(r) & 0xFFFFFFFF
ts = sorted(ts[3:])
print(f"{os.environ.get('DBHIP_LIB', 'default').split('libdbhip_')[-1]:12s} wgs={os.environ.get('DBHIP_RED_WGS', '-')}: {ts[len(ts) // 2]:7.1f} us ({4 * n / ts[len(ts) // 2] / 8e6 * 100:4.1f} %) {'ok' if got == want else 'WRONG'}", flush=True)
