"""development aid: time of the rank-level partition (dbhip_pjoin_partition_u32) of 2^27 rows into P buckets"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 27
keys = ops.gen_uniform_u32(n, 42, 0, (1 << 30) - 1)
for parts in (2, 4, 8, 16, 64, 256):
    ops.partition_by_hash(keys, 0, parts)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = ops.partition_by_hash(keys, 0, parts); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"{os.environ.get('DBHIP_LIB', 'default').split('libdbhip_')[-1]:12s} 2^27 rows -> {parts:4d} buckets: {ts[2]:8.1f} us (incl. output allocation)", flush=True)
    del r
